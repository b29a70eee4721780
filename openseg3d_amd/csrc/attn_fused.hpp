// Geometry and helpers shared by the fused window-attention kernels (attention_fused.hip: forward,
// attention_fused_bwd.hip: the two backward passes).  See attention_fused.hip for the design.
#pragma once
#include <cstdlib>
#include "attn_common.hpp"
#include "attn_dropout.hpp"

namespace attn_fused {

using namespace attn;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int DH>
struct Cfg {
    static constexpr bool kNarrow = DH <= 12;          // several heads of one 32-query tile per workgroup
    static constexpr int HG = kNarrow ? 4 : 1;         // heads per workgroup
    static constexpr int HPT = HG / 4;                 // narrow: whole heads per staging thread (4 threads per key)
    static constexpr int QT = kNarrow ? 1 : 4;         // 32-query tiles per workgroup
    static constexpr int UW = HG * QT / 4;             // (tile, head) units per wave
    static constexpr int DHS = (DH + 7) / 8 * 8;       // K channels stored per head
    static constexpr int KS = (DHS + 31) / 32;         // MFMA k-steps over the head dimension
    static constexpr bool kOnes = DH % 16 != 0;        // a spare V column holds ones: the row sum rides in the PV product
    static constexpr int VW = (DH + (kOnes ? 1 : 0) + 15) / 16 * 16;  // V channels stored per head
    static constexpr int NB = VW / 16;                 // 16-row d-blocks of O^T
    // LDS row strides in bytes (data + padding chosen so that the fragment reads spread over the banks)
    static constexpr int KRS = HG * DHS * 2 + (kNarrow ? 16 : (DH == 24 ? 0 : 0));
    static constexpr int VRS = HG * VW * 2 + (kNarrow ? 32 : (DH == 24 ? 32 : 0));
    // one plane (hi or lo) of a staged key tile + 16 zero bytes: fragment lanes whose channels lie past the stored width read
    // THAT block (a loop-invariant address) instead of zero-filling their registers under a branch per tile
    static constexpr int kPlaneData = 32 * (KRS + VRS);
    static constexpr int kPlane = kPlaneData + 16;
    static constexpr int kTile = 2 * kPlane;
    static constexpr int NBUF = kNarrow ? 1 : 2;       // narrow: one buffer, more workgroups per CU
    static constexpr int CT = kNarrow ? HPT * DH : DH / 4;  // fp32 values a staging thread converts per row
    // waves per SIMD the register allocator must leave room for (measured spill-free points)
    static constexpr int kWaves = DH == 6 ? 4 : DH == 12 ? 4 : DH == 24 ? 4 : 2;
};

__device__ __forceinline__ float quad_sum(float x) {
    // sum over the 4 lanes of a quad (DPP quad_perm xor 1, xor 2)
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    return x;
}

// 1 / max(|x|, eps) from the sum of squares (F.normalize, cosine_msa.py:152-153): v_rsq_f32 (1 ulp) instead of the
// correctly rounded sqrt + divide sequences (~40 instructions per key per tile); the operands are then rounded to
// 16 significant bits anyway
__device__ __forceinline__ float inv_norm(float ss) { return __builtin_amdgcn_rsqf(fmaxf(ss, kNormEps * kNormEps)); }

// two fp32 -> packed bf16 hi pair and lo pair
__device__ __forceinline__ void split2(float a, float b, uint32_t* hi, uint32_t* lo) {
    const uint32_t w = pack_bf16(a, b);
    *hi = w;
    *lo = pack_bf16(a - __builtin_bit_cast(float, w << 16), b - __builtin_bit_cast(float, w & 0xFFFF0000u));
}

// Work items are dealt to the 8 XCDs in BLOCKS of consecutive items (items are sorted by window: the 32-token tiles /
// 128-token chunks of one window are neighbours in the list and stream the SAME key / value rows, so they belong on one
// L2).  Item of the q-th item slot of group gx:  ((q / B) * XG + gx) * B + q % B.  Measured fabric traffic of the forward
// with single items dealt round-robin (B = 1): 1.5 - 2.7 x the footprint of q | k | v and the output (tools/collect_attn_traffic.sh).
__host__ __device__ __forceinline__ int xcd_block_item(int q, int gx, int xg, int b) { return ((q / b) * xg + gx) * b + q % b; }
// item slots of group gx that hold a real item (the list's last, partial block belongs to one group and sits at its end)
__host__ __device__ __forceinline__ int xcd_block_count(int n_items, int gx, int xg, int b) {
    const int nb = (n_items + b - 1) / b;               // blocks in the list
    const int mine = nb > gx ? (nb - gx + xg - 1) / xg : 0;  // blocks of this group
    if (mine == 0) return 0;
    const int last = nb - 1;
    return mine * b - (last % xg == gx ? nb * b - n_items : 0);
}
// items per block: SEG3D_ATTN_XCD_BLOCK (A/B; > 0 forces one size for every kernel).  Default (round 5): 8 consecutive
// 32-token tiles for the narrow heads (dh 6 / 12), 4 consecutive 128-token chunks for the wide heads -- measured on the
// headline scene's forward (tools/r5_s6.sh, two --pmc passes per setting): fabric traffic per launch at block sizes 1 / 4 / 8:
// dh 12 506 / 288 / 243 MB (footprint 186 MB: 2.7 x -> 1.3 x), dh 24 297 / 224 / 213, dh 48 185 / 140 / 135, dh 6 135 / 101 /
// 95; forward of the 18 layers 1.554 / 1.498 / 1.511 ms.  (Round 3 had measured forward + backward a tie at 1 / 2 / 4 and
// +2 % at 8 - 16 with ONE size for all kernels; whole windows on one XCD, 32, unbalance the groups: +6 %.)
// Forward + backward with dropout of the 18 layers, same box, old (1 everywhere) against these defaults: 7.29 -> 7.24 ms, the
// training step a tie (43.0 ms) -- the kernels are not bound by the fabric, the traffic is what falls.  Lists of fewer than
// 512 chunks (stage 4: 242) keep single items: blocks of 4 leave 60 blocks for 8 XCDs and cost that stage 5 %.
inline int xcd_block_items(bool narrow, int n_items) {
    static const int env = getenv("SEG3D_ATTN_XCD_BLOCK") ? atoi(getenv("SEG3D_ATTN_XCD_BLOCK")) : 0;
    return env > 0 ? env : (narrow ? 8 : (n_items >= 512 ? 4 : 1));
}

}  // namespace attn_fused
