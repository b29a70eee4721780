// Shared pieces of the split-bf16 MFMA window-attention kernels (attention_mfma.hip, attention_bwd.hip).
#pragma once
#include "common.hpp"

namespace attn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr float kNormEps = 1e-12f;  // F.normalize eps, cosine_msa.py:152-153
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ void split1(float v, __bf16* hi, __bf16* lo) {
    const __bf16 h = (__bf16)v;
    *hi = h;
    *lo = (__bf16)(v - (float)h);
}

// 8 fp32 values -> bf16 hi / lo MFMA fragments
__device__ __forceinline__ void split_frag(const float* v, bf16x8* hi, bf16x8* lo) {
    u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t w = pack_bf16(v[2 * i], v[2 * i + 1]);
        const float h0 = __builtin_bit_cast(float, w << 16);
        const float h1 = __builtin_bit_cast(float, w & 0xFFFF0000u);
        h[i] = w;
        l[i] = pack_bf16(v[2 * i] - h0, v[2 * i + 1] - h1);
    }
    *hi = __builtin_bit_cast(bf16x8, h);
    *lo = __builtin_bit_cast(bf16x8, l);
}

// the same from four float pairs: the low halves come out of one packed subtract per pair (v_pk_add_f32)
__device__ __forceinline__ void split_frag2(const f32x2* v, bf16x8* hi, bf16x8* lo) {
    u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t w = pack_bf16(v[i][0], v[i][1]);
        const f32x2 hf = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xFFFF0000u)};
        const f32x2 d = v[i] - hf;
        h[i] = w;
        l[i] = pack_bf16(d[0], d[1]);
    }
    *hi = __builtin_bit_cast(bf16x8, h);
    *lo = __builtin_bit_cast(bf16x8, l);
}

// Masking streamed tokens past a window's end INSIDE the score product: the K dimension of the score MFMAs is padded from
// the stored head width DHS to a multiple of 32, so channel DHS is spare.  The stationary (query) fragment carries 1.0
// there, the streamed (key) fragment kMaskScore for a token past the end and 0 otherwise: the score of such a token comes
// out of the matrix core as -16384 + (a real row's score), exp2 of it is exactly 0 under any admissible temperature
// (scores are bounded by log2e / tau_min = 144), and no per-element compare / select is spent on it.
constexpr uint32_t kSpareOne = 0x3F80u;    // bf16 1.0 in element 0 of a fragment dword
constexpr uint32_t kSpareMask = 0xC680u;   // bf16 -16384.0

// acc += a . b in split-bf16 (hi*hi + hi*lo + lo*hi)
__device__ __forceinline__ f32x4 mfma3(const bf16x8& a_hi, const bf16x8& a_lo, const bf16x8& b_hi, const bf16x8& b_lo,
                                       f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b_hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_hi, acc, 0, 0, 0);
    return acc;
}

// 16-B fragment load (8 bf16) from a hi block and the lo block `half` elements behind it; zero when !ok
__device__ __forceinline__ void load_frag(const __bf16* base, int64_t off, int64_t half, bool ok, bf16x8* hi, bf16x8* lo) {
    u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
    if (ok) {
        a = *reinterpret_cast<const u32x4*>(base + off);
        b = *reinterpret_cast<const u32x4*>(base + half + off);
    }
    *hi = __builtin_bit_cast(bf16x8, a);
    *lo = __builtin_bit_cast(bf16x8, b);
}

// slot of tile-local token r in the transposed arrays: token kappa(g, j) = (j < 4 ? 4g + j : 16 + 4g + j - 4)
// lives at slot 8g + j, so the 8 accumulator registers a lane holds after two 16-row MFMA tiles
// ([u=0: r 0..3, u=1: r 0..3]) are, in order, the k-slots of one 16-B transposed fragment.
__device__ __forceinline__ int perm_slot(int r) {
    return r < 16 ? (r >> 2) * 8 + (r & 3) : ((r - 16) >> 2) * 8 + 4 + ((r - 16) & 3);
}

template <int DH>
struct Geo {
    static constexpr int DHS = (DH + 7) / 8 * 8;  // stored channels per head (multiple of 8)
    static constexpr int KS = (DHS + 31) / 32;    // MFMA k-steps over the head dimension
    static constexpr int NB = (DH + 15) / 16;     // 16-row d-blocks of a transposed output
};

}  // namespace attn
