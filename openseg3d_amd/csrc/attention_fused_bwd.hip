// a19-a21 backward, fused: gradients of the ragged sparse-window cosine attention w.r.t. the raw q, k, v and tau
// (cosine_msa.py:115-177) in two launches per layer, flash-style recompute from the forward's LSE.
//
// The earlier backward made three launches: attn_prepare_bwd wrote seven bf16 hi/lo copies of q, k, v, dO (row-major and
// tile-transposed) plus LSE / delta through HBM, then a dQ pass and a dK/dV pass read them back.  Here each pass stages
// what it streams straight from the raw fp32 rows, exactly as the fused forward does (attention_fused.hip): 256 threads
// gather 32 streamed tokens, normalise / scale / split them and park TWO row-major images in LDS; row fragments come back
// with ds_read_b128, transposed fragments with ds_read_b64_tr_b16 out of the SAME images -- no transposed copies exist.
//   pass Q  (MODE 0): stationary = the workgroup's queries (Q~, dO fragments, LSE, delta = <dO, O> in registers);
//                     streamed keys: images K^ and V.       S^T = K^.Q~^T, dP^T = V.dO^T, dS = P (D dP - delta),
//                     dQ^^T += K^^T.dS^T (K^^T by tr-read of the K^ image), dtau += <dS, S>
//   pass KV (MODE 1): stationary = the workgroup's keys (K^, V fragments); streamed queries: images Q~ and dO (+ LSE and
//                     delta per streamed query in LDS).     S = Q~.K^^T, dP = dO.V^T, dV^T += dO^T.(D P), dK^^T += Q~^T.dS
// (D = the forward's dropout factors, regenerated from the counter-based mask; ^ = L2-normalised, ~ = times log2e / tau).
// As in the forward, the accumulator of the first products is the B operand of the last ones (tokens in the permuted order
// the two tr-reads deliver).  The gradient through the normalisation is applied in the epilogue from the raw row.
// Every wave owns its output rows; dtau = one plain store per wave, summed in a fixed order: no atomics, bit-reproducible.
#include <cstdlib>
#include <type_traits>

#include "attn_fused.hpp"

#ifdef SEG3D_ATTN_STAMP
// Diagnostic build only (tools/probes/attn_bwd_stamps.py): per-wave s_memtime (100 MHz) spans of the backward passes:
// slots [4 MODE + {0 prologue up to the first barrier, 1 main loop, 2 epilogue, 3 streamed tiles}] of wave blockIdx.x * 4 + wave
__device__ unsigned long long* g_attn_bwd_stamp_buf = nullptr;
extern "C" int seg3d_debug_attn_bwd_stamps(void* buf) {
    unsigned long long* p = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_bwd_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : 2;
}
#define BSTAMP(var)                                     \
    do {                                                \
        __builtin_amdgcn_sched_barrier(0);              \
        var = __builtin_amdgcn_s_memtime();             \
        __builtin_amdgcn_sched_barrier(0);              \
    } while (0)
#else
#define BSTAMP(var) do {} while (0)
#endif

namespace {

using namespace attn;
using namespace attn_fused;

// gradient through x_hat = x / max(|x|, eps) for the rows a wave owns.  grad[b][r] = d(x_hat)[d = 16 b + 4 g + r] of
// token column c16; returns d(x) in place.  Every lane of the column takes part in the shuffles.
template <int DH, int NBQ>
__device__ __forceinline__ void through_normalise(const float* __restrict__ xrow, int g, f32x4* grad) {
    float xr[NBQ][4];
    float nrm = 0.f;
#pragma unroll
    for (int b = 0; b < NBQ; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = 16 * b + 4 * g + r;
            xr[b][r] = d < DH ? xrow[d] : 0.f;
            // accumulator rows past the head width were fed by whatever lies behind the head in the transposed reads
            // (a neighbouring head, row padding): they are never stored, and must not reach the projection either
            if (d >= DH) grad[b][r] = 0.f;
            nrm = fmaf(xr[b][r], xr[b][r], nrm);
        }
    nrm += __shfl_xor(nrm, 16, SEG3D_WAVE);
    nrm += __shfl_xor(nrm, 32, SEG3D_WAVE);
    const float len = sqrtf(nrm);
    const float rinv = 1.0f / fmaxf(len, kNormEps);
    float proj = 0.f;
#pragma unroll
    for (int b = 0; b < NBQ; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xr[b][r] *= rinv;  // x_hat
            proj = fmaf(xr[b][r], grad[b][r], proj);
        }
    proj += __shfl_xor(proj, 16, SEG3D_WAVE);
    proj += __shfl_xor(proj, 32, SEG3D_WAVE);
    const bool clamped = len < kNormEps;
#pragma unroll
    for (int b = 0; b < NBQ; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) grad[b][r] = clamped ? grad[b][r] * rinv : (grad[b][r] - xr[b][r] * proj) * rinv;
}

// The same from the workgroup's own staged image: x_hat = (hi + lo) * unscale of the stationary image (q~ = q_hat log2e / tau
// in pass Q, k_hat in pass KV; 16 - 17 significant bits, the precision of every product it enters) and rinv = 1 / max(|x|, eps)
// as the staging thread took it -- no second gather of the raw row behind the main loop, no sqrt / divide sequence.
// img_row = the row's hi plane at the head's first channel, `plane` bytes to the lo plane; DHS = stored channels.
template <int DH, int DHS, int NBQ>
__device__ __forceinline__ void through_normalise_lds(const char* img_row, int plane, float unscale, float rinv, int g, f32x4* grad) {
    float xr[NBQ][4];
    float proj = 0.f;
#pragma unroll
    for (int b = 0; b < NBQ; ++b) {
        const int d0 = 16 * b + 4 * g;
        uint32_t h0 = 0u, h1 = 0u, l0 = 0u, l1 = 0u;
        if (d0 < DHS) {
            const uint2 hh = *reinterpret_cast<const uint2*>(img_row + d0 * 2);
            const uint2 ll = *reinterpret_cast<const uint2*>(img_row + plane + d0 * 2);
            h0 = hh.x; h1 = hh.y; l0 = ll.x; l1 = ll.y;
        }
        xr[b][0] = (__builtin_bit_cast(float, h0 << 16) + __builtin_bit_cast(float, l0 << 16)) * unscale;
        xr[b][1] = (__builtin_bit_cast(float, h0 & 0xFFFF0000u) + __builtin_bit_cast(float, l0 & 0xFFFF0000u)) * unscale;
        xr[b][2] = (__builtin_bit_cast(float, h1 << 16) + __builtin_bit_cast(float, l1 << 16)) * unscale;
        xr[b][3] = (__builtin_bit_cast(float, h1 & 0xFFFF0000u) + __builtin_bit_cast(float, l1 & 0xFFFF0000u)) * unscale;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (d0 + r >= DH) {  // (stored padding channels are zero; accumulator rows past the head width carry a neighbour's products)
                xr[b][r] = 0.f;
                grad[b][r] = 0.f;
            }
            proj = fmaf(xr[b][r], grad[b][r], proj);
        }
    }
    proj += __shfl_xor(proj, 16, SEG3D_WAVE);
    proj += __shfl_xor(proj, 32, SEG3D_WAVE);
    const bool clamped = rinv > 0.99f / kNormEps;  // |x| < eps: x_hat = x / eps, d x = d x_hat / eps
#pragma unroll
    for (int b = 0; b < NBQ; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) grad[b][r] = clamped ? grad[b][r] * rinv : (grad[b][r] - xr[b][r] * proj) * rinv;
}

// waves per SIMD the register allocator must leave room for (spill-free points; pass KV holds two accumulator sets:
// at dh 48 that is one wave per SIMD, accumulators in AGPRs)
#ifndef SEG3D_BWD_Q_WAVES  // (A/B: resident waves per SIMD of pass Q at dh <= 24 / dh 48)
#define SEG3D_BWD_Q_WAVES 3
#endif
#ifndef SEG3D_BWD_Q48_WAVES
#define SEG3D_BWD_Q48_WAVES 2
#endif
template <int DH, int MODE>
constexpr int kBwdWaves = (DH <= 24 && MODE == 0) ? SEG3D_BWD_Q_WAVES : (DH == 48 && MODE == 1) ? 1 : (DH == 48) ? SEG3D_BWD_Q48_WAVES : 2;

template <int DH, int MODE>
__global__ __launch_bounds__(256, (kBwdWaves<DH, MODE>)) void attn_fused_bwd(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ldq, int ldk, int ldv,
    const float* __restrict__ out, const float* __restrict__ dout, const float* __restrict__ lse,
    const int32_t* __restrict__ tok, const int32_t* __restrict__ win_start, const int32_t* __restrict__ win_count,
    const int4* __restrict__ items, int n_items, int heads, const float* __restrict__ tau, float tau_min, float* __restrict__ dq,
    int lddq, float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv, float* __restrict__ tau_part,
    float* __restrict__ delta_buf, DropoutParams drop, int xcd_block) {
    using C = Cfg<DH>;
    constexpr int HG = C::HG, QT = C::QT, DHS = C::DHS, KS = C::KS, VW = C::VW;
    constexpr int KRS = C::KRS, VRS = C::VRS, CT = C::CT;
    constexpr int NBQ = (DH + 15) / 16;  // 16-row d-blocks of a gradient
    // the spare channel of the score product (attn_common.hpp: kSpareOne / kSpareMask): channel DHS = fragment SP_KS,
    // lane group SP_G, element 0
    constexpr int SP_KS = DHS / 32, SP_G = (DHS % 32) / 8;
    static_assert(KS * 32 > DHS, "the score product needs a spare K channel");
    static_assert(C::UW == 1, "one (tile, head) unit per wave");
    // stationary side (this workgroup's own tokens): two images [row][head][DHS] (hi | lo planes each) that all 256 threads
    // fill with whole 16-B pieces of the rows -- as the forward stages its queries -- plus token / LSE / delta per row
    constexpr int SROWS = QT * 32;
    constexpr int SRS = HG * DHS * 2 + (DH == 48 ? 0 : 16);  // bytes per row per plane
    constexpr int kSPlane = SROWS * SRS;
    constexpr int kStreamBytes = C::NBUF * C::kTile + (MODE == 1 ? C::NBUF * (2 * HG * 32 + HG * 16) * 4 : 0);
    constexpr int kStatBytes = 4 * kSPlane + SROWS * 4 + 3 * SROWS * HG * 4;
    __shared__ __attribute__((aligned(16))) char lds[kStreamBytes + kStatBytes];
    char* const sa_lds = lds + kStreamBytes;            // image A: Q~ (MODE 0) / K^ (MODE 1)
    char* const sb_lds = sa_lds + 2 * kSPlane;          // image B: dO (MODE 0) / V (MODE 1)
    int32_t* const stok_lds = reinterpret_cast<int32_t*>(sb_lds + 2 * kSPlane);
    float* const slse_lds = reinterpret_cast<float*>(stok_lds + SROWS);  // MODE 0: [row][head] log2-domain LSE, then delta
    float* const sdel_lds = slse_lds + SROWS * HG;
    float* const snrm_lds = sdel_lds + SROWS * HG;      // [row][head] 1 / max(|x|, eps) of the stationary image-A rows (epilogue)
    float* ld_lds = reinterpret_cast<float*>(lds + C::NBUF * C::kTile);  // MODE 1: [buf][L | delta][head][32 tokens]
    // MODE 1: dropout hash state of the streamed query pairs, [buf][head][16 pairs] (attn_dropout.hpp: the two-round part of
    // the hash, once per query pair and tile instead of once per 2 x 2 block and lane)
    uint32_t* rs_lds = reinterpret_cast<uint32_t*>(ld_lds + C::NBUF * 2 * HG * 32);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
#ifdef SEG3D_ATTN_STAMP
    unsigned long long st_t0 = 0, st_t1 = 0, st_t2 = 0, st_t3 = 0;
    unsigned long long st_la = 0, st_lb = 0, st_lc = 0, st_lx = 0, st_compute = 0, st_wait = 0, st_store = 0, st_barrier = 0;
    BSTAMP(st_t0);
#endif
    if (tid < C::NBUF * 2 * 4)  // the zero block behind each plane of each streamed-tile buffer (never written again)
        *reinterpret_cast<uint32_t*>(lds + (tid >> 3) * C::kTile + ((tid >> 2) & 1) * C::kPlane + C::kPlaneData + (tid & 3) * 4) = 0u;
    // Block -> (item, head group): the head groups of one item run side by side on ONE XCD (blocks are dealt round-robin
    // over the 8 XCDs, so block % 8 labels the blocks that share an L2): a head's slice of a token row is a fraction of a
    // cache line, and the groups would otherwise pull the same lines over the fabric once per XCD (see attention_fused.hip)
    constexpr int XG = 8;
    const int hgn = heads / HG;
    const int gx = (int)blockIdx.x % XG, gu = (int)blockIdx.x / XG;
    const int item_i = xcd_block_item(gu / hgn, gx, XG, xcd_block);  // blocks of consecutive items per XCD (attn_fused.hpp)
    if (item_i >= n_items) {  // (padding of the last group)
        if (MODE == 0 && tau_part && lane == 0) tau_part[(size_t)blockIdx.x * 4 + wave] = 0.f;
        return;
    }
    const int4 item = items[item_i];  // {window, tile / chunk, first token slot, tokens}: one round trip less in front of the gathers
    const int n = item.w, start = item.z;
    const int n_t = (n + 31) >> 5;  // 32-token tiles of the window (streamed and stationary alike)
    const int h0 = (gu % hgn) * HG;
    const int c_all = heads * DH;
    const float tau_c = fmaxf(tau[0], tau_min);
    const float qscale = kLog2e / tau_c;

    // ---------------------------------------------------------------- staging role of this thread
    const int st_which = tid >> 7;  // 0: image A (normalised: K^ or Q~), 1: image B (V or dO)
    const int st_key = (tid & 127) >> 2, st_part = tid & 3;
    const float* st_src = MODE == 0 ? (st_which == 0 ? k : v) : (st_which == 0 ? q : dout);
    const int st_ld = MODE == 0 ? (st_which == 0 ? ldk : ldv) : (st_which == 0 ? ldq : c_all);
    const int st_col = C::kNarrow ? (h0 + C::HPT * st_part) * DH : h0 * DH + st_part * CT;
    const float st_scale = MODE == 0 ? 1.0f : qscale;
    // Dropout's 1 / keep factor rides on dO (image B of pass KV, the stationary dO image of pass Q): dP' = dP / keep and
    // dV = (keep-masked P)^T dO', so the elementwise part of both passes only SELECTS (no per-element factor)
    const float st_bscale = (MODE == 1 && drop.threshold) ? drop.inv_keep : 1.0f;
    float st_reg[CT];
    float st_lse = 0.f, st_delta = 0.f;
    auto load_tok = [&](int t) {
        int kk = t * 32 + st_key;
        kk = kk < n ? kk : n - 1;  // clamped rows are finite and masked by p = 0
        return tok[start + kk];
    };
    auto load_row = [&](const float* row, float* dst) {
        if constexpr ((CT * 4) % 16 == 0 && (C::kNarrow ? (C::HPT * DH * 4) % 16 == 0 : (DH * 4) % 16 == 0)) {
#pragma unroll
            for (int i = 0; i < CT / 4; ++i) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(row + 4 * i);
                dst[4 * i] = x[0]; dst[4 * i + 1] = x[1]; dst[4 * i + 2] = x[2]; dst[4 * i + 3] = x[3];
            }
        } else {
#pragma unroll
            for (int i = 0; i < CT / 2; ++i) {
                const f32x2 x = *reinterpret_cast<const f32x2*>(row + 2 * i);
                dst[2 * i] = x[0]; dst[2 * i + 1] = x[1];
            }
        }
    };
    auto stage_load = [&](int32_t token_row) {
        load_row(st_src + (int64_t)token_row * st_ld + st_col, st_reg);
        if constexpr (MODE == 1) {
            if (st_which == 1) {  // delta = <dO, O> (written per (token, head) by pass Q) and the LSE of the streamed query
                st_delta = delta_buf[(int64_t)token_row * heads + (C::kNarrow ? h0 + st_part : h0)];
                st_lse = lse[(int64_t)token_row * heads + (C::kNarrow ? h0 + st_part : h0)];
            }
        }
    };
    auto stage_store = [&](int buf, int t_of) {
        char* base = lds + buf * C::kTile;
        if (st_which == 0) {  // image A: L2-normalise per head (x scale), split, row-major [token][head][DHS]
            char* dst = base + st_key * KRS;
            if constexpr (C::kNarrow) {
                float ss = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) ss = fmaf(st_reg[d], st_reg[d], ss);
                const float r = st_scale * inv_norm(ss);
                uint32_t hi[DHS / 2], lo[DHS / 2];
#pragma unroll
                for (int i = 0; i < DHS / 2; ++i) {
                    const float a = 2 * i < DH ? st_reg[2 * i] * r : 0.f;
                    const float b = 2 * i + 1 < DH ? st_reg[2 * i + 1] * r : 0.f;
                    split2(a, b, &hi[i], &lo[i]);
                }
                char* p = dst + (st_part * DHS) * 2;
#pragma unroll
                for (int i = 0; i < DHS / 8; ++i) {
                    *reinterpret_cast<u32x4*>(p + 16 * i) = (u32x4){hi[4 * i], hi[4 * i + 1], hi[4 * i + 2], hi[4 * i + 3]};
                    *reinterpret_cast<u32x4*>(p + C::kPlane + 16 * i) = (u32x4){lo[4 * i], lo[4 * i + 1], lo[4 * i + 2], lo[4 * i + 3]};
                }
            } else {
                float ss = 0.f;
#pragma unroll
                for (int d = 0; d < CT; ++d) ss = fmaf(st_reg[d], st_reg[d], ss);
                ss = quad_sum(ss);
                const float r = st_scale * inv_norm(ss);
                char* p = dst + (st_part * CT) * 2;
#pragma unroll
                for (int i = 0; i < CT / 2; ++i) {
                    uint32_t hi, lo;
                    split2(st_reg[2 * i] * r, st_reg[2 * i + 1] * r, &hi, &lo);
                    *reinterpret_cast<uint32_t*>(p + 4 * i) = hi;
                    *reinterpret_cast<uint32_t*>(p + C::kPlane + 4 * i) = lo;
                }
            }
        } else {  // image B: split only, [token][head][VW]; channels DH .. VW-1 are zero
            char* dst = base + 32 * KRS + st_key * VRS;
            if constexpr (C::kNarrow) {
                uint32_t hi[VW / 2], lo[VW / 2];
#pragma unroll
                for (int i = 0; i < VW / 2; ++i) {
                    const float a = 2 * i < DH ? st_reg[2 * i] * st_bscale : 0.f;
                    const float b = 2 * i + 1 < DH ? st_reg[2 * i + 1] * st_bscale : 0.f;
                    split2(a, b, &hi[i], &lo[i]);
                }
                char* p = dst + (st_part * VW) * 2;
#pragma unroll
                for (int i = 0; i < VW / 8; ++i) {
                    *reinterpret_cast<u32x4*>(p + 16 * i) = (u32x4){hi[4 * i], hi[4 * i + 1], hi[4 * i + 2], hi[4 * i + 3]};
                    *reinterpret_cast<u32x4*>(p + C::kPlane + 16 * i) = (u32x4){lo[4 * i], lo[4 * i + 1], lo[4 * i + 2], lo[4 * i + 3]};
                }
            } else {
                char* p = dst + (st_part * CT) * 2;
#pragma unroll
                for (int i = 0; i < CT / 2; ++i) {
                    uint32_t hi, lo;
                    split2(st_reg[2 * i] * st_bscale, st_reg[2 * i + 1] * st_bscale, &hi, &lo);
                    *reinterpret_cast<uint32_t*>(p + 4 * i) = hi;
                    *reinterpret_cast<uint32_t*>(p + C::kPlane + 4 * i) = lo;
                }
                if (VW > DH && st_part == 3) {
#pragma unroll
                    for (int i = 0; i < (VW - DH) / 8; ++i) {
                        *reinterpret_cast<u32x4*>(dst + DH * 2 + 16 * i) = (u32x4){0u, 0u, 0u, 0u};
                        *reinterpret_cast<u32x4*>(dst + C::kPlane + DH * 2 + 16 * i) = (u32x4){0u, 0u, 0u, 0u};
                    }
                }
            }
            if constexpr (MODE == 1) {
                const float dsum = st_delta;
                const int hh = C::kNarrow ? st_part : 0;
                if (C::kNarrow || st_part == 0) {
                    float* ld = ld_lds + buf * 2 * HG * 32;
                    // a streamed query past the window's end (a clamped copy of the last row) gets an LSE no score reaches:
                    // p = exp2(s - 3e38) = 0 exactly, so its dS and P columns vanish without a per-element mask
                    ld[hh * 32 + st_key] = t_of * 32 + st_key < n ? st_lse * kLog2e : 3.0e38f;
                    ld[HG * 32 + hh * 32 + st_key] = dsum;
                    if (drop.threshold && (st_key & 1) == 0)
                        rs_lds[(buf * HG + hh) * 16 + (st_key >> 1)] =
                            dropout_row_state(dropout_head_state(drop, item.x, h0 + hh), t_of * 32 + st_key);
                }
            }
        }
    };

    // first streamed tile's gather in flight before the stationary prologue (see attention_fused.hip)
    int32_t tok_next = n_t > 1 ? load_tok(1) : 0;
    stage_load(load_tok(0));

    // ---------------------------------------------------------------- this wave's (tile, head) unit: stationary side
    const int n_st_here = min(QT, n_t - item.y * QT);
    const int unit = wave;
    const int st_tile = C::kNarrow ? 0 : unit;
    const int hh = C::kNarrow ? unit : 0;
    const int h = h0 + hh;
    const int s0 = (item.y * QT + st_tile) * 32;  // first stationary token of this wave
    const bool active = st_tile < n_st_here;      // wave-uniform
    const bool two = n - s0 > 16;                 // the tile's second 16-token group exists (wave-uniform)
    bf16x8 a_hi[2][KS], a_lo[2][KS], b_hi[2][KS], b_lo[2][KS];  // stationary fragments: MODE 0 (Q~, dO), MODE 1 (K^, V)
    // dropout, the stationary token's share of the hash: MODE 0 the query pair's state, MODE 1 the key pair's term
    uint32_t drop_st[2] = {0u, 0u};
    if (drop.threshold) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            drop_st[j] = MODE == 0 ? dropout_row_state(dropout_head_state(drop, item.x, h), s0 + 16 * j + c16)
                                   : dropout_key_term(s0 + 16 * j + c16);
    }
    float lq[2] = {0.f, 0.f}, dl[2] = {0.f, 0.f};                // MODE 0: log2-domain LSE and delta of the lane's queries
    int32_t token[2] = {-1, -1};
    // Stationary rows -> LDS.  Narrow heads: 32 rows x 4 heads, threads 0 .. 127 (one head of one row each); wide heads: up to
    // 128 rows of one head, two passes of 64 rows x 4 threads (a quarter of the head each).  The earlier prologue had every
    // lane fetch its fragment-shaped 8-B pieces straight from global memory (12-24 scattered loads per lane and tensor, the
    // normalisation repeated by the four lanes of a row): 25-40 % of a workgroup's life (tools/probes/attn_stamps.py).
    {
        constexpr int SP = C::kNarrow ? 1 : 2;
        const bool s_role = C::kNarrow ? tid < 128 : true;
        const int s_row0 = C::kNarrow ? (tid & 127) >> 2 : tid >> 2;
        const int n_srows = min(SROWS, n - item.y * SROWS);
        const float* a_src = MODE == 0 ? q : k;
        const int a_ld = MODE == 0 ? ldq : ldk;
        const float* b_src = MODE == 0 ? dout : v;
        const int b_ld = MODE == 0 ? c_all : ldv;
        const float sb_scale = (MODE == 0 && drop.threshold) ? drop.inv_keep : 1.0f;  // dO' = dO / keep (delta below takes the raw rows)
        float ra[SP][CT], rb[SP][CT], ro[MODE == 0 ? SP : 1][CT];
        int32_t stok[SP];
        float slse[SP];
#pragma unroll
        for (int p = 0; p < SP; ++p) {
            stok[p] = 0;
            slse[p] = 0.f;
            if (s_role && (p == 0 || n_srows > 64)) {
                const int si = item.y * SROWS + s_row0 + 64 * p;
                stok[p] = tok[start + (si < n ? si : n - 1)];
            }
        }
#pragma unroll
        for (int p = 0; p < SP; ++p)
            if (s_role && (p == 0 || n_srows > 64)) {
                load_row(a_src + (int64_t)stok[p] * a_ld + st_col, ra[p]);
                load_row(b_src + (int64_t)stok[p] * b_ld + st_col, rb[p]);
                if constexpr (MODE == 0) {
                    load_row(out + (int64_t)stok[p] * c_all + st_col, ro[p]);
                    slse[p] = lse[(int64_t)stok[p] * heads + (C::kNarrow ? h0 + st_part : h0)];
                }
            }
#pragma unroll
        for (int p = 0; p < SP; ++p)
            if (s_role && (p == 0 || n_srows > 64)) {
                const int row = s_row0 + 64 * p;
                const bool have = item.y * SROWS + row < n;
                // image A: normalised (x qscale for queries), split
                char* da = sa_lds + row * SRS;
                char* db = sb_lds + row * SRS;
                if constexpr (C::kNarrow) {
                    float ss = 0.f;
#pragma unroll
                    for (int d = 0; d < DH; ++d) ss = fmaf(ra[p][d], ra[p][d], ss);
                    const float rn = inv_norm(ss);
                    const float r = (MODE == 0 ? qscale : 1.0f) * rn;
                    snrm_lds[row * HG + st_part] = rn;
                    uint32_t hi[DHS / 2], lo[DHS / 2], bhi[DHS / 2], blo[DHS / 2];
#pragma unroll
                    for (int i = 0; i < DHS / 2; ++i) {
                        const float a0 = 2 * i < DH ? ra[p][2 * i] * r : 0.f, a1 = 2 * i + 1 < DH ? ra[p][2 * i + 1] * r : 0.f;
                        const float b0 = 2 * i < DH ? rb[p][2 * i] * sb_scale : 0.f, b1 = 2 * i + 1 < DH ? rb[p][2 * i + 1] * sb_scale : 0.f;
                        split2(a0, a1, &hi[i], &lo[i]);
                        split2(b0, b1, &bhi[i], &blo[i]);
                    }
#pragma unroll
                    for (int i = 0; i < DHS / 8; ++i) {
                        char* pa = da + (st_part * DHS) * 2 + 16 * i;
                        char* pb = db + (st_part * DHS) * 2 + 16 * i;
                        *reinterpret_cast<u32x4*>(pa) = (u32x4){hi[4 * i], hi[4 * i + 1], hi[4 * i + 2], hi[4 * i + 3]};
                        *reinterpret_cast<u32x4*>(pa + kSPlane) = (u32x4){lo[4 * i], lo[4 * i + 1], lo[4 * i + 2], lo[4 * i + 3]};
                        *reinterpret_cast<u32x4*>(pb) = (u32x4){bhi[4 * i], bhi[4 * i + 1], bhi[4 * i + 2], bhi[4 * i + 3]};
                        *reinterpret_cast<u32x4*>(pb + kSPlane) = (u32x4){blo[4 * i], blo[4 * i + 1], blo[4 * i + 2], blo[4 * i + 3]};
                    }
                } else {
                    float ss = 0.f;
#pragma unroll
                    for (int d = 0; d < CT; ++d) ss = fmaf(ra[p][d], ra[p][d], ss);
                    ss = quad_sum(ss);
                    const float rn = inv_norm(ss);
                    const float r = (MODE == 0 ? qscale : 1.0f) * rn;
                    if (st_part == 0) snrm_lds[row] = rn;
#pragma unroll
                    for (int i = 0; i < CT / 2; ++i) {
                        uint32_t hi, lo, bhi, blo;
                        split2(ra[p][2 * i] * r, ra[p][2 * i + 1] * r, &hi, &lo);
                        split2(rb[p][2 * i] * sb_scale, rb[p][2 * i + 1] * sb_scale, &bhi, &blo);
                        *reinterpret_cast<uint32_t*>(da + (st_part * CT) * 2 + 4 * i) = hi;
                        *reinterpret_cast<uint32_t*>(da + kSPlane + (st_part * CT) * 2 + 4 * i) = lo;
                        *reinterpret_cast<uint32_t*>(db + (st_part * CT) * 2 + 4 * i) = bhi;
                        *reinterpret_cast<uint32_t*>(db + kSPlane + (st_part * CT) * 2 + 4 * i) = blo;
                    }
                }
                if (C::kNarrow ? st_part == 0 : st_part == 0) stok_lds[row] = have ? stok[p] : -1;
                if constexpr (MODE == 0) {
                    float dsum = 0.f;
#pragma unroll
                    for (int d = 0; d < CT; ++d) dsum = fmaf(rb[p][d], ro[p][d], dsum);
                    if (!C::kNarrow) dsum = quad_sum(dsum);
                    const int hs = C::kNarrow ? st_part : 0;
                    if (C::kNarrow || st_part == 0) {
                        slse_lds[row * HG + hs] = slse[p] * kLog2e;
                        sdel_lds[row * HG + hs] = dsum;
                        if (have) delta_buf[(int64_t)stok[p] * heads + h0 + hs] = dsum;  // pass KV reads it instead of the O rows
                    }
                }
            }
    }

    // ---------------------------------------------------------------- fragment reads of a staged tile
    // row fragment (A operand: token u * 16 + c16, channels 32 s + 8 g .. + 7) of image A / image B
    // (lanes whose channels lie past the stored width read the plane's zero block, attn_fused.hpp: no branch, no register fill)
    auto read_arow = [&](const char* base, int u, int s, bf16x8* hi, bf16x8* lo) {
        const int c0 = 32 * s + 8 * g;
        const char* p = base + (c0 < DHS ? (u * 16 + c16) * KRS + (hh * DHS + c0) * 2 : C::kPlaneData);
        *hi = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
        *lo = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + C::kPlane));
    };
    auto read_brow = [&](const char* base, int u, int s, bf16x8* hi, bf16x8* lo) {
        const int c0 = 32 * s + 8 * g;
        const char* p = base + (c0 < VW ? 32 * KRS + (u * 16 + c16) * VRS + (hh * VW + c0) * 2 : C::kPlaneData);
        *hi = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
        *lo = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + C::kPlane));
    };
    // transposed fragment of d-block b (A operand: row d = 16 b + c16, token slots 8 g .. 8 g + 7 = tokens 4g..4g+3,
    // 16+4g..16+4g+3) out of a row-major image: two ds_read_b64_tr_b16 per plane, every lane takes part
    auto read_tr = [&](const char* img, int rs, int col0, bf16x8* hi, bf16x8* lo) {
        const int qq = c16 >> 2, pp = c16 & 3;
        const char* p0 = img + (4 * g + qq) * rs + (col0 + 4 * pp) * 2;
        const char* p1 = p0 + 16 * rs;
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + C::kPlane));
        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1 + C::kPlane));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        *hi = __builtin_bit_cast(bf16x8, (s16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]});
        *lo = __builtin_bit_cast(bf16x8, (s16x8){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]});
    };

    f32x4 acc0[2][NBQ], acc1[MODE == 1 ? 2 : 1][NBQ];  // MODE 0: dQ^^T; MODE 1: dK^^T (acc0) and dV^T (acc1)
    f32x2 tau2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int b = 0; b < NBQ; ++b) {
            acc0[j][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (MODE == 1) acc1[j][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    // dropout: the parity of the lane's stationary token picks a byte pair of every block hash (attn_dropout.hpp)
    const uint32_t drop_shift = MODE == 0 ? dropout_lane_shift_query(c16 & 1) : dropout_lane_shift_key(c16 & 1);
    // the spare score channel of a streamed key past the window's end (pass Q; lanes of group SP_G only)
    const uint32_t spare_mask = g == SP_G ? kSpareMask : 0u;

    // ---------------------------------------------------------------- one streamed tile
    // The elementwise part works on float PAIRS (the two accumulator registers of one hash block): packed subtract /
    // multiply / fma (v_pk_*_f32: two lanes' worth per issue slot), dropout as a select on dP and P, no masks -- tokens past
    // the window's end are switched off inside the score product (pass Q: spare channel) or by their LSE (pass KV).
    auto tile_step = [&](int t, int buf, auto drop_tag) {
        constexpr bool DROP = decltype(drop_tag)::value;
        const char* base = lds + buf * C::kTile;
        const bool last = t + 1 == n_t;
        const bool u_two = n - t * 32 > 16;  // the tile's second 16-token half holds tokens (false on a window's last tile only)
        bf16x8 ra_hi[2][KS], ra_lo[2][KS], rb_hi[2][KS], rb_lo[2][KS];  // row fragments of images A and B
        bf16x8 ta_hi[NBQ], ta_lo[NBQ], tb_hi[MODE == 1 ? NBQ : 1], tb_lo[MODE == 1 ? NBQ : 1];  // transposed fragments
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                read_arow(base, u, s, &ra_hi[u][s], &ra_lo[u][s]);
                read_brow(base, u, s, &rb_hi[u][s], &rb_lo[u][s]);
            }
        if constexpr (MODE == 0) {
            if (last) {  // (wave-uniform) keys past the end: -16384 in the spare channel, against the queries' 1.0
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    u32x4 w = __builtin_bit_cast(u32x4, ra_hi[u][SP_KS]);
                    w[0] |= t * 32 + u * 16 + c16 >= n ? spare_mask : 0u;
                    ra_hi[u][SP_KS] = __builtin_bit_cast(bf16x8, w);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < NBQ; ++b) {
            read_tr(base, KRS, hh * DHS + 16 * b, &ta_hi[b], &ta_lo[b]);
            if (MODE == 1) read_tr(base + 32 * KRS, VRS, hh * VW + 16 * b, &tb_hi[b], &tb_lo[b]);
        }
        f32x4 l4[2], d4[2];
        if constexpr (MODE == 1) {
            const float* ld = ld_lds + buf * 2 * HG * 32;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                l4[u] = *reinterpret_cast<const f32x4*>(ld + hh * 32 + u * 16 + 4 * g);
                d4[u] = *reinterpret_cast<const f32x4*>(ld + HG * 32 + hh * 32 + u * 16 + 4 * g);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && !two) break;
            f32x2 pv[4], dsv[4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !u_two) {  // (wave-uniform) the streamed tile's second 16-token half lies past the window's end
                    dsv[2] = dsv[3] = pv[2] = pv[3] = (f32x2){0.f, 0.f};
                    continue;
                }
                f32x4 s_acc = {0.f, 0.f, 0.f, 0.f}, p_acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    // MODE 0: S^T[key][query] = K^.Q~^T, dP^T = V.dO^T;  MODE 1: S[query][key] = Q~.K^^T, dP = dO.V^T
                    s_acc = mfma3(ra_hi[u][s], ra_lo[u][s], a_hi[j][s], a_lo[j][s], s_acc);
                    p_acc = mfma3(rb_hi[u][s], rb_lo[u][s], b_hi[j][s], b_lo[j][s], p_acc);
                }
                uint32_t adj[2] = {0u, 0u};
                if constexpr (DROP) {  // the forward's dropout mask, regenerated
                    // streamed tokens 4g .. 4g+3 of this 16-token half = two 2 x 2 blocks shared with lane c16 ^ 1 (same
                    // stationary pair): the even lane hashes the first, the odd lane the second, one DPP swap
                    const bool odd = c16 & 1;
                    const int sm0 = t * 32 + u * 16 + 4 * g;
                    uint32_t mine;
                    if constexpr (MODE == 0) mine = dropout_block_bits(drop_st[j], dropout_key_term(sm0 + (odd ? 2 : 0)));
                    else mine = dropout_block_bits(rs_lds[(buf * HG + hh) * 16 + u * 8 + 2 * g + (odd ? 1 : 0)], drop_st[j]);
                    uint32_t bits[2];
                    dropout_pair_bits(mine, odd, &bits[0], &bits[1]);
                    adj[0] = bits[0] >> drop_shift;
                    adj[1] = bits[1] >> drop_shift;
                }
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    const f32x2 s2 = {s_acc[2 * r2], s_acc[2 * r2 + 1]};
                    f32x2 dp2 = {p_acc[2 * r2], p_acc[2 * r2 + 1]};
                    f32x2 e2, d2;
                    if constexpr (MODE == 0) {  // the lane's own query: scalar operands (no register pair spent on a broadcast)
                        e2 = (f32x2){s2[0] - lq[j], s2[1] - lq[j]};
                        d2 = (f32x2){dl[j], dl[j]};
                    } else {
                        e2 = s2 - (f32x2){l4[u][2 * r2], l4[u][2 * r2 + 1]};
                        d2 = (f32x2){d4[u][2 * r2], d4[u][2 * r2 + 1]};
                    }
                    const f32x2 p2 = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
                    f32x2 pk2 = p2;
                    if constexpr (DROP) {
                        // the pair's second element is the next KEY (pass Q: byte 1) or the next QUERY (pass KV: byte 2)
                        const bool k0 = dropout_dropped_byte(drop, adj[r2], 0);
                        const bool k1 = dropout_dropped_byte(drop, adj[r2], MODE == 0 ? 1 : 2);
                        if (MODE == 0) {
                            dp2[0] = k0 ? 0.f : dp2[0];
                            dp2[1] = k1 ? 0.f : dp2[1];
                        } else {  // one select serves both products: dS = (D P) dP - P delta
                            pk2[0] = k0 ? 0.f : p2[0];
                            pk2[1] = k1 ? 0.f : p2[1];
                        }
                    }
                    f32x2 ds2;  // dS = P (D dP - delta)
                    if constexpr (MODE == 0) ds2 = p2 * (f32x2){dp2[0] - dl[j], dp2[1] - dl[j]};
                    else if constexpr (DROP) ds2 = __builtin_elementwise_fma(pk2, dp2, -(p2 * d2));
                    else ds2 = p2 * (dp2 - d2);
                    dsv[u * 2 + r2] = ds2;
                    if (MODE == 0) tau2[j] = __builtin_elementwise_fma(ds2, s2, tau2[j]);
                    else pv[u * 2 + r2] = pk2;          // dV = (D P)^T dO
                }
            }
            bf16x8 ds_hi, ds_lo;
            split_frag2(dsv, &ds_hi, &ds_lo);
#pragma unroll
            for (int b = 0; b < NBQ; ++b) acc0[j][b] = mfma3(ta_hi[b], ta_lo[b], ds_hi, ds_lo, acc0[j][b]);
            if constexpr (MODE == 1) {
                bf16x8 p_hi, p_lo;
                split_frag2(pv, &p_hi, &p_lo);
#pragma unroll
                for (int b = 0; b < NBQ; ++b) acc1[j][b] = mfma3(tb_hi[b], tb_lo[b], p_hi, p_lo, acc1[j][b]);
            }
        }
    };

    // ---------------------------------------------------------------- main loop over the window's streamed tiles
    {
        stage_store(0, 0);
        __syncthreads();
        BSTAMP(st_t1);
        // this wave's stationary fragments (row fragments of both images), token, LSE and delta of its rows
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool have_j = active && (j == 0 || two);
            const int row = st_tile * 32 + 16 * j + c16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c0 = 32 * ks + 8 * g;
                u32x4 ah = {0u, 0u, 0u, 0u}, al = ah, bh = ah, bl = ah;
                if (have_j && c0 < DHS) {
                    const char* pa = sa_lds + row * SRS + (hh * DHS + c0) * 2;
                    const char* pb = sb_lds + row * SRS + (hh * DHS + c0) * 2;
                    ah = *reinterpret_cast<const u32x4*>(pa);
                    al = *reinterpret_cast<const u32x4*>(pa + kSPlane);
                    bh = *reinterpret_cast<const u32x4*>(pb);
                    bl = *reinterpret_cast<const u32x4*>(pb + kSPlane);
                }
                if (MODE == 0 && ks == SP_KS && g == SP_G) ah[0] = kSpareOne;  // the queries' 1.0 in the spare score channel
                a_hi[j][ks] = __builtin_bit_cast(bf16x8, ah);
                a_lo[j][ks] = __builtin_bit_cast(bf16x8, al);
                b_hi[j][ks] = __builtin_bit_cast(bf16x8, bh);
                b_lo[j][ks] = __builtin_bit_cast(bf16x8, bl);
            }
            if (have_j) {
                token[j] = stok_lds[row];
                if constexpr (MODE == 0) {
                    lq[j] = slse_lds[row * HG + hh];
                    dl[j] = sdel_lds[row * HG + hh];
                }
            }
        }
        auto main_loop = [&](auto drop_tag) {
            for (int t = 0; t < n_t; ++t) {
                const bool more = t + 1 < n_t;
                const int buf = C::NBUF == 2 ? (t & 1) : 0;
                if (more) {
                    stage_load(tok_next);
                    if (t + 2 < n_t) tok_next = load_tok(t + 2);
                }
                BSTAMP(st_la);
                if (active) tile_step(t, buf, drop_tag);
                BSTAMP(st_lb);
                if (C::NBUF == 1) __syncthreads();
#ifdef SEG3D_ATTN_STAMP
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (diagnostic build) the wait for the next tile's rows, apart
#endif
                BSTAMP(st_lc);
                if (more) stage_store(C::NBUF == 2 ? (buf ^ 1) : 0, t + 1);
                BSTAMP(st_lx);
                __syncthreads();
#ifdef SEG3D_ATTN_STAMP
                {
                    unsigned long long st_le;
                    BSTAMP(st_le);
                    st_compute += st_lb - st_la;
                    st_wait += st_lc - st_lb;
                    st_store += st_lx - st_lc;
                    st_barrier += st_le - st_lx;
                }
#endif
            }
        };
        if (drop.threshold) main_loop(std::true_type{});
        else main_loop(std::false_type{});
        BSTAMP(st_t2);
    }

    // ---------------------------------------------------------------- epilogue
    float* tau_slot = tau_part ? tau_part + (size_t)blockIdx.x * 4 + wave : nullptr;
    if (!active) {
        if (MODE == 0 && lane == 0) *tau_slot = 0.f;
        return;
    }
    float tau_sum = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j == 1 && !two) break;
        const bool valid = token[j] >= 0;
        const int32_t trow = token[j];
        const int srow = st_tile * 32 + 16 * j + c16;  // the lane's row of the stationary images (all rows were staged)
        const char* img_row = sa_lds + srow * SRS + hh * DHS * 2;
        const float rinv = snrm_lds[srow * HG + hh];
        if constexpr (MODE == 0) {
            const float inv_tau = 1.0f / tau_c;
#pragma unroll
            for (int b = 0; b < NBQ; ++b) acc0[j][b] = acc0[j][b] * inv_tau;
            if (valid) tau_sum += tau2[j][0] + tau2[j][1];
            through_normalise_lds<DH, DHS, NBQ>(img_row, kSPlane, 1.0f / qscale, rinv, g, acc0[j]);
        } else {
#pragma unroll
            for (int b = 0; b < NBQ; ++b) acc0[j][b] = acc0[j][b] * kLn2;  // Q~ = q_hat log2e / tau  ->  q_hat / tau = Q~ ln2
            through_normalise_lds<DH, DHS, NBQ>(img_row, kSPlane, 1.0f, rinv, g, acc0[j]);
        }
        if (!valid) continue;
        float* o0 = MODE == 0 ? dq + (int64_t)trow * lddq + h * DH : dk + (int64_t)trow * lddk + h * DH;
        float* o1 = MODE == 1 ? dv + (int64_t)trow * lddv + h * DH : nullptr;
#pragma unroll
        for (int b = 0; b < NBQ; ++b) {
            const int d = 16 * b + 4 * g;
            if (DH % 4 == 0) {
                if (d < DH) {
                    *reinterpret_cast<f32x4*>(o0 + d) = acc0[j][b];
                    if (MODE == 1) *reinterpret_cast<f32x4*>(o1 + d) = acc1[j][b];
                }
            } else {
                if (d + 1 < DH) {
                    *reinterpret_cast<f32x2*>(o0 + d) = (f32x2){acc0[j][b][0], acc0[j][b][1]};
                    if (MODE == 1) *reinterpret_cast<f32x2*>(o1 + d) = (f32x2){acc1[j][b][0], acc1[j][b][1]};
                }
                if (d + 3 < DH) {
                    *reinterpret_cast<f32x2*>(o0 + d + 2) = (f32x2){acc0[j][b][2], acc0[j][b][3]};
                    if (MODE == 1) *reinterpret_cast<f32x2*>(o1 + d + 2) = (f32x2){acc1[j][b][2], acc1[j][b][3]};
                }
            }
        }
    }
    if constexpr (MODE == 0) {
        // d/dtau: s_nat = s2 ln2 = c / tau  ->  dL/dtau = -sum(ds s_nat) / tau   (zero while tau is clamped)
        for (int off = 32; off > 0; off >>= 1) tau_sum += __shfl_xor(tau_sum, off, SEG3D_WAVE);
        if (lane == 0) *tau_slot = tau[0] > tau_min ? -tau_sum * kLn2 / tau_c : 0.f;
    }
#ifdef SEG3D_ATTN_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BSTAMP(st_t3);
    if (g_attn_bwd_stamp_buf && lane == 0) {
        unsigned long long* o = g_attn_bwd_stamp_buf + ((size_t)blockIdx.x * 4 + wave) * 8 + 4 * MODE;
        o[0] = st_t1 - st_t0;
        o[1] = st_t2 - st_t1;
        o[2] = st_t3 - st_t2;
        // the tile loop by phase, in a second record block behind the first (8 x 4 x blocks words further)
        unsigned long long* o2 = g_attn_bwd_stamp_buf + ((size_t)gridDim.x * 4 + (size_t)blockIdx.x * 4 + wave) * 8 + 4 * MODE;
        o2[0] = st_compute;
        o2[1] = st_wait;
        o2[2] = st_store;
        o2[3] = st_barrier;
        o[3] = (unsigned long long)n_t;
    }
#endif
}

// dtau = sum of the per-wave partials in a fixed order
__global__ __launch_bounds__(1024) void tau_reduce_fused(const float* __restrict__ part, int count, float* __restrict__ dtau) {
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;  // independent chains: the loads overlap
    int i = threadIdx.x;
    for (; i + 3072 < count; i += 4096) {
        v0 += part[i];
        v1 += part[i + 1024];
        v2 += part[i + 2048];
        v3 += part[i + 3072];
    }
    for (; i < count; i += 1024) v0 += part[i];
    float v = (v0 + v1) + (v2 + v3);
    __shared__ float red[16];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, SEG3D_WAVE);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        dtau[0] = t;
    }
}

// blocks of a pass: whole groups of 8 items x head groups (the kernel's block -> (item, head group) map)
static size_t bwd_blocks(int n_items, int hgn, int xb) {
    const size_t per = (size_t)8 * xb;
    return (n_items + per - 1) / per * per * hgn;
}

template <int DH>
int launch(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const float* out, const float* dout,
           const float* lse, const int32_t* tok, const int32_t* win_start, const int32_t* win_count, const int4* tile_item,
           int n_tiles, const int4* chunk_item, int n_chunks, int heads, const float* tau, float tau_min, float* dq, float* dk,
           float* dv, int lddq, int lddk, int lddv, float* dtau, float* tau_part, float* delta_buf, const DropoutParams& drop,
           hipStream_t st) {
    using C = Cfg<DH>;
    const int4* items = C::kNarrow ? tile_item : chunk_item;
    const int n_items = C::kNarrow ? n_tiles : n_chunks;
    const int xb = xcd_block_items(C::kNarrow, n_items);
    const dim3 grid((unsigned)(bwd_blocks(n_items, heads / C::HG, xb)));
    hipLaunchKernelGGL((attn_fused_bwd<DH, 0>), grid, dim3(256), 0, st, q, k, v, ldq, ldk, ldv, out, dout, lse, tok, win_start,
                       win_count, items, n_items, heads, tau, tau_min, dq, lddq, dk, lddk, dv, lddv, tau_part, delta_buf, drop, xb);
    SEG3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(tau_reduce_fused, dim3(1), dim3(1024), 0, st, tau_part, (int)(grid.x * 4), dtau);
    SEG3D_CHECK_LAUNCH();
    hipLaunchKernelGGL((attn_fused_bwd<DH, 1>), grid, dim3(256), 0, st, q, k, v, ldq, ldk, ldv, out, dout, lse, tok, win_start,
                       win_count, items, n_items, heads, tau, tau_min, dq, lddq, dk, lddk, dv, lddv, nullptr, delta_buf, drop, xb);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // namespace

// dh 6 stays on the vector-ALU kernels (windows of ~15 voxels: 360 vs 475 us per layer on the headline scene).  dh 48:
// pass KV holds two 3-block accumulator sets + eight streamed fragments = 345 registers, i.e. ONE wave per SIMD (the
// accumulators in AGPRs), and is still ahead of the three-launch MFMA passes with their prepared-operand round trip
// (3 layers of the headline scene, dropout on: 1.58 -> 1.11 ms).  Measured per layer: dh 12 1540 -> 617 us, dh 24 780 -> 520 us
bool attn_fused_bwd_supported(int heads, int dh) {
    return (dh == 12 && heads % 4 == 0) || dh == 24 || dh == 48;
}

static size_t tau_part_bytes(int n_tiles, int n_chunks, int heads, int dh) {
    const size_t blocks = (dh <= 12) ? bwd_blocks(n_tiles, heads / 4, xcd_block_items(true, n_tiles)) : bwd_blocks(n_chunks, heads, xcd_block_items(false, n_chunks));
    return align_up(blocks * 4 * sizeof(float), 256);
}

// one dtau partial per wave of pass Q + delta = <dO, O> per (token, head), handed from pass Q to pass KV
size_t attn_fused_bwd_workspace_bytes(int64_t m, int n_tiles, int n_chunks, int heads, int dh) {
    return tau_part_bytes(n_tiles, n_chunks, heads, dh) + align_up((size_t)m * heads * sizeof(float), 256) + 256;
}

int attn_fused_bwd_launch(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const float* out,
                          const float* dout, const float* lse, const int32_t* tok, const int32_t* win_start,
                          const int32_t* win_count, const int32_t* tile_item, int n_tiles, const int32_t* chunk_item,
                          int n_chunks, int64_t m, int heads, int dh, const float* tau, float tau_min, float* dq, float* dk,
                          float* dv, int lddq, int lddk, int lddv, float* dtau, void* workspace, const DropoutParams& drop,
                          hipStream_t st) {
    (void)m;
    const int4* ti = reinterpret_cast<const int4*>(tile_item);
    const int4* ci = reinterpret_cast<const int4*>(chunk_item);
    float* tau_part = static_cast<float*>(workspace);
    float* delta_buf = reinterpret_cast<float*>(static_cast<char*>(workspace) + tau_part_bytes(n_tiles, n_chunks, heads, dh));
#define SEG3D_FB(D)                                                                                                        \
    case D:                                                                                                                \
        return launch<D>(q, k, v, ldq, ldk, ldv, out, dout, lse, tok, win_start, win_count, ti, n_tiles, ci, n_chunks, heads, \
                         tau, tau_min, dq, dk, dv, lddq, lddk, lddv, dtau, tau_part, delta_buf, drop, st)
    switch (dh) {
        SEG3D_FB(6);
        SEG3D_FB(12);
        SEG3D_FB(24);
        SEG3D_FB(48);
        default: return SEG3D_EINVAL;
    }
#undef SEG3D_FB
}
