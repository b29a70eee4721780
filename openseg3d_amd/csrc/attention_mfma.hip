// a19-a21 forward on the matrix cores: ragged sparse-window cosine attention with split-bf16 MFMA.
// Reference: flat2window -> CosineMultiheadAttention -> window2flat (swformer_utils.py:34-85,
// point_transformer_layer.py:233-258, cosine_msa.py:115-177) -- padded [W,T,C] tensors, -inf masks and a
// materialised (W*H, T, T) score tensor per encoder layer.
//
// Two launches per layer, no padding beyond 32-token tiles, no score tensor, no LDS in the core:
//  1. attn_prepare_fwd: one workgroup per 32-token tile of a window.  Gathers the q/k/v rows of the tile
//     (coalesced whole rows), L2-normalises q and k per head, folds log2(e)/max(tau, tau_min) into q, splits
//     every value into bf16 hi + lo and writes
//        Qp, Kp : [padded position][head][DHS]   (row-major; an MFMA A/B fragment = one 16-B load)
//        Vt     : [head][d][padded position]     (transposed, positions permuted inside a tile so that the
//                                                 score accumulator of the QK^T MFMAs is the B operand of PV)
//  2. attn_core_fwd: one wave per (window, 32-query tile, head).  Per 32-key tile and 16-query group: S^T = K.Q^T
//     as 2 x ksteps x 3 v_mfma_f32_16x16x32_bf16 (hi*hi + hi*lo + lo*hi), online softmax in registers (the 8
//     scores a lane holds all belong to its own query column; row max/sum = 2 shuffles), P split to bf16
//     hi/lo in place, O^T += V^T.P as d-blocks x 3 MFMAs.  Results go straight to flat voxel order.
// Work items (tiles, query groups) come from seg3d_window_partition, so no thread ever sees an empty window.
//
// Algorithmic FLOPs: 4*C*sum_w n_w^2 per layer; executed MFMA FLOPs are 3x that (split) plus
// tile padding.  Error: ~2^-16 relative per product (same budget as the sparse convs).
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

// attention_small.hip: exact-fp32 vector-ALU path for narrow heads (dh = 6, 12)
bool attn_small_supported(int heads, int dh);
int attn_small_fwd_launch(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const int32_t* tok,
                          const int32_t* win_start, const int32_t* win_count, const int32_t* tile_item, int n_tiles,
                          int heads, int dh, const float* tau, float tau_min, float* out, float* lse, hipStream_t st);
// attention_fused.hip: one launch per layer (gather + normalise + split fused into the MFMA core), dropout-capable
bool attn_fused_supported(int heads, int dh);
int attn_fused_fwd_launch(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const int32_t* tok,
                          const int32_t* win_start, const int32_t* win_count, const int32_t* tile_item, int n_tiles,
                          const int32_t* chunk_item, int n_chunks, int heads, int dh, const float* tau, float tau_min,
                          float* out, float* lse, float dropout_p, uint64_t seed, hipStream_t st);
bool attn_use_fused(int heads, int dh) {
    static const bool off = getenv("SEG3D_ATTN_FUSED") && atoi(getenv("SEG3D_ATTN_FUSED")) == 0;  // A/B switch
    return !off && attn_fused_supported(heads, dh);
}
bool attn_use_small(int heads, int dh) {
    static const bool mfma_only = getenv("SEG3D_ATTN_MFMA_ONLY") != nullptr;  // A/B switch for profiling
    return !mfma_only && attn_small_supported(heads, dh);
}

namespace {

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

// slot of tile-local token r in the transposed arrays: key kappa(g, j) = (j < 4 ? 4g + j : 16 + 4g + j - 4)
// lives at slot 8g + j, so a lane's 8 score registers [u=0: r 0..3, u=1: r 0..3] line up with one 16-B load.
__device__ __forceinline__ int perm_slot(int r) {
    return r < 16 ? (r >> 2) * 8 + (r & 3) : ((r - 16) >> 2) * 8 + 4 + ((r - 16) & 3);
}

template <int DH>
struct Geo {
    static constexpr int DHS = (DH + 7) / 8 * 8;    // stored channels per head (multiple of 8)
    static constexpr int KS = (DHS + 31) / 32;      // MFMA k-steps over the head dimension
    static constexpr int NB = (DH + 15) / 16;       // 16-row d-blocks of the output
};

// ------------------------------------------------------------------ prepare
template <int DH>
__global__ __launch_bounds__(256) void attn_prepare_fwd(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, int ldq, int ldk, int ldv,
                                                        const int32_t* __restrict__ tok, const int32_t* __restrict__ win_start,
                                                        const int32_t* __restrict__ win_count,
                                                        const int32_t* __restrict__ win_tile0, const int2* __restrict__ tile_item,
                                                        int heads, int64_t mpad, const float* __restrict__ tau, float tau_min,
                                                        __bf16* __restrict__ qp, __bf16* __restrict__ kp, __bf16* __restrict__ vt) {
    constexpr int DHS = Geo<DH>::DHS;
    extern __shared__ float smem[];
    const int c = heads * DH, cp = c + 1;
    float* buf = smem;                      // [32][cp]
    float* rn = smem + 32 * cp;             // [32][heads]
    int32_t* trow = reinterpret_cast<int32_t*>(rn + 32 * heads);  // [32] token row or -1

    const int2 item = tile_item[blockIdx.x];
    const int n = win_count[item.x], start = win_start[item.x];
    const int64_t pos0 = ((int64_t)win_tile0[item.x] + item.y) * 32;
    const int tid = threadIdx.x;
    if (tid < 32) {
        const int i = item.y * 32 + tid;
        trow[tid] = i < n ? tok[start + i] : -1;
    }
    const float qscale = kLog2e / fmaxf(tau[0], tau_min);
    __syncthreads();

    for (int which = 0; which < 3; ++which) {
        const float* src = which == 0 ? q : which == 1 ? k : v;
        const int ld = which == 0 ? ldq : which == 1 ? ldk : ldv;
        for (int e = tid; e < 8 * c; e += 256) {  // 16-B row pieces (c and every ld are multiples of 4)
            const int row = e / (c / 4), col = 4 * (e - row * (c / 4));
            const int t = trow[row];
            float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t >= 0) x4 = *reinterpret_cast<const float4*>(src + (int64_t)t * ld + col);
            float* b4 = buf + row * cp + col;
            b4[0] = x4.x; b4[1] = x4.y; b4[2] = x4.z; b4[3] = x4.w;
        }
        __syncthreads();
        if (which < 2) {
            for (int e = tid; e < 32 * heads; e += 256) {
                const int row = e / heads, h = e - row * heads;
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) {
                    const float x = buf[row * cp + h * DH + d];
                    s = fmaf(x, x, s);
                }
                rn[e] = (which == 0 ? qscale : 1.0f) / fmaxf(sqrtf(s), kNormEps);
            }
            __syncthreads();
            // 16-B stores: one item = 8 consecutive stored channels of one (row, head)
            __bf16* dst = which == 0 ? qp : kp;
            const int64_t half = mpad * heads * DHS;  // hi block then lo block
            constexpr int CH8 = DHS / 8;
            for (int e = tid; e < 32 * heads * CH8; e += 256) {
                const int c8 = e % CH8, h = (e / CH8) % heads, row = e / (CH8 * heads);
                const float r = rn[row * heads + h];
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = 8 * c8 + j < DH ? buf[row * cp + h * DH + 8 * c8 + j] * r : 0.f;
                u32x4 hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t w = pack_bf16(x[2 * j], x[2 * j + 1]);
                    hi[j] = w;
                    lo[j] = pack_bf16(x[2 * j] - __builtin_bit_cast(float, w << 16),
                                      x[2 * j + 1] - __builtin_bit_cast(float, w & 0xFFFF0000u));
                }
                const int64_t o = ((pos0 + row) * heads + h) * DHS + 8 * c8;
                *reinterpret_cast<u32x4*>(dst + o) = hi;
                *reinterpret_cast<u32x4*>(dst + half + o) = lo;
            }
        } else {
            // transposed: one item = the 8 token slots 8g .. 8g+7 of one channel row (tokens 4g..4g+3, 16+4g..16+4g+3)
            const int64_t half = (int64_t)heads * DH * mpad;
            for (int e = tid; e < 4 * c; e += 256) {
                const int sg = e & 3, ch = e >> 2;
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = buf[(j < 4 ? 4 * sg + j : 12 + 4 * sg + j) * cp + ch];
                u32x4 hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t w = pack_bf16(x[2 * j], x[2 * j + 1]);
                    hi[j] = w;
                    lo[j] = pack_bf16(x[2 * j] - __builtin_bit_cast(float, w << 16),
                                      x[2 * j + 1] - __builtin_bit_cast(float, w & 0xFFFF0000u));
                }
                const int64_t o = (int64_t)ch * mpad + pos0 + 8 * sg;
                *reinterpret_cast<u32x4*>(vt + o) = hi;
                *reinterpret_cast<u32x4*>(vt + half + o) = lo;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ core
// One wave per (32-query tile of a window, head), looping over the window's 32-key tiles: every K / V fragment it
// loads serves two 16-query groups (the K/V re-reads of the large windows are what bounds this kernel:
// sum_w n_w^2 / 32 * heads * 2 * (DHS + DH) * 2 bytes per layer); the second group is skipped (wave-uniform) when
// the tile holds <= 16 queries.
// Addressing: what depends on (item, head, tile) is wave-uniform and lives in scalar base pointers; the lane's share
// is a constant 32-bit byte offset.  No register prefetch -- measured: the extra stage registers cost more
// occupancy than the prefetch hides.  Only the last key tile masks keys >= n.
// Channel slices past DHS are zeroed in Q (so K needs no mask); V rows past DH only feed output rows that are never stored.
// waves per SIMD the register allocator must leave room for: dh 24 sits one register above the 4-wave line without it
template <int DH>
constexpr int kFwdWaves = DH <= 24 ? 4 : 2;

template <int DH>
__global__ __launch_bounds__(256, kFwdWaves<DH>) void attn_core_fwd(const __bf16* __restrict__ qp, const __bf16* __restrict__ kp,
                                                     const __bf16* __restrict__ vt, const int32_t* __restrict__ tok,
                                                     const int32_t* __restrict__ win_start, const int32_t* __restrict__ win_count,
                                                     const int32_t* __restrict__ win_tile0, const int2* __restrict__ tile_item,
                                                     int n_items, int heads, int64_t mpad, float* __restrict__ out,
                                                     float* __restrict__ lse) {
    constexpr int DHS = Geo<DH>::DHS, KS = Geo<DH>::KS, NB = Geo<DH>::NB;
    const int lane = threadIdx.x & 63;
    const int it = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (it >= n_items) return;
    const int g = lane >> 4, c16 = lane & 15;
    const int2 item = tile_item[it];
    const int n = win_count[item.x], start = win_start[item.x];
    const int64_t pos0 = (int64_t)win_tile0[item.x] * 32;
    const int n_kt = (n + 31) >> 5;
    const int q0 = item.y * 32;
    const bool two = n - q0 > 16;  // the tile's second 16-query group exists (wave-uniform)
    const int64_t qk_half = mpad * heads * DHS;
    const int64_t vt_half = (int64_t)heads * DH * mpad;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    // lane byte offsets; channel slices / d rows that do not exist read slice 0 / row 0
    uint32_t koff[2][KS], qoff[2][KS], voff[NB];
    bool slice_ok[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        slice_ok[s] = 32 * s + 8 * g < DHS;
        const int sl = slice_ok[s] ? 32 * s + 8 * g : 0;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            koff[u][s] = (uint32_t)(((u * 16 + c16) * heads * DHS + sl) * 2);
            qoff[u][s] = (uint32_t)(((q0 + u * 16 + c16) * heads * DHS + sl) * 2);
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) voff[b] = (uint32_t)(((int64_t)(16 * b + c16 < DH ? 16 * b + c16 : 0) * mpad + 8 * g) * 2);
    const int64_t kstep = (int64_t)32 * heads * DHS * 2;
    // token rows of this lane's two queries
    int32_t token[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) token[j] = q0 + 16 * j + c16 < n ? tok[start + q0 + 16 * j + c16] : -1;

    u32x4 kf[2][2][KS][2], vf[2][NB][2];  // [stage][..][hi, lo]
    auto fetch = [&](int h, int t, auto stage) {
        constexpr int S = decltype(stage)::value;
        const char* kb = reinterpret_cast<const char*>(kp + (pos0 * heads + h) * DHS) + t * kstep;
        const char* vb = reinterpret_cast<const char*>(vt + (int64_t)h * DH * mpad + pos0) + t * 64;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                kf[S][u][s][0] = *reinterpret_cast<const u32x4*>(kb + koff[u][s]);
                kf[S][u][s][1] = *reinterpret_cast<const u32x4*>(kb + qk_half * 2 + koff[u][s]);
            }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            vf[S][b][0] = *reinterpret_cast<const u32x4*>(vb + voff[b]);
            vf[S][b][1] = *reinterpret_cast<const u32x4*>(vb + vt_half * 2 + voff[b]);
        }
    };
    // Q fragments of the two query groups: B operand, lane (query c16, channels 32s + 8g .. +7)
    u32x4 qf[2][KS][2];
    auto load_q = [&](int h) {
        const char* qb = reinterpret_cast<const char*>(qp + (pos0 * heads + h) * DHS);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(qb + qoff[j][s]);
                const u32x4 b = *reinterpret_cast<const u32x4*>(qb + qk_half * 2 + qoff[j][s]);
                qf[j][s][0] = slice_ok[s] ? a : zero4;
                qf[j][s][1] = slice_ok[s] ? b : zero4;
            }
    };

    f32x4 o_acc[2][NB];
    float m_run[2], l_run[2];
    auto reset = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            m_run[j] = -INFINITY;
            l_run[j] = 0.f;
#pragma unroll
            for (int b = 0; b < NB; ++b) o_acc[j][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    // one key tile for query group j: scores, online softmax, O^T += V^T . P
    auto group_tile = [&](int j, int t, auto stage, bool last, f32x4 (&s_acc)[2]) {
        constexpr int S = decltype(stage)::value;
        float sc[8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[u * 4 + r] = s_acc[u][r];
        if (last) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (t * 32 + (i >> 2) * 16 + g * 4 + (i & 3) >= n) sc[i] = -INFINITY;
        }
        float tmax = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, SEG3D_WAVE));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, SEG3D_WAVE));
        const float m_new = fmaxf(m_run[j], tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run[j] - m_new);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sc[i] = __builtin_amdgcn_exp2f(sc[i] - m_new);
            psum += sc[i];
        }
        psum += __shfl_xor(psum, 16, SEG3D_WAVE);
        psum += __shfl_xor(psum, 32, SEG3D_WAVE);
        l_run[j] = fmaf(l_run[j], alpha, psum);
        m_run[j] = m_new;
        u32x4 ph, pl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w = pack_bf16(sc[2 * i], sc[2 * i + 1]);
            const float h0 = __builtin_bit_cast(float, w << 16);
            const float h1 = __builtin_bit_cast(float, w & 0xFFFF0000u);
            ph[i] = w;
            pl[i] = pack_bf16(sc[2 * i] - h0, sc[2 * i + 1] - h1);
        }
        const bf16x8 p_hi = __builtin_bit_cast(bf16x8, ph), p_lo = __builtin_bit_cast(bf16x8, pl);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const bf16x8 v_hi = __builtin_bit_cast(bf16x8, vf[S][b][0]), v_lo = __builtin_bit_cast(bf16x8, vf[S][b][1]);
            f32x4 acc = o_acc[j][b] * alpha;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v_lo, p_hi, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v_hi, p_lo, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v_hi, p_hi, acc, 0, 0, 0);
            o_acc[j][b] = acc;
        }
    };
    auto scores = [&](int j, auto stage, f32x4 (&s_acc)[2]) {
        constexpr int S = decltype(stage)::value;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            s_acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 k_hi = __builtin_bit_cast(bf16x8, kf[S][u][s][0]);
                const bf16x8 k_lo = __builtin_bit_cast(bf16x8, kf[S][u][s][1]);
                const bf16x8 q_hi = __builtin_bit_cast(bf16x8, qf[j][s][0]);
                const bf16x8 q_lo = __builtin_bit_cast(bf16x8, qf[j][s][1]);
                s_acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k_lo, q_hi, s_acc[u], 0, 0, 0);
                s_acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k_hi, q_lo, s_acc[u], 0, 0, 0);
                s_acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k_hi, q_hi, s_acc[u], 0, 0, 0);
            }
        }
    };
    // epilogue of a head: O^T[d = 16b + 4g + r][query c16] / l  ->  out[token][h*DH + d]
    auto finish = [&](int h) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (token[j] < 0) continue;
            const float inv = 1.0f / l_run[j];
            float* op = out + (int64_t)token[j] * (heads * DH) + h * DH;
            // a lane's 4 accumulator rows are 4 consecutive channels of its query: one 16-B store (8-B for DH = 6)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int d = 16 * b + 4 * g;
                const f32x4 o = o_acc[j][b] * inv;
                if (DH % 4 == 0) {
                    if (d < DH) *reinterpret_cast<f32x4*>(op + d) = o;
                } else {
                    if (d + 1 < DH) *reinterpret_cast<f32x2*>(op + d) = (f32x2){o[0], o[1]};
                    if (d + 3 < DH) *reinterpret_cast<f32x2*>(op + d + 2) = (f32x2){o[2], o[3]};
                }
            }
            if (lse && g == 0) lse[(int64_t)token[j] * heads + h] = (m_run[j] + __builtin_amdgcn_logf(l_run[j])) * kLn2;
        }
    };
    auto step = [&](int h, int t, auto stage) {
        const bool last = t + 1 == n_kt;
        f32x4 s0[2], s1[2];
        scores(0, stage, s0);
        if (two) scores(1, stage, s1);
        group_tile(0, t, stage, last, s0);
        if (two) group_tile(1, t, stage, last, s1);
        if (last) finish(h);
    };

    using St0 = std::integral_constant<int, 0>;
    const int total = n_kt;
    const int h = blockIdx.y;
    reset();
    load_q(h);
    for (int t = 0; t < total; ++t) {  // no register prefetch: the other waves of the SIMD cover the latency
        fetch(h, t, St0{});
        step(h, t, St0{});
    }
}

template <int DH>
__global__ __launch_bounds__(256, 2) void attn_core_fwd_lds(const __bf16* __restrict__ qp, const __bf16* __restrict__ kp,
                                                     const __bf16* __restrict__ vt, const int32_t* __restrict__ tok,
                                                     const int32_t* __restrict__ win_start, const int32_t* __restrict__ win_count,
                                                     const int32_t* __restrict__ win_tile0, const int2* __restrict__ tile_item,
                                                     int n_items, int heads, int64_t mpad, float* __restrict__ out,
                                                     float* __restrict__ lse) {
    constexpr int DHS = Geo<DH>::DHS, KS = Geo<DH>::KS, NB = Geo<DH>::NB;
    // One workgroup = four consecutive 32-query tiles of ONE window and one head: the K / V fragments of every key tile
    // are brought in once per workgroup (two 16-B loads per thread), parked in LDS in fragment order and read from there
    // by the four waves (conflict-free ds_read_b128) -- a quarter of the L1 fragment traffic of attn_core_fwd.  The grid
    // has one workgroup per tile; those whose tile is not the first of a group of four have nothing to do.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int2 item0 = tile_item[blockIdx.x];
    if (item0.y & 3) return;
    const int g = lane >> 4, c16 = lane & 15;
    const int n = win_count[item0.x], start = win_start[item0.x];
    const int64_t pos0 = (int64_t)win_tile0[item0.x] * 32;
    const int n_kt = (n + 31) >> 5;
    const int q0 = (item0.y + wave) * 32;
    const bool active = q0 < n;            // this wave's query tile exists (wave-uniform)
    const bool two = n - q0 > 16;  // the tile's second 16-query group exists (wave-uniform)
    (void)n_items;
    const int64_t qk_half = mpad * heads * DHS;
    const int64_t vt_half = (int64_t)heads * DH * mpad;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    // lane byte offsets; channel slices / d rows that do not exist read slice 0 / row 0
    uint32_t koff[2][KS], qoff[2][KS], voff[NB];
    bool slice_ok[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        slice_ok[s] = 32 * s + 8 * g < DHS;
        const int sl = slice_ok[s] ? 32 * s + 8 * g : 0;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            koff[u][s] = (uint32_t)(((u * 16 + c16) * heads * DHS + sl) * 2);
            qoff[u][s] = (uint32_t)(((q0 + u * 16 + c16) * heads * DHS + sl) * 2);
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) voff[b] = (uint32_t)(((int64_t)(16 * b + c16 < DH ? 16 * b + c16 : 0) * mpad + 8 * g) * 2);
    const int64_t kstep = (int64_t)32 * heads * DHS * 2;
    // token rows of this lane's two queries
    int32_t token[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) token[j] = q0 + 16 * j + c16 < n ? tok[start + q0 + 16 * j + c16] : -1;

    // LDS image of one key tile, double buffered: piece p = 64 lanes x 16 B in register-fragment order
    constexpr int kKPieces = 2 * KS * 2, kVPieces = NB * 2, kPieces = kKPieces + kVPieces;
    constexpr int kMine = (kPieces * 64 + 255) / 256;  // 16-B items per thread
    __shared__ u32x4 tile_lds[2][kPieces * 64];
    u32x4 kf[1][2][KS][2], vf[1][NB][2];
    u32x4 st_reg[kMine];
    auto stage_load = [&](int h, int t) {
        const char* kb = reinterpret_cast<const char*>(kp + (pos0 * heads + h) * DHS) + t * kstep;
        const char* vb = reinterpret_cast<const char*>(vt + (int64_t)h * DH * mpad + pos0) + t * 64;
#pragma unroll
        for (int j = 0; j < kMine; ++j) {
            const int item = j * 256 + threadIdx.x;
            const int piece = item >> 6, l = item & 63, lg = l >> 4, lc = l & 15;
            st_reg[j] = zero4;
            if (piece < kKPieces) {  // K piece (u, s, plane)
                const int plane = piece & 1, us = piece >> 1, s_ = us % KS, u = us / KS;
                const int sl = 32 * s_ + 8 * lg < DHS ? 32 * s_ + 8 * lg : 0;
                st_reg[j] = *reinterpret_cast<const u32x4*>(kb + plane * (qk_half * 2) + (uint32_t)(((u * 16 + lc) * heads * DHS + sl) * 2));
            } else if (piece < kPieces) {  // V piece (b, plane)
                const int pv = piece - kKPieces, plane = pv & 1, b_ = pv >> 1;
                const int64_t row = 16 * b_ + lc < DH ? 16 * b_ + lc : 0;
                st_reg[j] = *reinterpret_cast<const u32x4*>(vb + plane * (vt_half * 2) + (row * mpad + 8 * lg) * 2);
            }
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < kMine; ++j) {
            const int item = j * 256 + threadIdx.x;
            if (kPieces * 64 % 256 == 0 || item < kPieces * 64) tile_lds[buf][item] = st_reg[j];
        }
    };
    auto fetch = [&](int buf) {  // this wave's fragments of the staged tile
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                kf[0][u][s][0] = tile_lds[buf][((u * KS + s) * 2 + 0) * 64 + lane];
                kf[0][u][s][1] = tile_lds[buf][((u * KS + s) * 2 + 1) * 64 + lane];
            }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            vf[0][b][0] = tile_lds[buf][(kKPieces + b * 2 + 0) * 64 + lane];
            vf[0][b][1] = tile_lds[buf][(kKPieces + b * 2 + 1) * 64 + lane];
        }
    };
    // Q fragments of the two query groups: B operand, lane (query c16, channels 32s + 8g .. +7)
    u32x4 qf[2][KS][2];
    auto load_q = [&](int h) {
        const char* qb = reinterpret_cast<const char*>(qp + (pos0 * heads + h) * DHS);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(qb + qoff[j][s]);
                const u32x4 b = *reinterpret_cast<const u32x4*>(qb + qk_half * 2 + qoff[j][s]);
                qf[j][s][0] = slice_ok[s] ? a : zero4;
                qf[j][s][1] = slice_ok[s] ? b : zero4;
            }
    };

    f32x4 o_acc[2][NB];
    float m_run[2], l_run[2];
    auto reset = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            m_run[j] = -INFINITY;
            l_run[j] = 0.f;
#pragma unroll
            for (int b = 0; b < NB; ++b) o_acc[j][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    // one key tile for query group j: scores, online softmax, O^T += V^T . P
    auto group_tile = [&](int j, int t, auto stage, bool last, f32x4 (&s_acc)[2]) {
        constexpr int S = decltype(stage)::value;
        float sc[8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[u * 4 + r] = s_acc[u][r];
        if (last) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (t * 32 + (i >> 2) * 16 + g * 4 + (i & 3) >= n) sc[i] = -INFINITY;
        }
        float tmax = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, SEG3D_WAVE));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, SEG3D_WAVE));
        const float m_new = fmaxf(m_run[j], tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run[j] - m_new);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sc[i] = __builtin_amdgcn_exp2f(sc[i] - m_new);
            psum += sc[i];
        }
        psum += __shfl_xor(psum, 16, SEG3D_WAVE);
        psum += __shfl_xor(psum, 32, SEG3D_WAVE);
        l_run[j] = fmaf(l_run[j], alpha, psum);
        m_run[j] = m_new;
        u32x4 ph, pl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w = pack_bf16(sc[2 * i], sc[2 * i + 1]);
            const float h0 = __builtin_bit_cast(float, w << 16);
            const float h1 = __builtin_bit_cast(float, w & 0xFFFF0000u);
            ph[i] = w;
            pl[i] = pack_bf16(sc[2 * i] - h0, sc[2 * i + 1] - h1);
        }
        const bf16x8 p_hi = __builtin_bit_cast(bf16x8, ph), p_lo = __builtin_bit_cast(bf16x8, pl);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const bf16x8 v_hi = __builtin_bit_cast(bf16x8, vf[S][b][0]), v_lo = __builtin_bit_cast(bf16x8, vf[S][b][1]);
            f32x4 acc = o_acc[j][b] * alpha;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v_lo, p_hi, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v_hi, p_lo, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v_hi, p_hi, acc, 0, 0, 0);
            o_acc[j][b] = acc;
        }
    };
    auto scores = [&](int j, auto stage, f32x4 (&s_acc)[2]) {
        constexpr int S = decltype(stage)::value;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            s_acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 k_hi = __builtin_bit_cast(bf16x8, kf[S][u][s][0]);
                const bf16x8 k_lo = __builtin_bit_cast(bf16x8, kf[S][u][s][1]);
                const bf16x8 q_hi = __builtin_bit_cast(bf16x8, qf[j][s][0]);
                const bf16x8 q_lo = __builtin_bit_cast(bf16x8, qf[j][s][1]);
                s_acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k_lo, q_hi, s_acc[u], 0, 0, 0);
                s_acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k_hi, q_lo, s_acc[u], 0, 0, 0);
                s_acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k_hi, q_hi, s_acc[u], 0, 0, 0);
            }
        }
    };
    // epilogue of a head: O^T[d = 16b + 4g + r][query c16] / l  ->  out[token][h*DH + d]
    auto finish = [&](int h) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (token[j] < 0) continue;
            const float inv = 1.0f / l_run[j];
            float* op = out + (int64_t)token[j] * (heads * DH) + h * DH;
            // a lane's 4 accumulator rows are 4 consecutive channels of its query: one 16-B store (8-B for DH = 6)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int d = 16 * b + 4 * g;
                const f32x4 o = o_acc[j][b] * inv;
                if (DH % 4 == 0) {
                    if (d < DH) *reinterpret_cast<f32x4*>(op + d) = o;
                } else {
                    if (d + 1 < DH) *reinterpret_cast<f32x2*>(op + d) = (f32x2){o[0], o[1]};
                    if (d + 3 < DH) *reinterpret_cast<f32x2*>(op + d + 2) = (f32x2){o[2], o[3]};
                }
            }
            if (lse && g == 0) lse[(int64_t)token[j] * heads + h] = (m_run[j] + __builtin_amdgcn_logf(l_run[j])) * kLn2;
        }
    };
    auto step = [&](int h, int t, auto stage) {
        const bool last = t + 1 == n_kt;
        f32x4 s0[2], s1[2];
        scores(0, stage, s0);
        if (two) scores(1, stage, s1);
        group_tile(0, t, stage, last, s0);
        if (two) group_tile(1, t, stage, last, s1);
        if (last) finish(h);
    };

    using St0 = std::integral_constant<int, 0>;
    const int h = blockIdx.y;
    reset();
    if (active) load_q(h);
    stage_load(h, 0);
    stage_store(0);
    __syncthreads();
    int buf = 0;
    for (int t = 0; t < n_kt; ++t) {
        const bool more = t + 1 < n_kt;
        if (more) stage_load(h, t + 1);  // in flight while this tile is multiplied
        if (active) {
            fetch(buf);
            step(h, t, St0{});
        }
        if (more) stage_store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
}

template <int DH>
size_t prepared_bytes(int64_t mpad, int heads) {
    const size_t qk = (size_t)mpad * heads * Geo<DH>::DHS * 2 * sizeof(__bf16);  // hi + lo
    const size_t vt = (size_t)heads * DH * mpad * 2 * sizeof(__bf16);
    return 2 * align_up(qk, 256) + align_up(vt, 256);
}

template <int DH>
int run_fwd(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const int32_t* tok,
            const int32_t* win_start, const int32_t* win_count, const int32_t* win_tile0, const int2* tile_item,
            int n_tiles, const int2* qg_item, int n_qg, int heads, const float* tau, float tau_min, float* out,
            float* lse, void* workspace, hipStream_t st) {
    const int64_t mpad = (int64_t)n_tiles * 32;
    const size_t qk = align_up((size_t)mpad * heads * Geo<DH>::DHS * 2 * sizeof(__bf16), 256);
    char* base = static_cast<char*>(workspace);
    __bf16* qp = reinterpret_cast<__bf16*>(base);
    __bf16* kp = reinterpret_cast<__bf16*>(base + qk);
    __bf16* vt = reinterpret_cast<__bf16*>(base + 2 * qk);
    const int c = heads * DH;
    const size_t smem = (size_t)(32 * (c + 1) + 32 * heads) * sizeof(float) + 32 * sizeof(int32_t);
    hipLaunchKernelGGL(attn_prepare_fwd<DH>, dim3((unsigned)n_tiles), dim3(256), smem, st, q, k, v, ldq, ldk, ldv, tok,
                       win_start, win_count, win_tile0, tile_item, heads, mpad, tau, tau_min, qp, kp, vt);
    SEG3D_CHECK_LAUNCH();
    (void)qg_item;
    (void)n_qg;
    // default: K / V tiles staged once per four query tiles through LDS (dh 24: 152 -> 120 us per layer on the headline
    // scene, dh 48: 90 -> 83 us); SEG3D_ATTN_LDS=0 selects the wave-independent kernel for A/B runs
    static const int lds_env = getenv("SEG3D_ATTN_LDS") ? atoi(getenv("SEG3D_ATTN_LDS")) : 1;
    if (lds_env)
        hipLaunchKernelGGL(attn_core_fwd_lds<DH>, dim3((unsigned)n_tiles, (unsigned)heads), dim3(256), 0, st, qp, kp, vt, tok,
                           win_start, win_count, win_tile0, tile_item, n_tiles, heads, mpad, out, lse);
    else
        hipLaunchKernelGGL(attn_core_fwd<DH>, dim3((unsigned)((n_tiles + 3) / 4), (unsigned)heads), dim3(256), 0, st, qp, kp, vt, tok, win_start,
                           win_count, win_tile0, tile_item, n_tiles, heads, mpad, out, lse);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // namespace

size_t attn_mfma_workspace_bytes(int n_tiles, int heads, int dh) {
    const int64_t mpad = (int64_t)n_tiles * 32;
    switch (dh) {
        case 6: return prepared_bytes<6>(mpad, heads);
        case 12: return prepared_bytes<12>(mpad, heads);
        case 24: return prepared_bytes<24>(mpad, heads);
        case 48: return prepared_bytes<48>(mpad, heads);
        default: return 0;
    }
}

extern "C" int seg3d_window_attn_fwd(const float* q, const float* k, const float* v, int32_t ldq, int32_t ldk, int32_t ldv,
                                     const int32_t* tok, const int32_t* win_start, const int32_t* win_count,
                                     const int32_t* win_tile0, const int32_t* tile_item, int32_t n_tiles,
                                     const int32_t* qg_item, int32_t n_qgroups, int64_t m, int32_t n_windows,
                                     int32_t heads, int32_t dh, const float* tau, float tau_min, float dropout_p,
                                     uint64_t dropout_seed, float* out, float* lse, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (m == 0 || n_windows == 0 || n_tiles == 0 || n_qgroups == 0) return SEG3D_OK;
    if (!q || !k || !v || !tok || !win_start || !win_count || !win_tile0 || !tile_item || !qg_item || m < 0 ||
        n_windows < 0 || n_tiles < 0 || n_qgroups < 0 || heads <= 0 || heads > 16 || !tau || !out || !workspace ||
        !(dropout_p >= 0.f && dropout_p < 1.f))
        return SEG3D_EINVAL;
    if (attn_use_fused(heads, dh)) {
        if (((ldq | ldk | ldv) & 3) ||
            ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15))
            return SEG3D_EINVAL;
        return attn_fused_fwd_launch(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, tile_item, n_tiles, qg_item, n_qgroups,
                                     heads, dh, tau, tau_min, out, lse, dropout_p, dropout_seed, as_stream(stream));
    }
    if (dropout_p > 0.f) return SEG3D_EINVAL;  // only the fused kernels carry the dropout mask
    if (attn_use_small(heads, dh)) {
        if (((ldq | ldk | ldv) & 3) ||
            ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15))
            return SEG3D_EINVAL;
        return attn_small_fwd_launch(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, tile_item, n_tiles, heads, dh, tau,
                                     tau_min, out, lse, as_stream(stream));
    }
    // rows are gathered in 16-B pieces
    if (((ldq | ldk | ldv) & 3) || ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15))
        return SEG3D_EINVAL;
    const size_t need = attn_mfma_workspace_bytes(n_tiles, heads, dh);
    if (need == 0) return SEG3D_EINVAL;
    if (workspace_bytes < need) return SEG3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    const int2* ti = reinterpret_cast<const int2*>(tile_item);
    const int2* qi = reinterpret_cast<const int2*>(qg_item);
    switch (dh) {
        case 6: return run_fwd<6>(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, win_tile0, ti, n_tiles, qi, n_qgroups,
                                  heads, tau, tau_min, out, lse, workspace, st);
        case 12: return run_fwd<12>(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, win_tile0, ti, n_tiles, qi,
                                    n_qgroups, heads, tau, tau_min, out, lse, workspace, st);
        case 24: return run_fwd<24>(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, win_tile0, ti, n_tiles, qi,
                                    n_qgroups, heads, tau, tau_min, out, lse, workspace, st);
        case 48: return run_fwd<48>(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, win_tile0, ti, n_tiles, qi,
                                    n_qgroups, heads, tau, tau_min, out, lse, workspace, st);
        default: return SEG3D_EINVAL;
    }
}
