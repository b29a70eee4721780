// a19-a21 backward on the matrix cores: ragged sparse-window cosine attention, split-bf16 MFMA.
// Gradient of  out_i = softmax_j(<q_i/|q_i|, k_j/|k_j|> / max(tau, tau_min)) . v_j  (cosine_msa.py:115-177)
// w.r.t. the raw q, k, v (through the L2 normalisation) and tau.
//
// Three launches per layer, flash-style recompute (no score tensor is ever stored):
//  1. attn_prepare_bwd (one workgroup per 32-token tile): gathers q, k, v, dO rows, normalises q/k, splits to
//     bf16 hi/lo and writes row-major copies (Qp, Kp, Vp, Gp) and tile-permuted transposed copies (Qt, Kt,
//     Gt), plus per (token, head) the log2-domain LSE and delta = <dO, O>.
//  2. attn_bwd_q: one wave per (window, 32-query tile, head), loop over 32-key tiles, per 16-query group:
//        S^T = K.Q^T, dP^T = V.dO^T, P = exp2(S - L), dS = P (dP - delta), dQ^T += K^T.dS^T, dtau += <dS, S>
//  3. attn_bwd_kv: one wave per (window, 32-key tile, head), loop over 32-query tiles, per 16-key group:
//        S = Q.K^T, dP = dO.V^T, P, dS as above, dV^T += dO^T.P, dK^T += Q^T.dS
//  The accumulator layout of the first two products is exactly the B-operand layout of the last ones (tokens
//  permuted inside a tile, attn_common.hpp), so nothing moves between lanes and no LDS is used.  Each wave
//  owns its output rows: no atomics except one float add per wave for the scalar tau gradient.
#include <cstdlib>

#include "attn_common.hpp"
#include "attn_dropout.hpp"

size_t attn_mfma_workspace_bytes(int n_tiles, int heads, int dh);  // attention_mfma.hip
bool attn_use_fused(int heads, int dh);                            // attention_mfma.hip
struct DropoutParams;
size_t attn_fused_bwd_workspace_bytes(int n_tiles, int n_chunks, int heads, int dh);  // attention_fused_bwd.hip
bool attn_fused_bwd_supported(int heads, int dh);                                    // attention_fused_bwd.hip
int attn_fused_bwd_launch(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const float* out,
                          const float* dout, const float* lse, const int32_t* tok, const int32_t* win_start,
                          const int32_t* win_count, const int32_t* tile_item, int n_tiles, const int32_t* chunk_item,
                          int n_chunks, int heads, int dh, const float* tau, float tau_min, float* dq, float* dk, float* dv,
                          int lddq, int lddk, int lddv, float* dtau, void* workspace, const DropoutParams& drop,
                          hipStream_t st);                          // attention_fused_bwd.hip
bool attn_use_small(int heads, int dh);                            // attention_mfma.hip
int attn_small_bwd_launch(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const float* out,
                          const float* dout, const float* lse, const int32_t* tok, const int32_t* win_start,
                          const int32_t* win_count, const int32_t* tile_item, int n_tiles, int heads, int dh,
                          const float* tau, float tau_min, float* dq, float* dk, float* dv, int lddq, int lddk, int lddv,
                          float* dtau, void* workspace, const DropoutParams& drop, hipStream_t st);  // attention_small.hip

namespace {

using namespace attn;

template <int DH>
struct BwdWs {
    __bf16 *qp, *kp, *vp, *gp;  // row-major  [mpad][heads][DHS], hi block then lo block
    __bf16 *qt, *kt, *gt;       // transposed [heads][DH][mpad],  hi block then lo block
    float *lp, *dp;             // [heads][mpad]: log2-domain LSE, delta
    float* tau_part;            // one partial tau gradient per wave of pass A, summed in a fixed order by tau_reduce
    static size_t row_bytes(int64_t mpad, int heads) {
        return align_up((size_t)mpad * heads * Geo<DH>::DHS * 2 * sizeof(__bf16), 256);
    }
    static size_t tr_bytes(int64_t mpad, int heads) { return align_up((size_t)heads * DH * mpad * 2 * sizeof(__bf16), 256); }
    static size_t f_bytes(int64_t mpad, int heads) { return align_up((size_t)mpad * heads * sizeof(float), 256); }
    static size_t tau_count(int64_t mpad, int heads) { return (size_t)((mpad / 32 + 3) / 4 * 4) * heads; }
    static size_t tau_bytes(int64_t mpad, int heads) { return align_up(tau_count(mpad, heads) * sizeof(float), 256); }
    static size_t total(int64_t mpad, int heads) {
        return 4 * row_bytes(mpad, heads) + 3 * tr_bytes(mpad, heads) + 2 * f_bytes(mpad, heads) + tau_bytes(mpad, heads);
    }
    BwdWs(void* base, int64_t mpad, int heads) {
        char* p = static_cast<char*>(base);
        auto take = [&](size_t n) { char* r = p; p += n; return r; };
        qp = reinterpret_cast<__bf16*>(take(row_bytes(mpad, heads)));
        kp = reinterpret_cast<__bf16*>(take(row_bytes(mpad, heads)));
        vp = reinterpret_cast<__bf16*>(take(row_bytes(mpad, heads)));
        gp = reinterpret_cast<__bf16*>(take(row_bytes(mpad, heads)));
        qt = reinterpret_cast<__bf16*>(take(tr_bytes(mpad, heads)));
        kt = reinterpret_cast<__bf16*>(take(tr_bytes(mpad, heads)));
        gt = reinterpret_cast<__bf16*>(take(tr_bytes(mpad, heads)));
        lp = reinterpret_cast<float*>(take(f_bytes(mpad, heads)));
        dp = reinterpret_cast<float*>(take(f_bytes(mpad, heads)));
        tau_part = reinterpret_cast<float*>(take(tau_bytes(mpad, heads)));
    }
};

// ------------------------------------------------------------------ prepare
template <int DH>
__global__ __launch_bounds__(256) void attn_prepare_bwd(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, int ldq, int ldk, int ldv,
                                                        const float* __restrict__ dout, const float* __restrict__ out,
                                                        const float* __restrict__ lse, const int32_t* __restrict__ tok,
                                                        const int32_t* __restrict__ win_start, const int32_t* __restrict__ win_count,
                                                        const int32_t* __restrict__ win_tile0, const int2* __restrict__ tile_item,
                                                        int heads, int64_t mpad, const float* __restrict__ tau, float tau_min,
                                                        BwdWs<DH> ws) {
    constexpr int DHS = Geo<DH>::DHS;
    extern __shared__ float smem[];
    const int c = heads * DH, cp = c + 1;
    float* buf = smem;                                           // [32][cp]
    float* rn = smem + 32 * cp;                                  // [32][heads]
    int32_t* trow = reinterpret_cast<int32_t*>(rn + 32 * heads);  // [32]

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

    const int64_t row_half = mpad * heads * DHS;
    const int64_t tr_half = (int64_t)heads * DH * mpad;
    for (int which = 0; which < 4; ++which) {  // 0 q, 1 k, 2 v, 3 dO
        const float* src = which == 0 ? q : which == 1 ? k : which == 2 ? v : dout;
        const int ld = which == 0 ? ldq : which == 1 ? ldk : which == 2 ? ldv : c;
        for (int e = tid; e < 8 * c; e += 256) {  // 16-B row pieces (c and every ld are multiples of 4)
            const int row = e / (c / 4), col = 4 * (e - row * (c / 4));
            const int t = trow[row];
            float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t >= 0) x4 = *reinterpret_cast<const float4*>(src + (int64_t)t * ld + col);
            float* b4 = buf + row * cp + col;
            b4[0] = x4.x; b4[1] = x4.y; b4[2] = x4.z; b4[3] = x4.w;
        }
        __syncthreads();
        for (int e = tid; e < 32 * heads; e += 256) {
            const int row = e / heads, h = e - row * heads;
            const int t = trow[row];
            if (which < 2) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) {
                    const float x = buf[row * cp + h * DH + d];
                    s = fmaf(x, x, s);
                }
                rn[e] = (which == 0 ? qscale : 1.0f) / fmaxf(sqrtf(s), kNormEps);
            } else {
                rn[e] = 1.0f;
                if (which == 3) {  // delta = <dO, O>, LSE in the log2 domain
                    float s = 0.f;
                    if (t >= 0) {
                        const float* op = out + (int64_t)t * c + h * DH;
#pragma unroll
                        for (int d = 0; d < DH; ++d) s = fmaf(buf[row * cp + h * DH + d], op[d], s);
                    }
                    ws.dp[(int64_t)h * mpad + pos0 + row] = s;
                    ws.lp[(int64_t)h * mpad + pos0 + row] = t >= 0 ? lse[(int64_t)t * heads + h] * kLog2e : 0.f;
                }
            }
        }
        __syncthreads();
        // 16-B stores: one item = 8 consecutive stored channels of one (row, head)
        __bf16* rm = which == 0 ? ws.qp : which == 1 ? ws.kp : which == 2 ? ws.vp : ws.gp;
        constexpr int CH8 = DHS / 8;
        for (int e = tid; e < 32 * heads * CH8; e += 256) {
            const int c8 = e % CH8, h = (e / CH8) % heads, row = e / (CH8 * heads);
            const float r = rn[row * heads + h];
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = 8 * c8 + j < DH ? buf[row * cp + h * DH + 8 * c8 + j] * r : 0.f;
            bf16x8 hi, lo;
            split_frag(x, &hi, &lo);
            const int64_t o = ((pos0 + row) * heads + h) * DHS + 8 * c8;
            *reinterpret_cast<bf16x8*>(rm + o) = hi;
            *reinterpret_cast<bf16x8*>(rm + row_half + o) = lo;
        }
        if (which != 2) {
            // transposed: one item = the 8 token slots 8g .. 8g+7 of one channel row (tokens 4g..4g+3, 16+4g..16+4g+3)
            __bf16* tr = which == 0 ? ws.qt : which == 1 ? ws.kt : ws.gt;
            for (int e = tid; e < 4 * c; e += 256) {
                const int sg = e & 3, ch = e >> 2;
                const int h = ch / DH;
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int row = j < 4 ? 4 * sg + j : 12 + 4 * sg + j;
                    x[j] = buf[row * cp + ch] * rn[row * heads + h];
                }
                bf16x8 hi, lo;
                split_frag(x, &hi, &lo);
                const int64_t o = (int64_t)ch * mpad + pos0 + 8 * sg;
                *reinterpret_cast<bf16x8*>(tr + o) = hi;
                *reinterpret_cast<bf16x8*>(tr + tr_half + o) = lo;
            }
        }
        __syncthreads();
    }
}

// raw-row epilogue helper: gradient through x_hat = x / max(|x|, eps) for the rows a wave owns.
// Lane layout: grad[b][r] is d(x_hat)[d = 16b + 4g + r] of token column c16; returns d(x) in place.
template <int DH>
__device__ __forceinline__ void through_normalise(const float* __restrict__ xrow, int g, attn::f32x4* grad) {
    constexpr int NB = Geo<DH>::NB;
    float xr[NB][4];
    float nrm = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = 16 * b + 4 * g + r;
            xr[b][r] = d < DH ? xrow[d] : 0.f;
            nrm = fmaf(xr[b][r], xr[b][r], nrm);
        }
    nrm += __shfl_xor(nrm, 16, SEG3D_WAVE);
    nrm += __shfl_xor(nrm, 32, SEG3D_WAVE);
    const float len = sqrtf(nrm);
    const float rinv = 1.0f / fmaxf(len, kNormEps);
    float proj = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xr[b][r] *= rinv;  // x_hat
            proj = fmaf(xr[b][r], grad[b][r], proj);
        }
    proj += __shfl_xor(proj, 16, SEG3D_WAVE);
    proj += __shfl_xor(proj, 32, SEG3D_WAVE);
    const bool clamped = len < kNormEps;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) grad[b][r] = clamped ? grad[b][r] * rinv : (grad[b][r] - xr[b][r] * proj) * rinv;
}

// ------------------------------------------------------------------ shared addressing of the two passes
// One wave per (32-token tile of a window, head): the tile's tokens are the stationary operand (two 16-column
// groups; the second is skipped, wave-uniform, when the tile holds <= 16 tokens), the window's 32-token tiles
// stream past it and every streamed fragment serves both groups.  What depends on (item, head, streamed tile) is
// wave-uniform and lives in scalar base pointers; the lane's share is a constant 32-bit byte offset.  Channel
// slices past DHS are zeroed in the stationary fragments (the streamed ones need no mask); transposed rows
// past DH only feed output rows that are never stored.  Only the last streamed tile masks tokens >= n.
template <int DH>
struct Lanes {
    static constexpr int DHS = Geo<DH>::DHS, KS = Geo<DH>::KS, NB = Geo<DH>::NB;
    uint32_t row[2][KS];  // row-major fragment of streamed token 16u + c16, slice s
    uint32_t tr[NB];      // transposed fragment: row d = 16b + c16, token slots 8g .. 8g+7
    bool slice_ok[KS];
    __device__ Lanes(int g, int c16, int heads, int64_t mpad) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            slice_ok[s] = 32 * s + 8 * g < DHS;
            const int sl = slice_ok[s] ? 32 * s + 8 * g : 0;
#pragma unroll
            for (int u = 0; u < 2; ++u) row[u][s] = (uint32_t)(((u * 16 + c16) * heads * DHS + sl) * 2);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) tr[b] = (uint32_t)(((int64_t)(16 * b + c16 < DH ? 16 * b + c16 : 0) * mpad + 8 * g) * 2);
    }
};

__device__ __forceinline__ void ld2(const char* base, int64_t half_bytes, uint32_t off, bf16x8* hi, bf16x8* lo) {
    *hi = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + off));
    *lo = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + half_bytes + off));
}

// ------------------------------------------------------------------ pass A: dq, dtau
// waves per SIMD the register allocator must leave room for (the kernels are latency-bound: a wave more is worth more
// than a few registers): pass A at dh 24 sits 5 registers above the 3-wave line without it, at dh 48 24 above the 2-wave line
template <int DH>
constexpr int kBwdQWaves = DH <= 24 ? 3 : 2;

template <int DH>
__global__ __launch_bounds__(256, kBwdQWaves<DH>) void attn_bwd_q(BwdWs<DH> ws, const float* __restrict__ q, int ldq,
                                                  const int32_t* __restrict__ tok, const int32_t* __restrict__ win_start,
                                                  const int32_t* __restrict__ win_count, const int32_t* __restrict__ win_tile0,
                                                  const int2* __restrict__ tile_item, int n_items, int heads, int64_t mpad,
                                                  const float* __restrict__ tau, float tau_min, float* __restrict__ dq,
                                                  int lddq, DropoutParams drop) {
    constexpr int DHS = Geo<DH>::DHS, KS = Geo<DH>::KS, NB = Geo<DH>::NB;
    const int lane = threadIdx.x & 63;
    const int it = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    float* tau_slot = ws.tau_part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (it >= n_items) {
        if (lane == 0) *tau_slot = 0.f;
        return;
    }
    const int h = blockIdx.y, g = lane >> 4, c16 = lane & 15;
    const int2 item = tile_item[it];
    const int n = win_count[item.x], start = win_start[item.x];
    const int64_t pos0 = (int64_t)win_tile0[item.x] * 32;
    const int n_kt = (n + 31) >> 5;
    const int q0 = item.y * 32;
    const bool two = n - q0 > 16;
    const int64_t row_half = mpad * heads * DHS * 2, tr_half = (int64_t)heads * DH * mpad * 2;  // bytes
    const Lanes<DH> L(g, c16, heads, mpad);
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const bf16x8 zf = __builtin_bit_cast(bf16x8, zero4);

    // stationary: Q and dO fragments of the two query groups (B operands), LSE and delta of the lane's queries
    bf16x8 q_hi[2][KS], q_lo[2][KS], g_hi[2][KS], g_lo[2][KS];
    float lq[2], dl[2];
    {
        const char* qb = reinterpret_cast<const char*>(ws.qp + ((pos0 + q0) * heads + h) * DHS);
        const char* gb = reinterpret_cast<const char*>(ws.gp + ((pos0 + q0) * heads + h) * DHS);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                ld2(qb, row_half, L.row[j][s], &q_hi[j][s], &q_lo[j][s]);
                ld2(gb, row_half, L.row[j][s], &g_hi[j][s], &g_lo[j][s]);
                if (!L.slice_ok[s]) q_hi[j][s] = q_lo[j][s] = g_hi[j][s] = g_lo[j][s] = zf;
            }
            lq[j] = ws.lp[(int64_t)h * mpad + pos0 + q0 + 16 * j + c16];
            dl[j] = ws.dp[(int64_t)h * mpad + pos0 + q0 + 16 * j + c16];
        }
    }
    f32x4 acc[2][NB];
    float tau_acc[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[j][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int64_t kstep = (int64_t)32 * heads * DHS * 2;
    const char* kb = reinterpret_cast<const char*>(ws.kp + (pos0 * heads + h) * DHS);
    const char* vb = reinterpret_cast<const char*>(ws.vp + (pos0 * heads + h) * DHS);
    const char* ktb = reinterpret_cast<const char*>(ws.kt + (int64_t)h * DH * mpad + pos0);
    for (int t = 0; t < n_kt; ++t) {
        const bool last = t + 1 == n_kt;
        bf16x8 k_hi[2][KS], k_lo[2][KS], v_hi[2][KS], v_lo[2][KS], kt_hi[NB], kt_lo[NB];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                ld2(kb + t * kstep, row_half, L.row[u][s], &k_hi[u][s], &k_lo[u][s]);
                ld2(vb + t * kstep, row_half, L.row[u][s], &v_hi[u][s], &v_lo[u][s]);
            }
#pragma unroll
        for (int b = 0; b < NB; ++b) ld2(ktb + t * 64, tr_half, L.tr[b], &kt_hi[b], &kt_lo[b]);
        auto group = [&](int j) {
            float dsv[8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 s_acc = {0.f, 0.f, 0.f, 0.f}, p_acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    s_acc = mfma3(k_hi[u][s], k_lo[u][s], q_hi[j][s], q_lo[j][s], s_acc);  // S^T[key][query]
                    p_acc = mfma3(v_hi[u][s], v_lo[u][s], g_hi[j][s], g_lo[j][s], p_acc);  // dP^T[key][query]
                }
                if (drop.threshold) {  // dP = D * (dO . v): the forward's dropout factors, regenerated (wave-uniform branch)
                    const int qi_ = q0 + 16 * j + c16;
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        const int kj_ = t * 32 + u * 16 + g * 4 + 2 * r2;
                        const uint32_t bits = dropout_bits(drop, item.x, h, qi_, kj_);
                        p_acc[2 * r2] *= dropout_factor(drop, bits, qi_, kj_);
                        p_acc[2 * r2 + 1] *= dropout_factor(drop, bits, qi_, kj_ + 1);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(s_acc[r] - lq[j]);
                    if (last && t * 32 + u * 16 + g * 4 + r >= n) p = 0.f;
                    const float ds = p * (p_acc[r] - dl[j]);
                    dsv[u * 4 + r] = ds;
                    tau_acc[j] = fmaf(ds, s_acc[r], tau_acc[j]);
                }
            }
            bf16x8 ds_hi, ds_lo;
            split_frag(dsv, &ds_hi, &ds_lo);
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[j][b] = mfma3(kt_hi[b], kt_lo[b], ds_hi, ds_lo, acc[j][b]);  // dQhat^T[d][query] * tau_c
        };
        group(0);
        if (two) group(1);
    }

    const float tau_c = fmaxf(tau[0], tau_min);
    const float inv_tau = 1.0f / tau_c;
    float tau_sum = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j == 1 && !two) break;
        const int qi = q0 + 16 * j + c16;
        const bool valid = qi < n;
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[j][b] = acc[j][b] * inv_tau;
        if (valid) tau_sum += tau_acc[j];
        // every lane of a query column takes part in the shuffles; invalid columns read row 0 and store nothing
        const int32_t token = tok[start + (valid ? qi : 0)];
        through_normalise<DH>(q + (int64_t)token * ldq + h * DH, g, acc[j]);
        if (valid) {
            float* o = dq + (int64_t)token * lddq + h * DH;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int d = 16 * b + 4 * g;
                if (DH % 4 == 0) {
                    if (d < DH) *reinterpret_cast<f32x4*>(o + d) = acc[j][b];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (d + r < DH) o[d + r] = acc[j][b][r];
                }
            }
        }
    }
    // d/dtau: s_nat = s2 * ln2 = c / tau  ->  dL/dtau = -sum(ds * s_nat) / tau   (zero while tau is clamped)
    for (int off = 32; off > 0; off >>= 1) tau_sum += __shfl_xor(tau_sum, off, SEG3D_WAVE);
    if (lane == 0) *tau_slot = tau[0] > tau_min ? -tau_sum * kLn2 / tau_c : 0.f;  // one plain store per wave: no atomics
}

template <int DH>
__global__ __launch_bounds__(256, kBwdQWaves<DH>) void attn_bwd_q_lds(BwdWs<DH> ws, const float* __restrict__ q, int ldq,
                                                  const int32_t* __restrict__ tok, const int32_t* __restrict__ win_start,
                                                  const int32_t* __restrict__ win_count, const int32_t* __restrict__ win_tile0,
                                                  const int2* __restrict__ tile_item, int n_items, int heads, int64_t mpad,
                                                  const float* __restrict__ tau, float tau_min, float* __restrict__ dq,
                                                  int lddq, DropoutParams drop) {
    constexpr int DHS = Geo<DH>::DHS, KS = Geo<DH>::KS, NB = Geo<DH>::NB;
    // One workgroup = four consecutive 32-query tiles of ONE window and one head (see attn_core_fwd_lds): the streamed
    // K, V and K^T fragments of a key tile are loaded once per workgroup and shared through LDS.  Grid: one workgroup
    // per tile, those whose tile is not the first of a group of four return at once.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int2 item0 = tile_item[blockIdx.x];
    if (item0.y & 3) return;
    const int h = blockIdx.y, g = lane >> 4, c16 = lane & 15;
    const int n = win_count[item0.x], start = win_start[item0.x];
    const int64_t pos0 = (int64_t)win_tile0[item0.x] * 32;
    const int n_kt = (n + 31) >> 5;
    const int q0 = (item0.y + wave) * 32;
    const bool active = q0 < n;  // this wave's query tile exists (wave-uniform); its tile index is blockIdx.x + wave
    float* tau_slot = ws.tau_part + (size_t)blockIdx.y * n_items + blockIdx.x + wave;
    const bool two = n - q0 > 16;
    const int64_t row_half = mpad * heads * DHS * 2, tr_half = (int64_t)heads * DH * mpad * 2;  // bytes
    const Lanes<DH> L(g, c16, heads, mpad);
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const bf16x8 zf = __builtin_bit_cast(bf16x8, zero4);

    // stationary: Q and dO fragments of the two query groups (B operands), LSE and delta of the lane's queries
    bf16x8 q_hi[2][KS], q_lo[2][KS], g_hi[2][KS], g_lo[2][KS];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < KS; ++s) q_hi[j][s] = q_lo[j][s] = g_hi[j][s] = g_lo[j][s] = zf;
    float lq[2] = {0.f, 0.f}, dl[2] = {0.f, 0.f};
    if (active) {
        const char* qb = reinterpret_cast<const char*>(ws.qp + ((pos0 + q0) * heads + h) * DHS);
        const char* gb = reinterpret_cast<const char*>(ws.gp + ((pos0 + q0) * heads + h) * DHS);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                ld2(qb, row_half, L.row[j][s], &q_hi[j][s], &q_lo[j][s]);
                ld2(gb, row_half, L.row[j][s], &g_hi[j][s], &g_lo[j][s]);
                if (!L.slice_ok[s]) q_hi[j][s] = q_lo[j][s] = g_hi[j][s] = g_lo[j][s] = zf;
            }
            lq[j] = ws.lp[(int64_t)h * mpad + pos0 + q0 + 16 * j + c16];
            dl[j] = ws.dp[(int64_t)h * mpad + pos0 + q0 + 16 * j + c16];
        }
    }
    f32x4 acc[2][NB];
    float tau_acc[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[j][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int64_t kstep = (int64_t)32 * heads * DHS * 2;
    const char* kb = reinterpret_cast<const char*>(ws.kp + (pos0 * heads + h) * DHS);
    const char* vb = reinterpret_cast<const char*>(ws.vp + (pos0 * heads + h) * DHS);
    const char* ktb = reinterpret_cast<const char*>(ws.kt + (int64_t)h * DH * mpad + pos0);
    // LDS image of one key tile: fragment f (hi, lo) = pieces 2f, 2f + 1; K rows: f = u * KS + s, V rows: 2 KS + u * KS + s,
    // K^T: 4 KS + b; a piece = 64 lanes x 16 B in register order
    constexpr int kFrags = 4 * KS + NB, kPieces = 2 * kFrags, kMine = (kPieces * 64 + 255) / 256;
    __shared__ u32x4 tile_lds[2][kPieces * 64];
    u32x4 st_reg[kMine];
    auto stage_load = [&](int t) {
#pragma unroll
        for (int jj = 0; jj < kMine; ++jj) {
            const int item = jj * 256 + threadIdx.x;
            const int piece = item >> 6, l = item & 63, lg = l >> 4, lc = l & 15;
            const int f = piece >> 1, plane = piece & 1;
            st_reg[jj] = zero4;
            if (f < 4 * KS) {  // row-major K (f < 2 KS) or V fragment of streamed token 16 u + lc, slice s
                const int fr = f < 2 * KS ? f : f - 2 * KS, u = fr / KS, s_ = fr % KS;
                const int sl = 32 * s_ + 8 * lg < DHS ? 32 * s_ + 8 * lg : 0;
                const char* base = (f < 2 * KS ? kb : vb) + t * kstep + plane * row_half;
                st_reg[jj] = *reinterpret_cast<const u32x4*>(base + (uint32_t)(((u * 16 + lc) * heads * DHS + sl) * 2));
            } else if (f < kFrags) {  // transposed K fragment: row d = 16 b + lc, token slots 8 lg .. 8 lg + 7
                const int b_ = f - 4 * KS;
                const int64_t row = 16 * b_ + lc < DH ? 16 * b_ + lc : 0;
                st_reg[jj] = *reinterpret_cast<const u32x4*>(ktb + t * 64 + plane * tr_half + (row * mpad + 8 * lg) * 2);
            }
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int jj = 0; jj < kMine; ++jj) {
            const int item = jj * 256 + threadIdx.x;
            if (kPieces * 64 % 256 == 0 || item < kPieces * 64) tile_lds[buf][item] = st_reg[jj];
        }
    };
    auto frag = [&](int buf, int f, bf16x8* hi, bf16x8* lo) {
        *hi = __builtin_bit_cast(bf16x8, tile_lds[buf][(2 * f) * 64 + lane]);
        *lo = __builtin_bit_cast(bf16x8, tile_lds[buf][(2 * f + 1) * 64 + lane]);
    };
    stage_load(0);
    stage_store(0);
    __syncthreads();
    int buf = 0;
    for (int t = 0; t < n_kt; ++t) {
        const bool last = t + 1 == n_kt;
        if (!last) stage_load(t + 1);  // in flight while this tile is multiplied
        bf16x8 k_hi[2][KS], k_lo[2][KS], v_hi[2][KS], v_lo[2][KS], kt_hi[NB], kt_lo[NB];
        if (active) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    frag(buf, u * KS + s, &k_hi[u][s], &k_lo[u][s]);
                    frag(buf, 2 * KS + u * KS + s, &v_hi[u][s], &v_lo[u][s]);
                }
#pragma unroll
            for (int b = 0; b < NB; ++b) frag(buf, 4 * KS + b, &kt_hi[b], &kt_lo[b]);
        }
        auto group = [&](int j) {
            float dsv[8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 s_acc = {0.f, 0.f, 0.f, 0.f}, p_acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    s_acc = mfma3(k_hi[u][s], k_lo[u][s], q_hi[j][s], q_lo[j][s], s_acc);  // S^T[key][query]
                    p_acc = mfma3(v_hi[u][s], v_lo[u][s], g_hi[j][s], g_lo[j][s], p_acc);  // dP^T[key][query]
                }
                if (drop.threshold) {  // dP = D * (dO . v): the forward's dropout factors, regenerated (wave-uniform branch)
                    const int qi_ = q0 + 16 * j + c16;
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        const int kj_ = t * 32 + u * 16 + g * 4 + 2 * r2;
                        const uint32_t bits = dropout_bits(drop, item0.x, h, qi_, kj_);
                        p_acc[2 * r2] *= dropout_factor(drop, bits, qi_, kj_);
                        p_acc[2 * r2 + 1] *= dropout_factor(drop, bits, qi_, kj_ + 1);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(s_acc[r] - lq[j]);
                    if (last && t * 32 + u * 16 + g * 4 + r >= n) p = 0.f;
                    const float ds = p * (p_acc[r] - dl[j]);
                    dsv[u * 4 + r] = ds;
                    tau_acc[j] = fmaf(ds, s_acc[r], tau_acc[j]);
                }
            }
            bf16x8 ds_hi, ds_lo;
            split_frag(dsv, &ds_hi, &ds_lo);
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[j][b] = mfma3(kt_hi[b], kt_lo[b], ds_hi, ds_lo, acc[j][b]);  // dQhat^T[d][query] * tau_c
        };
        if (active) {
            group(0);
            if (two) group(1);
        }
        if (!last) stage_store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (!active) return;  // no barrier below

    const float tau_c = fmaxf(tau[0], tau_min);
    const float inv_tau = 1.0f / tau_c;
    float tau_sum = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j == 1 && !two) break;
        const int qi = q0 + 16 * j + c16;
        const bool valid = qi < n;
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[j][b] = acc[j][b] * inv_tau;
        if (valid) tau_sum += tau_acc[j];
        // every lane of a query column takes part in the shuffles; invalid columns read row 0 and store nothing
        const int32_t token = tok[start + (valid ? qi : 0)];
        through_normalise<DH>(q + (int64_t)token * ldq + h * DH, g, acc[j]);
        if (valid) {
            float* o = dq + (int64_t)token * lddq + h * DH;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int d = 16 * b + 4 * g;
                if (DH % 4 == 0) {
                    if (d < DH) *reinterpret_cast<f32x4*>(o + d) = acc[j][b];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (d + r < DH) o[d + r] = acc[j][b][r];
                }
            }
        }
    }
    // d/dtau: s_nat = s2 * ln2 = c / tau  ->  dL/dtau = -sum(ds * s_nat) / tau   (zero while tau is clamped)
    for (int off = 32; off > 0; off >>= 1) tau_sum += __shfl_xor(tau_sum, off, SEG3D_WAVE);
    if (lane == 0) *tau_slot = tau[0] > tau_min ? -tau_sum * kLn2 / tau_c : 0.f;  // one plain store per wave: no atomics
}

// dtau = sum of the per-wave partials in a fixed order
__global__ __launch_bounds__(1024) void tau_reduce(const float* __restrict__ part, int count, float* __restrict__ dtau) {
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;  // independent chains: loads overlap
    int i = threadIdx.x;
    for (; i + 3072 < count; i += 4096) {
        v0 += part[i];
        v1 += part[i + 1024];
        v2 += part[i + 2048];
        v3 += part[i + 3072];
    }
    for (; i < count; i += 1024) v0 += part[i];
    float v = (v0 + v1) + (v2 + v3);
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, SEG3D_WAVE);
    __shared__ float w[16];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < 16; ++k) t += w[k];
        dtau[0] = t;
    }
}

// ------------------------------------------------------------------ pass B: dk, dv
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_kv(BwdWs<DH> ws, const float* __restrict__ k, int ldk,
                                                   const int32_t* __restrict__ tok, const int32_t* __restrict__ win_start,
                                                   const int32_t* __restrict__ win_count, const int32_t* __restrict__ win_tile0,
                                                   const int2* __restrict__ tile_item, int n_items, int heads, int64_t mpad,
                                                   float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv,
                                                   DropoutParams drop) {
    constexpr int DHS = Geo<DH>::DHS, KS = Geo<DH>::KS, NB = Geo<DH>::NB;
    const int lane = threadIdx.x & 63;
    const int it = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (it >= n_items) return;
    const int h = blockIdx.y, g = lane >> 4, c16 = lane & 15;
    const int2 item = tile_item[it];
    const int n = win_count[item.x], start = win_start[item.x];
    const int64_t pos0 = (int64_t)win_tile0[item.x] * 32;
    const int n_qt = (n + 31) >> 5;
    const int k0 = item.y * 32;
    const bool two = n - k0 > 16;
    const int64_t row_half = mpad * heads * DHS * 2, tr_half = (int64_t)heads * DH * mpad * 2;  // bytes
    const Lanes<DH> L(g, c16, heads, mpad);
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const bf16x8 zf = __builtin_bit_cast(bf16x8, zero4);

    // stationary: K and V fragments of the two key groups: B operands (k = channel, column = key)
    bf16x8 k_hi[2][KS], k_lo[2][KS], v_hi[2][KS], v_lo[2][KS];
    {
        const char* kb = reinterpret_cast<const char*>(ws.kp + ((pos0 + k0) * heads + h) * DHS);
        const char* vb = reinterpret_cast<const char*>(ws.vp + ((pos0 + k0) * heads + h) * DHS);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                ld2(kb, row_half, L.row[j][s], &k_hi[j][s], &k_lo[j][s]);
                ld2(vb, row_half, L.row[j][s], &v_hi[j][s], &v_lo[j][s]);
                if (!L.slice_ok[s]) k_hi[j][s] = k_lo[j][s] = v_hi[j][s] = v_lo[j][s] = zf;
            }
    }
    f32x4 dk_acc[2][NB], dv_acc[2][NB];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            dk_acc[j][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dv_acc[j][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    const int64_t qstep = (int64_t)32 * heads * DHS * 2;
    const char* qb = reinterpret_cast<const char*>(ws.qp + (pos0 * heads + h) * DHS);
    const char* gb = reinterpret_cast<const char*>(ws.gp + (pos0 * heads + h) * DHS);
    const char* qtb = reinterpret_cast<const char*>(ws.qt + (int64_t)h * DH * mpad + pos0);
    const char* gtb = reinterpret_cast<const char*>(ws.gt + (int64_t)h * DH * mpad + pos0);
    const float* lpb = ws.lp + (int64_t)h * mpad + pos0 + 4 * g;  // + 32 t + 16 u: the accumulator's 4 query rows
    const float* dpb = ws.dp + (int64_t)h * mpad + pos0 + 4 * g;
    for (int t = 0; t < n_qt; ++t) {
        const bool last = t + 1 == n_qt;
        bf16x8 q_hi[2][KS], q_lo[2][KS], g_hi[2][KS], g_lo[2][KS], qt_hi[NB], qt_lo[NB], gt_hi[NB], gt_lo[NB];
        f32x4 lq[2], dl[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                ld2(qb + t * qstep, row_half, L.row[u][s], &q_hi[u][s], &q_lo[u][s]);
                ld2(gb + t * qstep, row_half, L.row[u][s], &g_hi[u][s], &g_lo[u][s]);
            }
            lq[u] = *reinterpret_cast<const f32x4*>(lpb + t * 32 + u * 16);
            dl[u] = *reinterpret_cast<const f32x4*>(dpb + t * 32 + u * 16);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            ld2(qtb + t * 64, tr_half, L.tr[b], &qt_hi[b], &qt_lo[b]);
            ld2(gtb + t * 64, tr_half, L.tr[b], &gt_hi[b], &gt_lo[b]);
        }
        auto group = [&](int j) {
            float pv[8], dsv[8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 s_acc = {0.f, 0.f, 0.f, 0.f}, p_acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    s_acc = mfma3(q_hi[u][s], q_lo[u][s], k_hi[j][s], k_lo[j][s], s_acc);  // S[query][key]
                    p_acc = mfma3(g_hi[u][s], g_lo[u][s], v_hi[j][s], v_lo[j][s], p_acc);  // dP[query][key]
                }
                // rows of this accumulator: queries 32t + 16u + 4g + r (4 consecutive positions)
                float dfac[4] = {1.f, 1.f, 1.f, 1.f};
                if (drop.threshold) {  // the forward's dropout factors of (query 32t + 16u + 4g + r, key k0 + 16j + c16)
                    const int kj_ = k0 + 16 * j + c16;
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        const int qi_ = t * 32 + u * 16 + g * 4 + 2 * r2;
                        const uint32_t bits = dropout_bits(drop, item.x, h, qi_, kj_);
                        dfac[2 * r2] = dropout_factor(drop, bits, qi_, kj_);
                        dfac[2 * r2 + 1] = dropout_factor(drop, bits, qi_ + 1, kj_);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(s_acc[r] - lq[u][r]);
                    if (last && t * 32 + u * 16 + g * 4 + r >= n) p = 0.f;
                    pv[u * 4 + r] = p * dfac[r];                              // dV = (D * P)^T dO
                    dsv[u * 4 + r] = p * (dfac[r] * p_acc[r] - dl[u][r]);     // dS = P * (D * dP - delta)
                }
            }
            bf16x8 p_hi, p_lo, ds_hi, ds_lo;
            split_frag(pv, &p_hi, &p_lo);
            split_frag(dsv, &ds_hi, &ds_lo);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                dv_acc[j][b] = mfma3(gt_hi[b], gt_lo[b], p_hi, p_lo, dv_acc[j][b]);    // dV^T[d][key]
                dk_acc[j][b] = mfma3(qt_hi[b], qt_lo[b], ds_hi, ds_lo, dk_acc[j][b]);  // dKhat^T[d][key] / ln2
            }
        };
        group(0);
        if (two) group(1);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j == 1 && !two) break;
        const int ki = k0 + 16 * j + c16;
        const bool valid = ki < n;
#pragma unroll
        for (int b = 0; b < NB; ++b) dk_acc[j][b] = dk_acc[j][b] * kLn2;  // Q~ = q_hat * log2e / tau  ->  q_hat / tau = Q~ * ln2
        const int32_t token = tok[start + (valid ? ki : 0)];
        through_normalise<DH>(k + (int64_t)token * ldk + h * DH, g, dk_acc[j]);
        if (valid) {
            float* ok_ = dk + (int64_t)token * lddk + h * DH;
            float* ov = dv + (int64_t)token * lddv + h * DH;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int d = 16 * b + 4 * g;
                if (DH % 4 == 0) {
                    if (d < DH) {
                        *reinterpret_cast<f32x4*>(ok_ + d) = dk_acc[j][b];
                        *reinterpret_cast<f32x4*>(ov + d) = dv_acc[j][b];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (d + r < DH) {
                            ok_[d + r] = dk_acc[j][b][r];
                            ov[d + r] = dv_acc[j][b][r];
                        }
                }
            }
        }
    }
}

template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_kv_lds(BwdWs<DH> ws, const float* __restrict__ k, int ldk,
                                                   const int32_t* __restrict__ tok, const int32_t* __restrict__ win_start,
                                                   const int32_t* __restrict__ win_count, const int32_t* __restrict__ win_tile0,
                                                   const int2* __restrict__ tile_item, int n_items, int heads, int64_t mpad,
                                                   float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv,
                                                   DropoutParams drop) {
    constexpr int DHS = Geo<DH>::DHS, KS = Geo<DH>::KS, NB = Geo<DH>::NB;
    // One workgroup = four consecutive 32-key tiles of ONE window and one head; the streamed Q, dO, Q^T, dO^T fragments
    // of a query tile are loaded once per workgroup and shared through LDS (see attn_core_fwd_lds).
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int2 item0 = tile_item[blockIdx.x];
    if (item0.y & 3) return;
    (void)n_items;
    const int h = blockIdx.y, g = lane >> 4, c16 = lane & 15;
    const int n = win_count[item0.x], start = win_start[item0.x];
    const int64_t pos0 = (int64_t)win_tile0[item0.x] * 32;
    const int n_qt = (n + 31) >> 5;
    const int k0 = (item0.y + wave) * 32;
    const bool active = k0 < n;  // this wave's key tile exists (wave-uniform)
    const bool two = n - k0 > 16;
    const int64_t row_half = mpad * heads * DHS * 2, tr_half = (int64_t)heads * DH * mpad * 2;  // bytes
    const Lanes<DH> L(g, c16, heads, mpad);
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const bf16x8 zf = __builtin_bit_cast(bf16x8, zero4);

    // stationary: K and V fragments of the two key groups: B operands (k = channel, column = key)
    bf16x8 k_hi[2][KS], k_lo[2][KS], v_hi[2][KS], v_lo[2][KS];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < KS; ++s) k_hi[j][s] = k_lo[j][s] = v_hi[j][s] = v_lo[j][s] = zf;
    if (active) {
        const char* kb = reinterpret_cast<const char*>(ws.kp + ((pos0 + k0) * heads + h) * DHS);
        const char* vb = reinterpret_cast<const char*>(ws.vp + ((pos0 + k0) * heads + h) * DHS);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                ld2(kb, row_half, L.row[j][s], &k_hi[j][s], &k_lo[j][s]);
                ld2(vb, row_half, L.row[j][s], &v_hi[j][s], &v_lo[j][s]);
                if (!L.slice_ok[s]) k_hi[j][s] = k_lo[j][s] = v_hi[j][s] = v_lo[j][s] = zf;
            }
    }
    f32x4 dk_acc[2][NB], dv_acc[2][NB];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            dk_acc[j][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dv_acc[j][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    const int64_t qstep = (int64_t)32 * heads * DHS * 2;
    const char* qb = reinterpret_cast<const char*>(ws.qp + (pos0 * heads + h) * DHS);
    const char* gb = reinterpret_cast<const char*>(ws.gp + (pos0 * heads + h) * DHS);
    const char* qtb = reinterpret_cast<const char*>(ws.qt + (int64_t)h * DH * mpad + pos0);
    const char* gtb = reinterpret_cast<const char*>(ws.gt + (int64_t)h * DH * mpad + pos0);
    const float* lpb = ws.lp + (int64_t)h * mpad + pos0 + 4 * g;  // + 32 t + 16 u: the accumulator's 4 query rows
    const float* dpb = ws.dp + (int64_t)h * mpad + pos0 + 4 * g;
    // LDS image of one query tile: fragment f (hi, lo) = pieces 2f, 2f + 1; Q rows: f = u * KS + s, dO rows: 2 KS + ..,
    // Q^T: 4 KS + b, dO^T: 4 KS + NB + b; LSE / delta (8 floats per lane) stay direct loads
    constexpr int kFrags = 4 * KS + 2 * NB, kPieces = 2 * kFrags, kMine = (kPieces * 64 + 255) / 256;
    __shared__ u32x4 tile_lds[2][kPieces * 64];
    u32x4 st_reg[kMine];
    auto stage_load = [&](int t) {
#pragma unroll
        for (int jj = 0; jj < kMine; ++jj) {
            const int item = jj * 256 + threadIdx.x;
            const int piece = item >> 6, l = item & 63, lg = l >> 4, lc = l & 15;
            const int f = piece >> 1, plane = piece & 1;
            st_reg[jj] = zero4;
            if (f < 4 * KS) {  // row-major Q (f < 2 KS) or dO fragment of streamed token 16 u + lc, slice s
                const int fr = f < 2 * KS ? f : f - 2 * KS, u = fr / KS, s_ = fr % KS;
                const int sl = 32 * s_ + 8 * lg < DHS ? 32 * s_ + 8 * lg : 0;
                const char* base = (f < 2 * KS ? qb : gb) + t * qstep + plane * row_half;
                st_reg[jj] = *reinterpret_cast<const u32x4*>(base + (uint32_t)(((u * 16 + lc) * heads * DHS + sl) * 2));
            } else if (f < kFrags) {  // transposed Q (f < 4 KS + NB) or dO fragment: row d = 16 b + lc, token slots 8 lg ..
                const int fb = f - 4 * KS, b_ = fb < NB ? fb : fb - NB;
                const int64_t row = 16 * b_ + lc < DH ? 16 * b_ + lc : 0;
                const char* base = (fb < NB ? qtb : gtb) + t * 64 + plane * tr_half;
                st_reg[jj] = *reinterpret_cast<const u32x4*>(base + (row * mpad + 8 * lg) * 2);
            }
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int jj = 0; jj < kMine; ++jj) {
            const int item = jj * 256 + threadIdx.x;
            if (kPieces * 64 % 256 == 0 || item < kPieces * 64) tile_lds[buf][item] = st_reg[jj];
        }
    };
    auto frag = [&](int buf, int f, bf16x8* hi, bf16x8* lo) {
        *hi = __builtin_bit_cast(bf16x8, tile_lds[buf][(2 * f) * 64 + lane]);
        *lo = __builtin_bit_cast(bf16x8, tile_lds[buf][(2 * f + 1) * 64 + lane]);
    };
    stage_load(0);
    stage_store(0);
    __syncthreads();
    int buf = 0;
    for (int t = 0; t < n_qt; ++t) {
        const bool last = t + 1 == n_qt;
        if (!last) stage_load(t + 1);  // in flight while this tile is multiplied
        bf16x8 q_hi[2][KS], q_lo[2][KS], g_hi[2][KS], g_lo[2][KS], qt_hi[NB], qt_lo[NB], gt_hi[NB], gt_lo[NB];
        f32x4 lq[2], dl[2];
        if (active) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    frag(buf, u * KS + s, &q_hi[u][s], &q_lo[u][s]);
                    frag(buf, 2 * KS + u * KS + s, &g_hi[u][s], &g_lo[u][s]);
                }
                lq[u] = *reinterpret_cast<const f32x4*>(lpb + t * 32 + u * 16);
                dl[u] = *reinterpret_cast<const f32x4*>(dpb + t * 32 + u * 16);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                frag(buf, 4 * KS + b, &qt_hi[b], &qt_lo[b]);
                frag(buf, 4 * KS + NB + b, &gt_hi[b], &gt_lo[b]);
            }
        }
        auto group = [&](int j) {
            float pv[8], dsv[8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 s_acc = {0.f, 0.f, 0.f, 0.f}, p_acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    s_acc = mfma3(q_hi[u][s], q_lo[u][s], k_hi[j][s], k_lo[j][s], s_acc);  // S[query][key]
                    p_acc = mfma3(g_hi[u][s], g_lo[u][s], v_hi[j][s], v_lo[j][s], p_acc);  // dP[query][key]
                }
                // rows of this accumulator: queries 32t + 16u + 4g + r (4 consecutive positions)
                float dfac[4] = {1.f, 1.f, 1.f, 1.f};
                if (drop.threshold) {  // the forward's dropout factors of (query 32t + 16u + 4g + r, key k0 + 16j + c16)
                    const int kj_ = k0 + 16 * j + c16;
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        const int qi_ = t * 32 + u * 16 + g * 4 + 2 * r2;
                        const uint32_t bits = dropout_bits(drop, item0.x, h, qi_, kj_);
                        dfac[2 * r2] = dropout_factor(drop, bits, qi_, kj_);
                        dfac[2 * r2 + 1] = dropout_factor(drop, bits, qi_ + 1, kj_);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(s_acc[r] - lq[u][r]);
                    if (last && t * 32 + u * 16 + g * 4 + r >= n) p = 0.f;
                    pv[u * 4 + r] = p * dfac[r];                              // dV = (D * P)^T dO
                    dsv[u * 4 + r] = p * (dfac[r] * p_acc[r] - dl[u][r]);     // dS = P * (D * dP - delta)
                }
            }
            bf16x8 p_hi, p_lo, ds_hi, ds_lo;
            split_frag(pv, &p_hi, &p_lo);
            split_frag(dsv, &ds_hi, &ds_lo);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                dv_acc[j][b] = mfma3(gt_hi[b], gt_lo[b], p_hi, p_lo, dv_acc[j][b]);    // dV^T[d][key]
                dk_acc[j][b] = mfma3(qt_hi[b], qt_lo[b], ds_hi, ds_lo, dk_acc[j][b]);  // dKhat^T[d][key] / ln2
            }
        };
        if (active) {
            group(0);
            if (two) group(1);
        }
        if (!last) stage_store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (!active) return;  // no barrier below
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j == 1 && !two) break;
        const int ki = k0 + 16 * j + c16;
        const bool valid = ki < n;
#pragma unroll
        for (int b = 0; b < NB; ++b) dk_acc[j][b] = dk_acc[j][b] * kLn2;  // Q~ = q_hat * log2e / tau  ->  q_hat / tau = Q~ * ln2
        const int32_t token = tok[start + (valid ? ki : 0)];
        through_normalise<DH>(k + (int64_t)token * ldk + h * DH, g, dk_acc[j]);
        if (valid) {
            float* ok_ = dk + (int64_t)token * lddk + h * DH;
            float* ov = dv + (int64_t)token * lddv + h * DH;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int d = 16 * b + 4 * g;
                if (DH % 4 == 0) {
                    if (d < DH) {
                        *reinterpret_cast<f32x4*>(ok_ + d) = dk_acc[j][b];
                        *reinterpret_cast<f32x4*>(ov + d) = dv_acc[j][b];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (d + r < DH) {
                            ok_[d + r] = dk_acc[j][b][r];
                            ov[d + r] = dv_acc[j][b][r];
                        }
                }
            }
        }
    }
}

template <int DH>
int run_bwd(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const float* out, const float* dout,
            const float* lse, const int32_t* tok, const int32_t* win_start, const int32_t* win_count,
            const int32_t* win_tile0, const int2* tile_item, int n_tiles, const int2* qg_item, int n_qg, int heads,
            const float* tau, float tau_min, float* dq, float* dk, float* dv, int lddq, int lddk, int lddv, float* dtau,
            void* workspace, const DropoutParams& drop, hipStream_t st) {
    const int64_t mpad = (int64_t)n_tiles * 32;
    BwdWs<DH> ws(workspace, mpad, heads);
    const int c = heads * DH;
    const size_t smem = (size_t)(32 * (c + 1) + 32 * heads) * sizeof(float) + 32 * sizeof(int32_t);
    hipLaunchKernelGGL(attn_prepare_bwd<DH>, dim3((unsigned)n_tiles), dim3(256), smem, st, q, k, v, ldq, ldk, ldv, dout, out,
                       lse, tok, win_start, win_count, win_tile0, tile_item, heads, mpad, tau, tau_min, ws);
    SEG3D_CHECK_LAUNCH();
    (void)qg_item;
    (void)n_qg;
    // default: the streamed fragments of a tile are staged once per four stationary tiles through LDS (headline scene, per
    // layer: pass A 276 -> 208 us at dh 24, 169 -> 141 us at dh 48; pass B 328 -> 305 us at dh 24);
    // SEG3D_ATTN_LDS=0 selects the wave-independent kernels for A/B runs
    static const int lds_env = getenv("SEG3D_ATTN_LDS") ? atoi(getenv("SEG3D_ATTN_LDS")) : 1;
    if (lds_env) {
        dim3 grid((unsigned)n_tiles, (unsigned)heads);
        hipLaunchKernelGGL(attn_bwd_q_lds<DH>, grid, dim3(256), 0, st, ws, q, ldq, tok, win_start, win_count, win_tile0,
                           tile_item, n_tiles, heads, mpad, tau, tau_min, dq, lddq, drop);
        SEG3D_CHECK_LAUNCH();
        hipLaunchKernelGGL(tau_reduce, dim3(1), dim3(1024), 0, st, ws.tau_part, (int)(n_tiles * heads), dtau);
        SEG3D_CHECK_LAUNCH();
        if (DH <= 24) {
            hipLaunchKernelGGL(attn_bwd_kv_lds<DH>, grid, dim3(256), 0, st, ws, k, ldk, tok, win_start, win_count, win_tile0,
                               tile_item, n_tiles, heads, mpad, dk, lddk, dv, lddv, drop);
        } else {  // dh 48: one wave per SIMD either way, and the 56 KiB LDS image costs more than it saves (211 vs 229 us)
            dim3 grid4((unsigned)((n_tiles + 3) / 4), (unsigned)heads);
            hipLaunchKernelGGL(attn_bwd_kv<DH>, grid4, dim3(256), 0, st, ws, k, ldk, tok, win_start, win_count, win_tile0,
                               tile_item, n_tiles, heads, mpad, dk, lddk, dv, lddv, drop);
        }
        SEG3D_CHECK_LAUNCH();
        return SEG3D_OK;
    }
    dim3 grid((unsigned)((n_tiles + 3) / 4), (unsigned)heads);
    hipLaunchKernelGGL(attn_bwd_q<DH>, grid, dim3(256), 0, st, ws, q, ldq, tok, win_start, win_count, win_tile0, tile_item,
                       n_tiles, heads, mpad, tau, tau_min, dq, lddq, drop);
    SEG3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(tau_reduce, dim3(1), dim3(1024), 0, st, ws.tau_part, (int)(grid.x * 4 * grid.y), dtau);
    SEG3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_kv<DH>, grid, dim3(256), 0, st, ws, k, ldk, tok, win_start, win_count, win_tile0, tile_item,
                       n_tiles, heads, mpad, dk, lddk, dv, lddv, drop);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

size_t bwd_bytes(int n_tiles, int heads, int dh) {
    const int64_t mpad = (int64_t)n_tiles * 32;
    switch (dh) {
        case 6: return BwdWs<6>::total(mpad, heads);
        case 12: return BwdWs<12>::total(mpad, heads);
        case 24: return BwdWs<24>::total(mpad, heads);
        case 48: return BwdWs<48>::total(mpad, heads);
        default: return 0;
    }
}

}  // namespace

extern "C" {

// which kernels the backward takes for this head geometry: 0 fused (two flash-style passes), 1 vector-ALU (dh 6),
// 2 legacy three-launch MFMA passes with prepared operands
static int attn_bwd_path(int heads, int dh) {
    static const bool fused_bwd_off = getenv("SEG3D_ATTN_FUSED_BWD") && atoi(getenv("SEG3D_ATTN_FUSED_BWD")) == 0;  // A/B
    if (!fused_bwd_off && attn_use_fused(heads, dh) && attn_fused_bwd_supported(heads, dh)) return 0;
    // narrow heads: the vector-ALU kernels win at dh 6 (364 vs 1552 us per layer on the headline scene), the MFMA passes
    // at dh 12 (1210 vs 1547 us)
    if (attn_use_small(heads, dh) && dh < 12) return 1;
    return 2;
}

// Bytes for the path seg3d_window_attn_fwd / _bwd will actually take with these arguments: the fused kernels keep
// nothing in HBM but one tau-gradient partial per wave; only the legacy kernels (SEG3D_ATTN_FUSED=0, head counts the fused
// kernels do not take) stage prepared operands -- gigabytes on a 2 M-point scene, which the callers used to allocate per
// call whatever the path.
size_t seg3d_window_attn_workspace_bytes(int64_t m, int32_t n_tiles, int32_t heads, int32_t dh) {
    if (m < 0 || heads <= 0 || n_tiles < 0) return 0;
    size_t fwd = 0, bwd = 0;
    if (!attn_use_fused(heads, dh) && !attn_use_small(heads, dh)) fwd = attn_mfma_workspace_bytes(n_tiles, heads, dh);
    switch (attn_bwd_path(heads, dh)) {
        case 0: bwd = attn_fused_bwd_workspace_bytes(n_tiles, n_tiles, heads, dh); break;  // chunks <= tiles
        case 1: bwd = (size_t)n_tiles * sizeof(float); break;
        default: bwd = bwd_bytes(n_tiles, heads, dh);
    }
    return (fwd > bwd ? fwd : bwd) + 256;
}

int seg3d_window_attn_bwd(const float* q, const float* k, const float* v, int32_t ldq, int32_t ldk, int32_t ldv,
                          const float* out, const float* dout, const float* lse, const int32_t* tok,
                          const int32_t* win_start, const int32_t* win_count, const int32_t* win_tile0,
                          const int32_t* tile_item, int32_t n_tiles, const int32_t* qg_item, int32_t n_qgroups, int64_t m,
                          int32_t n_windows, int32_t heads, int32_t dh, const float* tau, float tau_min, float dropout_p,
                          uint64_t dropout_seed, float* dq, float* dk, float* dv, int32_t lddq, int32_t lddk, int32_t lddv,
                          float* dtau, void* workspace, size_t workspace_bytes, void* stream) {
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return SEG3D_EINVAL;
    const DropoutParams drop = make_dropout(dropout_p, dropout_seed);  // same mask as the forward, given the same two values
    if (m == 0 || n_windows == 0 || n_tiles == 0 || n_qgroups == 0) {
        if (dtau) SEG3D_CHECK_HIP(hipMemsetAsync(dtau, 0, sizeof(float), as_stream(stream)));
        return SEG3D_OK;
    }
    if (!q || !k || !v || !out || !dout || !lse || !tok || !win_start || !win_count || !win_tile0 || !tile_item ||
        !qg_item || m < 0 || n_windows < 0 || n_tiles < 0 || n_qgroups < 0 || heads <= 0 || heads > 16 || !tau || !dq ||
        !dk || !dv || !dtau || !workspace)
        return SEG3D_EINVAL;
    if (((ldq | ldk | ldv | lddq | lddk | lddv) & 3) ||
        ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
          reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(dk) | reinterpret_cast<uintptr_t>(dv)) & 15))
        return SEG3D_EINVAL;  // rows are gathered / stored in 16-B pieces
    const int path = attn_bwd_path(heads, dh);
    if (path == 0) {
        if (workspace_bytes < attn_fused_bwd_workspace_bytes(n_tiles, n_qgroups, heads, dh)) return SEG3D_EWORKSPACE;
        return attn_fused_bwd_launch(q, k, v, ldq, ldk, ldv, out, dout, lse, tok, win_start, win_count, tile_item, n_tiles,
                                     qg_item, n_qgroups, heads, dh, tau, tau_min, dq, dk, dv, lddq, lddk, lddv, dtau,
                                     workspace, drop, as_stream(stream));
    }
    if (path == 1) {
        if (workspace_bytes < (size_t)n_tiles * sizeof(float)) return SEG3D_EWORKSPACE;
        return attn_small_bwd_launch(q, k, v, ldq, ldk, ldv, out, dout, lse, tok, win_start, win_count, tile_item, n_tiles,
                                     heads, dh, tau, tau_min, dq, dk, dv, lddq, lddk, lddv, dtau, workspace, drop,
                                     as_stream(stream));
    }
    const size_t need = bwd_bytes(n_tiles, heads, dh);
    if (need == 0) return SEG3D_EINVAL;
    if (workspace_bytes < need) return SEG3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    const int2* ti = reinterpret_cast<const int2*>(tile_item);
    const int2* qi = reinterpret_cast<const int2*>(qg_item);
#define SEG3D_BWD_CASE(D)                                                                                              \
    case D:                                                                                                            \
        return run_bwd<D>(q, k, v, ldq, ldk, ldv, out, dout, lse, tok, win_start, win_count, win_tile0, ti, n_tiles, qi,  \
                          n_qgroups, heads, tau, tau_min, dq, dk, dv, lddq, lddk, lddv, dtau, workspace, drop, st)
    switch (dh) {
        SEG3D_BWD_CASE(6);
        SEG3D_BWD_CASE(12);
        SEG3D_BWD_CASE(24);
        SEG3D_BWD_CASE(48);
        default: return SEG3D_EINVAL;
    }
#undef SEG3D_BWD_CASE
}

}  // extern "C"
