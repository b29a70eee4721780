// a19-a21 backward on the matrix cores: ragged sparse-window cosine attention, split-bf16 MFMA.
// Gradient of  out_i = softmax_j(<q_i/|q_i|, k_j/|k_j|> / max(tau, tau_min)) . v_j  (cosine_msa.py:115-177)
// w.r.t. the raw q, k, v (through the L2 normalisation) and tau.
//
// Three launches per layer, flash-style recompute (no score tensor is ever stored):
//  1. attn_prepare_bwd (one workgroup per 32-token tile): gathers q, k, v, dO rows, normalises q/k, splits to
//     bf16 hi/lo and writes row-major copies (Qp, Kp, Vp, Gp) and tile-permuted transposed copies (Qt, Kt,
//     Gt), plus per (token, head) the log2-domain LSE and delta = <dO, O>.
//  2. attn_bwd_q: one wave per (window, 16-query group, head), loop over 32-key tiles:
//        S^T = K.Q^T, dP^T = V.dO^T, P = exp2(S - L), dS = P (dP - delta), dQ^T += K^T.dS^T, dtau += <dS, S>
//  3. attn_bwd_kv: one wave per (window, 16-key group, head), loop over 32-query tiles:
//        S = Q.K^T, dP = dO.V^T, P, dS as above, dV^T += dO^T.P, dK^T += Q^T.dS
//  The accumulator layout of the first two products is exactly the B-operand layout of the last ones (tokens
//  permuted inside a tile, attn_common.hpp), so nothing moves between lanes and no LDS is used.  Each wave
//  owns its output rows: no atomics except one float add per wave for the scalar tau gradient.
#include "attn_common.hpp"

size_t attn_mfma_workspace_bytes(int n_tiles, int heads, int dh);  // attention_mfma.hip

namespace {

using namespace attn;

template <int DH>
struct BwdWs {
    __bf16 *qp, *kp, *vp, *gp;  // row-major  [mpad][heads][DHS], hi block then lo block
    __bf16 *qt, *kt, *gt;       // transposed [heads][DH][mpad],  hi block then lo block
    float *lp, *dp;             // [mpad][heads]: log2-domain LSE, delta
    float* tau_part;            // [256] partial tau gradients (one address would serialise ~1e5 atomics)
    static size_t row_bytes(int64_t mpad, int heads) {
        return align_up((size_t)mpad * heads * Geo<DH>::DHS * 2 * sizeof(__bf16), 256);
    }
    static size_t tr_bytes(int64_t mpad, int heads) { return align_up((size_t)heads * DH * mpad * 2 * sizeof(__bf16), 256); }
    static size_t f_bytes(int64_t mpad, int heads) { return align_up((size_t)mpad * heads * sizeof(float), 256); }
    static size_t total(int64_t mpad, int heads) {
        return 4 * row_bytes(mpad, heads) + 3 * tr_bytes(mpad, heads) + 2 * f_bytes(mpad, heads) + 1024;
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
        tau_part = reinterpret_cast<float*>(take(1024));
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
        for (int e = tid; e < 32 * c; e += 256) {
            const int row = e / c, col = e - row * c;
            const int t = trow[row];
            buf[row * cp + col] = t >= 0 ? src[(int64_t)t * ld + col] : 0.f;
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
                    ws.dp[(pos0 + row) * heads + h] = s;
                    ws.lp[(pos0 + row) * heads + h] = t >= 0 ? lse[(int64_t)t * heads + h] * kLog2e : 0.f;
                }
            }
        }
        __syncthreads();
        __bf16* rm = which == 0 ? ws.qp : which == 1 ? ws.kp : which == 2 ? ws.vp : ws.gp;
        for (int e = tid; e < 32 * heads * DHS; e += 256) {
            const int ds = e % DHS, h = (e / DHS) % heads, row = e / (DHS * heads);
            const float x = ds < DH ? buf[row * cp + h * DH + ds] * rn[row * heads + h] : 0.f;
            __bf16 hi, lo;
            split1(x, &hi, &lo);
            const int64_t o = ((pos0 + row) * heads + h) * DHS + ds;
            rm[o] = hi;
            rm[row_half + o] = lo;
        }
        if (which != 2) {
            __bf16* tr = which == 0 ? ws.qt : which == 1 ? ws.kt : ws.gt;
            for (int e = tid; e < 32 * c; e += 256) {
                const int row = e & 31, ch = e >> 5;
                const int h = ch / DH;
                __bf16 hi, lo;
                split1(buf[row * cp + ch] * rn[row * heads + h], &hi, &lo);
                const int64_t o = (int64_t)ch * mpad + pos0 + perm_slot(row);
                tr[o] = hi;
                tr[tr_half + o] = lo;
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

// ------------------------------------------------------------------ pass A: dq, dtau
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_q(BwdWs<DH> ws, const float* __restrict__ q, int ldq,
                                                  const int32_t* __restrict__ tok, const int32_t* __restrict__ win_start,
                                                  const int32_t* __restrict__ win_count, const int32_t* __restrict__ win_tile0,
                                                  const int2* __restrict__ qg_item, int n_items, int heads, int64_t mpad,
                                                  const float* __restrict__ tau, float tau_min, float* __restrict__ dq,
                                                  int lddq) {
    constexpr int DHS = Geo<DH>::DHS, KS = Geo<DH>::KS, NB = Geo<DH>::NB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int it = blockIdx.x * 4 + wave;
    if (it >= n_items) return;
    const int h = blockIdx.y, g = lane >> 4, c16 = lane & 15;
    const int2 item = qg_item[it];
    const int n = win_count[item.x], start = win_start[item.x];
    const int64_t pos0 = (int64_t)win_tile0[item.x] * 32;
    const int n_tiles = (n + 31) >> 5;
    const int64_t row_half = mpad * heads * DHS, tr_half = (int64_t)heads * DH * mpad;
    const int qi = item.y * 16 + c16;
    const int64_t qpos = pos0 + qi;

    bf16x8 q_hi[KS], q_lo[KS], g_hi[KS], g_lo[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bool ok = 32 * s + 8 * g < DHS;
        const int64_t o = (qpos * heads + h) * DHS + 32 * s + 8 * g;
        load_frag(ws.qp, o, row_half, ok, &q_hi[s], &q_lo[s]);
        load_frag(ws.gp, o, row_half, ok, &g_hi[s], &g_lo[s]);
    }
    const float lq = ws.lp[qpos * heads + h], dq_delta = ws.dp[qpos * heads + h];

    f32x4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float tau_acc = 0.f;

    for (int t = 0; t < n_tiles; ++t) {
        float dsv[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 s_acc = {0.f, 0.f, 0.f, 0.f}, p_acc = {0.f, 0.f, 0.f, 0.f};
            const int64_t krow = pos0 + t * 32 + u * 16 + c16;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bool ok = 32 * s + 8 * g < DHS;
                const int64_t o = (krow * heads + h) * DHS + 32 * s + 8 * g;
                bf16x8 a_hi, a_lo;
                load_frag(ws.kp, o, row_half, ok, &a_hi, &a_lo);
                s_acc = mfma3(a_hi, a_lo, q_hi[s], q_lo[s], s_acc);   // S^T[key][query]
                load_frag(ws.vp, o, row_half, ok, &a_hi, &a_lo);
                p_acc = mfma3(a_hi, a_lo, g_hi[s], g_lo[s], p_acc);   // dP^T[key][query]
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = t * 32 + u * 16 + g * 4 + r;
                const float p = key < n ? __builtin_amdgcn_exp2f(s_acc[r] - lq) : 0.f;
                const float ds = p * (p_acc[r] - dq_delta);
                dsv[u * 4 + r] = ds;
                tau_acc = fmaf(ds, s_acc[r], tau_acc);
            }
        }
        bf16x8 ds_hi, ds_lo;
        split_frag(dsv, &ds_hi, &ds_lo);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int d = 16 * b + c16;
            bf16x8 a_hi, a_lo;
            load_frag(ws.kt, ((int64_t)h * DH + d) * mpad + pos0 + t * 32 + 8 * g, tr_half, d < DH, &a_hi, &a_lo);
            acc[b] = mfma3(a_hi, a_lo, ds_hi, ds_lo, acc[b]);         // dQhat^T[d][query] * tau_c
        }
    }

    const float tau_c = fmaxf(tau[0], tau_min);
    if (qi < n) {
        const float inv_tau = 1.0f / tau_c;
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = acc[b] * inv_tau;
    } else {
        tau_acc = 0.f;
    }
    // every lane of a query column takes part in the shuffles; invalid columns read row 0 and store nothing
    const int32_t token = tok[start + (qi < n ? qi : 0)];
    through_normalise<DH>(q + (int64_t)token * ldq + h * DH, g, acc);
    if (qi < n) {
        float* o = dq + (int64_t)token * lddq + h * DH;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 16 * b + 4 * g + r;
                if (d < DH) o[d] = acc[b][r];
            }
    }
    // d/dtau: s_nat = s2 * ln2 = c / tau  ->  dL/dtau = -sum(ds * s_nat) / tau   (zero while tau is clamped)
    for (int off = 32; off > 0; off >>= 1) tau_acc += __shfl_xor(tau_acc, off, SEG3D_WAVE);
    if (lane == 0 && tau[0] > tau_min && tau_acc != 0.f) atomicAdd(&ws.tau_part[it & 255], -tau_acc * kLn2 / tau_c);
}

__global__ __launch_bounds__(256) void tau_reduce(const float* __restrict__ part, float* __restrict__ dtau) {
    float v = part[threadIdx.x];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, SEG3D_WAVE);
    __shared__ float w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) dtau[0] += w[0] + w[1] + w[2] + w[3];
}

// ------------------------------------------------------------------ pass B: dk, dv
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_kv(BwdWs<DH> ws, const float* __restrict__ k, int ldk,
                                                   const int32_t* __restrict__ tok, const int32_t* __restrict__ win_start,
                                                   const int32_t* __restrict__ win_count, const int32_t* __restrict__ win_tile0,
                                                   const int2* __restrict__ kg_item, int n_items, int heads, int64_t mpad,
                                                   float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv) {
    constexpr int DHS = Geo<DH>::DHS, KS = Geo<DH>::KS, NB = Geo<DH>::NB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int it = blockIdx.x * 4 + wave;
    if (it >= n_items) return;
    const int h = blockIdx.y, g = lane >> 4, c16 = lane & 15;
    const int2 item = kg_item[it];
    const int n = win_count[item.x], start = win_start[item.x];
    const int64_t pos0 = (int64_t)win_tile0[item.x] * 32;
    const int n_tiles = (n + 31) >> 5;
    const int64_t row_half = mpad * heads * DHS, tr_half = (int64_t)heads * DH * mpad;
    const int ki = item.y * 16 + c16;
    const int64_t kpos = pos0 + ki;

    // this wave's 16 keys: B operands (k = channel, column = key)
    bf16x8 k_hi[KS], k_lo[KS], v_hi[KS], v_lo[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bool ok = 32 * s + 8 * g < DHS;
        const int64_t o = (kpos * heads + h) * DHS + 32 * s + 8 * g;
        load_frag(ws.kp, o, row_half, ok, &k_hi[s], &k_lo[s]);
        load_frag(ws.vp, o, row_half, ok, &v_hi[s], &v_lo[s]);
    }
    f32x4 dk_acc[NB], dv_acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        dk_acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dv_acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    for (int t = 0; t < n_tiles; ++t) {
        float pv[8], dsv[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 s_acc = {0.f, 0.f, 0.f, 0.f}, p_acc = {0.f, 0.f, 0.f, 0.f};
            const int64_t qrow = pos0 + t * 32 + u * 16 + c16;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bool ok = 32 * s + 8 * g < DHS;
                const int64_t o = (qrow * heads + h) * DHS + 32 * s + 8 * g;
                bf16x8 a_hi, a_lo;
                load_frag(ws.qp, o, row_half, ok, &a_hi, &a_lo);
                s_acc = mfma3(a_hi, a_lo, k_hi[s], k_lo[s], s_acc);   // S[query][key]
                load_frag(ws.gp, o, row_half, ok, &a_hi, &a_lo);
                p_acc = mfma3(a_hi, a_lo, v_hi[s], v_lo[s], p_acc);   // dP[query][key]
            }
            // rows of this accumulator: queries 32t + 16u + 4g + r (4 consecutive positions)
            const int64_t rp = (pos0 + t * 32 + u * 16 + g * 4) * heads + h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qidx = t * 32 + u * 16 + g * 4 + r;
                const float lq = ws.lp[rp + (int64_t)r * heads], dl = ws.dp[rp + (int64_t)r * heads];
                const float p = qidx < n ? __builtin_amdgcn_exp2f(s_acc[r] - lq) : 0.f;
                pv[u * 4 + r] = p;
                dsv[u * 4 + r] = p * (p_acc[r] - dl);
            }
        }
        bf16x8 p_hi, p_lo, ds_hi, ds_lo;
        split_frag(pv, &p_hi, &p_lo);
        split_frag(dsv, &ds_hi, &ds_lo);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int d = 16 * b + c16;
            const int64_t o = ((int64_t)h * DH + d) * mpad + pos0 + t * 32 + 8 * g;
            bf16x8 a_hi, a_lo;
            load_frag(ws.gt, o, tr_half, d < DH, &a_hi, &a_lo);
            dv_acc[b] = mfma3(a_hi, a_lo, p_hi, p_lo, dv_acc[b]);     // dV^T[d][key]
            load_frag(ws.qt, o, tr_half, d < DH, &a_hi, &a_lo);
            dk_acc[b] = mfma3(a_hi, a_lo, ds_hi, ds_lo, dk_acc[b]);   // dKhat^T[d][key] / ln2
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) dk_acc[b] = dk_acc[b] * kLn2;  // Q~ = q_hat * log2e / tau  ->  q_hat / tau = Q~ * ln2
    const int32_t token = tok[start + (ki < n ? ki : 0)];
    through_normalise<DH>(k + (int64_t)token * ldk + h * DH, g, dk_acc);
    if (ki < n) {
        float* ok_ = dk + (int64_t)token * lddk + h * DH;
        float* ov = dv + (int64_t)token * lddv + h * DH;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 16 * b + 4 * g + r;
                if (d < DH) {
                    ok_[d] = dk_acc[b][r];
                    ov[d] = dv_acc[b][r];
                }
            }
    }
}

template <int DH>
int run_bwd(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const float* out, const float* dout,
            const float* lse, const int32_t* tok, const int32_t* win_start, const int32_t* win_count,
            const int32_t* win_tile0, const int2* tile_item, int n_tiles, const int2* qg_item, int n_qg, int heads,
            const float* tau, float tau_min, float* dq, float* dk, float* dv, int lddq, int lddk, int lddv, float* dtau,
            void* workspace, hipStream_t st) {
    const int64_t mpad = (int64_t)n_tiles * 32;
    BwdWs<DH> ws(workspace, mpad, heads);
    const int c = heads * DH;
    const size_t smem = (size_t)(32 * (c + 1) + 32 * heads) * sizeof(float) + 32 * sizeof(int32_t);
    hipLaunchKernelGGL(attn_prepare_bwd<DH>, dim3((unsigned)n_tiles), dim3(256), smem, st, q, k, v, ldq, ldk, ldv, dout, out,
                       lse, tok, win_start, win_count, win_tile0, tile_item, heads, mpad, tau, tau_min, ws);
    SEG3D_CHECK_LAUNCH();
    dim3 grid((unsigned)((n_qg + 3) / 4), (unsigned)heads);
    if (hipMemsetAsync(ws.tau_part, 0, 1024, st) != hipSuccess) return SEG3D_ELAUNCH;
    hipLaunchKernelGGL(attn_bwd_q<DH>, grid, dim3(256), 0, st, ws, q, ldq, tok, win_start, win_count, win_tile0, qg_item,
                       n_qg, heads, mpad, tau, tau_min, dq, lddq);
    SEG3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(tau_reduce, dim3(1), dim3(256), 0, st, ws.tau_part, dtau);
    SEG3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_kv<DH>, grid, dim3(256), 0, st, ws, k, ldk, tok, win_start, win_count, win_tile0, qg_item,
                       n_qg, heads, mpad, dk, lddk, dv, lddv);
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

size_t seg3d_window_attn_workspace_bytes(int64_t m, int32_t n_tiles, int32_t heads, int32_t dh) {
    if (m < 0 || heads <= 0 || n_tiles < 0) return 0;
    const size_t fwd = attn_mfma_workspace_bytes(n_tiles, heads, dh);
    const size_t bwd = bwd_bytes(n_tiles, heads, dh);
    return (fwd > bwd ? fwd : bwd) + 256;
}

int seg3d_window_attn_bwd(const float* q, const float* k, const float* v, int32_t ldq, int32_t ldk, int32_t ldv,
                          const float* out, const float* dout, const float* lse, const int32_t* tok,
                          const int32_t* win_start, const int32_t* win_count, const int32_t* win_tile0,
                          const int32_t* tile_item, int32_t n_tiles, const int32_t* qg_item, int32_t n_qgroups, int64_t m,
                          int32_t n_windows, int32_t heads, int32_t dh, const float* tau, float tau_min, float* dq,
                          float* dk, float* dv, int32_t lddq, int32_t lddk, int32_t lddv, float* dtau, void* workspace,
                          size_t workspace_bytes, void* stream) {
    if (m == 0 || n_windows == 0 || n_tiles == 0 || n_qgroups == 0) return SEG3D_OK;
    if (!q || !k || !v || !out || !dout || !lse || !tok || !win_start || !win_count || !win_tile0 || !tile_item ||
        !qg_item || m < 0 || n_windows < 0 || n_tiles < 0 || n_qgroups < 0 || heads <= 0 || heads > 16 || !tau || !dq ||
        !dk || !dv || !dtau || !workspace)
        return SEG3D_EINVAL;
    const size_t need = bwd_bytes(n_tiles, heads, dh);
    if (need == 0) return SEG3D_EINVAL;
    if (workspace_bytes < need) return SEG3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    const int2* ti = reinterpret_cast<const int2*>(tile_item);
    const int2* qi = reinterpret_cast<const int2*>(qg_item);
#define SEG3D_BWD_CASE(D)                                                                                              \
    case D:                                                                                                            \
        return run_bwd<D>(q, k, v, ldq, ldk, ldv, out, dout, lse, tok, win_start, win_count, win_tile0, ti, n_tiles, qi,  \
                          n_qgroups, heads, tau, tau_min, dq, dk, dv, lddq, lddk, lddv, dtau, workspace, st)
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
