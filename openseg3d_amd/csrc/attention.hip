// a19-a21 backward: sparse window cosine attention, variable-length (CSR) form (forward: attention_mfma.hip).
// Reference: flat2window -> CosineMultiheadAttention -> window2flat, i.e.
//   seg3d/utils/swformer_utils.py:34-85 (scatter into padded [W,T,C] per batching level and back, in
//   EVERY encoder layer), seg3d/models/layers/point_transformer_layer.py:233-258,
//   seg3d/models/layers/cosine_msa.py:115-177: normalize(q), normalize(k), bmm, /clamp(tau),
//   -inf padding mask, softmax, bmm -- with the (W*H, T, T) score tensor materialised.
//
// Here the windows stay ragged: a workgroup takes one (window, head), reads its tokens through the
// CSR built by seg3d_window_partition, and writes the result straight back in flat voxel order.
// No padding, no mask, no [W,T,C] round trip through HBM, no score tensor.  One lane owns one
// query; keys are wave-uniform so K/V rows are broadcast loads out of L1/L2 (a window's K/V is
// <= 800 x 48 floats).  Softmax is the usual running max/sum; everything is fp32.
//
// Algorithmic FLOPs: 4 * C * sum_w n_w^2 per layer; padded flops are not credited.
#include "common.hpp"

size_t attn_mfma_workspace_bytes(int n_tiles, int heads, int dh);  // attention_mfma.hip

namespace {

constexpr int kThreads = 256;
constexpr int kQueryZ = 4;  // grid.z: 4 x 256 queries covers the largest window (10*10*8 = 800 tokens)
constexpr float kNormEps = 1e-12f;  // F.normalize eps, cosine_msa.py:152-153

// inverse L2 norms of every (token, head) slice of q and k
template <int DH>
__global__ __launch_bounds__(kThreads) void rnorm_kernel(const float* __restrict__ q, const float* __restrict__ k, int ldq,
                                                         int ldk, int64_t m, int heads, float* __restrict__ rq,
                                                         float* __restrict__ rk) {
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (t >= m * heads) return;
    const int64_t i = t / heads;
    const int h = (int)(t - i * heads);
    const float* qp = q + i * ldq + h * DH;
    const float* kp = k + i * ldk + h * DH;
    float sq = 0.f, sk = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        sq = fmaf(qp[d], qp[d], sq);
        sk = fmaf(kp[d], kp[d], sk);
    }
    rq[t] = 1.0f / fmaxf(sqrtf(sq), kNormEps);
    rk[t] = 1.0f / fmaxf(sqrtf(sk), kNormEps);
}

// pass A: lane = query.  dq (through the normalisation) and the tau gradient.
template <int DH>
__global__ __launch_bounds__(kThreads) void attn_bwd_q_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, int ldq, int ldk, int ldv,
                                                              const float* __restrict__ dout, const float* __restrict__ o,
                                                              const float* __restrict__ lse, const int32_t* __restrict__ tok,
                                                              const int32_t* __restrict__ win_start,
                                                              const int32_t* __restrict__ win_count, int heads,
                                                              const float* __restrict__ rq, const float* __restrict__ rk,
                                                              const float* __restrict__ tau, float tau_min,
                                                              float* __restrict__ dq, int lddq, float* __restrict__ delta,
                                                              float* __restrict__ dtau) {
    const int w = blockIdx.x, h = blockIdx.y;
    const int n = win_count[w];
    const int qi = blockIdx.z * kThreads + threadIdx.x;
    if ((int)(blockIdx.z * kThreads) >= n) return;
    const int start = win_start[w];
    const float tau_c = fmaxf(tau[0], tau_min);
    const float inv_tau = 1.0f / tau_c;
    const bool active = qi < n;
    const int my_tok = active ? tok[start + qi] : tok[start];
    const int64_t th = (int64_t)my_tok * heads + h;

    float qh[DH], go[DH], gq[DH];
    const float rqi = rq[th];
    float dl = 0.f;
    {
        const float* qp = q + (int64_t)my_tok * ldq + h * DH;
        const float* gp = dout + (int64_t)my_tok * (heads * DH) + h * DH;
        const float* op = o + (int64_t)my_tok * (heads * DH) + h * DH;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            qh[d] = qp[d] * rqi;
            go[d] = gp[d];
            gq[d] = 0.f;
            dl = fmaf(go[d], op[d], dl);
        }
    }
    const float my_lse = lse[th];
    float tau_acc = 0.f;
    for (int j = 0; j < n; ++j) {
        const int tj = tok[start + j];
        const float* kp = k + (int64_t)tj * ldk + h * DH;
        const float* vp = v + (int64_t)tj * ldv + h * DH;
        const float rkj = rk[(int64_t)tj * heads + h];
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            s = fmaf(qh[d], kp[d], s);
            dp = fmaf(go[d], vp[d], dp);
        }
        s *= rkj * inv_tau;
        const float p = expf(s - my_lse);
        const float ds = p * (dp - dl);
        tau_acc = fmaf(ds, s, tau_acc);
        const float c = ds * rkj * inv_tau;
#pragma unroll
        for (int d = 0; d < DH; ++d) gq[d] = fmaf(c, kp[d], gq[d]);
    }
    if (active) {
        // back through q_hat = q / max(|q|, eps)
        float proj = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) proj = fmaf(qh[d], gq[d], proj);
        const bool clamped = rqi >= 1.0f / kNormEps;
        float* dqp = dq + (int64_t)my_tok * lddq + h * DH;
#pragma unroll
        for (int d = 0; d < DH; ++d) dqp[d] = clamped ? gq[d] * rqi : (gq[d] - qh[d] * proj) * rqi;
        delta[th] = dl;
    } else {
        tau_acc = 0.f;
    }
    // d/dtau of s = c/tau is -s/tau; zero when tau is clamped (cosine_msa.py:162)
    for (int off = 32; off > 0; off >>= 1) tau_acc += __shfl_xor(tau_acc, off, SEG3D_WAVE);
    if ((threadIdx.x & 63) == 0 && tau[0] > tau_min && tau_acc != 0.f) atomicAdd(dtau, -tau_acc * inv_tau);
}

// pass B: lane = key.  dk (through the normalisation) and dv.
template <int DH>
__global__ __launch_bounds__(kThreads) void attn_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, int ldq, int ldk, int ldv,
                                                               const float* __restrict__ dout, const float* __restrict__ lse,
                                                               const float* __restrict__ delta, const int32_t* __restrict__ tok,
                                                               const int32_t* __restrict__ win_start,
                                                               const int32_t* __restrict__ win_count, int heads,
                                                               const float* __restrict__ rq, const float* __restrict__ rk,
                                                               const float* __restrict__ tau, float tau_min,
                                                               float* __restrict__ dk, int lddk, float* __restrict__ dv, int lddv) {
    const int w = blockIdx.x, h = blockIdx.y;
    const int n = win_count[w];
    const int kj = blockIdx.z * kThreads + threadIdx.x;
    if ((int)(blockIdx.z * kThreads) >= n) return;
    const int start = win_start[w];
    const float inv_tau = 1.0f / fmaxf(tau[0], tau_min);
    const bool active = kj < n;
    const int my_tok = active ? tok[start + kj] : tok[start];
    const int64_t th = (int64_t)my_tok * heads + h;

    float kh[DH], vv[DH], gk[DH], gv[DH];
    const float rkj = rk[th];
    {
        const float* kp = k + (int64_t)my_tok * ldk + h * DH;
        const float* vp = v + (int64_t)my_tok * ldv + h * DH;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            kh[d] = kp[d] * rkj;
            vv[d] = vp[d];
            gk[d] = 0.f;
            gv[d] = 0.f;
        }
    }
    for (int i = 0; i < n; ++i) {
        const int ti = tok[start + i];  // wave-uniform query
        const int64_t tih = (int64_t)ti * heads + h;
        const float* qp = q + (int64_t)ti * ldq + h * DH;
        const float* gp = dout + (int64_t)ti * (heads * DH) + h * DH;
        const float rqi = rq[tih];
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            s = fmaf(qp[d], kh[d], s);
            dp = fmaf(gp[d], vv[d], dp);
        }
        s *= rqi * inv_tau;
        const float p = expf(s - lse[tih]);
        const float ds = p * (dp - delta[tih]);
        const float c = ds * rqi * inv_tau;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            gk[d] = fmaf(c, qp[d], gk[d]);
            gv[d] = fmaf(p, gp[d], gv[d]);
        }
    }
    if (active) {
        float proj = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) proj = fmaf(kh[d], gk[d], proj);
        const bool clamped = rkj >= 1.0f / kNormEps;
        float* dkp = dk + (int64_t)my_tok * lddk + h * DH;
        float* dvp = dv + (int64_t)my_tok * lddv + h * DH;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            dkp[d] = clamped ? gk[d] * rkj : (gk[d] - kh[d] * proj) * rkj;
            dvp[d] = gv[d];
        }
    }
}

struct AttnArgs {
    const float *q, *k, *v;
    int ldq, ldk, ldv;
    const int32_t *tok, *win_start, *win_count;
    int n_windows, heads;
    int64_t m;
    const float* tau;
    float tau_min;
};

template <int DH>
int run_rnorm(const AttnArgs& a, float* rq, float* rk, hipStream_t st) {
    hipLaunchKernelGGL(rnorm_kernel<DH>, dim3((unsigned)ceil_div64(a.m * a.heads, kThreads)), dim3(kThreads), 0, st, a.q,
                       a.k, a.ldq, a.ldk, a.m, a.heads, rq, rk);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

template <int DH>
int run_bwd(const AttnArgs& a, float* ws, const float* o, const float* dout, const float* lse, float* dq, float* dk,
            float* dv, int lddq, int lddk, int lddv, float* dtau, hipStream_t st) {
    float* rq = ws;
    float* rk = ws + a.m * a.heads;
    float* delta = ws + 2 * a.m * a.heads;
    int rc = run_rnorm<DH>(a, rq, rk, st);
    if (rc) return rc;
    dim3 grid((unsigned)a.n_windows, (unsigned)a.heads, kQueryZ);
    hipLaunchKernelGGL(attn_bwd_q_kernel<DH>, grid, dim3(kThreads), 0, st, a.q, a.k, a.v, a.ldq, a.ldk, a.ldv, dout, o,
                       lse, a.tok, a.win_start, a.win_count, a.heads, rq, rk, a.tau, a.tau_min, dq, lddq, delta, dtau);
    SEG3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_kv_kernel<DH>, grid, dim3(kThreads), 0, st, a.q, a.k, a.v, a.ldq, a.ldk, a.ldv, dout, lse,
                       delta, a.tok, a.win_start, a.win_count, a.heads, rq, rk, a.tau, a.tau_min, dk, lddk, dv, lddv);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

bool bad_common(const float* q, const float* k, const float* v, const int32_t* tok, const int32_t* ws, const int32_t* wc,
                int64_t m, int nw, int heads, int dh, const float* tau, const void* workspace) {
    return !q || !k || !v || !tok || !ws || !wc || m <= 0 || nw <= 0 || heads <= 0 || dh <= 0 || !tau || !workspace;
}

}  // namespace

extern "C" {

size_t seg3d_window_attn_workspace_bytes(int64_t m, int32_t n_tiles, int32_t heads, int32_t dh) {
    if (m < 0 || heads <= 0 || n_tiles < 0) return 0;
    const size_t bwd = (size_t)(3 * m * heads + 64) * sizeof(float);
    const size_t fwd = attn_mfma_workspace_bytes(n_tiles, heads, dh);
    return bwd > fwd ? bwd : fwd;
}

int seg3d_window_attn_bwd(const float* q, const float* k, const float* v, int32_t ldq, int32_t ldk, int32_t ldv,
                          const float* out, const float* dout, const float* lse, const int32_t* tok,
                          const int32_t* win_start, const int32_t* win_count, int64_t m, int32_t n_windows,
                          int32_t heads, int32_t dh, const float* tau, float tau_min, float* dq, float* dk, float* dv,
                          int32_t lddq, int32_t lddk, int32_t lddv, float* dtau, void* workspace, size_t workspace_bytes,
                          void* stream) {
    if (m == 0 || n_windows == 0) return SEG3D_OK;
    if (bad_common(q, k, v, tok, win_start, win_count, m, n_windows, heads, dh, tau, workspace) || !out || !dout ||
        !lse || !dq || !dk || !dv || !dtau)
        return SEG3D_EINVAL;
    if (workspace_bytes < (size_t)(3 * m * heads + 64) * sizeof(float)) return SEG3D_EWORKSPACE;
    AttnArgs a{q, k, v, ldq, ldk, ldv, tok, win_start, win_count, n_windows, heads, m, tau, tau_min};
    hipStream_t st = as_stream(stream);
    float* ws = static_cast<float*>(workspace);
    switch (dh) {
        case 6: return run_bwd<6>(a, ws, out, dout, lse, dq, dk, dv, lddq, lddk, lddv, dtau, st);
        case 12: return run_bwd<12>(a, ws, out, dout, lse, dq, dk, dv, lddq, lddk, lddv, dtau, st);
        case 24: return run_bwd<24>(a, ws, out, dout, lse, dq, dk, dv, lddq, lddk, lddv, dtau, st);
        case 48: return run_bwd<48>(a, ws, out, dout, lse, dq, dk, dv, lddq, lddk, lddv, dtau, st);
        default: return SEG3D_EINVAL;
    }
}

}  // extern "C"
