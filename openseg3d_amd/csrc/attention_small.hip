// a19-a21 for the narrowest heads (dh = 6: SWFormer stage 1, C = 48), forward and backward: exact-fp32 vector-ALU window
// attention.  (Forward and backward at dh >= 12 run on the fused MFMA kernels: attention_fused*.hip; so did the dh-6 forward
// until round 4 -- 68 us per layer in inference, 96 us with dropout, against 4 x 21 MB of rows to move.)
//
// At this stage the windows hold ~15 voxels on average and a head is 6 channels wide: a 16x16x32 MFMA tile would be
// 3/4 padding, and the matrix-core backward loses to this form (475 vs 360 us per layer on the headline scene).
// Here one thread owns one (token, head) pair and keeps its
// whole dh-vector in registers; the streamed side of a window (keys / queries) goes through LDS 32 tokens at a time,
// every LDS read is a broadcast (all lanes of a half-wave share the head), nothing crosses lanes, and the
// arithmetic is plain fp32 -- no operand split, no prepare pass, no workspace.
//   pass Q  : thread (query i, head h): dq_hat += ds * k_hat / tau, dtau += ds * s          (ds = p (dp - delta))
//   pass KV : thread (key j, head h)  : dv += p * dO, dk_hat += ds * q_hat / tau
// with the gradient through x_hat = x / max(|x|, eps) applied to the thread's own vector at the end.
// Work items are the (window, 32-token tile) list of seg3d_window_partition; block = 32 tokens x heads threads.
#include "attn_common.hpp"
#include "attn_dropout.hpp"

namespace {

using namespace attn;

// dh-vector of one (token, head) as float pairs: every multiply-add below is a v_pk_fma_f32 (two fp32 FMAs per lane
// per instruction), which is what these kernels are bound by
template <int DH>
struct Vec {
    f32x2 p[DH / 2];
};

template <int DH>
__device__ __forceinline__ Vec<DH> zero_vec() {
    Vec<DH> r;
#pragma unroll
    for (int i = 0; i < DH / 2; ++i) r.p[i] = (f32x2){0.f, 0.f};
    return r;
}

// DH floats from a 8-B (DH = 6) / 16-B (DH = 12) aligned address
template <int DH>
__device__ __forceinline__ Vec<DH> load_vec(const float* p) {
    Vec<DH> r;
    if (DH % 4 == 0) {
#pragma unroll
        for (int i = 0; i < DH / 4; ++i) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * i);
            r.p[2 * i] = (f32x2){t[0], t[1]};
            r.p[2 * i + 1] = (f32x2){t[2], t[3]};
        }
    } else {
#pragma unroll
        for (int i = 0; i < DH / 2; ++i) r.p[i] = *reinterpret_cast<const f32x2*>(p + 2 * i);
    }
    return r;
}

template <int DH>
__device__ __forceinline__ void store_vec(float* p, const Vec<DH>& r) {
    if (DH % 4 == 0) {
#pragma unroll
        for (int i = 0; i < DH / 4; ++i)
            *reinterpret_cast<f32x4*>(p + 4 * i) = (f32x4){r.p[2 * i][0], r.p[2 * i][1], r.p[2 * i + 1][0], r.p[2 * i + 1][1]};
    } else {
#pragma unroll
        for (int i = 0; i < DH / 2; ++i) *reinterpret_cast<f32x2*>(p + 2 * i) = r.p[i];
    }
}

template <int DH>
__device__ __forceinline__ float dot(const Vec<DH>& a, const Vec<DH>& b) {
    f32x2 s = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < DH / 2; ++i) s = __builtin_elementwise_fma(a.p[i], b.p[i], s);
    return s[0] + s[1];
}

// y += a * x
template <int DH>
__device__ __forceinline__ void axpy(float a, const Vec<DH>& x, Vec<DH>* y) {
    const f32x2 aa = {a, a};
#pragma unroll
    for (int i = 0; i < DH / 2; ++i) y->p[i] = __builtin_elementwise_fma(aa, x.p[i], y->p[i]);
}

template <int DH>
__device__ __forceinline__ void scale(float a, Vec<DH>* y) {
    const f32x2 aa = {a, a};
#pragma unroll
    for (int i = 0; i < DH / 2; ++i) y->p[i] = y->p[i] * aa;
}

// x -> x / max(|x|, eps) * scale_to; returns |x|
template <int DH>
__device__ __forceinline__ float normalise(Vec<DH>* x, float scale_to) {
    const float len = sqrtf(dot(*x, *x));
    scale<DH>(scale_to / fmaxf(len, kNormEps), x);
    return len;
}

// gradient through x_hat = x / max(|x|, eps): g (w.r.t. x_hat) -> g (w.r.t. x), cosine_msa.py:152-153
template <int DH>
__device__ __forceinline__ void through_normalise(const Vec<DH>& x_raw, Vec<DH>* g) {
    const float len = sqrtf(dot(x_raw, x_raw));
    const float rinv = 1.0f / fmaxf(len, kNormEps);
    if (len < kNormEps) {
        scale<DH>(rinv, g);
        return;
    }
    Vec<DH> xh = x_raw;
    scale<DH>(rinv, &xh);
    const float proj = dot(xh, *g);
    axpy<DH>(-proj, xh, g);
    scale<DH>(rinv, g);
}

constexpr int kTile = 32;

// ------------------------------------------------------------------ forward: out = softmax(q^.k^ / tau) (. dropout) v, LSE
// Thread (query i, head h) of a (window, 32-token tile) item walks the window's keys 32 at a time through LDS.  Cosine scores
// are bounded by log2e / tau: that bound is the fixed softmax maximum while exp2(-2 bound) stays a normal float (tau > ~0.036),
// otherwise the running-maximum form -- a block-uniform choice on the device scalar tau.  The row sum is taken BEFORE the
// dropout factor (cosine_msa.py:172-176: softmax, then F.dropout, then attn @ v).
template <int DH>
__global__ __launch_bounds__(256, 8) void attn_small_fwd(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, int ldq, int ldk, int ldv,
                                                      const int32_t* __restrict__ tok, const int32_t* __restrict__ win_start,
                                                      const int32_t* __restrict__ win_count, const int4* __restrict__ tile_item,
                                                      int heads, const float* __restrict__ tau, float tau_min,
                                                      float* __restrict__ out, float* __restrict__ lse, DropoutParams drop) {
    extern __shared__ float smem[];
    const int c = heads * DH;
    float* kbuf = smem;
    float* vbuf = smem + kTile * c;
    const int i = threadIdx.x & 31, h = threadIdx.x >> 5;
    const int4 item = tile_item[blockIdx.x];
    const int n = item.w, start = item.z;
    const int qi = item.y * kTile + i;
    const int32_t qtok = qi < n ? tok[start + qi] : -1;
    const float tau_c = fmaxf(tau[0], tau_min);
    const float bound = kLog2e / tau_c;  // every score (log2 domain) is <= bound
    const bool fixed_max = bound <= 40.0f;
    Vec<DH> qn = zero_vec<DH>(), acc = zero_vec<DH>();
    if (qtok >= 0) {
        qn = load_vec<DH>(q + (int64_t)qtok * ldq + h * DH);
        normalise<DH>(&qn, bound);
    }
    float l = 0.f, mx = fixed_max ? bound : -INFINITY;
    const uint32_t row_state = drop.threshold ? dropout_row_state(dropout_head_state(drop, item.x, h), qi) : 0u;
    Vec<DH> k_next, v_next;
    auto fetch_kv = [&](int t0) {
        const int kj = t0 + i;
        k_next = v_next = zero_vec<DH>();
        if (kj < n) {
            const int32_t kt = tok[start + kj];
            k_next = load_vec<DH>(k + (int64_t)kt * ldk + h * DH);
            v_next = load_vec<DH>(v + (int64_t)kt * ldv + h * DH);
        }
    };
    fetch_kv(0);
    for (int t0 = 0; t0 < n; t0 += kTile) {
        __syncthreads();
        {
            Vec<DH> kk = k_next;
            normalise<DH>(&kk, 1.0f);
            store_vec<DH>(kbuf + i * c + h * DH, kk);
            store_vec<DH>(vbuf + i * c + h * DH, v_next);
        }
        __syncthreads();
        if (t0 + kTile < n) fetch_kv(t0 + kTile);
        const int nk = n - t0 < kTile ? n - t0 : kTile;
        for (int j = 0; j < nk; j += 2) {  // key pairs: one dropout hash covers (query pair) x (key pair)
            const uint32_t bits = drop.threshold ? dropout_block_bits(row_state, dropout_key_term(t0 + j)) : 0u;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int jj = j + u;
                if (jj < nk) {
                    const Vec<DH> kk = load_vec<DH>(kbuf + jj * c + h * DH);
                    const Vec<DH> vv = load_vec<DH>(vbuf + jj * c + h * DH);
                    const float sc = dot(qn, kk);
                    float p;
                    if (fixed_max) {
                        p = __builtin_amdgcn_exp2f(sc - bound);
                    } else {
                        const float mnew = fmaxf(mx, sc);
                        const float resc = __builtin_amdgcn_exp2f(mx - mnew);  // (exp2(-inf) = 0 on the first key)
                        scale<DH>(resc, &acc);
                        l *= resc;
                        p = __builtin_amdgcn_exp2f(sc - mnew);
                        mx = mnew;
                    }
                    l += p;
                    if (drop.threshold) p *= dropout_factor(drop, bits, qi, t0 + jj);
                    axpy<DH>(p, vv, &acc);
                }
            }
        }
    }
    if (qtok >= 0) {
        scale<DH>(1.0f / l, &acc);
        store_vec<DH>(out + (int64_t)qtok * c + h * DH, acc);
        lse[(int64_t)qtok * heads + h] = (__builtin_amdgcn_logf(l) + mx) * kLn2;  // natural-log LSE (v_log_f32 = log2)
    }
}

// ------------------------------------------------------------------ backward, pass Q: dq, dtau
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_small_bwd_q(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, int ldq, int ldk, int ldv,
                                                        const float* __restrict__ out, const float* __restrict__ dout,
                                                        const float* __restrict__ lse, const int32_t* __restrict__ tok,
                                                        const int32_t* __restrict__ win_start, const int32_t* __restrict__ win_count,
                                                        const int4* __restrict__ tile_item, int heads,
                                                        const float* __restrict__ tau, float tau_min, float* __restrict__ dq,
                                                        int lddq, float* __restrict__ tau_part, DropoutParams drop) {
    extern __shared__ float smem[];
    const int c = heads * DH;
    float* kbuf = smem;
    float* vbuf = smem + kTile * c;
    __shared__ float tau_red[8];
    const int i = threadIdx.x & 31, h = threadIdx.x >> 5;
    const int4 item = tile_item[blockIdx.x];
    const int n = item.w, start = item.z;
    const int qi = item.y * kTile + i;
    const int32_t qtok = qi < n ? tok[start + qi] : -1;
    const float tau_c = fmaxf(tau[0], tau_min);
    Vec<DH> q_raw = zero_vec<DH>(), qn = zero_vec<DH>(), go = zero_vec<DH>(), acc = zero_vec<DH>();
    float l2 = 0.f, delta = 0.f;
    if (qtok >= 0) {
        q_raw = load_vec<DH>(q + (int64_t)qtok * ldq + h * DH);
        qn = q_raw;
        normalise<DH>(&qn, kLog2e / tau_c);
        go = load_vec<DH>(dout + (int64_t)qtok * c + h * DH);
        const Vec<DH> oo = load_vec<DH>(out + (int64_t)qtok * c + h * DH);
        delta = dot(go, oo);
        l2 = lse[(int64_t)qtok * heads + h] * kLog2e;
    }
    float tau_acc = 0.f;
    // dropout: the query's share of the hash (two mixing rounds) once per thread; its parity picks the byte pair of every block
    const uint32_t row_state = drop.threshold ? dropout_row_state(dropout_head_state(drop, item.x, h), qi) : 0u;
    const uint32_t drop_shift = dropout_lane_shift_query(qi & 1);
    Vec<DH> k_next, v_next;
    auto fetch_kv = [&](int t0) {
        const int kj = t0 + i;
        k_next = v_next = zero_vec<DH>();
        if (kj < n) {
            const int32_t kt = tok[start + kj];
            k_next = load_vec<DH>(k + (int64_t)kt * ldk + h * DH);
            v_next = load_vec<DH>(v + (int64_t)kt * ldv + h * DH);
        }
    };
    fetch_kv(0);
    for (int t0 = 0; t0 < n; t0 += kTile) {
        __syncthreads();
        {
            Vec<DH> kk = k_next;
            normalise<DH>(&kk, 1.0f);
            store_vec<DH>(kbuf + i * c + h * DH, kk);
            store_vec<DH>(vbuf + i * c + h * DH, v_next);
        }
        __syncthreads();
        if (t0 + kTile < n) fetch_kv(t0 + kTile);
        const int nk = n - t0 < kTile ? n - t0 : kTile;  // (block-uniform: stage-1 windows hold ~15 voxels, half a tile)
#pragma unroll 2
        for (int j = 0; j < nk; j += 2) {  // key pairs: one hash covers the 2 x 2 block of (query pair) x (key pair)
            // dP = D * (dO . v): the forward's dropout factors, regenerated; the lane's query parity is shifted out once per
            // hash, so the pair's keys read bytes 0 and 1 (attn_dropout.hpp)
            const uint32_t adj = drop.threshold ? dropout_block_bits(row_state, dropout_key_term(t0 + j)) >> drop_shift : 0u;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int jj = j + u;
                if (jj < nk) {
                    const Vec<DH> kk = load_vec<DH>(kbuf + jj * c + h * DH);
                    const Vec<DH> vv = load_vec<DH>(vbuf + jj * c + h * DH);
                    const float s = dot(qn, kk);
                    const float p = __builtin_amdgcn_exp2f(s - l2);
                    float dpv = dot(go, vv);
                    if (drop.threshold) dpv = dropout_dropped_byte(drop, adj, u) ? 0.f : dpv * drop.inv_keep;
                    const float ds = p * (dpv - delta);
                    tau_acc = fmaf(ds, s, tau_acc);
                    axpy<DH>(ds, kk, &acc);
                }
            }
        }
    }
    if (qtok >= 0) {
        scale<DH>(1.0f / tau_c, &acc);
        through_normalise<DH>(q_raw, &acc);
        store_vec<DH>(dq + (int64_t)qtok * lddq + h * DH, acc);
    } else {
        tau_acc = 0.f;
    }
    // d/dtau: s_nat = s2 * ln2 = c / tau  ->  dL/dtau = -sum(ds * s_nat) / tau   (zero while tau is clamped)
    for (int off = 32; off > 0; off >>= 1) tau_acc += __shfl_xor(tau_acc, off, SEG3D_WAVE);
    if ((threadIdx.x & 63) == 0) tau_red[threadIdx.x >> 6] = tau_acc;
    __syncthreads();
    if (threadIdx.x == 0) {  // one plain store per workgroup; tau_reduce_small adds them in a fixed order
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += tau_red[w];
        tau_part[blockIdx.x] = tau[0] > tau_min ? -t * kLn2 / tau_c : 0.f;
    }
}

// ------------------------------------------------------------------ backward, pass KV: dk, dv
template <int DH>
__global__ __launch_bounds__(256, 3) void attn_small_bwd_kv(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, int ldq, int ldk, int ldv,
                                                         const float* __restrict__ out, const float* __restrict__ dout,
                                                         const float* __restrict__ lse, const int32_t* __restrict__ tok,
                                                         const int32_t* __restrict__ win_start, const int32_t* __restrict__ win_count,
                                                         const int4* __restrict__ tile_item, int heads,
                                                         const float* __restrict__ tau, float tau_min, float* __restrict__ dk,
                                                         int lddk, float* __restrict__ dv, int lddv, DropoutParams drop) {
    extern __shared__ float smem[];  // q~ [32][c], dO [32][c], (L, delta) [32][heads][2]
    const int c = heads * DH;
    float* qbuf = smem;
    float* gbuf = smem + kTile * c;
    float* lbuf = smem + 2 * kTile * c;
    const int i = threadIdx.x & 31, h = threadIdx.x >> 5;
    const int4 item = tile_item[blockIdx.x];
    const int n = item.w, start = item.z;
    const int kj = item.y * kTile + i;
    const int32_t ktok = kj < n ? tok[start + kj] : -1;
    const float tau_c = fmaxf(tau[0], tau_min);
    Vec<DH> k_raw = zero_vec<DH>(), kn = zero_vec<DH>(), vv = zero_vec<DH>(), dk_acc = zero_vec<DH>(), dv_acc = zero_vec<DH>();
    if (ktok >= 0) {
        k_raw = load_vec<DH>(k + (int64_t)ktok * ldk + h * DH);
        kn = k_raw;
        normalise<DH>(&kn, 1.0f);
        vv = load_vec<DH>(v + (int64_t)ktok * ldv + h * DH);
    }
    // dropout: what does not depend on the streamed query -- the (window, head) state and the lane's key term
    const uint32_t head_state = drop.threshold ? dropout_head_state(drop, item.x, h) : 0u;
    const uint32_t key_term = dropout_key_term(kj);
    const uint32_t drop_shift = dropout_lane_shift_key(kj & 1);
    Vec<DH> q_next, g_next, o_next;
    float lse_next = 0.f;
    auto fetch_q = [&](int t0) {
        const int qi = t0 + i;
        q_next = g_next = o_next = zero_vec<DH>();
        lse_next = 0.f;
        if (qi < n) {
            const int32_t qt = tok[start + qi];
            q_next = load_vec<DH>(q + (int64_t)qt * ldq + h * DH);
            g_next = load_vec<DH>(dout + (int64_t)qt * c + h * DH);
            o_next = load_vec<DH>(out + (int64_t)qt * c + h * DH);
            lse_next = lse[(int64_t)qt * heads + h];
        }
    };
    fetch_q(0);
    for (int t0 = 0; t0 < n; t0 += kTile) {
        __syncthreads();
        {  // stage query row t0 + i, head h (all-zero rows past n: p is masked below)
            Vec<DH> qq = q_next;
            normalise<DH>(&qq, kLog2e / tau_c);
            store_vec<DH>(qbuf + i * c + h * DH, qq);
            store_vec<DH>(gbuf + i * c + h * DH, g_next);
            *reinterpret_cast<f32x2*>(lbuf + (i * heads + h) * 2) = (f32x2){lse_next * kLog2e, dot(g_next, o_next)};
        }
        __syncthreads();
        if (t0 + kTile < n) fetch_q(t0 + kTile);
        const int nq = n - t0 < kTile ? n - t0 : kTile;
#pragma unroll 2
        for (int j = 0; j < nq; j += 2) {  // query pairs: one hash (row state of the pair + one mixing round) per 2 x 2 block;
            // the lane's key parity is shifted out once, the pair's queries read bytes 0 and 2 (attn_dropout.hpp)
            const uint32_t adj = drop.threshold ? dropout_block_bits(dropout_row_state(head_state, t0 + j), key_term) >> drop_shift : 0u;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int jj = j + u;
                if (jj < nq) {
                    const Vec<DH> qq = load_vec<DH>(qbuf + jj * c + h * DH);
                    const Vec<DH> gg = load_vec<DH>(gbuf + jj * c + h * DH);
                    const f32x2 ld = *reinterpret_cast<const f32x2*>(lbuf + (jj * heads + h) * 2);
                    const float s = dot(qq, kn);
                    const float p = __builtin_amdgcn_exp2f(s - ld[0]);
                    float dfac = 1.0f;
                    if (drop.threshold) dfac = dropout_dropped_byte(drop, adj, 2 * u) ? 0.f : drop.inv_keep;
                    const float ds = p * (dfac * dot(gg, vv) - ld[1]);  // dS = P * (D * dP - delta)
                    axpy<DH>(p * dfac, gg, &dv_acc);                     // dV = (D * P)^T dO
                    axpy<DH>(ds, qq, &dk_acc);
                }
            }
        }
    }
    if (ktok >= 0) {
        scale<DH>(kLn2, &dk_acc);  // q~ = q_hat * log2e / tau  ->  q_hat / tau = q~ * ln2
        through_normalise<DH>(k_raw, &dk_acc);
        store_vec<DH>(dk + (int64_t)ktok * lddk + h * DH, dk_acc);
        store_vec<DH>(dv + (int64_t)ktok * lddv + h * DH, dv_acc);
    }
}

__global__ __launch_bounds__(1024) void tau_reduce_small(const float* __restrict__ part, int count, float* __restrict__ dtau) {
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

template <int DH>
int run_small_bwd(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const float* out,
                  const float* dout, const float* lse, const int32_t* tok, const int32_t* win_start,
                  const int32_t* win_count, const int4* tile_item, int n_tiles, int heads, const float* tau, float tau_min,
                  float* dq, float* dk, float* dv, int lddq, int lddk, int lddv, float* dtau, float* tau_part,
                  const DropoutParams& drop, hipStream_t st) {
    const size_t smem_q = (size_t)2 * kTile * heads * DH * sizeof(float);
    hipLaunchKernelGGL(attn_small_bwd_q<DH>, dim3((unsigned)n_tiles), dim3(32 * heads), smem_q, st, q, k, v, ldq, ldk, ldv,
                       out, dout, lse, tok, win_start, win_count, tile_item, heads, tau, tau_min, dq, lddq, tau_part, drop);
    SEG3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(tau_reduce_small, dim3(1), dim3(1024), 0, st, tau_part, n_tiles, dtau);
    SEG3D_CHECK_LAUNCH();
    const size_t smem_kv = smem_q + (size_t)kTile * heads * 2 * sizeof(float);
    hipLaunchKernelGGL(attn_small_bwd_kv<DH>, dim3((unsigned)n_tiles), dim3(32 * heads), smem_kv, st, q, k, v, ldq, ldk, ldv,
                       out, dout, lse, tok, win_start, win_count, tile_item, heads, tau, tau_min, dk, lddk, dv, lddv, drop);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // namespace

int attn_small_fwd_launch(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const int32_t* tok,
                          const int32_t* win_start, const int32_t* win_count, const int32_t* tile_item, int n_tiles, int heads,
                          int dh, const float* tau, float tau_min, float* out, float* lse, const DropoutParams& drop,
                          hipStream_t st) {
    if (dh != 6) return SEG3D_EINVAL;
    const size_t smem = (size_t)2 * kTile * heads * 6 * sizeof(float);
    hipLaunchKernelGGL(attn_small_fwd<6>, dim3((unsigned)n_tiles), dim3(32 * heads), smem, st, q, k, v, ldq, ldk, ldv, tok,
                       win_start, win_count, reinterpret_cast<const int4*>(tile_item), heads, tau, tau_min, out, lse, drop);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

// used by seg3d_window_attn_fwd / _bwd (attention.hip): dh 6, up to 8 heads
bool attn_small_supported(int heads, int dh) { return dh == 6 && heads >= 1 && heads <= 8; }

int attn_small_bwd_launch(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const float* out,
                          const float* dout, const float* lse, const int32_t* tok, const int32_t* win_start,
                          const int32_t* win_count, const int32_t* tile_item, int n_tiles, int heads, int dh,
                          const float* tau, float tau_min, float* dq, float* dk, float* dv, int lddq, int lddk, int lddv,
                          float* dtau, void* workspace, const DropoutParams& drop, hipStream_t st) {
    if (dh != 6) return SEG3D_EINVAL;
    const int4* ti = reinterpret_cast<const int4*>(tile_item);
    float* tau_part = static_cast<float*>(workspace);  // n_tiles floats
    return run_small_bwd<6>(q, k, v, ldq, ldk, ldv, out, dout, lse, tok, win_start, win_count, ti, n_tiles, heads, tau,
                            tau_min, dq, dk, dv, lddq, lddk, lddv, dtau, tau_part, drop, st);
}
