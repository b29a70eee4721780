// SURVEY 8(f) rank 1, DeepFusionBlock's cross attention (seg3d/models/layers/deep_fusion.py:26-45): every current-sweep
// point attends to the image features of its K nearest points,
//     out_i = sum_j softmax_j( <q_i, k_{nbr(i,j)}> * scale, masked where the neighbour has no image feature ) . v_{nbr(i,j)}
// with nan_to_num on rows whose neighbours are all masked (-> 0) and attention dropout on the probabilities.  The
// reference materialises k[nbr] and v[nbr] as [N, K, D] tensors (2 x 0.5 GB at N = 466 k, K = 16, D = 32), multiplies
// and reduces them with broadcast torch ops, and its backward scatters with index_put atomics.  Here:
//   forward   16 lanes per query: lane j holds neighbour j -- its dot product over the whole 128-B k row (no cross-lane
//             reduce), the softmax is four xor-shuffles inside the 16-lane group; for P.V the group switches roles:
//             lane l owns channels 2l, 2l+1 and walks the 16 neighbours (index and probability by shuffle), so every
//             v row is read once, coalesced, and nothing is materialised.  The probabilities (after dropout) are kept
//             [N, K] for the backward.
//   backward  pass Q (same lane roles): dP_ij = <dout_i, v_j>, dS = P (D dP - delta), dq_i = scale * sum_j dS_ij k_j; dS
//             and P D go to [N, K] scratch.  Pass KV walks the INVERSE neighbour lists (CSR over the flattened index
//             table, built by seg3d_group_index): source row r sums dS_e q_{i(e)} and P_e dout_{i(e)} over the pairs
//             e that reference it, in a fixed order -- no float atomics, bit-reproducible.
// D = 32 channels (hidden_channel of the block, segformer.py:51-53), K <= 16.
#include "common.hpp"

#include <math.h>

namespace {

constexpr int kD = 32;
constexpr int kThreads = 256;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float group16_sum(float x) {
    x += __shfl_xor(x, 1, SEG3D_WAVE);
    x += __shfl_xor(x, 2, SEG3D_WAVE);
    x += __shfl_xor(x, 4, SEG3D_WAVE);
    x += __shfl_xor(x, 8, SEG3D_WAVE);
    return x;
}
__device__ __forceinline__ float group16_max(float x) {
    x = fmaxf(x, __shfl_xor(x, 1, SEG3D_WAVE));
    x = fmaxf(x, __shfl_xor(x, 2, SEG3D_WAVE));
    x = fmaxf(x, __shfl_xor(x, 4, SEG3D_WAVE));
    x = fmaxf(x, __shfl_xor(x, 8, SEG3D_WAVE));
    return x;
}

// dot product of two 32-float rows held / addressed by one lane
__device__ __forceinline__ float dot32(const float* __restrict__ a, const float* __restrict__ b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kD / 4; ++i) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + 4 * i);
        const f32x4 y = *reinterpret_cast<const f32x4*>(b + 4 * i);
        s = fmaf(x[0], y[0], fmaf(x[1], y[1], fmaf(x[2], y[2], fmaf(x[3], y[3], s))));
    }
    return s;
}

__global__ __launch_bounds__(kThreads) void knn_attn_fwd(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const int32_t* __restrict__ idx,
                                                         const uint8_t* __restrict__ invalid, const float* __restrict__ keep,
                                                         int64_t n, int K, float scale, float* __restrict__ out,
                                                         float* __restrict__ prob) {
    const int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 4;  // query of this 16-lane group
    const int j = threadIdx.x & 15;
    const bool live = i < n;  // (whole groups are live or not: the shuffles below stay inside a group)
    const int64_t qi = live ? i : 0;
    const bool has = live && j < K;
    const int32_t nb = has ? idx[qi * K + j] : 0;
    float s = -INFINITY;
    if (has && !(invalid && invalid[nb])) s = dot32(q + qi * kD, k + (int64_t)nb * kD) * scale;
    const float m = group16_max(s);
    float p = (s == -INFINITY) ? 0.f : expf(s - m);  // m == -inf only when every neighbour is masked: p = 0 (nan_to_num)
    const float den = group16_sum(p);
    p = den > 0.f ? p / den : 0.f;
    if (prob && has) prob[qi * K + j] = p;    // the softmax itself (before dropout): what the backward differentiates
    if (keep && has) p *= keep[qi * K + j];  // F.dropout factors: 0 or 1 / (1 - p_drop)
    // P.V: lane l owns channels 2l, 2l + 1
    f32x2 acc = {0.f, 0.f};
    for (int t = 0; t < K; ++t) {
        const float pt = __shfl(p, (threadIdx.x & ~15) + t, SEG3D_WAVE);
        const int32_t nt = __shfl(nb, (threadIdx.x & ~15) + t, SEG3D_WAVE);
        const f32x2 vv = *reinterpret_cast<const f32x2*>(v + (int64_t)nt * kD + 2 * j);
        acc[0] = fmaf(pt, vv[0], acc[0]);
        acc[1] = fmaf(pt, vv[1], acc[1]);
    }
    if (live) *reinterpret_cast<f32x2*>(out + qi * kD + 2 * j) = acc;
}

__global__ __launch_bounds__(kThreads) void knn_attn_bwd_q(const float* __restrict__ k, const float* __restrict__ v,
                                                           const int32_t* __restrict__ idx, const float* __restrict__ keep,
                                                           const float* __restrict__ prob, const float* __restrict__ dout,
                                                           int64_t n, int K, float scale, float* __restrict__ dq,
                                                           float* __restrict__ ds_out, float* __restrict__ pd_out) {
    const int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 4;
    const int j = threadIdx.x & 15;
    const bool live = i < n;
    const int64_t qi = live ? i : 0;
    const bool has = live && j < K;
    const int32_t nb = has ? idx[qi * K + j] : 0;
    // out = sum_j P_j D_j v_j (P the softmax, D the dropout factor):  dP~_j = <dout, v_j>,  delta = sum_t P_t D_t dP~_t,
    // dS_j = P_j (D_j dP~_j - delta); masked neighbours have P = 0 and receive nothing
    const float pj = has ? prob[qi * K + j] : 0.f;
    const float dj = has ? (keep ? keep[qi * K + j] : 1.0f) : 0.f;
    const float dp = has ? dot32(dout + qi * kD, v + (int64_t)nb * kD) : 0.f;
    const float delta = group16_sum(pj * dj * dp);
    const float ds = pj * (dj * dp - delta);
    if (has) {
        ds_out[qi * K + j] = ds;
        pd_out[qi * K + j] = pj * dj;
    }
    f32x2 acc = {0.f, 0.f};
    for (int t = 0; t < K; ++t) {
        const float dt = __shfl(ds, (threadIdx.x & ~15) + t, SEG3D_WAVE);
        const int32_t nt = __shfl(nb, (threadIdx.x & ~15) + t, SEG3D_WAVE);
        const f32x2 kk = *reinterpret_cast<const f32x2*>(k + (int64_t)nt * kD + 2 * j);
        acc[0] = fmaf(dt, kk[0], acc[0]);
        acc[1] = fmaf(dt, kk[1], acc[1]);
    }
    if (live) *reinterpret_cast<f32x2*>(dq + qi * kD + 2 * j) = (f32x2){acc[0] * scale, acc[1] * scale};
}

// one 16-lane group per source row r: sums over the pairs e = (i, j) with idx[i][j] == r (order / offsets = CSR of the
// flattened index table by source row, ascending e inside a row: fixed summation order)
__global__ __launch_bounds__(kThreads) void knn_attn_bwd_kv(const float* __restrict__ q, const float* __restrict__ dout,
                                                            const float* __restrict__ pd, const float* __restrict__ ds,
                                                            const int32_t* __restrict__ order, const int32_t* __restrict__ offsets,
                                                            int64_t n_src, int K, float scale, float* __restrict__ dk,
                                                            float* __restrict__ dv) {
    const int64_t r = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    if (r >= n_src) return;
    f32x2 ak = {0.f, 0.f}, av = {0.f, 0.f};
    const int32_t b = offsets[r], e = offsets[r + 1];
    for (int32_t t = b; t < e; ++t) {
        const int32_t pair = order[t];
        const int64_t i = pair / K;
        const float dsv = ds[pair], pv = pd[pair];
        const f32x2 qq = *reinterpret_cast<const f32x2*>(q + i * kD + 2 * l);
        const f32x2 gg = *reinterpret_cast<const f32x2*>(dout + i * kD + 2 * l);
        ak[0] = fmaf(dsv, qq[0], ak[0]);
        ak[1] = fmaf(dsv, qq[1], ak[1]);
        av[0] = fmaf(pv, gg[0], av[0]);
        av[1] = fmaf(pv, gg[1], av[1]);
    }
    *reinterpret_cast<f32x2*>(dk + r * kD + 2 * l) = (f32x2){ak[0] * scale, ak[1] * scale};
    *reinterpret_cast<f32x2*>(dv + r * kD + 2 * l) = av;
}

bool bad_ptr16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

}  // namespace

extern "C" {

int seg3d_knn_attention_fwd(const float* q, const float* k, const float* v, const int32_t* idx, const uint8_t* invalid,
                            const float* keep, int64_t n, int64_t n_src, int32_t n_neighbors, int32_t d, float scale,
                            float* out, float* prob, void* stream) {
    if (n < 0 || n_src < 0 || d != kD || n_neighbors < 1 || n_neighbors > 16) return SEG3D_EINVAL;
    if (n == 0) return SEG3D_OK;
    if (!q || !k || !v || !idx || !out || n_src == 0 || bad_ptr16(q) || bad_ptr16(k) || bad_ptr16(v) || bad_ptr16(out))
        return SEG3D_EINVAL;
    const unsigned blocks = (unsigned)ceil_div64(n * 16, kThreads);
    hipLaunchKernelGGL(knn_attn_fwd, dim3(blocks), dim3(kThreads), 0, as_stream(stream), q, k, v, idx, invalid, keep, n,
                       n_neighbors, scale, out, prob);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_knn_attention_bwd(const float* q, const float* k, const float* v, const int32_t* idx, const float* keep,
                            const float* prob, const float* dout, const int32_t* pair_order, const int32_t* pair_offsets,
                            int64_t n, int64_t n_src, int32_t n_neighbors, int32_t d, float scale, float* dq, float* dk,
                            float* dv, float* scratch /* 2 * n * n_neighbors floats */, void* stream) {
    if (n < 0 || n_src < 0 || d != kD || n_neighbors < 1 || n_neighbors > 16) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    if (n_src > 0 && (!dk || !dv)) return SEG3D_EINVAL;
    if (n == 0) {
        if (n_src > 0) {
            SEG3D_CHECK_HIP(hipMemsetAsync(dk, 0, (size_t)n_src * kD * sizeof(float), st));
            SEG3D_CHECK_HIP(hipMemsetAsync(dv, 0, (size_t)n_src * kD * sizeof(float), st));
        }
        return SEG3D_OK;
    }
    if (!q || !k || !v || !idx || !prob || !dout || !pair_order || !pair_offsets || !dq || !scratch || n_src == 0 ||
        bad_ptr16(q) || bad_ptr16(k) || bad_ptr16(v) || bad_ptr16(dout) || n * n_neighbors >= 0x7FFFFFFF)
        return SEG3D_EINVAL;
    float* ds_scratch = scratch;
    float* pd_scratch = scratch + n * n_neighbors;
    hipLaunchKernelGGL(knn_attn_bwd_q, dim3((unsigned)ceil_div64(n * 16, kThreads)), dim3(kThreads), 0, st, k, v, idx, keep, prob,
                       dout, n, n_neighbors, scale, dq, ds_scratch, pd_scratch);
    SEG3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(knn_attn_bwd_kv, dim3((unsigned)ceil_div64(n_src * 16, kThreads)), dim3(kThreads), 0, st, q, dout,
                       pd_scratch, ds_scratch, pair_order, pair_offsets, n_src, n_neighbors, scale, dk, dv);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // extern "C"
