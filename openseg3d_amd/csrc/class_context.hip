// SURVEY 8(f) rank 3, OCR's SpatialGatherModule (seg3d/models/layers/ocr.py:10-36): per sample b, the class proxies
//     context[b, k, :] = sum_{r in sample b} softmax_r(scale * probs[r, k]) * feats[r, :]
// -- a softmax over the sample's VOXELS for every class column, then a [classes x rows] . [rows x C] product.  The
// reference loops over the samples in Python (boolean masks, a softmax and a matmul per sample); here the samples are
// spans of rows (voxel_id_offset-style cumulative counts: the stride-8 level is sorted by batch index) and one fixed launch
// sequence serves any batch size, forward and backward, without atomics:
//   forward   (1) per (sample, class) column: running maximum and sum of exponentials over the span, one workgroup per
//                 sample;  (2) the softmax weights w [rows, classes] and, per 128-row chunk, the partial products
//                 sum_r w[r, k] feats[r, c] (thread = channel, 32 class accumulators, w broadcast from LDS);
//             (3) fixed-order sum of the chunk partials.
//   backward  d feats[r, :] = sum_k w[r, k] d ctx[b, k, :];  g[r, k] = <feats[r, :], d ctx[b, k, :]>,
//             d probs[r, k] = scale * w[r, k] * (g[r, k] - sum_r' w[r', k] g[r', k]):  (4) per chunk: d feats, w * g and the
//             chunk's column sums of w * g;  (5) per (sample, class): fixed-order sum of the chunk sums, then (6) d probs.
// classes <= 32, channels a multiple of 4 and <= 1024.
#include "common.hpp"

#include <math.h>

namespace {

constexpr int kMaxK = 32;
constexpr int kChunk = 128;  // rows per partial product

__device__ __forceinline__ int sample_of(const int32_t* __restrict__ offsets, int batch, int64_t row) {
    int b = 0;
    while (b + 1 < batch && row >= offsets[b]) ++b;
    return b;
}

// (1) one workgroup per sample: stats[b][k] = (max over the span of scale * x, 1 / sum exp(scale * x - max))
__global__ __launch_bounds__(1024) void ctx_stats(const float* __restrict__ probs, const int32_t* __restrict__ offsets, int K,
                                                  float scale, float2* __restrict__ stats) {
    __shared__ float red[32][kMaxK + 1];
    const int b = blockIdx.x;
    const int64_t r0 = b ? offsets[b - 1] : 0, r1 = offsets[b];
    const int k = threadIdx.x & 31, lane_r = threadIdx.x >> 5;  // 32 row lanes x 32 class slots
    float mx = -INFINITY;
    if (k < K)
        for (int64_t r = r0 + lane_r; r < r1; r += 32) mx = fmaxf(mx, scale * probs[r * K + k]);
    red[lane_r][k] = mx;
    __syncthreads();
    float m = red[0][k];
    for (int l = 1; l < 32; ++l) m = fmaxf(m, red[l][k]);
    __syncthreads();
    float s = 0.f;
    if (k < K)
        for (int64_t r = r0 + lane_r; r < r1; r += 32) s += expf(scale * probs[r * K + k] - m);
    red[lane_r][k] = s;
    __syncthreads();
    if (lane_r == 0 && k < K) {
        float t = red[0][k];
        for (int l = 1; l < 32; ++l) t += red[l][k];
        stats[b * K + k] = make_float2(m, t > 0.f ? 1.0f / t : 0.f);
    }
}

// (2) chunk of kChunk rows (chunks never straddle samples: the chunk list is per sample): weights + partial products
__global__ __launch_bounds__(256) void ctx_partial(const float* __restrict__ feats, const float* __restrict__ probs,
                                                   const int32_t* __restrict__ offsets, const int2* __restrict__ chunks,
                                                   const float2* __restrict__ stats, int K, int C, float scale,
                                                   float* __restrict__ w_out, float* __restrict__ part) {
    __shared__ float w_lds[kChunk][kMaxK];
    const int2 ch = chunks[blockIdx.x];  // (sample, first row)
    const int b = ch.x;
    const int64_t r0 = ch.y, r_end = offsets[b];
    const int rows = (int)min((int64_t)kChunk, r_end - r0);
    for (int i = threadIdx.x; i < rows * K; i += 256) {
        const int r = i / K, k = i - r * K;
        const float2 st = stats[b * K + k];
        const float w = expf(scale * probs[(r0 + r) * K + k] - st.x) * st.y;
        w_lds[r][k] = w;
        w_out[(r0 + r) * K + k] = w;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc[kMaxK];
#pragma unroll
        for (int k = 0; k < kMaxK; ++k) acc[k] = 0.f;
        for (int r = 0; r < rows; ++r) {
            const float f = feats[(r0 + r) * C + c];
#pragma unroll
            for (int k = 0; k < kMaxK; ++k) acc[k] = fmaf(k < K ? w_lds[r][k] : 0.f, f, acc[k]);
        }
#pragma unroll
        for (int k = 0; k < kMaxK; ++k)
            if (k < K) part[((size_t)blockIdx.x * K + k) * C + c] = acc[k];
    }
}

// (3) context[b][k][c] = sum of the sample's chunk partials, ascending chunk order
__global__ __launch_bounds__(256) void ctx_reduce(const float* __restrict__ part, const int32_t* __restrict__ chunk_offsets,
                                                  int K, int C, float* __restrict__ ctx) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;  // (k, c) flattened
    if (i >= K * C) return;
    const int c0 = b ? chunk_offsets[b - 1] : 0, c1 = chunk_offsets[b];
    float s = 0.f;
    for (int ch = c0; ch < c1; ++ch) s += part[(size_t)ch * K * C + i];
    ctx[(size_t)b * K * C + i] = s;
}

// (4) backward per chunk: d feats rows, wg = w * g, and the chunk's column sums of wg
__global__ __launch_bounds__(256) void ctx_bwd_rows(const float* __restrict__ feats, const float* __restrict__ w,
                                                    const float* __restrict__ dctx, const int32_t* __restrict__ offsets,
                                                    const int2* __restrict__ chunks, int K, int C, float* __restrict__ dfeats,
                                                    float* __restrict__ wg, float* __restrict__ colsum /*[chunks][K]*/) {
    extern __shared__ float lds[];  // d ctx[b] [K][C] + wg of the chunk [kChunk][kMaxK]
    float* d_lds = lds;
    float* wg_lds = lds + (size_t)K * C;
    const int2 ch = chunks[blockIdx.x];
    const int b = ch.x;
    const int64_t r0 = ch.y, r_end = offsets[b];
    const int rows = (int)min((int64_t)kChunk, r_end - r0);
    for (int i = threadIdx.x; i < K * C; i += 256) d_lds[i] = dctx[(size_t)b * K * C + i];
    __syncthreads();
    // d feats[r][c] = sum_k w[r][k] d ctx[k][c]: thread = channel, rows in turn
    for (int c = threadIdx.x; c < C; c += 256)
        for (int r = 0; r < rows; ++r) {
            float s = 0.f;
            for (int k = 0; k < K; ++k) s = fmaf(w[(r0 + r) * K + k], d_lds[k * C + c], s);
            dfeats[(r0 + r) * C + c] = s;
        }
    // g[r][k] = <feats[r], d ctx[k]>: one (row, class) pair per thread slot
    for (int i = threadIdx.x; i < rows * K; i += 256) {
        const int r = i / K, k = i - r * K;
        const float* f = feats + (r0 + r) * C;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(f[c], d_lds[k * C + c], s);
        const float v = w[(r0 + r) * K + k] * s;
        wg_lds[r * kMaxK + k] = v;
        wg[(r0 + r) * K + k] = v;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += wg_lds[r * kMaxK + threadIdx.x];
        colsum[(size_t)blockIdx.x * K + threadIdx.x] = s;
    }
}

// (5) + (6): d probs[r][k] = scale * (wg[r][k] - w[r][k] * delta[b][k]), delta = fixed-order sum of the chunk sums
__global__ __launch_bounds__(256) void ctx_bwd_probs(const float* __restrict__ w, const float* __restrict__ wg,
                                                     const float* __restrict__ colsum, const int32_t* __restrict__ offsets,
                                                     const int32_t* __restrict__ chunk_offsets, int batch, int K, int64_t m,
                                                     float scale, float* __restrict__ dprobs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m * K) return;
    const int64_t r = i / K;
    const int k = (int)(i - r * K);
    const int b = sample_of(offsets, batch, r);
    const int c0 = b ? chunk_offsets[b - 1] : 0, c1 = chunk_offsets[b];
    float delta = 0.f;
    for (int ch = c0; ch < c1; ++ch) delta += colsum[(size_t)ch * K + k];
    dprobs[i] = scale * (wg[i] - w[i] * delta);
}

// classes x channels is capped by the backward's dynamic LDS block (classes * C + 128 rows * 32 classes floats) at the
// 64 KiB a kernel gets without opting in to more: the validated range (SPNet: 22 x 128 = 27 KiB)
bool bad_shape(int K, int C) {
    return K < 1 || K > kMaxK || C < 4 || (C & 3) || C > 1024 || ((size_t)K * C + (size_t)kChunk * kMaxK) * sizeof(float) > 64 * 1024;
}

}  // namespace

extern "C" {

/* chunks [n_chunks][2] = (sample, first row) of every 128-row chunk, samples in order; chunk_offsets [batch] = cumulative
 * chunk counts; both built by the caller from the cumulative row offsets (host integers: the stride-8 level's per-sample
 * row counts are read back once per batch with the rest of the index plan). */
int seg3d_class_context_fwd(const float* feats, const float* probs, const int32_t* offsets, const int32_t* chunks,
                            const int32_t* chunk_offsets, int32_t n_chunks, int32_t batch, int64_t m, int32_t classes,
                            int32_t c, float scale, float* weights /*[m, classes]*/, float* partials /*[n_chunks, classes, c]*/,
                            float* stats /*[batch, classes, 2]*/, float* context /*[batch, classes, c]*/, void* stream) {
    if (batch < 0 || m < 0 || n_chunks < 0 || bad_shape(classes, c)) return SEG3D_EINVAL;
    if (batch == 0) return SEG3D_OK;
    if (!offsets || !chunk_offsets || !context || !stats || (m > 0 && (!feats || !probs || !chunks || !weights || !partials)))
        return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(ctx_stats, dim3((unsigned)batch), dim3(1024), 0, st, probs, offsets, classes, scale,
                       reinterpret_cast<float2*>(stats));
    SEG3D_CHECK_LAUNCH();
    if (n_chunks > 0) {
        hipLaunchKernelGGL(ctx_partial, dim3((unsigned)n_chunks), dim3(256), 0, st, feats, probs, offsets,
                           reinterpret_cast<const int2*>(chunks), reinterpret_cast<const float2*>(stats), classes, c, scale,
                           weights, partials);
        SEG3D_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(ctx_reduce, dim3((unsigned)((classes * c + 255) / 256), (unsigned)batch), dim3(256), 0, st, partials,
                       chunk_offsets, classes, c, context);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_class_context_bwd(const float* feats, const float* weights, const float* dcontext, const int32_t* offsets,
                            const int32_t* chunks, const int32_t* chunk_offsets, int32_t n_chunks, int32_t batch, int64_t m,
                            int32_t classes, int32_t c, float scale, float* dfeats, float* dprobs,
                            float* scratch /*[m * classes + n_chunks * classes]*/, void* stream) {
    if (batch < 0 || m < 0 || n_chunks < 0 || bad_shape(classes, c)) return SEG3D_EINVAL;
    if (batch == 0 || m == 0) return SEG3D_OK;
    if (!feats || !weights || !dcontext || !offsets || !chunks || !chunk_offsets || !dfeats || !dprobs || !scratch)
        return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    float* wg = scratch;
    float* colsum = scratch + m * classes;
    const size_t smem = ((size_t)classes * c + (size_t)kChunk * kMaxK) * sizeof(float);
    hipLaunchKernelGGL(ctx_bwd_rows, dim3((unsigned)n_chunks), dim3(256), smem, st, feats, weights, dcontext, offsets,
                       reinterpret_cast<const int2*>(chunks), classes, c, dfeats, wg, colsum);
    SEG3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(ctx_bwd_probs, dim3((unsigned)ceil_div64(m * classes, 256)), dim3(256), 0, st, weights, wg, colsum, offsets,
                       chunk_offsets, batch, classes, m, scale, dprobs);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // extern "C"
