// device cross-entropy over per-point / per-voxel logits [n, C] (C = 22 classes on Waymo): the 'ce' and
// 'ohem_ce' terms of build_criterion (seg3d/models/builder.py:26-40).  With keep_thresh > 0 it is
// OHEMCrossEntropyLoss(keep_thresh) (ohem_cross_entropy_loss.py:23-38): only rows whose softmax probability of the
// target class is < keep_thresh are averaged.  The Lovasz term lives in lovasz.hip.
// torch's nll_loss forward / backward reduce kernels run in ONE workgroup (113 / 88 us at n = 175 k); this is one
// pass each way: a thread owns a row (C <= 64 floats), rows with label == ignore_index contribute nothing.
//   forward : lse[r] = log sum exp(x[r]) on counted rows, +inf on the others;
//             acc[0] += lse[r] - x[r][label],  acc[1] += 1                                   (per-block partials)
//   backward: dx[r][c] = (softmax(x[r])[c] - [c == label]) * g / count on counted rows, 0 elsewhere
#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;

__global__ __launch_bounds__(kThreads) void ce_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ label,
                                                          int64_t n, int c, int64_t ignore_index, float keep_thresh,
                                                          float* __restrict__ lse,
                                                          float* __restrict__ part /*[gridDim.x][2]*/) {
    float loss = 0.f, cnt = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x; r < n; r += (int64_t)gridDim.x * kThreads) {
        const float* row = x + r * c;
        float m = row[0];
        for (int j = 1; j < c; ++j) m = fmaxf(m, row[j]);
        float s = 0.f;
        for (int j = 0; j < c; ++j) s += expf(row[j] - m);
        const float l = m + logf(s);
        const int64_t y = label[r];
        bool counted = y != ignore_index && y >= 0 && y < c;
        if (counted && keep_thresh > 0.f) counted = expf(row[y] - m) / s < keep_thresh;  // softmax(x)[y] < keep_thresh
        lse[r] = counted ? l : INFINITY;
        if (counted) {
            loss += l - row[y];
            cnt += 1.f;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        loss += __shfl_xor(loss, off, SEG3D_WAVE);
        cnt += __shfl_xor(cnt, off, SEG3D_WAVE);
    }
    __shared__ float red[2][kThreads / 64];
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = loss;
        red[1][threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < kThreads / 64; ++w) {
            a += red[0][w];
            b += red[1][w];
        }
        part[2 * blockIdx.x] = a;
        part[2 * blockIdx.x + 1] = b;
    }
}

// out[0] = mean loss over counted rows (0 when none), out[1] = count; fixed summation order
__global__ __launch_bounds__(kThreads) void ce_finalize_kernel(const float* __restrict__ part, int nblocks,
                                                               float* __restrict__ out) {
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += kThreads) {
        a += part[2 * i];
        b += part[2 * i + 1];
    }
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, SEG3D_WAVE);
        b += __shfl_xor(b, off, SEG3D_WAVE);
    }
    __shared__ float red[2][kThreads / 64];
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sa = 0.f, sb = 0.f;
        for (int w = 0; w < kThreads / 64; ++w) {
            sa += red[0][w];
            sb += red[1][w];
        }
        out[0] = sb > 0.f ? sa / sb : 0.f;
        out[1] = sb;
    }
}

__global__ __launch_bounds__(kThreads) void ce_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ label,
                                                          const float* __restrict__ lse, const float* __restrict__ stats,
                                                          const float* __restrict__ gout, int64_t n, int c,
                                                          float* __restrict__ dx) {
    const float cnt = stats[1];
    const float scale = cnt > 0.f ? gout[0] / cnt : 0.f;
    const int64_t total = n * c;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t r = e / c;
        const int j = (int)(e - r * c);
        const float l = lse[r];
        float g = 0.f;
        if (l != INFINITY) g = (expf(x[e] - l) - (j == label[r] ? 1.f : 0.f)) * scale;
        dx[e] = g;
    }
}

}  // namespace

extern "C" size_t seg3d_cross_entropy_workspace_bytes(int64_t n) {
    return n < 0 ? 0 : (size_t)kMaxBlocks * 2 * sizeof(float);
}

extern "C" int seg3d_cross_entropy_fwd(const float* logits, const int64_t* labels, int64_t n, int32_t c,
                                       int64_t ignore_index, float keep_thresh, float* lse,
                                       float* stats /*[2]: mean loss, count*/,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || c <= 0 || c > 4096 || !stats || !workspace || workspace_bytes < seg3d_cross_entropy_workspace_bytes(n))
        return SEG3D_EINVAL;
    if (n > 0 && (!logits || !labels || !lse)) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    int nb = (int)(n > 0 ? ceil_div64(n, kThreads) : 0);
    if (nb > kMaxBlocks) nb = kMaxBlocks;
    if (nb > 0) {
        hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)nb), dim3(kThreads), 0, st, logits, labels, n, c, ignore_index,
                           keep_thresh, lse, part);
        SEG3D_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(kThreads), 0, st, part, nb, stats);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

extern "C" int seg3d_cross_entropy_bwd(const float* logits, const int64_t* labels, const float* lse, const float* stats,
                                       const float* grad_out /*[1]*/, int64_t n, int32_t c, float* dlogits,
                                       void* stream) {
    if (n < 0 || c <= 0 || c > 4096 || !stats || !grad_out) return SEG3D_EINVAL;
    if (n == 0) return SEG3D_OK;
    if (!logits || !labels || !lse || !dlogits) return SEG3D_EINVAL;
    int64_t nb = ceil_div64(n * c, (int64_t)kThreads * 4);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)nb), dim3(kThreads), 0, as_stream(stream), logits, labels, lse, stats,
                       grad_out, n, c, dlogits);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}
