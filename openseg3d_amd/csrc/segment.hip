// a7, a24, a25, a26: segmented reduce (max / mean / sum) and row gather, forward and backward.
// Reference: torch_scatter.scatter at seg3d/models/voxel_encoders/vfe.py:24-25 and
// seg3d/models/layers/se_layer.py:24-28; voxel_pooling_ext at seg3d/ops/voxel_pooling/src/
// voxel_pooling_cuda.cu:10-79 (one float atomicAdd per element, <<<N, c>>> blocks smaller than a
// wave); VoxelToPoint at seg3d/ops/voxel_to_point/voxel_to_point.py:4-17.
//
// Here the reduce walks a CSR (order, offsets) built once per batch by seg3d_group_index: every
// output element is produced by exactly one thread in a fixed order -- deterministic, no float
// atomics (MI355X global float atomics top out near 1.3 TB/s, plain stores run 4-5x that).
// One thread per (segment, 4 channels): 16-B loads, a wave covers whole 256-B+ rows.
// Algorithmic bytes: (N + M) * C * 4 + N * 8 per launch.
#include "common.hpp"

#include <math.h>

namespace {

constexpr int kThreads = 256;

template <int MODE, int V>
__global__ __launch_bounds__(kThreads) void seg_reduce_kernel(const float* __restrict__ x, int c, const int32_t* __restrict__ order,
                                                              const int32_t* __restrict__ offsets, int64_t n_seg,
                                                              float* __restrict__ out, int32_t* __restrict__ argmax) {
    const int cv = c / V;
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n_seg * cv) return;
    const int64_t s = t / cv;
    const int c0 = (int)(t - s * cv) * V;
    const int32_t b = offsets[s], e = offsets[s + 1];
    float acc[V];
    int32_t arg[V];
#pragma unroll
    for (int u = 0; u < V; ++u) {
        acc[u] = (MODE == SEG3D_REDUCE_MAX) ? -INFINITY : 0.f;
        arg[u] = -1;
    }
    for (int32_t j = b; j < e; ++j) {
        const int32_t row = order[j];
        const float* xp = x + (int64_t)row * c + c0;
        float val[V];
        if (V == 4) {
            const float4 t4 = *reinterpret_cast<const float4*>(xp);
            val[0] = t4.x; val[1 % V] = t4.y; val[2 % V] = t4.z; val[3 % V] = t4.w;
        } else {
#pragma unroll
            for (int u = 0; u < V; ++u) val[u] = xp[u];
        }
#pragma unroll
        for (int u = 0; u < V; ++u) {
            if (MODE == SEG3D_REDUCE_MAX) {
                if (val[u] > acc[u] || arg[u] < 0) {
                    acc[u] = val[u];
                    arg[u] = row;
                }
            } else {
                acc[u] += val[u];
            }
        }
    }
    const float scale = (MODE == SEG3D_REDUCE_MEAN && e > b) ? 1.0f / (float)(e - b) : 1.0f;
#pragma unroll
    for (int u = 0; u < V; ++u) {
        float r = acc[u];
        if (MODE == SEG3D_REDUCE_MAX && e == b) r = 0.f;
        if (MODE == SEG3D_REDUCE_MEAN) r = r * scale;
        out[s * c + c0 + u] = r;
        if (MODE == SEG3D_REDUCE_MAX && argmax) argmax[s * c + c0 + u] = arg[u];
    }
}

// SUM / MEAN backward: dx[i] = dout[seg[i]] * (1 or 1/count)
template <int MODE>
__global__ __launch_bounds__(kThreads) void seg_bwd_bcast_kernel(const float* __restrict__ dout, int c,
                                                                 const int32_t* __restrict__ seg_of_row, int64_t n,
                                                                 const int32_t* __restrict__ offsets, float* __restrict__ dx) {
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n * c) return;
    const int64_t i = t / c;
    const int ch = (int)(t - i * c);
    const int32_t s = seg_of_row[i];
    float g = 0.f;
    if (s >= 0) {
        g = dout[(int64_t)s * c + ch];
        if (MODE == SEG3D_REDUCE_MEAN) g = g / (float)(offsets[s + 1] - offsets[s]);
    }
    dx[t] = g;
}

// MAX backward: route each output gradient to its arg-max row (unique writer per (row, channel))
__global__ __launch_bounds__(kThreads) void seg_bwd_max_kernel(const float* __restrict__ dout, int c,
                                                               const int32_t* __restrict__ argmax, int64_t n_seg,
                                                               float* __restrict__ dx) {
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n_seg * c) return;
    const int32_t row = argmax[t];
    if (row >= 0) dx[(int64_t)row * c + (t % c)] = dout[t];
}

template <int V>
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(const float* __restrict__ feats, const int32_t* __restrict__ ids,
                                                               int64_t n, int c, float* __restrict__ out) {
    const int cv = c / V;
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (t >= n * cv) return;
    const int64_t i = t / cv;
    const int c0 = (int)(t - i * cv) * V;
    const int32_t id = ids[i];
    if (V == 4) {
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id >= 0) val = *reinterpret_cast<const float4*>(feats + (int64_t)id * c + c0);
        *reinterpret_cast<float4*>(out + i * c + c0) = val;
    } else {
        out[i * c + c0] = id >= 0 ? feats[(int64_t)id * c + c0] : 0.f;
    }
}

template <int MODE>
int launch_reduce(const float* x, int c, const int32_t* order, const int32_t* offsets, int64_t n_seg, float* out,
                  int32_t* argmax, hipStream_t st) {
    if (c % 4 == 0) {
        hipLaunchKernelGGL((seg_reduce_kernel<MODE, 4>), dim3((unsigned)ceil_div64(n_seg * (c / 4), kThreads)),
                           dim3(kThreads), 0, st, x, c, order, offsets, n_seg, out, argmax);
    } else {
        hipLaunchKernelGGL((seg_reduce_kernel<MODE, 1>), dim3((unsigned)ceil_div64(n_seg * c, kThreads)), dim3(kThreads),
                           0, st, x, c, order, offsets, n_seg, out, argmax);
    }
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

// prepare_voxel_labels (seg3d/datasets/waymo_dataset.py:213-246): per voxel the most frequent
// label of its (current-sweep) points, ties to the smallest label (np.argmax of a 256-bin counter), voxels without a
// point keep ignore_index.  One thread per voxel over the point->voxel CSR; a voxel holds ~1.6 points on average, so
// counting every point's label against the others (n^2) beats a 256-bin histogram.
__global__ __launch_bounds__(kThreads) void voxel_majority_kernel(const uint8_t* __restrict__ labels,
                                                                  const int32_t* __restrict__ order,
                                                                  const int32_t* __restrict__ offsets, int64_t n_seg,
                                                                  int ignore_index, uint8_t* __restrict__ out) {
    const int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (v >= n_seg) return;
    const int b = offsets[v], e = offsets[v + 1];
    int best = ignore_index, best_cnt = 0;
    for (int i = b; i < e; ++i) {
        const int li = labels[order[i]];
        int cnt = 0;
        for (int j = b; j < e; ++j) cnt += labels[order[j]] == li;
        if (cnt > best_cnt || (cnt == best_cnt && li < best)) {
            best = li;
            best_cnt = cnt;
        }
    }
    out[v] = (uint8_t)best;
}

}  // namespace

extern "C" {

int seg3d_voxel_majority_labels(const uint8_t* point_labels, const int32_t* order, const int32_t* offsets, int64_t n_voxels,
                                int32_t ignore_index, uint8_t* voxel_labels, void* stream) {
    if (n_voxels < 0 || ignore_index < 0 || ignore_index > 255) return SEG3D_EINVAL;
    if (n_voxels == 0) return SEG3D_OK;
    if (!point_labels || !order || !offsets || !voxel_labels) return SEG3D_EINVAL;
    hipLaunchKernelGGL(voxel_majority_kernel, dim3((unsigned)ceil_div64(n_voxels, kThreads)), dim3(kThreads), 0,
                       as_stream(stream), point_labels, order, offsets, n_voxels, (int)ignore_index, voxel_labels);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_segment_reduce_fwd(const float* x, int32_t c, const int32_t* order, const int32_t* offsets, int64_t n_seg,
                             int32_t mode, float* out, int32_t* argmax, void* stream) {
    if (n_seg < 0 || c <= 0) return SEG3D_EINVAL;
    if (n_seg == 0) return SEG3D_OK;
    if (!x || !order || !offsets || !out) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    switch (mode) {
        case SEG3D_REDUCE_SUM: return launch_reduce<SEG3D_REDUCE_SUM>(x, c, order, offsets, n_seg, out, argmax, st);
        case SEG3D_REDUCE_MEAN: return launch_reduce<SEG3D_REDUCE_MEAN>(x, c, order, offsets, n_seg, out, argmax, st);
        case SEG3D_REDUCE_MAX: return launch_reduce<SEG3D_REDUCE_MAX>(x, c, order, offsets, n_seg, out, argmax, st);
        default: return SEG3D_EINVAL;
    }
}

int seg3d_segment_reduce_bwd(const float* dout, int32_t c, const int32_t* seg_of_row, int64_t n, const int32_t* offsets,
                             const int32_t* argmax, int64_t n_seg, int32_t mode, float* dx, void* stream) {
    if (n < 0 || n_seg < 0 || c <= 0) return SEG3D_EINVAL;
    if (n == 0 || n_seg == 0) return SEG3D_OK;
    if (!dout || !dx) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    if (mode == SEG3D_REDUCE_MAX) {
        if (!argmax) return SEG3D_EINVAL;
        hipLaunchKernelGGL(seg_bwd_max_kernel, dim3((unsigned)ceil_div64(n_seg * c, kThreads)), dim3(kThreads), 0, st, dout,
                           c, argmax, n_seg, dx);
    } else if (mode == SEG3D_REDUCE_MEAN) {
        if (!seg_of_row || !offsets) return SEG3D_EINVAL;
        hipLaunchKernelGGL(seg_bwd_bcast_kernel<SEG3D_REDUCE_MEAN>, dim3((unsigned)ceil_div64(n * c, kThreads)),
                           dim3(kThreads), 0, st, dout, c, seg_of_row, n, offsets, dx);
    } else if (mode == SEG3D_REDUCE_SUM) {
        if (!seg_of_row) return SEG3D_EINVAL;
        hipLaunchKernelGGL(seg_bwd_bcast_kernel<SEG3D_REDUCE_SUM>, dim3((unsigned)ceil_div64(n * c, kThreads)),
                           dim3(kThreads), 0, st, dout, c, seg_of_row, n, offsets, dx);
    } else {
        return SEG3D_EINVAL;
    }
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_gather_rows(const float* feats, const int32_t* ids, int64_t n, int32_t c, float* out, void* stream) {
    if (n < 0 || c <= 0) return SEG3D_EINVAL;
    if (n == 0) return SEG3D_OK;
    if (!feats || !ids || !out) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    if (c % 4 == 0)
        hipLaunchKernelGGL(gather_rows_kernel<4>, dim3((unsigned)ceil_div64(n * (c / 4), kThreads)), dim3(kThreads), 0, st,
                           feats, ids, n, c, out);
    else
        hipLaunchKernelGGL(gather_rows_kernel<1>, dim3((unsigned)ceil_div64(n * c, kThreads)), dim3(kThreads), 0, st, feats,
                           ids, n, c, out);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // extern "C"
