// batched brute-force k-nearest-neighbour query.
// Reference: seg3d/ops/knn_query/src/knn_query_cuda.cu:67-112 (one thread per query scanning its whole batch
// segment from global memory with a 100-entry max-heap in local memory, then a heap sort), called from
// DeepFusionBlock (deep_fusion.py:31, k = 16) and the auxiliary-label lookup (tools/train.py:103, k = 1).
//
// Same contract -- xyz [n,3] / new_xyz [m,3] contiguous float32 with stride 3, cumulative int32 offsets per
// batch sample, outputs idx int32 [m,k] and squared distances [m,k] in ascending order, unfilled slots =
// (1e10, segment start) -- with this canonical tie rule: equal distances are ordered by ascending candidate
// index (the reference's heap order among exact ties is an artefact of its sift-down sequence).
// d2 = (dx*dx + dy*dy) + dz*dz is evaluated without FMA contraction, so indices are reproducible bit for bit.
//
// MI355X design: a workgroup of 256 consecutive queries walks the candidate range of the segments it spans in
// 1024-point tiles staged through LDS (12 KiB, coalesced loads, broadcast reads), each lane keeps its K best
// in a sorted register list (insertion only when a candidate beats the current worst).  O(m*n_b) like the
// reference; a grid-hash variant is the follow-up for the 180k x 180k fusion query.
#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kTile = 1024;

__device__ __forceinline__ int segment_of(int i, const int32_t* __restrict__ offs, int b) {
    int s = 0;
    while (s < b - 1 && i >= offs[s]) ++s;
    return s;
}

template <int K>
__global__ __launch_bounds__(kThreads) void knn_kernel(const float* __restrict__ xyz, const float* __restrict__ qxyz,
                                                       const int32_t* __restrict__ offset,
                                                       const int32_t* __restrict__ new_offset, int b, int m, int k,
                                                       int32_t* __restrict__ idx_out, float* __restrict__ d2_out) {
    __shared__ float tile[kTile * 3];
    const int q = blockIdx.x * kThreads + threadIdx.x;
    const bool live = q < m;
    const int qc = live ? q : m - 1;
    // segments covered by this block's queries
    const int q_first = blockIdx.x * kThreads;
    const int q_last = min(q_first + kThreads, m) - 1;
    const int s_first = segment_of(q_first, new_offset, b), s_last = segment_of(q_last, new_offset, b);
    const int my_seg = segment_of(qc, new_offset, b);
    const int my_start = my_seg == 0 ? 0 : offset[my_seg - 1], my_end = offset[my_seg];
    const float qx = qxyz[3 * (int64_t)qc + 0], qy = qxyz[3 * (int64_t)qc + 1], qz = qxyz[3 * (int64_t)qc + 2];

    float bd[K];
    int32_t bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        bd[j] = 1e10f;
        bi[j] = my_start;
    }
    const int c_begin = s_first == 0 ? 0 : offset[s_first - 1], c_end = offset[s_last];
    for (int t0 = c_begin; t0 < c_end; t0 += kTile) {
        const int cnt = min(kTile, c_end - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * 3; e += kThreads) tile[e] = xyz[3 * (int64_t)t0 + e];
        __syncthreads();
        // candidates of this tile that belong to my segment
        const int lo = max(my_start, t0) - t0, hi = min(my_end, t0 + cnt) - t0;
        if (live) {
            for (int c = lo; c < hi; ++c) {
                const float dx = qx - tile[3 * c + 0], dy = qy - tile[3 * c + 1], dz = qz - tile[3 * c + 2];
                const float d2 = (dx * dx + dy * dy) + dz * dz;
                if (d2 < bd[K - 1]) {  // the list keeps K >= k entries; the first k are what is asked for
                    // stable insertion into the ascending list (after every entry with d <= d2)
                    float cd = d2;
                    int32_t ci = t0 + c;
                    bool shifting = false;  // once inserted, everything behind moves down one slot (ties included)
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        if (shifting || cd < bd[j]) {
                            shifting = true;
                            const float td = bd[j];
                            const int32_t ti = bi[j];
                            bd[j] = cd;
                            bi[j] = ci;
                            cd = td;
                            ci = ti;
                        }
                    }
                }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (j < k) {
                idx_out[(int64_t)q * k + j] = bi[j];
                d2_out[(int64_t)q * k + j] = bd[j];
            }
        }
    }
}

}  // namespace

extern "C" int seg3d_knn_query(const float* xyz, int64_t n, const float* new_xyz, int64_t m, const int32_t* offset,
                               const int32_t* new_offset, int32_t batch_size, int32_t k, int32_t* idx, float* dist2,
                               void* stream) {
    if (n < 0 || m < 0 || batch_size <= 0 || k <= 0 || k > 64 || m >= (int64_t)0x7FFFFFF0 || n >= (int64_t)0x7FFFFFF0)
        return SEG3D_EINVAL;
    if (m == 0) return SEG3D_OK;
    if (!xyz || !new_xyz || !offset || !new_offset || !idx || !dist2 || n == 0) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    const unsigned nb = (unsigned)ceil_div64(m, kThreads);
#define SEG3D_KNN(KK)                                                                                              \
    hipLaunchKernelGGL(knn_kernel<KK>, dim3(nb), dim3(kThreads), 0, st, xyz, new_xyz, offset, new_offset, batch_size, \
                       (int)m, (int)k, idx, dist2)
    if (k == 1) SEG3D_KNN(1);
    else if (k <= 4) SEG3D_KNN(4);
    else if (k <= 16) SEG3D_KNN(16);
    else if (k <= 32) SEG3D_KNN(32);
    else SEG3D_KNN(64);
#undef SEG3D_KNN
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}
