// a14 + CSR build: deterministic in-group rank and grouped order.
// Reference: seg3d/ops/ingroup_inds/src/ingroup_inds_cuda.cu:12-25 (rank = atomicAdd arrival order,
// cudaMalloc/cudaMemset/cudaFree and a .max().item() sync per call, ingroup_inds.cpp:36).
//
// Here: count per group (integer atomics) -> exclusive scan -> unordered fill -> every element
// ranks itself inside its group's list by counting smaller element ids.  The result is the
// stable rank by element index, independent of atomic arrival order; no allocation, no sync.
// Work is sum_g n_g^2 compares on L2-resident lists (windows <= 800 tokens, voxels a few points).
#include "common.hpp"

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void grp_count(const int32_t* __restrict__ gid, int64_t n, int64_t n_groups,
                                                      uint32_t* __restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const int32_t g = gid[i];
    if (g >= 0 && g < n_groups) atomicAdd(&count[g], 1u);
}

__global__ __launch_bounds__(kThreads) void grp_fill(const int32_t* __restrict__ gid, int64_t n, int64_t n_groups,
                                                     const uint32_t* __restrict__ offs, uint32_t* __restrict__ cursor,
                                                     int32_t* __restrict__ unordered) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const int32_t g = gid[i];
    if (g >= 0 && g < n_groups) unordered[offs[g] + atomicAdd(&cursor[g], 1u)] = (int32_t)i;
}

__global__ __launch_bounds__(kThreads) void grp_rank(const int32_t* __restrict__ gid, int64_t n, int64_t n_groups,
                                                     const uint32_t* __restrict__ offs, const uint32_t* __restrict__ count,
                                                     const int32_t* __restrict__ unordered, int32_t* __restrict__ rank,
                                                     int32_t* __restrict__ order) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const int32_t g = gid[i];
    if (g < 0 || g >= n_groups) {
        if (rank) rank[i] = -1;
        return;
    }
    const uint32_t s = offs[g], c = count[g];
    int32_t r = 0;
    for (uint32_t j = 0; j < c; ++j) r += (unordered[s + j] < (int32_t)i) ? 1 : 0;
    if (rank) rank[i] = r;
    if (order) order[s + r] = (int32_t)i;
}

__global__ __launch_bounds__(kThreads) void grp_offsets_out(const uint32_t* __restrict__ offs, const uint32_t* __restrict__ total,
                                                            int64_t n_groups, int32_t* __restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (g < n_groups) out[g] = (int32_t)offs[g];
    if (g == n_groups) out[g] = (int32_t)*total;
}

struct GrpWs {
    uint32_t *count, *offs, *cursor, *total, *tmp;
    int32_t* unordered;
};

GrpWs carve(void* ws, int64_t n, int64_t ng) {
    WsCarver c(ws);
    GrpWs w;
    w.count = c.take<uint32_t>((size_t)ng + 1);
    w.cursor = c.take<uint32_t>((size_t)ng + 1);
    w.offs = c.take<uint32_t>((size_t)ng + 1);
    w.total = c.take<uint32_t>(4);
    w.tmp = c.take<uint32_t>(scan_tmp_count(ng));
    w.unordered = c.take<int32_t>((size_t)n + 1);
    return w;
}

}  // namespace

// internal entry shared with window.hip / segment users
int group_index_launch(const int32_t* gid, int64_t n, int64_t ng, int32_t* rank, int32_t* order, int32_t* offsets,
                       void* workspace, hipStream_t st, uint32_t** count_out, uint32_t** offs_out) {
    GrpWs w = carve(workspace, n, ng);
    // count and cursor are adjacent pieces: one memset covers both
    const size_t zero_bytes = (size_t)(reinterpret_cast<char*>(w.offs) - reinterpret_cast<char*>(w.count));
    SEG3D_CHECK_HIP(hipMemsetAsync(w.count, 0, zero_bytes, st));
    const unsigned nb = (unsigned)ceil_div64(n > 0 ? n : 1, kThreads);
    if (n > 0) {
        hipLaunchKernelGGL(grp_count, dim3(nb), dim3(kThreads), 0, st, gid, n, ng, w.count);
        SEG3D_CHECK_LAUNCH();
    }
    int rc = scan_exclusive_u32(w.count, w.offs, ng, w.total, w.tmp, st);
    if (rc != SEG3D_OK) return rc;
    if (n > 0) {
        hipLaunchKernelGGL(grp_fill, dim3(nb), dim3(kThreads), 0, st, gid, n, ng, w.offs, w.cursor, w.unordered);
        SEG3D_CHECK_LAUNCH();
        hipLaunchKernelGGL(grp_rank, dim3(nb), dim3(kThreads), 0, st, gid, n, ng, w.offs, w.count, w.unordered, rank,
                           order);
        SEG3D_CHECK_LAUNCH();
    }
    if (offsets) {
        hipLaunchKernelGGL(grp_offsets_out, dim3((unsigned)ceil_div64(ng + 1, kThreads)), dim3(kThreads), 0, st,
                           w.offs, w.total, ng, offsets);
        SEG3D_CHECK_LAUNCH();
    }
    if (count_out) *count_out = w.count;
    if (offs_out) *offs_out = w.offs;
    return SEG3D_OK;
}

extern "C" {

size_t seg3d_group_index_workspace_bytes(int64_t n, int64_t n_groups) {
    if (n < 0) n = 0;
    if (n_groups < 0) n_groups = 0;
    WsCarver c(nullptr);
    c.take<uint32_t>((size_t)n_groups + 1);
    c.take<uint32_t>((size_t)n_groups + 1);
    c.take<uint32_t>((size_t)n_groups + 1);
    c.take<uint32_t>(4);
    c.take<uint32_t>(scan_tmp_count(n_groups));
    c.take<int32_t>((size_t)n + 1);
    return c.off;
}

int seg3d_group_index(const int32_t* group_ids, int64_t n, int64_t n_groups, int32_t* rank, int32_t* order,
                      int32_t* offsets, void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || n_groups < 0 || n >= (int64_t)0x7F000000 || !workspace) return SEG3D_EINVAL;
    if (n > 0 && !group_ids) return SEG3D_EINVAL;
    if (workspace_bytes < seg3d_group_index_workspace_bytes(n, n_groups)) return SEG3D_EWORKSPACE;
    return group_index_launch(group_ids, n, n_groups, rank, order, offsets, workspace, as_stream(stream), nullptr,
                              nullptr);
}

}  // extern "C"
