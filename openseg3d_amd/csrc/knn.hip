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
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

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

// ---------------------------------------------------------------------------------------------------------------
// Grid-accelerated exact variant (same results, same tie rule) for large clouds: the 180k x 180k fusion query costs
// 46 ms by brute force.  Points are binned into cubic cells (key = batch | cx | cy | cz, 16 bits each), sorted by key
// (rocPRIM device radix sort), the non-empty cells go into an open-addressing table, and every query walks the cube shells
// r = 0, 1, 2, .. around its own cell until its K-th distance is strictly below (r * cell)^2 -- every point not yet
// visited is at least that far.  Queries that have not converged after kMaxRing shells (isolated points) scan their
// whole batch segment.  Entries are ordered by (d2, original index), i.e. exactly the brute-force order.
constexpr int kCellBias = 32768;

__device__ __forceinline__ int cell_of(float p, float inv_cell) {
    const float f = floorf(p * inv_cell);
    const int c = (int)fminf(fmaxf(f, -32768.f), 32767.f) + kCellBias;
    return c;
}

__device__ __forceinline__ unsigned long long cell_key(int b, int cx, int cy, int cz) {
    return ((((unsigned long long)b << 16 | (unsigned)cx) << 16 | (unsigned)cy) << 16) | (unsigned)cz;
}

__global__ __launch_bounds__(kThreads) void knn_cell_keys(const float* __restrict__ xyz, int n, const int32_t* __restrict__ offset,
                                                          int b, float inv_cell, unsigned long long* __restrict__ keys,
                                                          uint32_t* __restrict__ rows) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const int seg = segment_of(i, offset, b);
    keys[i] = cell_key(seg, cell_of(xyz[3 * (int64_t)i], inv_cell), cell_of(xyz[3 * (int64_t)i + 1], inv_cell),
                       cell_of(xyz[3 * (int64_t)i + 2], inv_cell));
    rows[i] = (uint32_t)i;
}

// after the (key, row) sort: gather the points into cell order; the first point of every cell finds the end of its
// run by bisection and publishes key -> first position
__global__ __launch_bounds__(kThreads) void knn_level_finish(const unsigned long long* __restrict__ skeys,
                                                             const uint32_t* __restrict__ srows, const float* __restrict__ xyz,
                                                             int n, float* __restrict__ sxyz, int32_t* __restrict__ src,
                                                             int32_t* __restrict__ cell_end,
                                                             unsigned long long* __restrict__ tkeys,
                                                             int32_t* __restrict__ tvals, unsigned cap_mask) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = skeys[i];
    const uint32_t r = srows[i];
    src[i] = (int32_t)r;
    sxyz[3 * (int64_t)i] = xyz[3 * (int64_t)r];
    sxyz[3 * (int64_t)i + 1] = xyz[3 * (int64_t)r + 1];
    sxyz[3 * (int64_t)i + 2] = xyz[3 * (int64_t)r + 2];
    if (i > 0 && skeys[i - 1] == key) return;
    int lo = i + 1, hi = n;  // first position in (i, n] whose key differs
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (skeys[mid] == key) lo = mid + 1;
        else hi = mid;
    }
    cell_end[i] = lo;
    unsigned slot = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 40) & cap_mask;
    for (;;) {  // unique keys, table at most half full: terminates
        const unsigned long long prev = atomicCAS(&tkeys[slot], ~0ull, key);
        if (prev == ~0ull) {
            tvals[slot] = i;
            return;
        }
        slot = (slot + 1) & cap_mask;
    }
}

// key of a site for the parity order of the strided / inverse conv tables: (z & 1, y & 1, x & 1)
__global__ __launch_bounds__(kThreads) void parity_keys(const int32_t* __restrict__ coords, int m,
                                                        unsigned long long* __restrict__ keys, uint32_t* __restrict__ rows) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= m) return;
    const int32_t* c = coords + 4 * (int64_t)i;
    keys[i] = (unsigned long long)(((c[1] & 1) << 2) | ((c[2] & 1) << 1) | (c[3] & 1));
    rows[i] = (uint32_t)i;
}

__global__ __launch_bounds__(kThreads) void knn_copy_order(const uint32_t* __restrict__ srows, int m, int32_t* __restrict__ order) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < m) order[i] = (int32_t)srows[i];
}

struct GridLevel {
    const float* sxyz;        // points in this level's cell order
    const int32_t* src;       // original row of each sorted point
    const int32_t* cell_end;  // [n], valid at the first sorted position of a cell
    const unsigned long long* tkeys;
    const int32_t* tvals;
    unsigned cap_mask;
    float cell;
    int max_ring;
};
struct GridLevels {
    GridLevel lv[4];
    int n;
    int nested;  // every level's cell is exactly 8x the previous one: dense cells may be searched through their sub-cells
};

constexpr int kDenseCell = 128;  // points; above this a cell is searched through the next finer level

// Lidar density falls as 1/r^2: no single cell size serves both the ground rings next to the sensor and the walls at
// 70 m.  Up to four grids (fine -> coarse); a query climbs to the next level (list reset, shells restart) when its
// K-th distance is still open after max_ring shells, and scans its whole segment after the last level.  A coarse cell
// that holds many points is not scanned but searched through its 8^3 sub-cells of the next finer level, skipping
// sub-cells whose box is farther than the current K-th distance -- so a dense clump next to a sparse query costs a
// thin slice of it, not all of it.
template <int K>
struct KnnState {
    float bd[K];
    int32_t bi[K];
    float qx, qy, qz;
    int seg;

    __device__ __forceinline__ void reset(int seg_start) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            bd[j] = 1e10f;
            bi[j] = seg_start;
        }
    }
    __device__ __forceinline__ void offer(const GridLevel& L, int c) {  // candidate at sorted position c of level L
        const float dx = qx - L.sxyz[3 * (int64_t)c], dy = qy - L.sxyz[3 * (int64_t)c + 1], dz = qz - L.sxyz[3 * (int64_t)c + 2];
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 > bd[K - 1]) return;
        const int32_t id = L.src[c];
        if (d2 == bd[K - 1] && id > bi[K - 1]) return;
        float cd = d2;
        int32_t ci = id;
        bool shifting = false;  // ordered by (d2, original index): once inserted, everything behind moves down one slot
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (shifting || cd < bd[j] || (cd == bd[j] && ci < bi[j])) {
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
    // squared distance from the query to the box of (biased) cell (x, y, z) of size s
    __device__ __forceinline__ float box_d2(int x, int y, int z, float s) const {
        const float lx = (float)(x - kCellBias) * s, ly = (float)(y - kCellBias) * s, lz = (float)(z - kCellBias) * s;
        const float ex = fmaxf(fmaxf(lx - qx, qx - (lx + s)), 0.f);
        const float ey = fmaxf(fmaxf(ly - qy, qy - (ly + s)), 0.f);
        const float ez = fmaxf(fmaxf(lz - qz, qz - (lz + s)), 0.f);
        return (ex * ex + ey * ey) + ez * ez;
    }
    template <int L>
    __device__ void visit(const GridLevels& g, int x, int y, int z) {
        if ((unsigned)x > 65535u || (unsigned)y > 65535u || (unsigned)z > 65535u) return;
        const GridLevel& lv = g.lv[L];
        const unsigned long long key = cell_key(seg, x, y, z);
        unsigned slot = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 40) & lv.cap_mask;
        for (;;) {
            const unsigned long long tk = lv.tkeys[slot];
            if (tk == key) break;
            if (tk == ~0ull) return;  // empty cell
            slot = (slot + 1) & lv.cap_mask;
        }
        const int c0 = lv.tvals[slot], c1 = lv.cell_end[c0];
        if constexpr (L > 0) {
            if (g.nested && c1 - c0 > kDenseCell) {
                const float s = g.lv[L - 1].cell;
                // the conservative factor keeps a sub-cell whose box distance rounds just above the K-th distance
                const int fx = (x - kCellBias) * 8 + kCellBias, fy = (y - kCellBias) * 8 + kCellBias, fz = (z - kCellBias) * 8 + kCellBias;
                for (int dz = 0; dz < 8; ++dz)
                    for (int dy = 0; dy < 8; ++dy)
                        for (int dx = 0; dx < 8; ++dx)
                            if (box_d2(fx + dx, fy + dy, fz + dz, s) * 0.99999f <= bd[K - 1]) visit<L - 1>(g, fx + dx, fy + dy, fz + dz);
                return;
            }
        }
        for (int c = c0; c < c1; ++c) offer(lv, c);
    }
    template <int L>
    __device__ bool search_level(const GridLevels& g, int seg_start) {  // true when the K-th distance is settled
        const GridLevel& lv = g.lv[L];
        const float inv_cell = 1.0f / lv.cell;
        const int cx = cell_of(qx, inv_cell), cy = cell_of(qy, inv_cell), cz = cell_of(qz, inv_cell);
        reset(seg_start);
        for (int r = 0; r <= lv.max_ring; ++r) {
            for (int dz = -r; dz <= r; ++dz)
                for (int dy = -r; dy <= r; ++dy) {
                    if (dz == -r || dz == r || dy == -r || dy == r) {
                        for (int dx = -r; dx <= r; ++dx) visit<L>(g, cx + dx, cy + dy, cz + dz);
                    } else {
                        visit<L>(g, cx - r, cy + dy, cz + dz);
                        if (r > 0) visit<L>(g, cx + r, cy + dy, cz + dz);
                    }
                }
            // every unvisited point lies outside the cube of r cells around the query's cell: farther than r * cell
            const float bound = (float)r * lv.cell;
            if (bd[K - 1] < bound * bound * 0.999999f) return true;
        }
        return false;
    }
};

template <int K>
__global__ __launch_bounds__(kThreads) void knn_grid_kernel(GridLevels g, const float* __restrict__ qxyz,
                                                            const int32_t* __restrict__ qorder,
                                                            const int32_t* __restrict__ offset,
                                                            const int32_t* __restrict__ new_offset, int b, int m, int k,
                                                            int32_t* __restrict__ idx_out, float* __restrict__ d2_out) {
    const int t = blockIdx.x * kThreads + threadIdx.x;
    if (t >= m) return;
    // queries are walked in (finest) cell order when the caller provides it: the lanes of a wave then visit the same
    // cells in the same order (uniform trip counts, identical candidate addresses)
    const int q = qorder ? qorder[t] : t;
    KnnState<K> st;
    st.seg = segment_of(q, new_offset, b);
    const int seg_start = st.seg == 0 ? 0 : offset[st.seg - 1], seg_end = offset[st.seg];
    st.qx = qxyz[3 * (int64_t)q];
    st.qy = qxyz[3 * (int64_t)q + 1];
    st.qz = qxyz[3 * (int64_t)q + 2];
    bool done = st.template search_level<0>(g, seg_start);
    if (!done && g.n > 1) done = st.template search_level<1>(g, seg_start);
    if (!done && g.n > 2) done = st.template search_level<2>(g, seg_start);
    if (!done && g.n > 3) done = st.template search_level<3>(g, seg_start);
    if (!done) {  // isolated query: exact scan of the whole segment (same order relation)
        st.reset(seg_start);
        for (int c = seg_start; c < seg_end; ++c) st.offer(g.lv[0], c);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (j < k) {
            idx_out[(int64_t)q * k + j] = st.bi[j];
            d2_out[(int64_t)q * k + j] = st.bd[j];
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

namespace {

using SortKey = unsigned long long;
constexpr unsigned kKeyBits = 56;  // batch (8) | cx (16) | cy (16) | cz (16)

bool sort_scratch_bytes(int64_t n, size_t* bytes) {
    *bytes = 0;
    if (n == 0) return true;
    rocprim::double_buffer<SortKey> k(nullptr, nullptr);
    rocprim::double_buffer<uint32_t> v(nullptr, nullptr);
    return rocprim::radix_sort_pairs(nullptr, *bytes, k, v, (size_t)n, 0u, kKeyBits) == hipSuccess;
}

struct SortBuffers {
    SortKey *k0, *k1;
    uint32_t *v0, *v1;
    void* tmp;
    size_t tmp_bytes;
};

bool carve_sort(void* workspace, size_t workspace_bytes, int64_t n, SortBuffers* sb) {
    if (!sort_scratch_bytes(n, &sb->tmp_bytes)) return false;
    if (!workspace || workspace_bytes < seg3d_knn_level_workspace_bytes(n)) return false;
    WsCarver ws(workspace);
    sb->k0 = ws.take<SortKey>((size_t)n);
    sb->k1 = ws.take<SortKey>((size_t)n);
    sb->v0 = ws.take<uint32_t>((size_t)n);
    sb->v1 = ws.take<uint32_t>((size_t)n);
    sb->tmp = ws.take<char>(sb->tmp_bytes);
    return true;
}

// keys + rows of xyz on a grid of `cell`, sorted by key; *skeys / *srows point at the sorted halves
int sorted_cells(const float* xyz, int64_t n, const int32_t* offset, int32_t batch_size, float cell, SortBuffers& sb,
                 hipStream_t st, const SortKey** skeys, const uint32_t** srows) {
    hipLaunchKernelGGL(knn_cell_keys, dim3((unsigned)ceil_div64(n, kThreads)), dim3(kThreads), 0, st, xyz, (int)n, offset,
                       batch_size, 1.0f / cell, sb.k0, sb.v0);
    SEG3D_CHECK_LAUNCH();
    rocprim::double_buffer<SortKey> kb(sb.k0, sb.k1);
    rocprim::double_buffer<uint32_t> vb(sb.v0, sb.v1);
    size_t bytes = sb.tmp_bytes;
    SEG3D_CHECK_HIP(rocprim::radix_sort_pairs(sb.tmp, bytes, kb, vb, (size_t)n, 0u, kKeyBits, st));
    *skeys = kb.current();
    *srows = vb.current();
    return SEG3D_OK;
}

}  // namespace

/* Grid-accelerated exact kNN (see the kernels above): level build and query order are launch sequences without any
 * host read-back; the sort is rocPRIM's device radix sort of (cell key, row). */
extern "C" size_t seg3d_knn_level_workspace_bytes(int64_t n) {
    if (n < 0 || n >= (int64_t)0x7FFFFFF0) return 0;
    size_t tmp = 0;
    if (!sort_scratch_bytes(n, &tmp)) return 0;
    return 2 * align_up((size_t)n * sizeof(SortKey), 256) + 2 * align_up((size_t)n * sizeof(uint32_t), 256) + align_up(tmp, 256) + 256;
}

extern "C" int seg3d_knn_level_build(const float* xyz, int64_t n, const int32_t* offset, int32_t batch_size, float cell,
                                     float* sorted_xyz, int32_t* src_index, int32_t* cell_end, void* table_keys,
                                     int32_t* table_vals, int64_t capacity, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (n < 0 || batch_size <= 0 || batch_size > 255 || !(cell > 0.f) || n >= (int64_t)0x7FFFFFF0) return SEG3D_EINVAL;
    if (capacity < 2 * n || capacity <= 0 || (capacity & (capacity - 1)) || capacity > (1ll << 31)) return SEG3D_EINVAL;
    if (!table_keys || !table_vals) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    SEG3D_CHECK_HIP(hipMemsetAsync(table_keys, 0xFF, (size_t)capacity * 8, st));
    if (n == 0) return SEG3D_OK;
    if (!xyz || !offset || !sorted_xyz || !src_index || !cell_end) return SEG3D_EINVAL;
    SortBuffers sb;
    if (!carve_sort(workspace, workspace_bytes, n, &sb)) return SEG3D_EWORKSPACE;
    const SortKey* skeys;
    const uint32_t* srows;
    const int rc = sorted_cells(xyz, n, offset, batch_size, cell, sb, st, &skeys, &srows);
    if (rc != SEG3D_OK) return rc;
    hipLaunchKernelGGL(knn_level_finish, dim3((unsigned)ceil_div64(n, kThreads)), dim3(kThreads), 0, st, skeys, srows, xyz, (int)n,
                       sorted_xyz, src_index, cell_end, static_cast<unsigned long long*>(table_keys), table_vals,
                       (unsigned)(capacity - 1));
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

extern "C" int seg3d_knn_query_order(const float* new_xyz, int64_t m, const int32_t* new_offset, int32_t batch_size, float cell,
                                     int32_t* order, void* workspace, size_t workspace_bytes, void* stream) {
    if (m < 0 || batch_size <= 0 || batch_size > 255 || !(cell > 0.f) || m >= (int64_t)0x7FFFFFF0) return SEG3D_EINVAL;
    if (m == 0) return SEG3D_OK;
    if (!new_xyz || !new_offset || !order) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    SortBuffers sb;
    if (!carve_sort(workspace, workspace_bytes, m, &sb)) return SEG3D_EWORKSPACE;
    const SortKey* skeys;
    const uint32_t* srows;
    const int rc = sorted_cells(new_xyz, m, new_offset, batch_size, cell, sb, st, &skeys, &srows);
    if (rc != SEG3D_OK) return rc;
    hipLaunchKernelGGL(knn_copy_order, dim3((unsigned)ceil_div64(m, kThreads)), dim3(kThreads), 0, st, srows, (int)m, order);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

// Rows of a site list grouped by the parity of (z, y, x), original order kept inside a group (a stable 3-bit radix
// sort): the processing order of the strided / inverse conv tables (spconv.SiteLevel.parity_order).  Workspace as
// seg3d_knn_level_workspace_bytes(m).
extern "C" int seg3d_parity_order(const int32_t* coords, int64_t m, int32_t* order, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    if (m < 0 || m >= (int64_t)0x7FFFFFF0) return SEG3D_EINVAL;
    if (m == 0) return SEG3D_OK;
    if (!coords || !order) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    SortBuffers sb;
    if (!carve_sort(workspace, workspace_bytes, m, &sb)) return SEG3D_EWORKSPACE;
    hipLaunchKernelGGL(parity_keys, dim3((unsigned)ceil_div64(m, kThreads)), dim3(kThreads), 0, st, coords, (int)m, sb.k0, sb.v0);
    SEG3D_CHECK_LAUNCH();
    rocprim::double_buffer<SortKey> kb(sb.k0, sb.k1);
    rocprim::double_buffer<uint32_t> vb(sb.v0, sb.v1);
    size_t bytes = sb.tmp_bytes;
    SEG3D_CHECK_HIP(rocprim::radix_sort_pairs(sb.tmp, bytes, kb, vb, (size_t)m, 0u, 3u, st));
    hipLaunchKernelGGL(knn_copy_order, dim3((unsigned)ceil_div64(m, kThreads)), dim3(kThreads), 0, st, vb.current(), (int)m, order);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

extern "C" int seg3d_knn_grid_query(const seg3d_knn_level* levels, int32_t n_levels, const float* new_xyz,
                                    const int32_t* query_order, int64_t m, const int32_t* offset,
                                    const int32_t* new_offset, int32_t batch_size, int32_t k, int32_t* idx, float* dist2,
                                    void* stream) {
    if (m < 0 || batch_size <= 0 || batch_size > 255 || k <= 0 || k > 64 || n_levels < 1 || n_levels > 4 || !levels ||
        m >= (int64_t)0x7FFFFFF0)
        return SEG3D_EINVAL;
    if (m == 0) return SEG3D_OK;
    if (!new_xyz || !offset || !new_offset || !idx || !dist2) return SEG3D_EINVAL;
    GridLevels g;
    g.n = n_levels;
    g.nested = 1;
    for (int l = 1; l < n_levels; ++l)
        if (fabsf(levels[l].cell - 8.0f * levels[l - 1].cell) > 1e-6f * levels[l].cell) g.nested = 0;
    for (int l = 0; l < n_levels; ++l) {
        const seg3d_knn_level& a = levels[l];
        if (!a.sorted_xyz || !a.src_index || !a.cell_end || !a.table_keys || !a.table_vals || a.capacity <= 0 ||
            (a.capacity & (a.capacity - 1)) || !(a.cell > 0.f) || a.max_ring < 0 || a.max_ring > 16)
            return SEG3D_EINVAL;
        g.lv[l] = GridLevel{a.sorted_xyz, a.src_index, a.cell_end, static_cast<const unsigned long long*>(a.table_keys),
                            a.table_vals, (unsigned)(a.capacity - 1), a.cell, a.max_ring};
    }
    hipStream_t st = as_stream(stream);
    const unsigned nb = (unsigned)ceil_div64(m, kThreads);
#define SEG3D_KNNG(KK)                                                                                                  \
    hipLaunchKernelGGL(knn_grid_kernel<KK>, dim3(nb), dim3(kThreads), 0, st, g, new_xyz, query_order, offset, new_offset,  \
                       batch_size, (int)m, (int)k, idx, dist2)
    if (k == 1) SEG3D_KNNG(1);
    else if (k <= 4) SEG3D_KNNG(4);
    else if (k <= 16) SEG3D_KNNG(16);
    else if (k <= 32) SEG3D_KNNG(32);
    else SEG3D_KNNG(64);
#undef SEG3D_KNNG
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}
