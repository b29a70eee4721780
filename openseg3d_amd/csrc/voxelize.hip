// a1/a2/a4: hard voxelisation with first-seen voxel order, on device.
// Reference: seg3d/core/voxel/voxel_generator.py:55-153 (serial numba loop over a dense 531 MB lookup grid).
//
// MI355X design: the dense grid is replaced by an open-addressing hash keyed on the linear cell
// index.  Pass 1 inserts every in-range point and keeps, per cell, the smallest point index
// (atomicMin) -- the point that would have created the voxel in the serial loop.  Lidar points
// arrive in scan order, so neighbouring lanes very often hit the same cell: a wave ballot finds
// the head lane of every run of equal keys and only heads touch the table.  Pass 2 flags the
// creating points, an exclusive scan over the flags yields the first-seen rank, pass 3 writes
// ids and coordinates.  HBM-bound integer work: N*(12 B read + 4 B write) + M*16 B algorithmic.
#include <cstdio>

#include "common.hpp"

namespace {

constexpr int kThreads = 256;

struct VoxParams {
    int32_t grid[3];  // x, y, z
    int32_t row_stride, xyz_col, batch_col;
};

template <typename T>
struct VoxConst {
    T lo[3];
    T vs[3];
};

__device__ __forceinline__ float floor_t(float v) { return floorf(v); }
__device__ __forceinline__ double floor_t(double v) { return floor(v); }

template <typename T>
__global__ __launch_bounds__(kThreads) void vox_insert(const T* __restrict__ pts, int64_t n, VoxParams p,
                                                       VoxConst<T> c, HashView h, uint32_t* __restrict__ slot_of) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool valid = i < n;
    uint64_t key = 0;
    if (valid) {
        const T* row = pts + i * p.row_stride;
        int32_t cc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            // voxel_generator.py:139 -- subtract and TRUE divide in the point dtype, then floor
            T v = floor_t((row[p.xyz_col + j] - c.lo[j]) / c.vs[j]);
            if (v < (T)0 || v >= (T)p.grid[j]) valid = false;
            cc[j] = (int32_t)v;
        }
        if (valid) {
            int64_t b = p.batch_col >= 0 ? (int64_t)row[p.batch_col] : 0;
            key = (uint64_t)(((b * p.grid[2] + cc[2]) * p.grid[1] + cc[1]) * (int64_t)p.grid[0] + cc[0]);
        }
    }
    // head of a run of equal keys inside the wave (ascending point index => head has the run minimum)
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    const uint32_t plo = __shfl_up(klo, 1, SEG3D_WAVE), phi = __shfl_up(khi, 1, SEG3D_WAVE);
    const int pvalid = __shfl_up((int)valid, 1, SEG3D_WAVE);
    const bool head = valid && (lane == 0 || !pvalid || plo != klo || phi != khi);
    const unsigned long long heads = __ballot(head);
    uint32_t slot = 0xFFFFFFFFu;
    if (head) {
        slot = (uint32_t)hash_insert_slot(h, key);
        atomicMin(&h.vals[slot], (int32_t)i);
    }
    const unsigned long long below = heads & ((2ull << lane) - 1ull);
    int src = below ? 63 - __clzll((long long)below) : 0;
    const uint32_t run_slot = __shfl(slot, src, SEG3D_WAVE);
    if (i < n) slot_of[i] = valid ? run_slot : 0xFFFFFFFFu;
}

__global__ __launch_bounds__(kThreads) void vox_flag(const uint32_t* __restrict__ slot_of, int64_t n, HashView h,
                                                     uint32_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = slot_of[i];
    flag[i] = (s != 0xFFFFFFFFu && h.vals[s] == (int32_t)i) ? 1u : 0u;
}

__global__ __launch_bounds__(kThreads) void vox_emit(const uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ rank,
                                                     int64_t n, HashView h, VoxParams p, int32_t* __restrict__ coords,
                                                     int32_t* __restrict__ ids) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = slot_of[i];
    if (s == 0xFFFFFFFFu) {
        ids[i] = -1;
        return;
    }
    const int32_t first = h.vals[s];
    const int32_t id = (int32_t)rank[first];
    ids[i] = id;
    if (first == (int32_t)i) {
        uint64_t key = h.keys[s];
        const int32_t x = (int32_t)(key % (uint64_t)p.grid[0]); key /= (uint64_t)p.grid[0];
        const int32_t y = (int32_t)(key % (uint64_t)p.grid[1]); key /= (uint64_t)p.grid[1];
        const int32_t z = (int32_t)(key % (uint64_t)p.grid[2]); key /= (uint64_t)p.grid[2];
        int4 row = make_int4((int32_t)key, z, y, x);
        reinterpret_cast<int4*>(coords)[id] = row;
    }
}

struct VoxWs {
    void* table;
    uint64_t cap;
    uint32_t *slot_of, *flag, *rank, *tmp;
};

VoxWs carve(void* ws, int64_t n) {
    WsCarver c(ws);
    VoxWs w;
    w.cap = hash_capacity(n);
    w.table = c.take<char>(w.cap * 12);
    w.slot_of = c.take<uint32_t>((size_t)n + 1);
    w.flag = c.take<uint32_t>((size_t)n + 1);
    w.rank = c.take<uint32_t>((size_t)n + 1);
    w.tmp = c.take<uint32_t>(scan_tmp_count(n));
    return w;
}

size_t ws_bytes(int64_t n) {
    WsCarver c(nullptr);
    const uint64_t cap = hash_capacity(n);
    c.take<char>(cap * 12);
    c.take<uint32_t>((size_t)n + 1);
    c.take<uint32_t>((size_t)n + 1);
    c.take<uint32_t>((size_t)n + 1);
    c.take<uint32_t>(scan_tmp_count(n));
    return c.off;
}

template <typename T>
int voxelize_impl(const T* points, int64_t n, int32_t row_stride, int32_t xyz_col, int32_t batch_col,
                  const float* voxel_size, const float* range, int32_t* voxel_coords, int32_t* point_voxel_ids,
                  int32_t* n_voxels, void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || n >= (int64_t)0x7F000000 || !voxel_size || !range || !voxel_coords || !n_voxels || !workspace ||
        row_stride < 3 || xyz_col < 0 || xyz_col + 3 > row_stride || batch_col >= row_stride)
        return SEG3D_EINVAL;
    if (n > 0 && (!points || !point_voxel_ids)) return SEG3D_EINVAL;
    if (workspace_bytes < ws_bytes(n)) return SEG3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    VoxParams p;
    if (seg3d_grid_size(voxel_size, range, p.grid) != SEG3D_OK) return SEG3D_EINVAL;
    p.row_stride = row_stride;
    p.xyz_col = xyz_col;
    p.batch_col = batch_col;
    VoxConst<T> c;
    for (int j = 0; j < 3; ++j) {
        c.lo[j] = (T)range[j];  // float32 constants widened exactly for the f64 entry
        c.vs[j] = (T)voxel_size[j];
    }
    VoxWs w = carve(workspace, n);
    HashView h = hash_view(w.table, w.cap);
    SEG3D_CHECK_HIP(hipMemsetAsync(h.keys, 0xFF, w.cap * 8, st));
    SEG3D_CHECK_HIP(hipMemsetAsync(h.vals, 0x7F, w.cap * 4, st));
    if (n == 0) {
        SEG3D_CHECK_HIP(hipMemsetAsync(n_voxels, 0, 4, st));
        return SEG3D_OK;
    }
    const unsigned nb = (unsigned)ceil_div64(n, kThreads);
    hipLaunchKernelGGL(vox_insert<T>, dim3(nb), dim3(kThreads), 0, st, points, n, p, c, h, w.slot_of);
    SEG3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(vox_flag, dim3(nb), dim3(kThreads), 0, st, w.slot_of, n, h, w.flag);
    SEG3D_CHECK_LAUNCH();
    int rc = scan_exclusive_u32(w.flag, w.rank, n, reinterpret_cast<uint32_t*>(n_voxels), w.tmp, st);
    if (rc != SEG3D_OK) return rc;
    hipLaunchKernelGGL(vox_emit, dim3(nb), dim3(kThreads), 0, st, w.slot_of, w.rank, n, h, p, voxel_coords,
                       point_voxel_ids);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

// a3: cart2polar (seg3d/utils/pointops_utils.py:8-11) + the row re-assembly of waymo_dataset.py:270-273.
// One thread per point.  rho = sqrt(x*x + y*y): two roundings of the products, one of the sum, IEEE sqrt -- the same
// operations numpy performs in the point dtype (-ffp-contract=off: no fused multiply-add).  phi: numpy calls the host
// libm's atan2f / atan2, which is NOT correctly rounded (glibc documents 1 ulp), so "bit-exact with the reference" is
// machine-dependent for this one column; here float32 phi is atan2 evaluated in double and rounded once (correctly
// rounded float32 but for double-rounding ties), float64 phi is the device library's atan2.
__device__ __forceinline__ float atan2_t(float y, float x) { return (float)atan2((double)y, (double)x); }
__device__ __forceinline__ double atan2_t(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ float sqrt_t(float v) { return sqrtf(v); }
__device__ __forceinline__ double sqrt_t(double v) { return sqrt(v); }

template <typename T>
__global__ __launch_bounds__(kThreads) void cart2polar_rows(const T* __restrict__ pts, int64_t n, int row_stride, int xyz_col,
                                                            T* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const T* row = pts + i * row_stride;
    T* o = out + i * (row_stride + 2);
    for (int j = 0; j < xyz_col; ++j) o[j] = row[j];  // batch-index column(s) in front of the coordinates
    const T x = row[xyz_col], y = row[xyz_col + 1], z = row[xyz_col + 2];
    o[xyz_col] = sqrt_t(x * x + y * y);
    o[xyz_col + 1] = atan2_t(y, x);
    o[xyz_col + 2] = z;
    o[xyz_col + 3] = x;
    o[xyz_col + 4] = y;
    for (int j = xyz_col + 3; j < row_stride; ++j) o[j + 2] = row[j];
}

template <typename T>
int cart2polar_impl(const T* points, int64_t n, int32_t row_stride, int32_t xyz_col, T* out, void* stream) {
    if (n < 0 || row_stride < 3 || xyz_col < 0 || xyz_col + 3 > row_stride) return SEG3D_EINVAL;
    if (n == 0) return SEG3D_OK;
    if (!points || !out) return SEG3D_EINVAL;
    hipLaunchKernelGGL(cart2polar_rows<T>, dim3((unsigned)ceil_div64(n, kThreads)), dim3(kThreads), 0, as_stream(stream), points, n,
                       row_stride, xyz_col, out);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}


// (b) the CPU entry of VoxelGenerator.generate (voxel_generator.py:24-26 -> points_to_voxel :55-95 ->
// _points_to_voxel_reverse_kernel :98-153), for DataLoader workers and TTA, which run without a GPU context.  The same
// serial loop as the reference -- first-seen order falls out of it -- with the dense coor_to_voxelidx grid (531 MB of page
// faults per call at 1440 x 1440 x 64) replaced by an open-addressing table over the linear cell index in the caller's
// workspace (12 bytes per slot, >= 2 slots per point).  Plain host C++: no HIP call, safe in a forked worker.
inline float floor_h(float v) { return __builtin_floorf(v); }
inline double floor_h(double v) { return __builtin_floor(v); }

inline uint64_t host_capacity(int64_t n) {
    uint64_t cap = 64;
    while (cap < (uint64_t)(2 * n + 2)) cap <<= 1;
    return cap;
}

template <typename T>
int voxelize_host_impl(const T* points, int64_t n, int32_t row_stride, int32_t xyz_col, int32_t batch_col,
                       const float* voxel_size, const float* range, int32_t* voxel_coords, int32_t* point_voxel_ids,
                       int32_t* n_voxels, void* workspace, size_t workspace_bytes) {
    if (n < 0 || n >= (int64_t)0x7F000000 || !voxel_size || !range || !n_voxels || row_stride < 3 || xyz_col < 0 ||
        xyz_col + 3 > row_stride || batch_col >= row_stride)
        return SEG3D_EINVAL;
    if (n > 0 && (!points || !point_voxel_ids || !voxel_coords || !workspace)) return SEG3D_EINVAL;
    const uint64_t cap = host_capacity(n);
    if (n > 0 && workspace_bytes < cap * 12) return SEG3D_EWORKSPACE;
    int32_t grid[3];
    if (seg3d_grid_size(voxel_size, range, grid) != SEG3D_OK) return SEG3D_EINVAL;
    *n_voxels = 0;
    if (n == 0) return SEG3D_OK;
    uint64_t* keys = static_cast<uint64_t*>(workspace);
    int32_t* vals = reinterpret_cast<int32_t*>(keys + cap);
    for (uint64_t s = 0; s < cap; ++s) keys[s] = ~0ull;
    // volatile: the subtract and the divide are two IEEE operations in the point dtype (no contraction, no reassociation
    // whatever flags a host compiler is given), float32 constants widened exactly for the f64 entry
    T lo[3], vs[3];
    for (int j = 0; j < 3; ++j) {
        lo[j] = (T)range[j];
        vs[j] = (T)voxel_size[j];
    }
    int32_t count = 0;
    for (int64_t i = 0; i < n; ++i) {
        const T* row = points + i * row_stride;
        int32_t cc[3];
        bool ok = true;
        for (int j = 0; j < 3 && ok; ++j) {
            volatile T d = row[xyz_col + j] - lo[j];
            volatile T q = d / vs[j];
            const T v = floor_h((T)q);  // voxel_generator.py:139
            if (!(v >= (T)0) || v >= (T)grid[j]) ok = false;  // (a NaN coordinate fails both of the reference's tests and
            else cc[j] = (int32_t)v;                           //  would index the grid with garbage there: rejected here)
        }
        if (!ok) {
            point_voxel_ids[i] = -1;
            continue;
        }
        const int64_t b = batch_col >= 0 ? (int64_t)row[batch_col] : 0;
        const uint64_t key = (uint64_t)(((b * grid[2] + cc[2]) * grid[1] + cc[1]) * (int64_t)grid[0] + cc[0]);
        uint64_t h = key * 0x9E3779B97F4A7C15ull;
        uint64_t s = (h ^ (h >> 29)) & (cap - 1);
        while (keys[s] != key && keys[s] != ~0ull) s = (s + 1) & (cap - 1);
        if (keys[s] == ~0ull) {
            keys[s] = key;
            vals[s] = count;
            int32_t* o = voxel_coords + 4 * (int64_t)count;
            o[0] = (int32_t)b;
            o[1] = cc[2];
            o[2] = cc[1];
            o[3] = cc[0];
            ++count;
        }
        point_voxel_ids[i] = vals[s];
    }
    *n_voxels = count;
    return SEG3D_OK;
}

}  // namespace

extern "C" {

int seg3d_cart2polar_f32(const float* points, int64_t n_points, int32_t row_stride, int32_t xyz_col, float* out, void* stream) {
    return cart2polar_impl<float>(points, n_points, row_stride, xyz_col, out, stream);
}

int seg3d_cart2polar_f64(const double* points, int64_t n_points, int32_t row_stride, int32_t xyz_col, double* out,
                         void* stream) {
    return cart2polar_impl<double>(points, n_points, row_stride, xyz_col, out, stream);
}

int seg3d_abi_version(void) { return 40; }

namespace {
thread_local char g_last_error[320] = "";
}

const char* seg3d_last_error(void) { return g_last_error; }

int seg3d_grid_size(const float* voxel_size, const float* range, int32_t* grid_xyz) {
    if (!voxel_size || !range || !grid_xyz) return SEG3D_EINVAL;
    for (int j = 0; j < 3; ++j) {
        // voxel_generator.py:15-18: float32 arrays, np.round (half to even)
        volatile float span = range[3 + j] - range[j];
        volatile float g = span / voxel_size[j];
        grid_xyz[j] = (int32_t)__builtin_rintf(g);
        if (grid_xyz[j] <= 0) return SEG3D_EINVAL;
    }
    return SEG3D_OK;
}

size_t seg3d_voxelize_workspace_bytes(int64_t n_points) { return ws_bytes(n_points < 0 ? 0 : n_points); }

int seg3d_voxelize_f32(const float* points, int64_t n_points, int32_t row_stride, int32_t xyz_col, int32_t batch_col,
                       const float* voxel_size, const float* range, int32_t* voxel_coords, int32_t* point_voxel_ids,
                       int32_t* n_voxels, void* workspace, size_t workspace_bytes, void* stream) {
    return voxelize_impl<float>(points, n_points, row_stride, xyz_col, batch_col, voxel_size, range, voxel_coords,
                                point_voxel_ids, n_voxels, workspace, workspace_bytes, stream);
}

int seg3d_voxelize_f64(const double* points, int64_t n_points, int32_t row_stride, int32_t xyz_col, int32_t batch_col,
                       const float* voxel_size, const float* range, int32_t* voxel_coords, int32_t* point_voxel_ids,
                       int32_t* n_voxels, void* workspace, size_t workspace_bytes, void* stream) {
    return voxelize_impl<double>(points, n_points, row_stride, xyz_col, batch_col, voxel_size, range, voxel_coords,
                                 point_voxel_ids, n_voxels, workspace, workspace_bytes, stream);
}

size_t seg3d_voxelize_host_workspace_bytes(int64_t n_points) { return (size_t)host_capacity(n_points < 0 ? 0 : n_points) * 12; }

int seg3d_voxelize_host_f32(const float* points, int64_t n_points, int32_t row_stride, int32_t xyz_col, int32_t batch_col,
                            const float* voxel_size, const float* range, int32_t* voxel_coords, int32_t* point_voxel_ids,
                            int32_t* n_voxels, void* workspace, size_t workspace_bytes) {
    return voxelize_host_impl<float>(points, n_points, row_stride, xyz_col, batch_col, voxel_size, range, voxel_coords,
                                     point_voxel_ids, n_voxels, workspace, workspace_bytes);
}

int seg3d_voxelize_host_f64(const double* points, int64_t n_points, int32_t row_stride, int32_t xyz_col, int32_t batch_col,
                            const float* voxel_size, const float* range, int32_t* voxel_coords, int32_t* point_voxel_ids,
                            int32_t* n_voxels, void* workspace, size_t workspace_bytes) {
    return voxelize_host_impl<double>(points, n_points, row_stride, xyz_col, batch_col, voxel_size, range, voxel_coords,
                                      point_voxel_ids, n_voxels, workspace, workspace_bytes);
}

}  // extern "C"

// SEG3D_CHECK_LAUNCH / SEG3D_CHECK_HIP land here: keep the runtime's own words for seg3d_last_error()
void seg3d_note_hip_error(int code, const char* file, int line) {
    const char* base = file;
    for (const char* p = file; *p; ++p)
        if (*p == '/') base = p + 1;
    snprintf(g_last_error, sizeof(g_last_error), "%s (%s) at %s:%d", hipGetErrorString((hipError_t)code),
             hipGetErrorName((hipError_t)code), base, line);
}
