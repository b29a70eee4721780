// a8-a11: sparse-conv indice generation ("rulebook") for 3x3x3 kernels.
// Replaces spconv's indice pair generation at the call sites seg3d/utils/spconv_utils.py:13-32 and
// seg3d/models/backbones/pointtransformer.py:26-34,73-81,159-166,184-189.
//
// Output-stationary tables: nbr[k][r] = input row feeding output row r through kernel offset k
// (offset-major so a tile of consecutive output rows reads a contiguous run per offset).
//   * coordinate hash: 64-bit linear site key -> row, open addressing (common.hpp)
//   * submanifold table: 27 probes per site
//   * strided (k3,s2,p1) output sites: bitmap over the coarse grid (1 bit per cell, <= 2 MB per
//     sample at 1440x1440x64) + popcount scan -> unique sites in ascending (b,z,y,x) order with
//     no sort and no second hash
//   * strided / inverse tables: 27 probes per coarse site into the fine hash; the inverse table
//     is the same pair list scattered by input row (each (input, k) has at most one output).
// All integer work; parity with the oracle is bit-exact.
#include "common.hpp"

namespace {

constexpr int kThreads = 256;

struct Shape3 {
    int32_t z, y, x;
};

__device__ __forceinline__ uint64_t site_key(int64_t b, int64_t z, int64_t y, int64_t x, Shape3 s) {
    return (uint64_t)(((b * s.z + z) * s.y + y) * s.x + x);
}

__global__ __launch_bounds__(kThreads) void hash_fill(const int32_t* __restrict__ coords, int64_t m, Shape3 s, HashView h) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= m) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const uint64_t slot = hash_insert_slot(h, site_key(c.x, c.y, c.z, c.w, s));
    h.vals[slot] = (int32_t)i;
}

// one thread per (offset k, site i); consecutive threads = consecutive sites of one offset
__global__ __launch_bounds__(kThreads) void subm_table(const int32_t* __restrict__ coords, int64_t m, Shape3 s, HashView h,
                                                       int32_t* __restrict__ nbr) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int k = blockIdx.y;
    if (i >= m) return;
    if (k == 13) {
        nbr[(int64_t)k * m + i] = (int32_t)i;
        return;
    }
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const int z = c.y + k / 9 - 1, y = c.z + (k / 3) % 3 - 1, x = c.w + k % 3 - 1;
    int32_t j = -1;
    if (z >= 0 && z < s.z && y >= 0 && y < s.y && x >= 0 && x < s.x) j = hash_lookup(h, site_key(c.x, z, y, x, s));
    nbr[(int64_t)k * m + i] = j;
}

// ---- strided output sites via bitmap
__global__ __launch_bounds__(kThreads) void down_mark(const int32_t* __restrict__ coords, int64_t m,
                                                      const int32_t* __restrict__ m_dev, Shape3 so,
                                                      uint32_t* __restrict__ bitmap) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= m || (m_dev && i >= (int64_t)*m_dev)) return;  // m_dev: exact row count still on the device (chained levels)
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    // per dimension: o = (c + 1 - k) / 2 for the k in {0,1,2} that make it integral and in range
    int oz[2], oy[2], ox[2], nz = 0, ny = 0, nx = 0;
    for (int k = 0; k < 3; ++k) {
        int t = c.y + 1 - k;
        if (t >= 0 && !(t & 1) && (t >> 1) < so.z) oz[nz++] = t >> 1;
        t = c.z + 1 - k;
        if (t >= 0 && !(t & 1) && (t >> 1) < so.y) oy[ny++] = t >> 1;
        t = c.w + 1 - k;
        if (t >= 0 && !(t & 1) && (t >> 1) < so.x) ox[nx++] = t >> 1;
    }
    for (int a = 0; a < nz; ++a)
        for (int b = 0; b < ny; ++b)
            for (int d = 0; d < nx; ++d) {
                const uint64_t key = site_key(c.x, oz[a], oy[b], ox[d], so);
                atomicOr(&bitmap[key >> 5], 1u << (key & 31));
            }
}

__global__ __launch_bounds__(kThreads) void down_popc(const uint32_t* __restrict__ bitmap, int64_t n_words,
                                                      uint32_t* __restrict__ cnt) {
    const int64_t w = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (w < n_words) cnt[w] = __popc(bitmap[w]);
}

__global__ __launch_bounds__(kThreads) void down_emit(const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ prefix,
                                                      int64_t n_words, Shape3 so, int64_t cap, int32_t* __restrict__ coords_out) {
    const int64_t w = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (w >= n_words) return;
    uint32_t bits = bitmap[w];
    int64_t row = prefix[w];
    while (bits) {
        const int b = __ffs(bits) - 1;
        bits &= bits - 1;
        uint64_t key = ((uint64_t)w << 5) + b;
        const int32_t x = (int32_t)(key % (uint64_t)so.x); key /= (uint64_t)so.x;
        const int32_t y = (int32_t)(key % (uint64_t)so.y); key /= (uint64_t)so.y;
        const int32_t z = (int32_t)(key % (uint64_t)so.z); key /= (uint64_t)so.z;
        if (row < cap) reinterpret_cast<int4*>(coords_out)[row] = make_int4((int32_t)key, z, y, x);
        ++row;
    }
}

__global__ __launch_bounds__(kThreads) void strided_table(const int32_t* __restrict__ coords_out, int64_t m_out, int64_t m_in,
                                                          Shape3 si, HashView h, int32_t* __restrict__ nbr_fwd,
                                                          int32_t* __restrict__ nbr_inv) {
    const int64_t o = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int k = blockIdx.y;
    if (o >= m_out) return;
    const int4 c = reinterpret_cast<const int4*>(coords_out)[o];
    const int z = 2 * c.y + k / 9 - 1, y = 2 * c.z + (k / 3) % 3 - 1, x = 2 * c.w + k % 3 - 1;
    int32_t i = -1;
    if (z >= 0 && z < si.z && y >= 0 && y < si.y && x >= 0 && x < si.x) i = hash_lookup(h, site_key(c.x, z, y, x, si));
    nbr_fwd[(int64_t)k * m_out + o] = i;
    if (i >= 0 && nbr_inv) nbr_inv[(int64_t)k * m_in + i] = (int32_t)o;
}

inline Shape3 shape_of(const int32_t* s) { return Shape3{s[0], s[1], s[2]}; }
inline Shape3 down_shape(const int32_t* s) {
    return Shape3{(s[0] + 2 - 3) / 2 + 1, (s[1] + 2 - 3) / 2 + 1, (s[2] + 2 - 3) / 2 + 1};
}
inline int64_t down_words(int32_t batch, const int32_t* shape_in) {
    const Shape3 so = down_shape(shape_in);
    const int64_t cells = (int64_t)batch * so.z * so.y * so.x;
    return (cells + 31) / 32;
}

}  // namespace

extern "C" {

size_t seg3d_coord_hash_bytes(int64_t m) { return (size_t)hash_capacity(m < 0 ? 0 : m) * 12; }

int seg3d_coord_hash_build(const int32_t* coords, int64_t m, const int32_t* shape_zyx, void* table, size_t table_bytes,
                           void* stream) {
    if (m < 0 || !shape_zyx || !table || (m > 0 && !coords)) return SEG3D_EINVAL;
    const uint64_t cap = hash_capacity(m);
    if (table_bytes < cap * 12) return SEG3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    HashView h = hash_view(table, cap);
    SEG3D_CHECK_HIP(hipMemsetAsync(h.keys, 0xFF, cap * 8, st));
    if (m == 0) return SEG3D_OK;
    hipLaunchKernelGGL(hash_fill, dim3((unsigned)ceil_div64(m, kThreads)), dim3(kThreads), 0, st, coords, m,
                       shape_of(shape_zyx), h);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_rulebook_subm(const int32_t* coords, int64_t m, const int32_t* shape_zyx, const void* table, size_t table_bytes,
                        int32_t* nbr, void* stream) {
    if (m < 0 || !shape_zyx || !table || (m > 0 && (!coords || !nbr))) return SEG3D_EINVAL;
    const uint64_t cap = hash_capacity(m);
    if (table_bytes < cap * 12) return SEG3D_EWORKSPACE;
    if (m == 0) return SEG3D_OK;
    HashView h = hash_view(const_cast<void*>(table), cap);
    hipLaunchKernelGGL(subm_table, dim3((unsigned)ceil_div64(m, kThreads), 27), dim3(kThreads), 0, as_stream(stream),
                       coords, m, shape_of(shape_zyx), h, nbr);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

size_t seg3d_downsample_workspace_bytes(int32_t batch_size, const int32_t* shape_in_zyx) {
    if (batch_size <= 0 || !shape_in_zyx) return 0;
    const int64_t nw = down_words(batch_size, shape_in_zyx);
    WsCarver c(nullptr);
    c.take<uint32_t>((size_t)nw + 1);
    c.take<uint32_t>((size_t)nw + 1);
    c.take<uint32_t>(scan_tmp_count(nw));
    return c.off;
}

int seg3d_downsample_coords(const int32_t* coords_in, int64_t m_in, const int32_t* m_in_dev, int32_t batch_size,
                            const int32_t* shape_in_zyx, int32_t* coords_out, int64_t cap_out, int32_t* m_out,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (m_in < 0 || batch_size <= 0 || !shape_in_zyx || !coords_out || !m_out || !workspace || cap_out < 0 ||
        (m_in > 0 && !coords_in))
        return SEG3D_EINVAL;
    if (workspace_bytes < seg3d_downsample_workspace_bytes(batch_size, shape_in_zyx)) return SEG3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    const int64_t nw = down_words(batch_size, shape_in_zyx);
    const Shape3 so = down_shape(shape_in_zyx);
    WsCarver c(workspace);
    uint32_t* bitmap = c.take<uint32_t>((size_t)nw + 1);
    uint32_t* cnt = c.take<uint32_t>((size_t)nw + 1);
    uint32_t* tmp = c.take<uint32_t>(scan_tmp_count(nw));
    SEG3D_CHECK_HIP(hipMemsetAsync(bitmap, 0, (size_t)nw * 4, st));
    if (m_in > 0) {
        hipLaunchKernelGGL(down_mark, dim3((unsigned)ceil_div64(m_in, kThreads)), dim3(kThreads), 0, st, coords_in, m_in,
                           m_in_dev, so, bitmap);
        SEG3D_CHECK_LAUNCH();
    }
    const unsigned nbw = (unsigned)ceil_div64(nw, kThreads);
    hipLaunchKernelGGL(down_popc, dim3(nbw), dim3(kThreads), 0, st, bitmap, nw, cnt);
    SEG3D_CHECK_LAUNCH();
    int rc = scan_exclusive_u32(cnt, cnt, nw, reinterpret_cast<uint32_t*>(m_out), tmp, st);
    if (rc != SEG3D_OK) return rc;
    hipLaunchKernelGGL(down_emit, dim3(nbw), dim3(kThreads), 0, st, bitmap, cnt, nw, so, cap_out, coords_out);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_rulebook_strided(const int32_t* coords_out, int64_t m_out, int64_t m_in, const int32_t* shape_in_zyx,
                           const void* table_in, size_t table_bytes, int32_t* nbr_fwd, int32_t* nbr_inv, void* stream) {
    if (m_out < 0 || m_in < 0 || !shape_in_zyx || !table_in || (m_out > 0 && (!coords_out || !nbr_fwd)))
        return SEG3D_EINVAL;
    const uint64_t cap = hash_capacity(m_in);
    if (table_bytes < cap * 12) return SEG3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    if (nbr_inv && m_in > 0)
        SEG3D_CHECK_HIP(hipMemsetAsync(nbr_inv, 0xFF, (size_t)27 * m_in * 4, st));
    if (m_out == 0) return SEG3D_OK;
    HashView h = hash_view(const_cast<void*>(table_in), cap);
    hipLaunchKernelGGL(strided_table, dim3((unsigned)ceil_div64(m_out, kThreads), 27), dim3(kThreads), 0, st, coords_out,
                       m_out, m_in, shape_of(shape_in_zyx), h, nbr_fwd, nbr_inv);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // extern "C"
