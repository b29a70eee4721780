// a13, a15, a16, a17, a18: sparse window partition + positional embedding.
// Reference: seg3d/utils/swformer_utils.py:8-31,108-171 (get_window_coors, get_flat2win_inds,
// make_continuous_inds) and seg3d/models/layers/point_transformer_layer.py:71-87,141-220
// (batching_single_shift, get_pos_embed, get_key_padding_mask) -- ~25 small torch kernels, a
// unique+sort, a dense canvas and several .item()/.any() host syncs per (stage, shift).
//
// Here one call per (stage, shift) builds, without any host sync:
//   window id / in-window coordinate / in-window rank / batching level / flat2window slot per voxel
//   (the reference's intermediates, kept for bit-exact parity), and the CSR of non-empty windows
//   (tok, win_start, win_count) that the variable-length attention kernel consumes directly.
// Window ids index a dense canvas (batch * Wx*Wy*Wz counters); an exclusive scan of the counters
// gives CSR offsets, a 4-lane scan of per-level flags gives the compact window index per level in
// ascending id order (= make_continuous_inds), and the deterministic in-group rank of group.hip
// replaces the atomic-order rank of the reference's CUDA op.
#include "common.hpp"

int group_index_launch(const int32_t* gid, int64_t n, int64_t ng, int32_t* rank, int32_t* order, int32_t* offsets,
                       void* workspace, hipStream_t st, uint32_t** count_out, uint32_t** offs_out);

namespace {

constexpr int kThreads = 256;
constexpr int kMaxLevels = 4;

struct WinGeom {
    int32_t win[3];    // x, y, z
    int32_t nwin[3];   // x, y, z
    int32_t shift[3];  // x, y, z
};

struct Levels {
    int32_t n;
    int32_t lo[kMaxLevels], hi[kMaxLevels], cap[kMaxLevels];
};

__global__ __launch_bounds__(kThreads) void win_ids(const int32_t* __restrict__ coords, int64_t m, WinGeom g,
                                                    int32_t* __restrict__ win_id, int32_t* __restrict__ in_win) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= m) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];  // b, z, y, x
    const int sx = c.w + g.shift[0], sy = c.z + g.shift[1], sz = c.y + g.shift[2];
    // coordinates and shifts are non-negative: C division == floor division (swformer_utils.py:137-139)
    const int wx = sx / g.win[0], wy = sy / g.win[1], wz = sz / g.win[2];
    const int per_sample = g.nwin[0] * g.nwin[1] * g.nwin[2];
    win_id[i] = c.x * per_sample + wx * (g.nwin[1] * g.nwin[2]) + wy * g.nwin[2] + wz;
    if (in_win) {
        in_win[3 * i + 0] = sz - wz * g.win[2];
        in_win[3 * i + 1] = sy - wy * g.win[1];
        in_win[3 * i + 2] = sx - wx * g.win[0];
    }
}

__device__ __forceinline__ int level_of(uint32_t n, const Levels& lv) {
    // later levels overwrite earlier ones, as the loop at point_transformer_layer.py:80-85 does
    int l = -1;
#pragma unroll
    for (int j = 0; j < kMaxLevels; ++j)
        if (j < lv.n && (int32_t)n >= lv.lo[j] && (int32_t)n < lv.hi[j]) l = j;
    return l;
}

__global__ __launch_bounds__(kThreads) void win_level_flags(const uint32_t* __restrict__ count, int64_t n_canvas, Levels lv,
                                                            uint4* __restrict__ flags) {
    const int64_t w = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (w >= n_canvas) return;
    const uint32_t n = count[w];
    // windows whose count matches no range still take a compact slot (lane 3 + marker in level -1)
    const int l = n ? level_of(n, lv) : -2;
    flags[w] = make_uint4(l == 0, l == 1, l == 2, (l == 3) || (l == -1));
}

__global__ __launch_bounds__(kThreads) void win_per_voxel(const int32_t* __restrict__ win_id, const int32_t* __restrict__ rank,
                                                          int64_t m, const uint32_t* __restrict__ count,
                                                          const uint4* __restrict__ lvl_prefix, Levels lv,
                                                          int32_t* __restrict__ level, int32_t* __restrict__ slot,
                                                          int32_t* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= m) return;
    const int32_t w = win_id[i];
    const int l = level_of(count[w], lv);
    if (level) level[i] = l;
    int32_t s = -1;
    if (l >= 0) {
        const uint4 p = lvl_prefix[w];
        const uint32_t cw = l == 0 ? p.x : l == 1 ? p.y : l == 2 ? p.z : p.w;
        const int32_t r = rank[i];
        if (r < lv.cap[l]) s = (int32_t)cw * lv.cap[l] + r;
    }
    if (slot) slot[i] = s;
    if (s < 0) atomicAdd(&counts[1], 1);
}

// per canvas entry: (non-empty, 32-token tiles, 128-token query chunks, 0) -- scanned to give every window its
// compact index, its first tile in the 32-padded token space and its first attention work items
__global__ __launch_bounds__(kThreads) void win_geom_flags(const uint32_t* __restrict__ count, int64_t n_canvas,
                                                           uint4* __restrict__ flags) {
    const int64_t w = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (w >= n_canvas) return;
    const uint32_t n = count[w];
    flags[w] = make_uint4(n ? 1u : 0u, (n + 31u) >> 5, (n + 127u) >> 7, 0u);
}

__global__ __launch_bounds__(kThreads) void win_compact(const uint32_t* __restrict__ count, const uint32_t* __restrict__ offs,
                                                        const uint4* __restrict__ geom_prefix, int64_t n_canvas,
                                                        int32_t* __restrict__ win_start, int32_t* __restrict__ win_count,
                                                        int32_t* __restrict__ win_tile0, int4* __restrict__ tile_item,
                                                        int4* __restrict__ qg_item, int32_t* __restrict__ counts) {
    const int64_t w = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (w >= n_canvas) return;
    const uint32_t n = count[w];
    const uint4 p = geom_prefix[w];
    const uint32_t cw = p.x;  // non-empty windows with a smaller id
    if (n) {
        win_start[cw] = (int32_t)offs[w];
        win_count[cw] = (int32_t)n;
        if (win_tile0) win_tile0[cw] = (int32_t)p.y;
        if (tile_item)
            for (uint32_t t = 0; t < ((n + 31u) >> 5); ++t) tile_item[p.y + t] = make_int4((int)cw, (int)t, (int)offs[w], (int)n);
        if (qg_item)
            for (uint32_t q = 0; q < ((n + 127u) >> 7); ++q) qg_item[p.z + q] = make_int4((int)cw, (int)q, (int)offs[w], (int)n);
    }
    if (w == n_canvas - 1) {
        counts[0] = (int32_t)(cw + (n ? 1u : 0u));
        counts[2] = (int32_t)(p.y + ((n + 31u) >> 5));
        counts[3] = (int32_t)(p.z + ((n + 127u) >> 7));
    }
}

// a17 -- one thread per (voxel, output channel)
__global__ __launch_bounds__(kThreads) void pos_embed_kernel(const int32_t* __restrict__ in_win, int64_t m, int wx, int wy,
                                                             int wz, const float* __restrict__ inv_freq, int c,
                                                             float* __restrict__ pos) {
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (t >= m * c) return;
    const int64_t i = t / c;
    const int ch = (int)(t - i * c);
    const int plen = c / 3;
    const int d = ch / plen;  // 0: x, 1: y, 2: z  (torch.cat([embed_x, embed_y, embed_z]), :199-203)
    const int j = ch - d * plen;
    // coors_in_win is (z, y, x); v = coord - win/2 (:171)
    float v;
    if (d == 0) v = (float)in_win[3 * i + 2] - (float)wx / 2.0f;
    else if (d == 1) v = (float)in_win[3 * i + 1] - (float)wy / 2.0f;
    else v = (float)in_win[3 * i + 0] - (float)wz / 2.0f;
    const float e = v / inv_freq[j];
    pos[t] = (j & 1) ? cosf(e) : sinf(e);
}

size_t canvas_of(int32_t batch, const int32_t* nwin) { return (size_t)batch * nwin[0] * nwin[1] * nwin[2]; }

}  // namespace

extern "C" {

size_t seg3d_window_partition_workspace_bytes(int64_t m, int32_t batch_size, const int32_t* nwin_xyz) {
    if (m < 0 || batch_size <= 0 || !nwin_xyz) return 0;
    const size_t nc = canvas_of(batch_size, nwin_xyz);
    WsCarver c(nullptr);
    c.take<char>(seg3d_group_index_workspace_bytes(m, (int64_t)nc));
    c.take<uint4>(nc + 1);
    c.take<uint4>(nc + 1);
    c.take<uint4>(scan_tmp_count((int64_t)nc));
    c.take<int32_t>((size_t)m + 1);  // win ids when the caller passes NULL
    c.take<int32_t>((size_t)m + 1);  // ranks when the caller passes NULL
    return c.off;
}

int seg3d_window_partition(const int32_t* coords, int64_t m, int32_t batch_size, const int32_t* win_xyz,
                           const int32_t* nwin_xyz, const int32_t* shift_xyz, int32_t n_levels, const int32_t* level_lo,
                           const int32_t* level_hi, const int32_t* level_cap, int32_t* win_id, int32_t* in_win,
                           int32_t* rank, int32_t* level, int32_t* slot, int32_t* tok, int32_t* win_start,
                           int32_t* win_count, int32_t* win_tile0, int32_t* tile_item, int32_t* qg_item,
                           int32_t* counts, void* workspace, size_t workspace_bytes, void* stream) {
    if (m < 0 || batch_size <= 0 || !win_xyz || !nwin_xyz || !shift_xyz || n_levels < 1 || n_levels > kMaxLevels ||
        !level_lo || !level_hi || !level_cap || !tok || !win_start || !win_count || !counts || !workspace ||
        (m > 0 && !coords))
        return SEG3D_EINVAL;
    for (int j = 0; j < 3; ++j)
        if (win_xyz[j] <= 0 || nwin_xyz[j] <= 0 || shift_xyz[j] < 0) return SEG3D_EINVAL;
    if (workspace_bytes < seg3d_window_partition_workspace_bytes(m, batch_size, nwin_xyz)) return SEG3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    const size_t nc = canvas_of(batch_size, nwin_xyz);
    if (nc >= 0x7F000000u) return SEG3D_EINVAL;

    WsCarver c(workspace);
    void* grp_ws = c.take<char>(seg3d_group_index_workspace_bytes(m, (int64_t)nc));
    uint4* lvl = c.take<uint4>(nc + 1);
    uint4* geom = c.take<uint4>(nc + 1);
    uint4* lvl_tmp = c.take<uint4>(scan_tmp_count((int64_t)nc));
    int32_t* wid_ws = c.take<int32_t>((size_t)m + 1);
    int32_t* rank_ws = c.take<int32_t>((size_t)m + 1);
    if (!win_id) win_id = wid_ws;
    if (!rank) rank = rank_ws;

    WinGeom g;
    Levels lv;
    for (int j = 0; j < 3; ++j) {
        g.win[j] = win_xyz[j];
        g.nwin[j] = nwin_xyz[j];
        g.shift[j] = shift_xyz[j];
    }
    lv.n = n_levels;
    for (int j = 0; j < kMaxLevels; ++j) {
        lv.lo[j] = j < n_levels ? level_lo[j] : 0;
        lv.hi[j] = j < n_levels ? level_hi[j] : 0;
        lv.cap[j] = j < n_levels ? level_cap[j] : 0;
    }
    SEG3D_CHECK_HIP(hipMemsetAsync(counts, 0, 4 * sizeof(int32_t), st));
    if (m == 0) return SEG3D_OK;

    const unsigned nbv = (unsigned)ceil_div64(m, kThreads), nbc = (unsigned)ceil_div64((int64_t)nc, kThreads);
    hipLaunchKernelGGL(win_ids, dim3(nbv), dim3(kThreads), 0, st, coords, m, g, win_id, in_win);
    SEG3D_CHECK_LAUNCH();
    uint32_t *count = nullptr, *offs = nullptr;
    int rc = group_index_launch(win_id, m, (int64_t)nc, rank, tok, nullptr, grp_ws, st, &count, &offs);
    if (rc != SEG3D_OK) return rc;
    // batching levels / flat2window slots (the reference's intermediates) also detect dropped voxels
    hipLaunchKernelGGL(win_level_flags, dim3(nbc), dim3(kThreads), 0, st, count, (int64_t)nc, lv, lvl);
    SEG3D_CHECK_LAUNCH();
    rc = scan_exclusive_u32x4(lvl, lvl, (int64_t)nc, nullptr, lvl_tmp, st);
    if (rc != SEG3D_OK) return rc;
    hipLaunchKernelGGL(win_per_voxel, dim3(nbv), dim3(kThreads), 0, st, win_id, rank, m, count, lvl, lv, level, slot,
                       counts);
    SEG3D_CHECK_LAUNCH();
    // CSR of non-empty windows + 32-padded tile geometry + attention work items
    hipLaunchKernelGGL(win_geom_flags, dim3(nbc), dim3(kThreads), 0, st, count, (int64_t)nc, geom);
    SEG3D_CHECK_LAUNCH();
    rc = scan_exclusive_u32x4(geom, geom, (int64_t)nc, nullptr, lvl_tmp, st);
    if (rc != SEG3D_OK) return rc;
    hipLaunchKernelGGL(win_compact, dim3(nbc), dim3(kThreads), 0, st, count, offs, geom, (int64_t)nc, win_start,
                       win_count, win_tile0, reinterpret_cast<int4*>(tile_item), reinterpret_cast<int4*>(qg_item),
                       counts);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_pos_embed(const int32_t* in_win, int64_t m, const int32_t* win_xyz, const float* inv_freq, int32_t c,
                    float* pos, void* stream) {
    if (m < 0 || !win_xyz || c <= 0 || c % 3 != 0 || ((c / 3) & 1)) return SEG3D_EINVAL;
    if (m == 0) return SEG3D_OK;
    if (!in_win || !inv_freq || !pos) return SEG3D_EINVAL;
    hipLaunchKernelGGL(pos_embed_kernel, dim3((unsigned)ceil_div64(m * c, kThreads)), dim3(kThreads), 0,
                       as_stream(stream), in_win, m, win_xyz[0], win_xyz[1], win_xyz[2], inv_freq, c, pos);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // extern "C"
