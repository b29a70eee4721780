// Dense Linear weight gradient (a6/a22 wgrad) in split-bf16 arithmetic, workgroup-tiled and deterministic:
//   dw[co][ci] = sum_r dy[r][co] * x[r][ci],   db[co] = sum_r dy[r][co]
// A tall-skinny GEMM: outputs of C x C (48..768), reduction over 1e4..2e5 rows; the row index is the MFMA K
// dimension of v_mfma_f32_16x16x32_bf16.
//
// A workgroup of WA x WB waves owns a (64*WA) x (64*WB) block of dw over one chunk of rows; wave (wa, wb)
// accumulates the 64 x 64 sub-block in 16 accumulator tiles.  Per 32-row step the workgroup stages WA slabs of
// dy and WB slabs of x (a slab = 32 rows x 64 channels): the wave that owns a slab loads 8 rows x 4 channels
// per lane (16-B loads along the channel axis), converts to bf16 hi/lo and packs the 8 rows of one channel into
// one 16-B record -- the transpose happens in registers -- and writes the records to a double-buffered LDS
// image [slab][hi|lo][row group][64 records]; every wave then reads its A (dy) and B (x) fragments with
// conflict-free 16-B reads and issues 16*3 MFMAs.  The loads of step s+1 are issued before the MFMAs of step
// s, so global latency hides behind the matrix pipe; one __syncthreads per step.
// Each operand row is read (64*WA + 64*WB) / (64*WA * 64*WB) times per output column instead of 1/32: the
// (WA, WB) shape is picked per layer to minimise MFMA padding + operand traffic (plan()).
// Channel c of a slab sits in record (c%4)*16 + c/4, so MFMA tile t holds channels {4*i + t}.
//
// No atomics: every (row chunk, block) writes its partial block to a workspace with plain stores and a second
// kernel sums the chunks in a fixed order -- the gradient is bit-reproducible run to run, and neither dw nor
// db needs a memset.  gridDim.x is padded to a multiple of 8 so that the blocks sharing a row chunk land
// on the same XCD (round-robin dispatch) and share its L2.
#include "attn_common.hpp"

namespace {

using namespace attn;

struct Plan {
    int wa, wb;      // waves along cout / cin
    int nbo, nbi;    // blocks along cout / cin
    int chunks;      // row chunks (partial sums)
    int64_t rows;    // rows per chunk (multiple of 32)
};

Plan plan(int64_t m, int cin, int cout) {
    static const int shapes[][2] = {{1, 1}, {1, 2}, {1, 3}, {1, 4}, {1, 6}, {2, 1}, {2, 2}, {2, 3}, {2, 4},
                                    {3, 1}, {3, 2}, {4, 1}, {4, 2}, {6, 1}};
    Plan best{};
    double best_cost = 1e30;
    for (const auto& s : shapes) {
        const int wa = s[0], wb = s[1];
        const int nbo = (cout + 64 * wa - 1) / (64 * wa), nbi = (cin + 64 * wb - 1) / (64 * wb);
        // per 32-row step on one CU: MFMA cycles (48 MFMAs x 16 cycles per wave tile, 4 SIMDs) + operand fetch
        // cycles (64 B/clk per CU), counted without overlap
        const double mfma = (double)nbo * wa * nbi * wb * 768.0 / 4.0;
        const double mem = (double)nbo * nbi * (wa + wb) * 64.0 * 32.0 * 4.0 / 64.0;
        const double cost = mfma + mem;
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = Plan{wa, wb, nbo, nbi, 0, 0};
        }
    }
    // one resident workgroup per CU when the partial blocks are large (their write + re-read is the overhead
    // that grows with the chunk count), two when they are small; chunks of at least 256 rows
    const int tiles = best.nbo * best.nbi;
    const int target = (int64_t)cin * cout >= 65536 ? 256 : 512;
    int64_t chunks = (target + tiles - 1) / tiles;
    if (chunks < 1) chunks = 1;
    int64_t rows = m > 0 ? (m + chunks - 1) / chunks : 32;
    if (rows < 256) rows = 256;
    rows = (rows + 31) / 32 * 32;
    best.rows = rows;
    best.chunks = m > 0 ? (int)((m + rows - 1) / rows) : 0;
    return best;
}

template <int WA, int WB>
__global__ __launch_bounds__(64 * WA * WB) void wgrad_dense_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ dy, int64_t m_rows,
                                                                    int cin, int cout, int rows_per_chunk, int nbi,
                                                                    float* __restrict__ part_w,
                                                                    float* __restrict__ part_b) {
    constexpr int NW = WA * WB, NS = WA + WB, ITEMS = (NS + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];  // [2 buffers][NS slabs][2 hi/lo][4 row groups][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cq = lane & 15, rg = lane >> 4;  // load role: channel quad, row group (8 rows); MFMA role: c16, g
    const int wa = wave / WB, wb = wave % WB;
    const int bi = blockIdx.y % nbi, bo = blockIdx.y / nbi;
    const int ci0 = bi * 64 * WB, co0 = bo * 64 * WA;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_chunk;
    if (r_begin >= m_rows) return;  // padding blocks of the XCD-aligned grid (whole workgroup)
    const int64_t r_end = r_begin + rows_per_chunk < m_rows ? r_begin + rows_per_chunk : m_rows;
    const int n_steps = (int)((r_end - r_begin + 31) / 32);

    auto image = [&](int buf, int slab, int hl, int row_group) -> uint4* {
        return lds + ((((buf * NS + slab) * 2 + hl) * 4 + row_group) * 64);
    };

    // slabs this wave stages: slab < WA is dy channels co0 + 64*slab.., otherwise x channels ci0 + 64*(slab-WA)..
    const float* src[ITEMS];
    int ld[ITEMS];
    bool has[ITEMS], ch_ok[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int s = wave + it * NW;
        has[it] = s < NS;
        const bool is_dy = s < WA;
        const int c = is_dy ? co0 + 64 * s + 4 * cq : ci0 + 64 * (s - WA) + 4 * cq;
        ld[it] = is_dy ? cout : cin;
        ch_ok[it] = has[it] && c < ld[it];
        src[it] = (is_dy ? dy : x) + (ch_ok[it] ? c : 0);
    }
    const bool want_db = part_b != nullptr && bi == 0 && wave < WA;  // dy slabs are always item 0 of waves < WA
    f32x4 db_acc = {0.f, 0.f, 0.f, 0.f};

    f32x4 pre[ITEMS][8];
    auto fetch = [&](int step) {
        const int64_t r0 = r_begin + 32 * (int64_t)step + 8 * rg;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            if (!has[it]) continue;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t r = r0 + i;
                const bool ok = r < r_end && ch_ok[it];
                const f32x4 v = *reinterpret_cast<const f32x4*>(src[it] + (ok ? r : r_begin) * ld[it]);
                pre[it][i] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            if (!has[it]) continue;
            const int s = wave + it * NW;
            if (it == 0 && want_db) {
#pragma unroll
                for (int i = 0; i < 8; ++i) db_acc += pre[0][i];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = pre[it][i][j];
                bf16x8 hi, lo;
                split_frag(v, &hi, &lo);
                image(buf, s, 0, rg)[j * 16 + cq] = __builtin_bit_cast(uint4, hi);
                image(buf, s, 1, rg)[j * 16 + cq] = __builtin_bit_cast(uint4, lo);
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    fetch(0);
    stash(0);
    __syncthreads();
    for (int step = 0; step < n_steps; ++step) {
        const int buf = step & 1;
        const bool more = step + 1 < n_steps;
        if (more) fetch(step + 1);
        bf16x8 b_hi[4], b_lo[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            b_hi[b] = __builtin_bit_cast(bf16x8, image(buf, WA + wb, 0, rg)[b * 16 + cq]);
            b_lo[b] = __builtin_bit_cast(bf16x8, image(buf, WA + wb, 1, rg)[b * 16 + cq]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const bf16x8 a_hi = __builtin_bit_cast(bf16x8, image(buf, wa, 0, rg)[a * 16 + cq]);
            const bf16x8 a_lo = __builtin_bit_cast(bf16x8, image(buf, wa, 1, rg)[a * 16 + cq]);
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = mfma3(a_hi, a_lo, b_hi[b], b_lo[b], acc[a][b]);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
    }

    const int64_t chunk = blockIdx.x;
    if (want_db) {  // lanes cq, cq+16, cq+32, cq+48 hold the four row groups of the same channel quad
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = db_acc[j];
            v += __shfl_xor(v, 16, SEG3D_WAVE);
            v += __shfl_xor(v, 32, SEG3D_WAVE);
            const int co = co0 + 64 * wave + 4 * cq + j;
            if (rg == 0 && co < cout) part_b[chunk * cout + co] = v;
        }
    }
    // ---- epilogue: acc[a][b][r] = dw[co = 4*(4g + r) + a][ci = 4*c16 + b] of this wave's sub-block; staged in two
    // halves of 32 output rows through 8 KiB of LDS per wave, stored as whole contiguous rows of the partial
    float* st = reinterpret_cast<float*>(lds) + wave * 2048;
    float* pw = part_w + chunk * (int64_t)cout * cin;
    const int ci = ci0 + 64 * wb + lane;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((rg >> 1) == h) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[(4 * (4 * (rg & 1) + r) + a) * 64 + 4 * cq + b] = acc[a][b][r];
        }
        __builtin_amdgcn_wave_barrier();
        if (ci < cin) {
            for (int row = 0; row < 32; ++row) {
                const int co = co0 + 64 * wa + 32 * h + row;
                if (co >= cout) break;
                pw[(int64_t)co * cin + ci] = st[row * 64 + lane];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// out[i] = sum over chunks of part[c][i], fixed order
__global__ __launch_bounds__(256) void wgrad_dense_reduce(const float* __restrict__ part, int chunks, int64_t n,
                                                          float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 4 <= chunks; c += 4) {
        s0 += part[(int64_t)c * n + i];
        s1 += part[(int64_t)(c + 1) * n + i];
        s2 += part[(int64_t)(c + 2) * n + i];
        s3 += part[(int64_t)(c + 3) * n + i];
    }
    for (; c < chunks; ++c) s0 += part[(int64_t)c * n + i];
    out[i] = (s0 + s1) + (s2 + s3);
}

template <int WA, int WB>
int launch(const Plan& p, const float* x, const float* dy, int64_t m, int cin, int cout, float* part_w, float* part_b,
           hipStream_t st) {
    constexpr int NS = WA + WB;
    constexpr size_t lds_bytes = (size_t)2 * NS * 2 * 4 * 64 * sizeof(uint4);
    static bool configured = false;
    if (!configured) {
        if (lds_bytes > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_dense_kernel<WA, WB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return SEG3D_ELAUNCH;
        configured = true;
    }
    const unsigned gx = (unsigned)((p.chunks + 7) / 8 * 8);
    hipLaunchKernelGGL((wgrad_dense_kernel<WA, WB>), dim3(gx, (unsigned)(p.nbo * p.nbi)), dim3(64 * WA * WB), lds_bytes,
                       st, x, dy, m, cin, cout, (int)p.rows, p.nbi, part_w, part_b);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // namespace

extern "C" size_t seg3d_linear_wgrad_workspace_bytes(int64_t m, int32_t cin, int32_t cout) {
    if (m < 0 || cin <= 0 || cout <= 0) return 0;
    const Plan p = plan(m, cin, cout);
    return ((size_t)p.chunks * ((size_t)cin * cout + cout) + 64) * sizeof(float);
}

extern "C" int seg3d_linear_wgrad(const float* x, const float* dy, int64_t m, int32_t cin, int32_t cout, float* dw,
                                  float* db, void* workspace, size_t workspace_bytes, void* stream) {
    if (m < 0 || cin <= 0 || cout <= 0 || (cin & 3) || (cout & 3) || !dw) return SEG3D_EINVAL;
    if (m > 0 && (!x || !dy)) return SEG3D_EINVAL;
    if (workspace_bytes < seg3d_linear_wgrad_workspace_bytes(m, cin, cout) || (m > 0 && !workspace)) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    const Plan p = plan(m, cin, cout);
    float* part_w = static_cast<float*>(workspace);
    float* part_b = part_w + (size_t)p.chunks * cin * cout;
    if (m > 0) {
        int rc = SEG3D_EINVAL;
#define SEG3D_WG(A, B) \
    if (p.wa == A && p.wb == B) rc = launch<A, B>(p, x, dy, m, cin, cout, part_w, db ? part_b : nullptr, st)
        SEG3D_WG(1, 1);
        SEG3D_WG(1, 2);
        SEG3D_WG(1, 3);
        SEG3D_WG(1, 4);
        SEG3D_WG(1, 6);
        SEG3D_WG(2, 1);
        SEG3D_WG(2, 2);
        SEG3D_WG(2, 3);
        SEG3D_WG(2, 4);
        SEG3D_WG(3, 1);
        SEG3D_WG(3, 2);
        SEG3D_WG(4, 1);
        SEG3D_WG(4, 2);
        SEG3D_WG(6, 1);
#undef SEG3D_WG
        if (rc != SEG3D_OK) return rc;
    }
    const int64_t nw = (int64_t)cin * cout;
    hipLaunchKernelGGL(wgrad_dense_reduce, dim3((unsigned)ceil_div64(nw, 256)), dim3(256), 0, st, part_w, p.chunks, nw, dw);
    SEG3D_CHECK_LAUNCH();
    if (db) {
        hipLaunchKernelGGL(wgrad_dense_reduce, dim3((unsigned)ceil_div64(cout, 256)), dim3(256), 0, st, part_b, p.chunks,
                           (int64_t)cout, db);
        SEG3D_CHECK_LAUNCH();
    }
    return SEG3D_OK;
}
