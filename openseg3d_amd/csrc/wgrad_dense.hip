// Dense Linear weight gradient (a6/a22 wgrad) in split-bf16 arithmetic, deterministic:
//   dw[co][ci] = sum_r dy[r][co] * x[r][ci],   db[co] = sum_r dy[r][co]
// A tall-skinny GEMM: outputs of C x C (48..768), reduction over 1e4..2e5 rows; the row index is the MFMA K
// dimension of v_mfma_f32_16x16x32_bf16.
//
// A workgroup of 4 waves owns one 64 x 64 block of dw (co x ci, masked at the edges) over one chunk of rows; its
// waves take the 32-row steps of the chunk round-robin (split-K inside the workgroup) and each accumulates the
// whole block in 16 accumulator tiles.  Operands go from global memory straight into MFMA fragments: lane
// (cq, rg) loads rows 8*rg .. 8*rg+7 of channels 4*cq .. 4*cq+3 with 16-B loads, so for channel j of its quad
// it holds exactly the 8 consecutive K (row) values the MFMA wants from lane (c16 = cq, g = rg) of tile j --
// tile t holds channels {4*i + t}.  No LDS and no barrier in the main loop; the loads of the wave's next step
// are issued as soon as the registers they land in have been converted, and 8+ independent waves per CU cover
// the rest.
//
// No atomics: the four accumulators of a workgroup are summed through LDS in wave order, every (row chunk,
// block) writes its partial to a workspace with plain stores and a second kernel sums the chunks in a fixed
// order -- the gradient is bit-reproducible run to run, and neither dw nor db needs a memset.
// Block order: id -> (xcd = id % 8, k = id / 8), chunk = 8 * (k / tiles) + xcd, block = k % tiles, so the
// workgroups that read the same rows (all blocks of a chunk) are dispatched back to back on the same XCD
// and share its L2.
#include <cstdlib>
#include <type_traits>

#include "attn_common.hpp"

// wgrad_dense_lds.hip: the LDS-shared form for cout % 96 == 0, cin % 48 == 0 (1 = it takes the shape)
int wgrad_dense_lds_plan(int64_t m, int cin, int cout, int* chunks, int64_t* rows);
int wgrad_dense_lds_launch(const float* x, const float* dy, int64_t m, int cin, int cout, int chunks, int64_t rows, float* part,
                           int want_bias, hipStream_t st);

namespace {

using namespace attn;

constexpr int kWaves = 4;
constexpr int kThreads = 64 * kWaves;

struct Plan {
    int nbo, nbi;  // 64-channel blocks along cout / cin
    int chunks;    // row chunks (partial sums)
    int64_t rows;  // rows per chunk (multiple of 32)
    int group;     // 0 = the waves of a workgroup split the rows of one block; else QB << 4 | QC blocks per workgroup
    int lds;       // 1 = the LDS-shared kernel of wgrad_dense_lds.hip (fp32 rows only)
};

// SEG3D_WGRAD_GROUP=1 (A/B, OFF by default: it lost).  Measured (tools/wgrad_bench.py, one box, split-K -> groups): 96 -> 192
// @121 k rows 36.7 -> 45.9 us, 96 -> 96 29.2 -> 37.6, 192 -> 384 @58 k 52.3 -> 55.6, 192 -> 192 32.2 -> 35.1; only the 19 k-row
// level gains (384 -> 768 70.4 -> 63.7, 384 -> 384 39.6 -> 35.9); training step 42.3 -> 42.7 ms.  The waves of a group drift
// apart by more rows than the 32 KB L1 holds, so the shared slab is fetched from L2 by each of them anyway, and a wave's chain
// is four times as long.
static const bool g_wgrad_group = [] {
    const char* e = getenv("SEG3D_WGRAD_GROUP");
    return e && atoi(e) == 1;
}();

Plan plan(int64_t m, int cin, int cout, bool allow_lds = false) {
    Plan p{(cout + 63) / 64, (cin + 63) / 64, 0, 32, 0, 0};
    if (m <= 0) return p;
    if (allow_lds) {
        int c = 0;
        int64_t r = 0;
        if (wgrad_dense_lds_plan(m, cin, cout, &c, &r)) {
            p.chunks = c;
            p.rows = r;
            p.lds = 1;
            return p;
        }
    }
    if (g_wgrad_group && p.nbo * p.nbi >= 4 && m >= 8192) {
        // the block group with the fewest operand slabs per block among the shapes that divide the block grid; at least three waves
        int best_b = 0, best_c = 0;
        float best = 1e9f;
        for (int qb = 1; qb <= 4; ++qb)
            for (int qc = 1; qb * qc <= 4; ++qc) {
                if (p.nbo % qb || p.nbi % qc || qb * qc < 3) continue;
                const float slabs = (float)(qb + qc) / (float)(qb * qc) - 0.01f * (float)(qb * qc);
                if (slabs < best) best = slabs, best_b = qb, best_c = qc;
            }
        if (best_b) {
            p.group = best_b << 4 | best_c;
            const int groups = (p.nbo / best_b) * (p.nbi / best_c);
            // one resident round of workgroups (512), every wave walking its whole chunk: >= 8 steps per wave
            int64_t chunks = 512 / groups / 8 * 8;
            if (chunks < 8) chunks = 8;
            int64_t rows = (m + chunks - 1) / chunks;
            if (rows < 32 * 8) rows = 32 * 8;
            rows = (rows + 31) / 32 * 32;
            p.rows = rows;
            p.chunks = (int)((m + rows - 1) / rows);
            return p;
        }
    }
    const int tiles = p.nbo * p.nbi;
    // 2 waves per SIMD = 2 workgroups per CU = 512 resident workgroups.  Few blocks per chunk: one resident
    // round of long chunks (per-workgroup prologue / reduction overhead matters); many blocks per chunk: two
    // rounds.  Chunks map to XCDs round-robin, so their number is kept a multiple of 8.
    int target = tiles >= 16 ? 1024 : 512;
    if (const char* e = getenv("SEG3D_WGRAD_TARGET")) target = atoi(e);
    int64_t chunks = target / tiles / 8 * 8;
    if (chunks < 8) chunks = 8;
    int64_t rows = (m + chunks - 1) / chunks;
    if (rows < 32 * kWaves * 4) rows = 32 * kWaves * 4;
    rows = (rows + 31) / 32 * 32;
    p.rows = rows;
    p.chunks = (int)((m + rows - 1) / rows);
    return p;
}

// bf16 fragment of channel `comp` (0..3) of a lane's quad from 8 rows of 4 bf16 each (q[i] = row i: {ch0 | ch1 << 16, ch2 | ch3 << 16})
__device__ __forceinline__ bf16x8 frag_from_bf16_rows(const uint2* q, int comp) {
    u32x4 h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t a = (comp & 2) ? q[2 * j].y : q[2 * j].x, b = (comp & 2) ? q[2 * j + 1].y : q[2 * j + 1].x;
        h[j] = (comp & 1) ? ((a >> 16) | (b & 0xFFFF0000u)) : ((a & 0xFFFFu) | (b << 16));
    }
    return __builtin_bit_cast(bf16x8, h);
}

// acc += a . b for a split-bf16 `a` and a `b` that IS bf16 (a bf16 row is its own high half: no a_hi * b_lo term)
__device__ __forceinline__ f32x4 mfma2(const bf16x8& a_hi, const bf16x8& a_lo, const bf16x8& b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b, acc, 0, 0, 0);
    return acc;
}

// XB: the x rows are stored as bf16 (the opt-in bf16 copies a training forward saves for its backward, SEG3D_TRAIN_STORAGE=bf16):
// 8-byte loads, no split of x, two MFMAs per product.  dy is always fp32.
// GROUP (round 5): instead of splitting the ROWS of a chunk over the workgroup's waves (all on ONE 64 x 64 block, summed
// through LDS at the end), the waves take NEIGHBOURING blocks -- QB x QC of them (group = QB << 4 | QC, <= 4 waves) -- and every
// wave walks all rows of the chunk: the dy slab is read by QC waves and the x slab by QB waves at about the same time, so
// the second reader finds it in the CU's L1 and the L2 -> L1 stream that bounds this kernel falls by a third to a half
// (192 <-> 384: 18 blocks x 128 channel-reads per row = 2 304 -> 6 groups x 256 = 1 536).  No cross-wave sum, no barrier: a wave
// stores its own block.  Per-wave work is 4 x longer, so the chunks are 2 x shorter (the partial blocks double: still a
// quarter of the operand bytes).
// DEPTH (round 5, fp32 rows, split-K form): steps of a wave in flight.  The kernel is bound by its waves' round trips (~3 us per
// 16 KB step at two waves per SIMD, section 6); a second register set does not fit 256 registers (101 - 223 spills), so the deep
// form runs ONE wave per SIMD with DEPTH sets of x and dy rows (512 registers).
template <bool XB, bool GROUP, int DEPTH = 1>
__global__ __launch_bounds__(kThreads, (DEPTH > 1 ? 1 : 2)) void wgrad_dense_kernel(const void* __restrict__ x_v, const float* __restrict__ dy,
                                                               int64_t m_rows, int cin, int cout, int rows_per_chunk,
                                                               int nbi, int tiles, float* __restrict__ part,
                                                               int want_bias, int group) {
    const float* x = static_cast<const float*>(x_v);
    __shared__ __attribute__((aligned(16))) float red[GROUP ? 4 : 64 * 64];  // block sum [co_local][ci_local]
    __shared__ float red_b[GROUP ? 1 : kWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cq = lane & 15, rg = lane >> 4;  // channel quad, row group: load role == MFMA role (c16, g)
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int chunk = (k / tiles) * 8 + xcd, tile = k % tiles;
    int bi = tile % nbi, bo = tile / nbi;  // GROUP: `tiles` counts groups and nbi groups along cin
    if constexpr (GROUP) {
        const int qb = group >> 4, qc = group & 15;
        if (wave >= qb * qc) return;  // (a group of 2 or 3 blocks leaves wave slots empty; nothing below synchronises the workgroup)
        bo = bo * qb + wave / qc;
        bi = bi * qc + wave % qc;
    }
    const int ci0 = bi * 64, co0 = bo * 64;
    const int64_t r_begin = (int64_t)chunk * rows_per_chunk;
    if (r_begin >= m_rows) return;  // padding of the XCD-aligned grid (whole workgroup)
    const int64_t r_end = r_begin + rows_per_chunk < m_rows ? r_begin + rows_per_chunk : m_rows;
    // Channel quads past cin / cout read quad 0 instead: they only feed output rows / columns that are never
    // stored, so no masking is needed.  Lane offsets are 32-bit and added to a wave-uniform row pointer.
    const bool a_ok = co0 + 4 * cq < cout, b_ok = ci0 + 4 * cq < cin;
    const uint32_t yoff = (uint32_t)((8 * rg * cout + (a_ok ? co0 + 4 * cq : 0)) * 4);
    const uint32_t xoff = (uint32_t)((8 * rg * cin + (b_ok ? ci0 + 4 * cq : 0)) * (XB ? 2 : 4));
    const bool want_db = want_bias && bi == 0;

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 db_acc = {0.f, 0.f, 0.f, 0.f};

    f32x4 xr[XB ? 1 : 8], yr[8];
    f32x4 xd[DEPTH > 1 ? DEPTH : 1][8], yd[DEPTH > 1 ? DEPTH : 1][8];  // the register sets of the deep form
    uint2 xq[XB ? 8 : 1];
    bf16x8 b_hi[4], b_lo[XB ? 1 : 4];
    auto load_rows = [&](const float* src, int ld, uint32_t off, int step, f32x4(&dst)[8]) {
        const char* base = reinterpret_cast<const char*>(src + (r_begin + 32 * (int64_t)step) * ld);  // wave-uniform
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = *reinterpret_cast<const f32x4*>(base + (size_t)i * ld * 4 + off);
    };
    auto load_x = [&](int step, auto& xs) {
        if constexpr (XB) {
            const char* base = static_cast<const char*>(x_v) + (r_begin + 32 * (int64_t)step) * cin * 2;
#pragma unroll
            for (int i = 0; i < 8; ++i) xq[i] = *reinterpret_cast<const uint2*>(base + (size_t)i * cin * 2 + xoff);
        } else {
            const char* base = reinterpret_cast<const char*>(x + (r_begin + 32 * (int64_t)step) * cin);
#pragma unroll
            for (int i = 0; i < 8; ++i) xs[i] = *reinterpret_cast<const f32x4*>(base + (size_t)i * cin * 4 + xoff);
        }
    };
    auto make_b = [&](auto& xs) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if constexpr (XB) {
                b_hi[b] = frag_from_bf16_rows(xq, b);
            } else {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = xs[i][b];
                split_frag(v, &b_hi[b], &b_lo[b]);
            }
        }
    };
    auto multiply = [&](auto& ys) {
        if (want_db) {
#pragma unroll
            for (int i = 0; i < 8; ++i) db_acc += ys[i];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = ys[i][a];
            bf16x8 a_hi, a_lo;
            split_frag(v, &a_hi, &a_lo);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if constexpr (XB) acc[a][b] = mfma2(a_hi, a_lo, b_hi[b], acc[a][b]);
                else acc[a][b] = mfma3(a_hi, a_lo, b_hi[b], b_lo[b], acc[a][b]);
            }
        }
    };
    // full 32-row steps: the x rows of the wave's next step are requested as soon as this step's are converted,
    // its dy rows right after their last use
    const int n_full = (int)((r_end - r_begin) / 32);
    constexpr int kStride = GROUP ? 1 : kWaves;  // GROUP: every wave walks all steps of the chunk
    int s = GROUP ? 0 : wave;
    if constexpr (DEPTH > 1) {
        static_assert(!XB && !GROUP, "the deep form is the fp32 split-K form's");
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (s + d * kStride < n_full) {
                load_x(s + d * kStride, xd[d]);
                load_rows(dy, cout, yoff, s + d * kStride, yd[d]);
            }
        for (; s < n_full; s += DEPTH * kStride) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int cur = s + d * kStride;
                if (cur < n_full) {  // (wave-uniform)
                    const bool more = cur + DEPTH * kStride < n_full;
                    make_b(xd[d]);
                    if (more) load_x(cur + DEPTH * kStride, xd[d]);
                    multiply(yd[d]);
                    if (more) load_rows(dy, cout, yoff, cur + DEPTH * kStride, yd[d]);
                }
            }
        }
    } else {
        if (s < n_full) {
            load_x(s, xr);
            load_rows(dy, cout, yoff, s, yr);
        }
        for (; s < n_full; s += kStride) {
            const bool more = s + kStride < n_full;
            make_b(xr);
            if (more) load_x(s + kStride, xr);
            multiply(yr);
            if (more) load_rows(dy, cout, yoff, s + kStride, yr);
        }
    }
    // the chunk's last, partial step (only the last chunk of the tensor has one): rows clamped and masked
    if ((r_end - r_begin) % 32 != 0 && (GROUP || n_full % kWaves == wave)) {
        const int64_t r0 = r_begin + 32 * (int64_t)n_full + 8 * rg;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool ok = r0 + i < r_end;
            const int64_t r = ok ? r0 + i : r_end - 1;
            const f32x4 vy = *reinterpret_cast<const f32x4*>(dy + r * cout + (a_ok ? co0 + 4 * cq : 0));
            if constexpr (XB) {
                const uint2 vx = *reinterpret_cast<const uint2*>(static_cast<const char*>(x_v) + (r * cin + (b_ok ? ci0 + 4 * cq : 0)) * 2);
                xq[i] = ok ? vx : make_uint2(0u, 0u);
            } else {
                const f32x4 vx = *reinterpret_cast<const f32x4*>(x + r * cin + (b_ok ? ci0 + 4 * cq : 0));
                xr[i] = ok ? vx : z;
            }
            yr[i] = ok ? vy : z;
        }
        make_b(xr);
        multiply(yr);
    }

    float* pw = part + (int64_t)chunk * ((int64_t)cout * cin + cout);
    if constexpr (GROUP) {
        // a wave stores its own block straight from the accumulators: acc[a][b][r] = dw[co = 4*(4g + r) + a][ci = 4*c16 + b]
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + 4 * (4 * rg + r) + a, ci = ci0 + 4 * cq;
                if (co < cout && ci < cin)
                    *reinterpret_cast<f32x4*>(pw + (int64_t)co * cin + ci) = (f32x4){acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
            }
        if (want_db) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = db_acc[j];
                v += __shfl_xor(v, 16, SEG3D_WAVE);
                v += __shfl_xor(v, 32, SEG3D_WAVE);
                if (rg == 0 && co0 + 4 * cq + j < cout) pw[(int64_t)cout * cin + co0 + 4 * cq + j] = v;
            }
        }
        return;
    }
    // ---- sum the waves' blocks in wave order: acc[a][b][r] = dw[co = 4*(4g + r) + a][ci = 4*c16 + b]
    for (int w = 0; w < kWaves; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f32x4* slot = reinterpret_cast<f32x4*>(&red[(4 * (4 * rg + r) + a) * 64 + 4 * cq]);
                    f32x4 v = {acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
                    if (w > 0) v += *slot;
                    *slot = v;
                }
        }
        __syncthreads();
    }
    // partial of this chunk: [cout][cin] block sums followed by [cout] column sums of dy
    for (int e = threadIdx.x; e < 64 * 16; e += kThreads) {
        const int row = e >> 4, q = e & 15;
        const int co = co0 + row, ci = ci0 + 4 * q;
        if (co < cout && ci < cin)
            *reinterpret_cast<f32x4*>(pw + (int64_t)co * cin + ci) = *reinterpret_cast<const f32x4*>(&red[row * 64 + 4 * q]);
    }
    if (want_db) {  // lanes cq, cq+16, cq+32, cq+48 hold the four row groups of the same channel quad
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = db_acc[j];
            v += __shfl_xor(v, 16, SEG3D_WAVE);
            v += __shfl_xor(v, 32, SEG3D_WAVE);
            if (rg == 0) red_b[wave][4 * cq + j] = v;
        }
        __syncthreads();
        if (threadIdx.x < 64 && co0 + threadIdx.x < cout) {
            float t = red_b[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) t += red_b[w][threadIdx.x];
            pw[(int64_t)cout * cin + co0 + threadIdx.x] = t;
        }
    }
}

// dw[i] / db[i - nw] = sum over chunks of part[c][i] in a fixed order: a workgroup owns 64 column quads (16-B loads, 1 KiB
// per wave instruction), its 4 chunk lanes take chunks q, q+4, .. with four independent accumulators (loads in flight
// instead of a serial chain) and are combined through LDS in lane order.  n and nw are multiples of 4.
__device__ __forceinline__ void reduce_block(const float* __restrict__ part, int chunks, int64_t n, int64_t nw,
                                             float* __restrict__ dw, float* __restrict__ db, int64_t block) {
    __shared__ float4 red[4][64];
    const int cq = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t i = (block * 64 + cq) * 4;
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        int c = q;
        for (; c + 12 < chunks; c += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)(c + 4 * u) * n + i);
                acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
            }
        }
        for (; c < chunks; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)c * n + i);
            acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
        }
    }
    float4 s;
    s.x = (acc[0].x + acc[1].x) + (acc[2].x + acc[3].x);
    s.y = (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y);
    s.z = (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z);
    s.w = (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w);
    red[q][cq] = s;
    __syncthreads();
    if (q == 0 && i < n) {
        float4 t = red[0][cq];
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float4 v = red[k][cq];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        if (i < nw) *reinterpret_cast<float4*>(dw + i) = t;
        else if (db) *reinterpret_cast<float4*>(db + (i - nw)) = t;
    }
}

__global__ __launch_bounds__(256) void wgrad_dense_reduce(const float* __restrict__ part, int chunks, int64_t n,
                                                          int64_t nw, float* __restrict__ dw,
                                                          float* __restrict__ db) {
    reduce_block(part, chunks, n, nw, dw, db, (int64_t)blockIdx.x);
}

// Many such reduces in ONE launch (all parameter-gradient partials of a backward pass whose join was deferred to the
// pass's end: ~160 launches of 5-7 us each otherwise).  jobs are sorted by first_block; a block finds its job by bisection.
struct ReduceJob {
    const float* part;
    float* dw;
    float* db;
    int64_t n, nw;
    int32_t chunks, reserved;
    int64_t first_block;
};
static_assert(sizeof(ReduceJob) == 56, "ReduceJob layout is part of the C ABI (seg3d_reduce_partials_batched)");

__global__ __launch_bounds__(256) void reduce_partials_batched(const ReduceJob* __restrict__ jobs, int n_jobs) {
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {  // last job with first_block <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int64_t)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const ReduceJob jb = jobs[lo];
    reduce_block(jb.part, jb.chunks, jb.n, jb.nw, jb.dw, jb.db, (int64_t)blockIdx.x - jb.first_block);
}

}  // namespace

// part[chunks][n] -> dw[0, nw) and db[0, n - nw) (db nullable); also used by the sparse wgrad (wgrad_split.hip)
int wgrad_chunk_reduce(const float* part, int chunks, int64_t n, int64_t nw, float* dw, float* db, hipStream_t st) {
    hipLaunchKernelGGL(wgrad_dense_reduce, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, st, part, chunks, n, nw, dw, db);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

extern "C" size_t seg3d_linear_wgrad_workspace_bytes(int64_t m, int32_t cin, int32_t cout) {
    if (m < 0 || cin <= 0 || cout <= 0) return 0;
    const Plan p = plan(m, cin, cout), q = plan(m, cin, cout, true);  // either kernel may run (bf16 rows take the first)
    return ((size_t)(p.chunks > q.chunks ? p.chunks : q.chunks) * ((size_t)cin * cout + cout) + 64) * sizeof(float);
}

// The two halves of seg3d_linear_wgrad apart: partial blocks now, their fixed-order sum later (alone, or batched with the
// other pending sums of a backward pass).  *chunks receives the number of partial blocks written (0 when m == 0).
// one launch of the partial-block kernel for plan p
static void launch_dense(const Plan& p, const void* x, bool x_bf16, const float* dy, int64_t m, int cin, int cout, float* part,
                         int want_bias, hipStream_t st) {
    if (p.lds) {
        wgrad_dense_lds_launch(static_cast<const float*>(x), dy, m, cin, cout, p.chunks, p.rows, part, want_bias, st);
        return;
    }
    if (p.group) {
        const int qb = p.group >> 4, qc = p.group & 15;
        const int nbi_g = p.nbi / qc, groups = (p.nbo / qb) * nbi_g;
        const unsigned blocks = (unsigned)((p.chunks + 7) / 8 * 8) * (unsigned)groups;
        if (x_bf16)
            hipLaunchKernelGGL((wgrad_dense_kernel<true, true>), dim3(blocks), dim3(kThreads), 0, st, x, dy, m, cin, cout, (int)p.rows,
                               nbi_g, groups, part, want_bias, p.group);
        else
            hipLaunchKernelGGL((wgrad_dense_kernel<false, true>), dim3(blocks), dim3(kThreads), 0, st, x, dy, m, cin, cout, (int)p.rows,
                               nbi_g, groups, part, want_bias, p.group);
        return;
    }
    const int tiles = p.nbo * p.nbi;
    const unsigned blocks = (unsigned)((p.chunks + 7) / 8 * 8) * (unsigned)tiles;
    // SEG3D_WGRAD_DENSE_DEPTH=3 (A/B, OFF: it lost): three steps of a wave in flight at ONE wave per SIMD (fp32 rows; 472
    // registers; four sets spill).  Measured (tools/wgrad_bench.py --partials, one box): the family 3.46 -> 4.88 ms (192 -> 192
    // @58 k 27.2 -> 37.1 us, 384 -> 768 @19 k 69.4 -> 91.7); with 256 / 512 workgroups 4.95 / 4.79 ms.  A second set at two
    // waves per SIMD does not fit (101 - 223 spilled registers).  The kernel needs its second wave per SIMD (the other wave's
    // conversions under this wave's MFMAs) more than it needs bytes in flight.
    static const int depth = [] {
        const char* e = getenv("SEG3D_WGRAD_DENSE_DEPTH");
        return (e && atoi(e) == 3) ? 3 : 1;
    }();
    if (!x_bf16 && depth == 3) {
        hipLaunchKernelGGL((wgrad_dense_kernel<false, false, 3>), dim3(blocks), dim3(kThreads), 0, st, x, dy, m, cin, cout, (int)p.rows,
                           p.nbi, tiles, part, want_bias, 0);
        return;
    }
    if (x_bf16)
        hipLaunchKernelGGL((wgrad_dense_kernel<true, false>), dim3(blocks), dim3(kThreads), 0, st, x, dy, m, cin, cout, (int)p.rows,
                           p.nbi, tiles, part, want_bias, 0);
    else
        hipLaunchKernelGGL((wgrad_dense_kernel<false, false>), dim3(blocks), dim3(kThreads), 0, st, x, dy, m, cin, cout, (int)p.rows,
                           p.nbi, tiles, part, want_bias, 0);
}

static int linear_wgrad_partials(const void* x, bool x_bf16, const float* dy, int64_t m, int32_t cin, int32_t cout,
                                 int32_t with_bias, void* workspace, size_t workspace_bytes, int32_t* chunks, void* stream) {
    if (m < 0 || cin <= 0 || cout <= 0 || (cin & 3) || (cout & 3) || !chunks) return SEG3D_EINVAL;
    if (m > 0 && (!x || !dy)) return SEG3D_EINVAL;
    if (workspace_bytes < seg3d_linear_wgrad_workspace_bytes(m, cin, cout) || (m > 0 && !workspace)) return SEG3D_EINVAL;
    *chunks = 0;
    if (m == 0) return SEG3D_OK;
    const Plan p = plan(m, cin, cout, !x_bf16);
    launch_dense(p, x, x_bf16, dy, m, cin, cout, static_cast<float*>(workspace), with_bias ? 1 : 0, as_stream(stream));
    SEG3D_CHECK_LAUNCH();
    *chunks = p.chunks;
    return SEG3D_OK;
}

extern "C" int seg3d_linear_wgrad_partials(const float* x, const float* dy, int64_t m, int32_t cin, int32_t cout,
                                           int32_t with_bias, void* workspace, size_t workspace_bytes, int32_t* chunks,
                                           void* stream) {
    return linear_wgrad_partials(x, false, dy, m, cin, cout, with_bias, workspace, workspace_bytes, chunks, stream);
}

extern "C" int seg3d_linear_wgrad_partials_xbf16(const uint16_t* x_bf16, const float* dy, int64_t m, int32_t cin, int32_t cout,
                                                 int32_t with_bias, void* workspace, size_t workspace_bytes, int32_t* chunks,
                                                 void* stream) {
    return linear_wgrad_partials(x_bf16, true, dy, m, cin, cout, with_bias, workspace, workspace_bytes, chunks, stream);
}

extern "C" int seg3d_reduce_partials(const float* part, int32_t chunks, int64_t n, int64_t nw, float* dw, float* db,
                                     void* stream) {
    if (chunks < 0 || n <= 0 || nw < 0 || nw > n || (n & 3) || (nw & 3) || !dw || (chunks > 0 && !part)) return SEG3D_EINVAL;
    return wgrad_chunk_reduce(part, chunks, n, nw, dw, db, as_stream(stream));
}

extern "C" int seg3d_reduce_partials_batched(const void* jobs, int32_t n_jobs, int64_t total_blocks, void* stream) {
    if (n_jobs < 0 || total_blocks < 0 || total_blocks > 0x7FFFFFFF || (n_jobs > 0 && !jobs)) return SEG3D_EINVAL;
    if (n_jobs == 0 || total_blocks == 0) return SEG3D_OK;
    hipLaunchKernelGGL(reduce_partials_batched, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream),
                       static_cast<const ReduceJob*>(jobs), (int)n_jobs);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

extern "C" int seg3d_linear_wgrad(const float* x, const float* dy, int64_t m, int32_t cin, int32_t cout, float* dw,
                                  float* db, void* workspace, size_t workspace_bytes, void* stream) {
    if (m < 0 || cin <= 0 || cout <= 0 || (cin & 3) || (cout & 3) || !dw) return SEG3D_EINVAL;
    if (m > 0 && (!x || !dy)) return SEG3D_EINVAL;
    if (workspace_bytes < seg3d_linear_wgrad_workspace_bytes(m, cin, cout) || (m > 0 && !workspace)) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    const Plan p = plan(m, cin, cout, true);
    float* part = static_cast<float*>(workspace);
    if (m > 0) {
        launch_dense(p, x, false, dy, m, cin, cout, part, db ? 1 : 0, st);
        SEG3D_CHECK_LAUNCH();
    }
    // without db the trailing [cout] columns of each partial are never written; their sums are discarded
    const int64_t nw = (int64_t)cin * cout;
    return wgrad_chunk_reduce(part, p.chunks, nw + cout, nw, dw, db, st);
}
