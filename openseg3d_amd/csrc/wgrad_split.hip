// Sparse-conv weight gradient (a9-a11 wgrad) in split-bf16 arithmetic, deterministic:
//   dw[co][k][ci] = sum_r x[nbr[k][r]][ci] * dy[r][co]          (rows with nbr[k][r] < 0 contribute nothing)
// 27 tall-skinny GEMMs over the valid (input row, output row) pairs of each kernel offset; the pair index is
// the MFMA K dimension of v_mfma_f32_16x16x32_bf16.
//
// A workgroup of 4 waves owns one 64 x 64 block of dw[:, k, :] over one chunk of output rows.  Its waves take
// the 64-row groups of the chunk round-robin, compact the valid pairs of the offset into a per-wave LDS ring
// (ballot + prefix popcount, output-row order kept) and consume them 32 at a time.  Operands go from global
// memory straight into MFMA fragments, as in wgrad_dense.hip: lane (cq, rg) loads pairs 8*rg .. 8*rg+7 of channels
// 4*cq .. 4*cq+3 (16-B loads; x rows gathered through the pair list), which are the 8 consecutive K values
// lane (c16 = cq, g = rg) of tile j needs for channel j of its quad.  No LDS image, no barrier in the main
// loop; the pair list is kept one step ahead so that the x rows of the next step are requested as soon as
// this step's are converted, and its dy rows right after their last use.
// Channel quads past cin / cout read quad 0 instead: they only feed output rows / columns that are never stored.
//
// No atomics: the four accumulators of a workgroup are summed through LDS in wave order, every (chunk, offset,
// block) writes its partial to a workspace and wgrad_chunk_reduce (wgrad_dense.hip) sums the chunks in a fixed
// order -- bit-reproducible, no memset of dw.
// Block order: id -> (xcd = id % 8, j = id / 8), unit = 8 * (j / tiles) + xcd = (chunk, offset), block = j % tiles:
// the workgroups that read the same gathered rows are dispatched back to back on one XCD and share its L2.
#include <cstdlib>

#include "attn_common.hpp"

int wgrad_chunk_reduce(const float* part, int chunks, int64_t n, int64_t nw, float* dw, float* db, hipStream_t st);

#ifdef SEG3D_WGRAD_STAMP
// In-kernel phase clocks of the wide kernel (tools/probes/wgrad_stamps.py; -DSEG3D_WGRAD_STAMP build only, never shipped):
// per wave s_memtime sums of [0] pair compaction + its barriers, [1] ring reads + gather issue, [2] fragment reads + MFMAs,
// [3] wait for the gathered rows + split + image store, [4] epilogue (store of the block), [5] steps, [6] whole wave.
__device__ unsigned long long* g_wgrad_stamp_buf = nullptr;
extern "C" int seg3d_debug_wgrad_stamps(void* buf) {
    unsigned long long* p = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : 2;
}
#define WSTAMP(i)                                                    \
    do {                                                             \
        __builtin_amdgcn_sched_barrier(0);                           \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
        __builtin_amdgcn_sched_barrier(0);                           \
        st_acc[i] += t_ - st_last;                                   \
        st_last = t_;                                                \
    } while (0)
#else
#define WSTAMP(i) do {} while (0)
#endif

namespace {

using namespace attn;

constexpr int kWaves = 4;
constexpr int kThreads = 64 * kWaves;
constexpr int kRing = 128;  // pairs per wave ring: at most 31 left over + 64 new

// XB variants (x rows stored as bf16: the opt-in copies a training forward saves for its backward, SEG3D_TRAIN_STORAGE=bf16):
// a bf16 row is its own high half -- 8-byte gathers, no split of x, two MFMAs per product.  dy is always fp32.
// bf16 fragment of channel `comp` (0..3) of a lane's quad from 8 rows of 4 bf16 each (q[i] = {ch0 | ch1 << 16, ch2 | ch3 << 16})
__device__ __forceinline__ bf16x8 frag_from_bf16_rows(const uint2* q, int comp) {
    u32x4 h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t a = (comp & 2) ? q[2 * j].y : q[2 * j].x, b = (comp & 2) ? q[2 * j + 1].y : q[2 * j + 1].x;
        h[j] = (comp & 1) ? ((a >> 16) | (b & 0xFFFF0000u)) : ((a & 0xFFFFu) | (b << 16));
    }
    return __builtin_bit_cast(bf16x8, h);
}

__device__ __forceinline__ f32x4 mfma2(const bf16x8& a_hi, const bf16x8& a_lo, const bf16x8& b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b, acc, 0, 0, 0);
    return acc;
}

struct Plan {
    int nbo, nbi;
    int chunks;
    int64_t rows;  // output rows per chunk (multiple of 64)
};

// Longest units first.  In a submanifold table the centre offset (k = 13) pairs EVERY row with itself while the other 26
// offsets hold a quarter of the rows or fewer: a centre unit is a workgroup four times as long as the rest, and in (chunk, k)
// order the last chunk's centre unit is dispatched in the launch's last round and ends long after everything else (level 4:
// 54 workgroups of ~100 steps among 1 404 of ~25 on 512 slots).  Units 0 .. chunks - 1 are therefore the centre units of all
// chunks, the other offsets follow chunk by chunk.  (Strided / inverse tables have no dominant offset: the order is harmless.)
__device__ __forceinline__ void unit_to_chunk_offset(int unit, int n_chunks, int* chunk, int* k) {
    if (unit < n_chunks) {
        *chunk = unit;
        *k = 13;
    } else {
        const int u = unit - n_chunks;
        const int kk = u % 26;
        *chunk = u / 26;
        *k = kk + (kk >= 13 ? 1 : 0);
    }
}

Plan plan(int64_t m, int cin, int cout) {
    Plan p{(cout + 63) / 64, (cin + 63) / 64, 0, 64};
    if (m <= 0) return p;
    const int tiles = p.nbo * p.nbi;
    // a wave should see >= ~8 steps (about a third of the rows of an offset are valid pairs): chunks of ~3000
    // rows; at least ~1000 workgroups per launch
    static const int rows_target = [] {
        const char* e = getenv("SEG3D_WGRAD_SPARSE_ROWS");
        const int v = e ? atoi(e) : 3072;
        return v >= 64 ? v : 3072;  // 0 / non-numeric / tiny values: the default (never a division by zero)
    }();
    int64_t chunks = m / rows_target;
    const int64_t need = (1024 + 27 * tiles - 1) / (27 * tiles);
    if (chunks < need) chunks = need;
    if (chunks < 1) chunks = 1;
    int64_t rows = (m + chunks - 1) / chunks;
    rows = (rows + 63) / 64 * 64;
    p.rows = rows;
    p.chunks = (int)((m + rows - 1) / rows);
    return p;
}

template <bool XB>
__global__ __launch_bounds__(kThreads, 2) void wgrad_sparse_kernel(const void* __restrict__ x_v, const float* __restrict__ dy,
                                                                    const int32_t* __restrict__ nbr, int64_t m_rows,
                                                                    int cin, int cout, int rows_per_chunk, int nbi,
                                                                    int tiles, int units, float* __restrict__ part,
                                                                    int center_first) {
    const float* x = static_cast<const float*>(x_v);
    __shared__ __attribute__((aligned(16))) float red[64 * 64];  // block sum [co_local][ci_local]
    __shared__ int2 ring[kWaves][kRing];                         // (input row, output row) pairs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cq = lane & 15, rg = lane >> 4;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    // units are dealt to the XCDs in blocks of `xcd_block` consecutive units (center_first >> 8): the offsets of one row chunk
    // gather the same dy rows and overlapping x rows -- on ONE L2 when they are neighbours in the block
    const int xb = (center_first >> 8) > 0 ? (center_first >> 8) : 1;
    const int qs = j / tiles, tile = j % tiles;
    const int unit = ((qs / xb) * 8 + xcd) * xb + qs % xb;
    if (unit >= units) return;  // padding of the XCD-aligned grid (whole workgroup)
    int chunk = unit / 27, k = unit % 27;
    if (center_first & 1) unit_to_chunk_offset(unit, units / 27, &chunk, &k);
    const int bi = tile % nbi, bo = tile / nbi;
    const int ci0 = bi * 64, co0 = bo * 64;
    const int64_t r_begin = (int64_t)chunk * rows_per_chunk;
    const int64_t r_end = r_begin + rows_per_chunk < m_rows ? r_begin + rows_per_chunk : m_rows;
    const int32_t* nk = nbr + (int64_t)k * m_rows;
    const float* px = x + (ci0 + 4 * cq < cin ? ci0 + 4 * cq : 0);
    const uint16_t* pxb = static_cast<const uint16_t*>(x_v) + (ci0 + 4 * cq < cin ? ci0 + 4 * cq : 0);
    const float* py = dy + (co0 + 4 * cq < cout ? co0 + 4 * cq : 0);
    int2* my = ring[wave];

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- pair compaction: the table entries of the wave's next kAhead 64-row groups are always in flight (idxq): a sparse
    // table (strided / inverse levels: 3 - 10 of 27 entries valid) needs several groups per 32-pair step, and with one group
    // of look-ahead every one of them but the first paid a full load latency
    constexpr int kAhead = 4;
    int head = 0, tail = 0;  // wave-uniform ring cursors (monotonic, masked on use)
    int64_t g_row = r_begin + 64 * (int64_t)wave;
    int32_t idxq[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
        const int64_t r = g_row + (int64_t)j * 64 * kWaves + lane;
        idxq[j] = r < r_end ? nk[r] : -1;
    }
    auto scan_group = [&]() {
        const int32_t idx = idxq[0];
        const int64_t row = g_row + lane;
        g_row += 64 * kWaves;
#pragma unroll
        for (int j = 0; j + 1 < kAhead; ++j) idxq[j] = idxq[j + 1];
        const int64_t rn = g_row + (int64_t)(kAhead - 1) * 64 * kWaves + lane;
        idxq[kAhead - 1] = rn < r_end ? nk[rn] : -1;
        const unsigned long long mask = __ballot(idx >= 0);
        if (idx >= 0) {
            const int p = tail + __popcll(mask & ((1ull << lane) - 1ull));
            my[p & (kRing - 1)] = make_int2(idx, (int)row);
        }
        tail += __popcll(mask);
    };
    auto refill = [&]() {
        while (tail - head < 32 && g_row < r_end) scan_group();
        __builtin_amdgcn_wave_barrier();
    };

    f32x4 xr[XB ? 1 : 8], yr[8];
    uint2 xq[XB ? 8 : 1];
    bf16x8 b_hi[4], b_lo[XB ? 1 : 4];
    // rows of the step that starts at ring position h0; slots >= valid are zero
    auto load_x = [&](int h0, int valid) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = 8 * rg + i;
            const int2 p = my[(h0 + (e < valid ? e : 0)) & (kRing - 1)];
            if constexpr (XB) xq[i] = *reinterpret_cast<const uint2*>(pxb + (int64_t)p.x * cin);
            else xr[i] = *reinterpret_cast<const f32x4*>(px + (int64_t)p.x * cin);
        }
        if (valid < 32) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (8 * rg + i >= valid) {
                    if constexpr (XB) xq[i] = make_uint2(0u, 0u);
                    else xr[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
        }
    };
    auto load_y = [&](int h0, int valid) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = 8 * rg + i;
            const int2 p = my[(h0 + (e < valid ? e : 0)) & (kRing - 1)];
            yr[i] = *reinterpret_cast<const f32x4*>(py + (int64_t)p.y * cout);
        }
        if (valid < 32) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (8 * rg + i >= valid) yr[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto make_b = [&]() {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if constexpr (XB) {
                b_hi[b] = frag_from_bf16_rows(xq, b);
            } else {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = xr[i][b];
                split_frag(v, &b_hi[b], &b_lo[b]);
            }
        }
    };
    auto multiply = [&]() {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = yr[i][a];
            bf16x8 a_hi, a_lo;
            split_frag(v, &a_hi, &a_lo);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if constexpr (XB) acc[a][b] = mfma2(a_hi, a_lo, b_hi[b], acc[a][b]);
                else acc[a][b] = mfma3(a_hi, a_lo, b_hi[b], b_lo[b], acc[a][b]);
            }
        }
    };

    refill();
    int valid = tail - head < 32 ? tail - head : 32;
    if (valid > 0) {
        load_x(head, valid);
        load_y(head, valid);
    }
    while (valid > 0) {
        head += valid;
        refill();  // the ring entries of the step in flight are already in registers
        const int next = tail - head < 32 ? tail - head : 32;
        make_b();
        if (next > 0) load_x(head, next);
        multiply();
        if (next > 0) load_y(head, next);
        valid = next;
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
    float* pw = part + (int64_t)chunk * ((int64_t)cout * 27 * cin);
    for (int e = threadIdx.x; e < 64 * 16; e += kThreads) {
        const int row = e >> 4, q = e & 15;
        const int co = co0 + row, ci = ci0 + 4 * q;
        if (co < cout && ci < cin)
            *reinterpret_cast<f32x4*>(pw + ((int64_t)co * 27 + k) * cin + ci) =
                *reinterpret_cast<const f32x4*>(&red[row * 64 + 4 * q]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Wide channels (C >= 128): the 64 x 64 kernel above re-reads its gathered rows once per block and is bound by that
// L2 -> L1 stream.  Here a workgroup of 2 x 2 waves owns a 128 x 128 block of dw[:, k, :]: per 32-pair step wave w
// loads ONE 64-channel slab (waves 0,1: dy channels co0.., co0+64..; waves 2,3: x channels ci0.., ci0+64..) with the
// same 8-pairs x 4-channels-per-lane pattern, converts it to bf16 hi/lo records and writes them to a double-buffered
// LDS image [slab][hi|lo][row group][64 records] (record (c%4)*16 + c/4 of row group g = channel c, pairs 8g..8g+7,
// conflict-free 16-B accesses both ways); after one barrier wave (wa, wb) reads slab wa as A and slab 2+wb as B and
// issues its 48 MFMAs.  Half the operand traffic and half the conversions per MFMA.  The pair list is compacted once
// per workgroup: each wave ballots 64 of 256 rows, wave totals are exchanged through LDS.  The slab loads of step
// s+1 are issued before the MFMAs of step s.
constexpr int kRing2 = 512;  // pairs: at most 31 left over + 256 new

// SB (round 5 experiment): ONE image buffer instead of two and three workgroups per CU instead of two -- a second barrier per
// step (all waves must have read the image before the next step's rows overwrite it) against half more resident waves to
// hide the gathers' round trips behind (the stamps say 44 % of a step is waiting for rows).
template <bool XB, int DEPTH, bool SB = false>
__global__ __launch_bounds__(kThreads, SB ? 3 : 2) void wgrad_sparse_wide_kernel(const void* __restrict__ x_v, const float* __restrict__ dy,
                                                                         const int32_t* __restrict__ nbr, int64_t m_rows,
                                                                         int cin, int cout, int rows_per_chunk, int nbi,
                                                                         int tiles, int units, float* __restrict__ part,
                                                                         int center_first) {
    const float* x = static_cast<const float*>(x_v);
    __shared__ __attribute__((aligned(16))) uint4 img[SB ? 1 : 2][4][2][4][64];  // [buffer][slab][hi|lo][row group][record] 64 KiB
    __shared__ int2 ring[kRing2];
    __shared__ int wave_cnt[2][kWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cq = lane & 15, rg = lane >> 4;
    const int wa = wave >> 1, wb = wave & 1;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    // units are dealt to the XCDs in blocks of `xcd_block` consecutive units (center_first >> 8): the offsets of one row chunk
    // gather the same dy rows and overlapping x rows -- on ONE L2 when they are neighbours in the block
    const int xb = (center_first >> 8) > 0 ? (center_first >> 8) : 1;
    const int qs = j / tiles, tile = j % tiles;
    const int unit = ((qs / xb) * 8 + xcd) * xb + qs % xb;
    if (unit >= units) return;  // padding of the XCD-aligned grid (whole workgroup)
    int chunk = unit / 27, k = unit % 27;
    if (center_first & 1) unit_to_chunk_offset(unit, units / 27, &chunk, &k);
    const int bi = tile % nbi, bo = tile / nbi;
    const int ci0 = bi * 128, co0 = bo * 128;
    const int64_t r_begin = (int64_t)chunk * rows_per_chunk;
    const int64_t r_end = r_begin + rows_per_chunk < m_rows ? r_begin + rows_per_chunk : m_rows;
    const int32_t* nk = nbr + (int64_t)k * m_rows;
    // this wave's slab: waves 0,1 stage dy (rows = output rows), waves 2,3 stage x (rows = gathered input rows)
    const bool stage_dy = wave < 2;
    const int ld = stage_dy ? cout : cin;
    const int ch = (stage_dy ? co0 : ci0) + 64 * (wave & 1) + 4 * cq;
    const float* src = (stage_dy ? dy : x) + (ch < ld ? ch : 0);
    const uint16_t* srcb = static_cast<const uint16_t*>(x_v) + (ch < ld ? ch : 0);  // (XB: the x slabs of waves 2, 3)

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- pair compaction, 256 rows per round (64 per wave); cursors are identical in every wave
    int head = 0, tail = 0, round = 0;
    int64_t g_row = r_begin;
    // table entries of the next two rounds are in flight while this round is compacted
    int32_t idx_n1 = g_row + 64 * wave + lane < r_end ? nk[g_row + 64 * wave + lane] : -1;
    int32_t idx_n2 = g_row + 64 * (kWaves + wave) + lane < r_end ? nk[g_row + 64 * (kWaves + wave) + lane] : -1;
    auto scan_round = [&]() {
        const int64_t row = g_row + 64 * wave + lane;
        const int32_t idx = idx_n1;
        idx_n1 = idx_n2;
        {
            const int64_t rn = row + 2 * 64 * kWaves;
            idx_n2 = rn < r_end ? nk[rn] : -1;
        }
        const unsigned long long mask = __ballot(idx >= 0);
        const int mine = __popcll(mask);
        int* cnt = wave_cnt[round & 1];
        if (lane == 0) cnt[wave] = mine;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const int c = cnt[w];
            before += w < wave ? c : 0;
            total += c;
        }
        if (idx >= 0) {
            const int p = tail + before + __popcll(mask & ((1ull << lane) - 1ull));
            ring[p & (kRing2 - 1)] = make_int2(idx, (int)row);
        }
        tail += total;
        g_row += 64 * kWaves;
        ++round;
    };
    auto refill = [&]() {  // every wave takes the same branches
        while (tail - head < 32 && g_row < r_end) scan_round();
        __syncthreads();  // ring writes visible
    };

    f32x4 raw[8], raw2[DEPTH == 2 ? 8 : 1];
    uint2 rawb[XB ? 8 : 1], rawb2[(XB && DEPTH == 2) ? 8 : 1];
    // (the row format is a wave-uniform choice made ONCE per call, outside the gather loops: a branch per gathered row puts
    // every load into a basic block of its own and the waits at the block boundaries serialise the eight gathers --
    // measured: the bf16 variant ran 1.7 x SLOWER than the fp32 one that way)
    auto load_into = [&](f32x4* r, uint2* rb, int h0, int valid) {
        int32_t rows[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = 8 * rg + i;
            const int2 p = ring[(h0 + (e < valid ? e : 0)) & (kRing2 - 1)];
            rows[i] = stage_dy ? p.y : p.x;
        }
        if (XB && !stage_dy) {
            if constexpr (XB) {
#pragma unroll
                for (int i = 0; i < 8; ++i) rb[i] = *reinterpret_cast<const uint2*>(srcb + (int64_t)rows[i] * ld);
                if (valid < 32) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (8 * rg + i >= valid) rb[i] = make_uint2(0u, 0u);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = *reinterpret_cast<const f32x4*>(src + (int64_t)rows[i] * ld);
            if (valid < 32) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (8 * rg + i >= valid) r[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto stash_from = [&](const f32x4* r, const uint2* rb, int buf) {
        if (XB && !stage_dy) {  // bf16 rows: the record is the row's own bits, there is no low plane
            if constexpr (XB) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    img[buf][wave][0][rg][jj * 16 + cq] = __builtin_bit_cast(uint4, frag_from_bf16_rows(rb, jj));
            }
            return;
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = r[i][jj];
            bf16x8 hi, lo;
            split_frag(v, &hi, &lo);
            img[buf][wave][0][rg][jj * 16 + cq] = __builtin_bit_cast(uint4, hi);
            img[buf][wave][1][rg][jj * 16 + cq] = __builtin_bit_cast(uint4, lo);
        }
    };
    auto load_slab = [&](int h0, int valid) { load_into(raw, rawb, h0, valid); };
    auto stash = [&](int buf) { stash_from(raw, rawb, buf); };
    auto multiply = [&](int buf) {
        bf16x8 b_hi[4], b_lo[XB ? 1 : 4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            b_hi[b] = __builtin_bit_cast(bf16x8, img[buf][2 + wb][0][rg][b * 16 + cq]);
            if constexpr (!XB) b_lo[b] = __builtin_bit_cast(bf16x8, img[buf][2 + wb][1][rg][b * 16 + cq]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const bf16x8 a_hi = __builtin_bit_cast(bf16x8, img[buf][wa][0][rg][a * 16 + cq]);
            const bf16x8 a_lo = __builtin_bit_cast(bf16x8, img[buf][wa][1][rg][a * 16 + cq]);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if constexpr (XB) acc[a][b] = mfma2(a_hi, a_lo, b_hi[b], acc[a][b]);
                else acc[a][b] = mfma3(a_hi, a_lo, b_hi[b], b_lo[b], acc[a][b]);
            }
        }
    };

    int buf = 0;
#ifdef SEG3D_WGRAD_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_last;
#endif
    if constexpr (DEPTH == 1) {
        refill();
        int valid = tail - head < 32 ? tail - head : 32;
        if (valid > 0) {
            load_slab(head, valid);
            stash(0);
        }
        WSTAMP(7);
        while (valid > 0) {
            head += valid;
            refill();  // also the barrier that publishes image `buf`
            WSTAMP(0);
            const int next = tail - head < 32 ? tail - head : 32;
            if (next > 0) load_slab(head, next);
            WSTAMP(1);
            multiply(buf);
            WSTAMP(2);
            if constexpr (SB) {
                __syncthreads();  // every wave has read the image: the next step's rows may overwrite it
                if (next > 0) stash(0);
            } else {
                if (next > 0) stash(buf ^ 1);
                buf ^= 1;
            }
            WSTAMP(3);
            valid = next;
#ifdef SEG3D_WGRAD_STAMP
            st_acc[5] += 1;
#endif
        }
    } else {
        // Two steps of gathers in flight: the rows of step s+2 are requested before the MFMAs of step s and converted after
        // the MFMAs of step s+1 (two register sets, roles alternate), so a gather has two multiplies to arrive instead of one.
        // head = ring position of the next step to be requested; fetch() is executed by every wave alike (barriers inside).
        auto fetch = [&](f32x4* r, uint2* rb) {
            refill();  // (>= 32 pairs ahead or the table exhausted) + the barrier that publishes the image stashed last
            const int v = tail - head < 32 ? tail - head : 32;
            if (v > 0) load_into(r, rb, head, v);
            head += v;
            return v;
        };
        int va = fetch(raw, rawb);                      // step 0
        int vb = va > 0 ? fetch(raw2, rawb2) : 0;       // step 1
        if (va > 0) stash_from(raw, rawb, 0);
        while (va > 0) {
            // image `buf` = step s (from set A), set B holds the rows of step s+1
            va = vb > 0 ? fetch(raw, rawb) : (refill(), 0);    // step s+2 into set A; its barrier publishes image `buf`
            multiply(buf);
            if (vb > 0) stash_from(raw2, rawb2, buf ^ 1);
            buf ^= 1;
            if (vb == 0) break;
            // image `buf` = step s+1 (from set B), set A holds the rows of step s+2
            vb = va > 0 ? fetch(raw2, rawb2) : (refill(), 0);  // step s+3 into set B
            multiply(buf);
            if (va > 0) stash_from(raw, rawb, buf ^ 1);
            buf ^= 1;
        }
    }
    __syncthreads();  // all reads of the images done: reuse them as the store staging area

    // ---- store: acc[a][b][r] = dw[co = 4*(4g + r) + a][ci = 4*c16 + b] of this wave's 64 x 64 sub-block
    if constexpr (SB) {  // (the single image is too small to stage four blocks: 16-byte stores straight from the accumulators)
        float* pw = part + (int64_t)chunk * ((int64_t)cout * 27 * cin);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + 64 * wa + 4 * (4 * rg + r) + a, ci = ci0 + 64 * wb + 4 * cq;
                if (co < cout && ci < cin)
                    *reinterpret_cast<f32x4*>(pw + ((int64_t)co * 27 + k) * cin + ci) =
                        (f32x4){acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
            }
        return;
    }
    float* st = reinterpret_cast<float*>(&img[0][0][0][0][0]) + wave * 4096;  // 16 KiB per wave
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<f32x4*>(&st[(4 * (4 * rg + r) + a) * 64 + 4 * cq]) =
                (f32x4){acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
    __builtin_amdgcn_wave_barrier();
    float* pw = part + (int64_t)chunk * ((int64_t)cout * 27 * cin);
    for (int e = lane; e < 64 * 16; e += 64) {
        const int row = e >> 4, q = e & 15;
        const int co = co0 + 64 * wa + row, ci = ci0 + 64 * wb + 4 * q;
        if (co < cout && ci < cin)
            *reinterpret_cast<f32x4*>(pw + ((int64_t)co * 27 + k) * cin + ci) =
                *reinterpret_cast<const f32x4*>(&st[row * 64 + 4 * q]);
    }
#ifdef SEG3D_WGRAD_STAMP
    if constexpr (DEPTH == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WSTAMP(4);
        if (g_wgrad_stamp_buf && lane == 0) {
            st_acc[6] = __builtin_amdgcn_s_memtime() - st_begin;
            unsigned long long* o = g_wgrad_stamp_buf + (k == 13 ? 8 : 0);  // centre units apart from the rest
            for (int i = 0; i < 8; ++i) atomicAdd(o + i, st_acc[i]);
            atomicAdd(g_wgrad_stamp_buf + 16 + (k == 13 ? 1 : 0), 1ull);  // waves counted
        }
    }
#endif
}

}  // namespace

// used by seg3d_spconv_wgrad (spconv.hip)
size_t wgrad_split_sparse_workspace_bytes(int64_t m_out, int cin, int cout) {
    const Plan p = plan(m_out, cin, cout);
    return ((size_t)p.chunks * 27 * (size_t)cin * cout + 64) * sizeof(float);
}

// dw == nullptr: the partial blocks only (their fixed-order sum is queued by the caller); *chunks_out = their count
int wgrad_split_sparse(const void* x, const float* dy, const int32_t* nbr, int64_t m_out, int cin, int cout, float* dw,
                       void* workspace, size_t workspace_bytes, hipStream_t st, int32_t* chunks_out, bool x_bf16) {
    if (workspace_bytes < wgrad_split_sparse_workspace_bytes(m_out, cin, cout) || !workspace) return SEG3D_EINVAL;
    const Plan p = plan(m_out, cin, cout);
    float* part = static_cast<float*>(workspace);
    const int units = p.chunks * 27;
    static const bool narrow_only = getenv("SEG3D_WGRAD_NARROW") != nullptr;  // A/B switch for profiling
    // SEG3D_WGRAD_CENTER_FIRST (A/B): 0 = units in (chunk, offset) order (round 4)
    static const int center_first = [] {
        const char* e = getenv("SEG3D_WGRAD_CENTER_FIRST");
        const int cf = (e && atoi(e) == 0) ? 0 : 1;
        // SEG3D_WGRAD_XCD_BLOCK (A/B): consecutive units per XCD (1 = round-robin, round 4's order)
        const char* b = getenv("SEG3D_WGRAD_XCD_BLOCK");
        int xb = b ? atoi(b) : 1;
        if (xb < 1 || xb > 64) xb = 1;
        return cf | (xb << 8);
    }();
    const int xcd_block = center_first >> 8;
    const int unit_slots = (units + 8 * xcd_block - 1) / (8 * xcd_block) * (8 * xcd_block);  // grid padding: whole blocks on every XCD
    // 128-wide blocks only where they add no padding (C = 256, 384, 768; not 192 = 1.5 blocks)
    const bool fits128 = ((cin + 127) / 128) * 2 == (cin + 63) / 64 && ((cout + 127) / 128) * 2 == (cout + 63) / 64;
    // ... and, padded, on the rectangular wide layers (384 <-> 192, 192 <-> 96: the strided levels' convs and the 2C -> C
    // bottlenecks): there the compaction done once per 128 x 128 block and the halved operand re-reads outweigh the padded
    // MFMAs (384 -> 192 inverse 274 -> 196 us, 192 -> 96 inverse 190 -> 151 us, 384 -> 192 submanifold 766 -> 719 us); square
    // 192 x 192 (78 % padding on MFMA-heavy submanifold tables) loses: 395 -> 492 us.  SEG3D_WGRAD_WIDE_MIN=n (A/B) pads every
    // layer with both widths >= n.
    static const int wide_min = [] {
        const char* e = getenv("SEG3D_WGRAD_WIDE_MIN");
        return e ? atoi(e) : 0;
    }();
    const int cmin = cin < cout ? cin : cout, cmax = cin < cout ? cout : cin;
    const bool wide_padded = (wide_min > 0 && cmin >= wide_min) || (cin != cout && cmin >= 96 && cmax >= 192);
    if ((fits128 || wide_padded) && !narrow_only) {
        const int nbo = (cout + 127) / 128, nbi = (cin + 127) / 128, tiles = nbo * nbi;
        const unsigned blocks = (unsigned)unit_slots * (unsigned)tiles;
        // SEG3D_WGRAD_DEPTH (A/B): gather steps in flight per wave (1 or 2)
        static const int depth = [] {
            const char* e = getenv("SEG3D_WGRAD_DEPTH");
            const int v = e ? atoi(e) : 1;
            return v == 2 ? 2 : 1;
        }();
#define SEG3D_LAUNCH_WIDE(XB_, D_)                                                                                             \
    hipLaunchKernelGGL((wgrad_sparse_wide_kernel<XB_, D_>), dim3(blocks), dim3(kThreads), 0, st, x, dy, nbr, m_out, cin, cout, \
                       (int)p.rows, nbi, tiles, units, part, center_first)
        // SEG3D_WGRAD_SB (A/B): 0 = round 4's double-buffered image at two workgroups per CU.  Default 1 (round 5): ONE image
        // buffer (37 KB of LDS, 162 VGPRs) and THREE workgroups per CU -- the kernel waits for gathered rows 44 % of the time, and
        // half more resident waves hide more of it than the second barrier per step costs: the 11 wide layers of the headline
        // step 4.95 -> 4.76 ms alone (inverse tables -11 %, 768 -> 384 -6 %), training step 42.95 -> 42.77 ms, same bits.
        static const bool single = [] {
            const char* e = getenv("SEG3D_WGRAD_SB");
            return !(e && atoi(e) == 0);
        }();
#define SEG3D_LAUNCH_WIDE_SB(XB_)                                                                                                   \
    hipLaunchKernelGGL((wgrad_sparse_wide_kernel<XB_, 1, true>), dim3(blocks), dim3(kThreads), 0, st, x, dy, nbr, m_out, cin, cout, \
                       (int)p.rows, nbi, tiles, units, part, center_first)
        if (single && depth == 1) {
            if (x_bf16) SEG3D_LAUNCH_WIDE_SB(true);
            else SEG3D_LAUNCH_WIDE_SB(false);
        } else if (x_bf16) {  // (two register sets of both row formats do not fit: 464 bytes of scratch at depth 2)
            SEG3D_LAUNCH_WIDE(true, 1);
        } else {
            if (depth == 2) SEG3D_LAUNCH_WIDE(false, 2);
            else SEG3D_LAUNCH_WIDE(false, 1);
        }
#undef SEG3D_LAUNCH_WIDE_SB
#undef SEG3D_LAUNCH_WIDE
    } else {
        const int tiles = p.nbo * p.nbi;
        const unsigned blocks = (unsigned)unit_slots * (unsigned)tiles;
        if (x_bf16)
            hipLaunchKernelGGL(wgrad_sparse_kernel<true>, dim3(blocks), dim3(kThreads), 0, st, x, dy, nbr, m_out, cin, cout,
                               (int)p.rows, p.nbi, tiles, units, part, center_first);
        else
            hipLaunchKernelGGL(wgrad_sparse_kernel<false>, dim3(blocks), dim3(kThreads), 0, st, x, dy, nbr, m_out, cin, cout,
                               (int)p.rows, p.nbi, tiles, units, part, center_first);
    }
    SEG3D_CHECK_LAUNCH();
    if (chunks_out) *chunks_out = p.chunks;
    if (!dw) return SEG3D_OK;
    const int64_t n = (int64_t)27 * cin * cout;
    return wgrad_chunk_reduce(part, p.chunks, n, n, dw, nullptr, st);
}
