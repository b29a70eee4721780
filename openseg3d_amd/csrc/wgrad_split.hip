// Weight gradients in split-bf16 arithmetic: sparse conv (a9-a11 wgrad); the dense Linear case of the same
// kernel body (SPARSE = false) is superseded by wgrad_dense.hip and no longer instantiated.
//   sparse:  dw[co][k][ci] = sum_r x[nbr[k][r]][ci] * dy[r][co]
//   dense :  dw[co][ci]    = sum_r x[r][ci]         * dy[r][co]
// Both are "tall-skinny" GEMMs: tiny outputs (C x C), reduction over 1e4..1e5 rows.  rocBLAS/hipBLASLt picks
// 32x32 macro-tiles for them (250 us per Linear layer); the exact-fp32 MFMA version of this kernel was bound
// by the fp32 matrix pipe.  Here the row index is the MFMA K dimension of v_mfma_f32_16x16x32_bf16.
//
// One wave owns a 64 x 64 channel block (co x ci, masked at the edges) of one kernel offset over a chunk of rows.
// Per 32-row step every lane loads 8 rows x 4 channels of x and of dy (16-B loads along the channel
// axis, rows gathered through the compacted pair list in the sparse case), converts to bf16 hi/lo, packs the
// 8 rows of one channel into one 16-B record and writes it to a per-wave LDS image [hi|lo][row group][64
// channel records] -- i.e. the transpose happens in registers, LDS writes and reads are both conflict-free
// 16-B accesses -- then reads the A (dy) and B (x) fragments back and issues NA*NB*3 MFMAs.
// Channel c of a block sits in record (c%4)*16 + c/4, so MFMA tile t holds channels {4*i + t}: always 4 x 4
// tiles per block; a block narrower than 64 channels just has fewer live rows per tile.
// The finished block goes through LDS once more and is added to dw in whole contiguous rows (float
// atomics at full rate); partial sums over row chunks are the only cross-wave traffic.
#include "attn_common.hpp"

namespace {

using namespace attn;

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;

template <bool SPARSE>
__global__ __launch_bounds__(kThreads) void wgrad_split_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               const int32_t* __restrict__ nbr, int64_t m_rows, int cin,
                                                               int cout, int rows_per_wave, int kk, float* __restrict__ dw,
                                                               float* __restrict__ db) {
    // per wave: operand images 2 x [2 hi/lo][4 row groups][64 records] x 16 B = 16 KiB, pair queue 1 KiB
    __shared__ __attribute__((aligned(16))) uint4 img[kWaves][2][2][4][64];
    __shared__ int32_t q_in[kWaves][128];
    __shared__ int32_t q_out[kWaves][128];

    constexpr int NA = 4, NB = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cq = lane & 15, rg = lane >> 4;  // load role: channel quad, row group (8 rows)
    const int c16 = lane & 15, g = lane >> 4;  // MFMA role
    const int k = SPARSE ? blockIdx.y : 0;
    const int nbi = (cin + 16 * NB - 1) / (16 * NB);
    const int bi = blockIdx.z % nbi, bo = blockIdx.z / nbi;
    const int ci0 = bi * 16 * NB, co0 = bo * 16 * NA;
    const int64_t r_begin = ((int64_t)blockIdx.x * kWaves + wave) * rows_per_wave;
    if (r_begin >= m_rows) return;
    const int64_t r_end = r_begin + rows_per_wave < m_rows ? r_begin + rows_per_wave : m_rows;
    const bool a_ok = 4 * cq < 16 * NA && co0 + 4 * cq < cout;  // this lane's dy channel quad exists
    const bool b_ok = 4 * cq < 16 * NB && ci0 + 4 * cq < cin;

    f32x4 acc[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint4(*my)[2][4][64] = img[wave];  // [operand][hi/lo][row group][record]
    // bias gradient (dense layers): column sums of dy, taken by the waves of the first ci block only
    const bool want_db = !SPARSE && db != nullptr && bi == 0;
    f32x4 db_acc = {0.f, 0.f, 0.f, 0.f};

    // one 32-pair step: rows come from (in_of(e), out_of(e)), e = 0..31; entries >= valid are zero
    auto step = [&](auto in_of, auto out_of, int valid) {
        f32x4 xa[8], ya[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = 8 * rg + i;
            const bool ok = e < valid;
            const int64_t ri = ok ? in_of(e) : 0, ro = ok ? out_of(e) : 0;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 lx = *reinterpret_cast<const f32x4*>(x + ri * cin + (b_ok ? ci0 + 4 * cq : 0));
            const f32x4 ly = *reinterpret_cast<const f32x4*>(dy + ro * cout + (a_ok ? co0 + 4 * cq : 0));
            xa[i] = ok && b_ok ? lx : z;
            ya[i] = ok && a_ok ? ly : z;
            if (want_db) db_acc += ya[i];
        }
        // transpose in registers: channel j of this lane's quad, 8 rows -> one 16-B record
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float vx[8], vy[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                vx[i] = xa[i][j];
                vy[i] = ya[i][j];
            }
            bf16x8 hi, lo;
            split_frag(vy, &hi, &lo);
            my[0][0][rg][j * 16 + cq] = __builtin_bit_cast(uint4, hi);
            my[0][1][rg][j * 16 + cq] = __builtin_bit_cast(uint4, lo);
            split_frag(vx, &hi, &lo);
            my[1][0][rg][j * 16 + cq] = __builtin_bit_cast(uint4, hi);
            my[1][1][rg][j * 16 + cq] = __builtin_bit_cast(uint4, lo);
        }
        __builtin_amdgcn_wave_barrier();
        bf16x8 b_hi[NB], b_lo[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            b_hi[b] = __builtin_bit_cast(bf16x8, my[1][0][g][b * 16 + c16]);
            b_lo[b] = __builtin_bit_cast(bf16x8, my[1][1][g][b * 16 + c16]);
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const bf16x8 a_hi = __builtin_bit_cast(bf16x8, my[0][0][g][a * 16 + c16]);
            const bf16x8 a_lo = __builtin_bit_cast(bf16x8, my[0][1][g][a * 16 + c16]);
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = mfma3(a_hi, a_lo, b_hi[b], b_lo[b], acc[a][b]);
        }
        __builtin_amdgcn_wave_barrier();
    };

    if (SPARSE) {
        const int32_t* nk = nbr + (int64_t)k * m_rows;
        int head = 0, tail = 0;  // wave-uniform queue cursors (monotonic, masked on use)
        for (int64_t r0 = r_begin; r0 < r_end; r0 += 64) {
            const int64_t r = r0 + lane;
            const int32_t idx = r < r_end ? nk[r] : -1;
            const unsigned long long mask = __ballot(idx >= 0);
            if (mask != 0ull) {
                if (idx >= 0) {
                    const int p = tail + __popcll(mask & ((1ull << lane) - 1ull));
                    q_in[wave][p & 127] = idx;
                    q_out[wave][p & 127] = (int32_t)(r - r_begin);
                }
                tail += __popcll(mask);
                __builtin_amdgcn_wave_barrier();
            }
            while (tail - head >= 32) {
                const int h0 = head;
                step([&](int e) { return (int64_t)q_in[wave][(h0 + e) & 127]; },
                     [&](int e) { return r_begin + q_out[wave][(h0 + e) & 127]; }, 32);
                head += 32;
            }
        }
        if (tail > head) {
            const int h0 = head;
            step([&](int e) { return (int64_t)q_in[wave][(h0 + e) & 127]; },
                 [&](int e) { return r_begin + q_out[wave][(h0 + e) & 127]; }, tail - head);
        }
    } else {
        for (int64_t r0 = r_begin; r0 < r_end; r0 += 32) {
            const int valid = r_end - r0 < 32 ? (int)(r_end - r0) : 32;
            step([&](int e) { return r0 + e; }, [&](int e) { return r0 + e; }, valid);
        }
    }

    if (want_db) {  // lanes cq, cq+16, cq+32, cq+48 hold the four row groups of the same channel quad
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = db_acc[j];
            v += __shfl_xor(v, 16, SEG3D_WAVE);
            v += __shfl_xor(v, 32, SEG3D_WAVE);
            if (rg == 0 && a_ok && v != 0.0f) atomicAdd(&db[co0 + 4 * cq + j], v);
        }
    }
    // ---- epilogue: D[a][b][r] = dw[co = co0 + 4*(4g + r) + a][ci = ci0 + 4*c16 + b]; stage [co_local][ci_local]
    float* st = reinterpret_cast<float*>(&my[0][0][0][0]);  // 16 KiB = 64 x 64 floats
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[(4 * (4 * g + r) + a) * 64 + 4 * c16 + b] = acc[a][b][r];
    __builtin_amdgcn_wave_barrier();
    const int ci = ci0 + lane;
    if (lane < 16 * NB && ci < cin) {
        for (int col = 0; col < 16 * NA; ++col) {
            const int co = co0 + col;
            if (co >= cout) break;
            const float v = st[col * 64 + lane];
            if (v != 0.0f) atomicAdd(&dw[((int64_t)co * kk + k) * cin + ci], v);
        }
    }
}

template <bool SPARSE>
int dispatch(const float* x, const float* dy, const int32_t* nbr, int64_t m, int cin, int cout, float* dw, float* db,
             hipStream_t st) {
    const int nblk = ((cin + 63) / 64) * ((cout + 63) / 64);
    const int blocks = nblk * (SPARSE ? 27 : 1);
    // aim for ~8k waves in flight; chunks are multiples of 64 rows
    int64_t rows = m * blocks / 8192;
    rows = (rows + 63) / 64 * 64;
    if (rows < 256) rows = 256;
    if (rows > 4096) rows = 4096;
    dim3 grid((unsigned)ceil_div64(m, rows * kWaves), SPARSE ? 27 : 1, (unsigned)nblk);
    hipLaunchKernelGGL(wgrad_split_kernel<SPARSE>, grid, dim3(kThreads), 0, st, x, dy, nbr, m, cin, cout, (int)rows,
                       SPARSE ? 27 : 1, dw, db);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // namespace

// used by seg3d_spconv_wgrad (spconv.hip)
int wgrad_split_sparse(const float* x, const float* dy, const int32_t* nbr, int64_t m_out, int cin, int cout, float* dw,
                       hipStream_t st) {
    return dispatch<true>(x, dy, nbr, m_out, cin, cout, dw, nullptr, st);
}
