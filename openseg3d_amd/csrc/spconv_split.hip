// a9-a11 forward / dgrad in split-bf16 ("bf16x3") arithmetic.
//
// gfx950's fp32 MFMA runs at the fp32 vector rate (1/16 of bf16 MFMA), so every sparse conv with C >= 96
// is bound by the exact-fp32 matrix pipe rather than by the gather.  Here each fp32 operand is split into
// two bf16 terms (x = hi + lo, 16 significant bits) and a product is three bf16 MFMAs
// (hi*hi + hi*lo + lo*hi, fp32 accumulate): ~2^-16 relative error per product at 16/3 the fp32-MFMA
// rate.  Effect on the full model: max |logit error| 9e-6 against the 1e-3 budget.
//
// Tile: 4 waves x (RB x 16) output rows x (NBT x 16) columns.  Work is a stream of chunks
// (32 input channels, active kernel offset k) -- channel slice OUTERMOST: the rows a tile gathers for its different
// offsets are largely the same rows (the tile and its halo), so within one channel slice the same 128-B lines are
// read again a few chunks later and come from the CU's L1 / the XCD's L2 instead of crossing the fabric once per
// offset (offset-outermost order: 2.4 GB of fabric reads for a 384 -> 384 layer whose operands total 76 MB).
// The tile's neighbour table (27 x 128 entries) is loaded once into LDS -- the same loads find the offsets that are
// active at all.  An optional row order lets the caller group output rows that share their set of active offsets
// (strided / inverse convs: the set is fixed by the row's coordinate parity, 1-8 of 27), so that a tile visits only
// those offsets instead of all 27.  Per chunk the workgroup copies the pre-split W_k fragments
// (NBT x 2 KiB, contiguous in the packed stream) straight into a double-buffered LDS slot with
// register-staged 16-B loads, every wave gathers its neighbour rows (32 B per lane), splits them in
// registers and issues RB x NBT x 3 v_mfma_f32_16x16x32_bf16.  The W loads and the gather of chunk c+1 are
// in flight while chunk c computes (their waits sit behind the MFMAs); one barrier per chunk.  Offsets with no active
// neighbour in the whole tile are never visited; a wave whose own rows have none skips the MFMAs.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

#ifdef SEG3D_CONV_STAMP
__device__ unsigned long long* g_stamp_buf = nullptr;
extern "C" int seg3d_debug_conv_stamps(void* buf) {
    unsigned long long* p = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : 2;
}
#define STAMP(i)                                                     \
    do {                                                             \
        __builtin_amdgcn_sched_barrier(0);                           \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
        __builtin_amdgcn_sched_barrier(0);                           \
        st_acc[i] += t_ - st_last;                                   \
        st_last = t_;                                                \
    } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// 8 floats -> (hi, lo) bf16 fragments; hi = RNE(x), lo = RNE(x - hi)
__device__ __forceinline__ void split8(const f32x4& p, const f32x4& q, bf16x8* hi, bf16x8* lo) {
    u32x4 h, l;
    const float v[8] = {p[0], p[1], p[2], p[3], q[0], q[1], q[2], q[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t ph = pack_bf16(v[2 * i], v[2 * i + 1]);
        const float h0 = __builtin_bit_cast(float, ph << 16);
        const float h1 = __builtin_bit_cast(float, ph & 0xFFFF0000u);
        h[i] = ph;
        l[i] = pack_bf16(v[2 * i] - h0, v[2 * i + 1] - h1);
    }
    *hi = __builtin_bit_cast(bf16x8, h);
    *lo = __builtin_bit_cast(bf16x8, l);
}

// packed stream: [k'][cin_op/32 (padded)][cout_op/16][2: hi,lo][64 lanes][8 bf16];
// lane = (ci%32)/8 * 16 + co%16, element j = ci%8.  One thread produces one lane's 8 elements of both halves
// (two 16-B stores); item index = ((k' * cb_n + cb) * nb_n + nb) * 64 + lane.
__device__ __forceinline__ void pack_item(const float* __restrict__ w, int cin_src, int cout_src, int kk, int transpose,
                                          int flip, int64_t item, __bf16* __restrict__ wp) {
    const int cin_op = transpose ? cout_src : cin_src;
    const int cout_op = transpose ? cin_src : cout_src;
    const int cb_n = (cin_op + 31) / 32, nb_n = cout_op / 16;
    int64_t r = item;
    const int lane = (int)(r & 63); r >>= 6;
    const int nb = (int)(r % nb_n); r /= nb_n;
    const int cb = (int)(r % cb_n); r /= cb_n;
    const int kp = (int)r;
    const int k = flip ? kk - 1 - kp : kp;
    const int ci0 = cb * 32 + (lane >> 4) * 8;
    const int co_op = nb * 16 + (lane & 15);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci_op = ci0 + j;
        const int ci = transpose ? co_op : ci_op;
        const int co = transpose ? ci_op : co_op;
        v[j] = ci_op < cin_op ? w[((int64_t)co * kk + k) * cin_src + ci] : 0.f;
    }
    bf16x8 hi, lo;
    split8((f32x4){v[0], v[1], v[2], v[3]}, (f32x4){v[4], v[5], v[6], v[7]}, &hi, &lo);
    const int64_t base = ((item >> 6) * 2) * 512 + (int64_t)lane * 8;  // (k', cb, nb) block of 2 x 64 x 8 elements
    *reinterpret_cast<bf16x8*>(wp + base) = hi;
    *reinterpret_cast<bf16x8*>(wp + base + 512) = lo;
}

__global__ __launch_bounds__(256) void pack_weight_split(const float* __restrict__ w, int cin_src, int cout_src, int kk,
                                                         int transpose, int flip, int64_t items, __bf16* __restrict__ wp) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < items) pack_item(w, cin_src, cout_src, kk, transpose, flip, t, wp);
}

// One launch for many weights (all conv and Linear packs of a training step: the weights change every optimizer
// step and every layer needs W and W^T).  jobs are sorted by first_block; a block finds its job by bisection.
struct PackJob {
    const float* src;
    __bf16* dst;
    int32_t cin_src, cout_src, kk, transpose, flip, reserved;
    int64_t first_block;
};
static_assert(sizeof(PackJob) == 48, "PackJob layout is part of the C ABI (seg3d_pack_weights_batched)");

__global__ __launch_bounds__(256) void pack_weights_batched(const PackJob* __restrict__ jobs, int n_jobs) {
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {  // last job with first_block <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int64_t)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const PackJob jb = jobs[lo];
    const int cin_op = jb.transpose ? jb.cout_src : jb.cin_src;
    const int cout_op = jb.transpose ? jb.cin_src : jb.cout_src;
    const int64_t items = (int64_t)jb.kk * ((cin_op + 31) / 32) * (cout_op / 16) * 64;
    const int64_t t = ((int64_t)blockIdx.x - jb.first_block) * 256 + threadIdx.x;
    if (t < items) pack_item(jb.src, jb.cin_src, jb.cout_src, jb.kk, jb.transpose, jb.flip, t, jb.dst);
}

// IO: storage of the activations.  0 = float32 in, float32 out (the default path); 3 = the same with a second float32
// summand on the A operand (Linear layers only: y = (x + x_add) W^T + b); 4 = float32, the epilogue MULTIPLIES by the
// `addend` tensor instead of adding it (Linear layers only: d h = gelu'(h) * (d m W2)); 1 = float32 in, bf16 out;
// 2 = bf16 in, bf16 out (the opt-in bf16-storage mode of the sparse-conv feature maps, BASELINE configs[4]): a bf16 row
// IS its own hi part (lo = 0), so the a_lo . w_hi product disappears -- two MFMAs per product instead of three -- and a
// gathered row is half the bytes.  Residual addend and output share the output's storage type; accumulation, bias and
// the activation stay float32.
// IO = 5 (Linear only, one column group = the whole row): LayerNorm + residual in the epilogue,
// y = res + LN(x W^T + b) (point_transformer_layer.py:289-298: x = x + norm1(attn(..)), x = x + norm2(mlp(x)))
struct LnEpilogue {
    const float* gamma;
    const float* beta;
    float eps;
};

template <int NBT, int RB, bool DENSE, int IO = 0, int DEPTH = 1>
__global__ __launch_bounds__(256, 2) void spconv_split_kernel(const void* __restrict__ x_v, const int32_t* __restrict__ nbr,
                                                              int64_t m_out, const uint4* __restrict__ wp,
                                                              const float* __restrict__ bias,
                                                              const void* __restrict__ addend_v,
                                                              const int32_t* __restrict__ row_order, int cin, int cout,
                                                              void* __restrict__ y_v, int relu_flags,
                                                              const float* __restrict__ x_add, LnEpilogue ln) {
    const int relu = relu_flags & 1;
    const float* __restrict__ x = static_cast<const float*>(x_v);
    const float* __restrict__ addend = static_cast<const float*>(addend_v);
    float* __restrict__ y = static_cast<float*>(y_v);
    constexpr int kW = 4;
    constexpr int kSlot = NBT * 128;  // uint4 per staged chunk (NBT x {hi, lo} x 64 lanes)
    constexpr int kPieces = NBT * 2;  // 1-KiB wave-instructions per chunk
    __shared__ __attribute__((aligned(16))) uint4 wlds[2 * kSlot];
    __shared__ uint32_t wave_mask[kW];
    // neighbour table of the tile, [wave][offset][row of the wave]: loaded once (the same 27 loads find the active
    // offsets), read per chunk -- the chunk order below changes the offset every chunk
    constexpr int kRowsW = RB * 16;
    __shared__ int32_t idx_lds[DENSE ? 1 : kW * 27 * kRowsW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
#ifdef SEG3D_CONV_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_last;
#endif
    // relu_flags bit 1 (sparse tables, SEG3D_CONV_XCD_RUN): workgroup -> row tile so that each XCD (workgroup id % 8) walks ONE
    // contiguous run of tiles -- neighbouring tiles gather overlapping parent rows, which then meet in one L2 instead of
    // being fetched by eight (the grid is padded to a multiple of 8 tiles; surplus workgroups leave before any barrier)
    int64_t tile_id = blockIdx.x;
    int col_group = blockIdx.y;
    if (relu_flags & 4) {
        // relu_flags bit 2 (dense Linear layers with several column groups, 1-D grid): the column groups of ONE row tile are
        // workgroups id, id + 8, id + 16, .. -- the same XCD, back to back -- so the row tile they all read crosses the fabric
        // once and comes out of that XCD's L2 afterwards (a 2-D grid puts them gridDim.x workgroups apart, on any XCD)
        const int ny = (cout >> 4) / NBT;
        const int64_t n_tiles = (m_out + kW * RB * 16 - 1) / (kW * RB * 16);
        const int64_t id = blockIdx.x;
        const int64_t grp = id / (8 * ny);
        const int t = (int)(id - grp * 8 * ny);
        tile_id = grp * 8 + (t & 7);
        col_group = t >> 3;
        if (tile_id >= n_tiles) return;
    }
    if (relu_flags & 2) {
        const int64_t n_tiles = (m_out + kW * RB * 16 - 1) / (kW * RB * 16);
        const int64_t per = (n_tiles + 7) >> 3;
        tile_id = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (tile_id >= n_tiles) return;
    }
    const int64_t row0 = tile_id * (kW * RB * 16) + wave * (RB * 16);
    const int nb0 = col_group * NBT;
    const int cb_n = (cin + 31) >> 5, nb_n = cout >> 4;
    const int64_t last_row = m_out - 1;

    // ---- which offsets does this tile touch?  (27 independent loads, then ballots)
    uint32_t my_mask = 0;
    {
        const int64_t pos = row0 + (lane & (kRowsW - 1));  // tile position -> output row (optional processing order)
        const bool ok = pos < m_out;
        const int64_t rc = ok ? (row_order ? (int64_t)row_order[pos] : pos) : last_row;
        if (DENSE) {  // dense rows (Linear layer): one "offset", neighbour of row r is row r
            my_mask = 1u;
        } else {
            int32_t v[27];
#pragma unroll
            for (int k = 0; k < 27; ++k) v[k] = nbr[(int64_t)k * m_out + rc];
#pragma unroll
            for (int k = 0; k < 27; ++k) {
                v[k] = ok ? v[k] : -1;
                if (__ballot(v[k] >= 0) != 0ull) my_mask |= 1u << k;
                if (lane < kRowsW) idx_lds[(wave * 27 + k) * kRowsW + lane] = v[k];
            }
        }
    }
    if (lane == 0) wave_mask[wave] = my_mask;
    __syncthreads();
    const uint32_t todo_all = wave_mask[0] | wave_mask[1] | wave_mask[2] | wave_mask[3];

    f32x4 acc[RB][NBT];
#pragma unroll
    for (int n = 0; n < NBT; ++n) {
        const float b = bias ? bias[(nb0 + n) * 16 + c16] : 0.0f;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb][n] = (f32x4){b, b, b, b};
    }

    if (todo_all != 0u) {
        // Chunk order: 32-channel slice outermost, offsets inside.  The rows a tile gathers for its different offsets
        // are mostly the same rows (its own rows and their halo), so for one channel slice they are 128-B lines that
        // the next offsets read again within a few chunks -- they stay in the CU's L1 / the XCD's L2 instead of coming
        // over the fabric once per offset (deep levels: x alone is 30 MB per level, 4 MiB of L2 per XCD).
        int64_t grow[RB];  // DENSE: rows this lane streams (clamped so that every load is in bounds)
        bool grow_ok[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int64_t pos = row0 + rb * 16 + c16;
            grow_ok[rb] = pos < m_out;
            grow[rb] = grow_ok[rb] ? pos : last_row;
        }
        // neighbour rows of offset k for this lane's RB rows
        auto fetch_idx = [&](int k, int32_t* idx) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                idx[rb] = DENSE ? (grow_ok[rb] ? (int32_t)grow[rb] : -1) : idx_lds[(wave * 27 + k) * kRowsW + rb * 16 + c16];
        };
        // W chunk -> registers now, -> LDS slot after the MFMAs of the current chunk: piece j (1 KiB) belongs to wave
        // j % 4.  (An LDS-DMA copy would save the registers, but the compiler then drains vmcnt before the first
        // ds_read of the current chunk -- it cannot tell the slots apart -- and nothing overlaps.)
        constexpr int kMine = (kPieces + kW - 1) / kW;
        // DEPTH register sets: loads of up to DEPTH chunks are in flight while the current one is multiplied (set index =
        // a compile-time constant everywhere, so the sets are plain registers)
        u32x4 wreg[DEPTH][kMine];  // ext-vector type: HIP's uint4 struct is not promoted out of scratch here
        auto stage_w = [&](auto S, int k, int cb) {
            const uint4* src = wp + (((int64_t)k * cb_n + cb) * nb_n + nb0) * 128;
#pragma unroll
            for (int j = 0; j < kMine; ++j) {
                const int piece = j * kW + wave;
                wreg[S][j] = *reinterpret_cast<const u32x4*>(src + (kPieces % kW == 0 || piece < kPieces ? piece : 0) * 64 + lane);
            }
        };
        auto commit_w = [&](auto S, int buf) {
#pragma unroll
            for (int j = 0; j < kMine; ++j) {
                const int piece = j * kW + wave;
                if (kPieces % kW == 0 || piece < kPieces)
                    *reinterpret_cast<u32x4*>(wlds + buf * kSlot + piece * 64 + lane) = wreg[S][j];
            }
        };
        f32x4 areg[DEPTH][RB][2];
        constexpr bool XADD = DENSE && IO == 3;  // Linear with a second summand: y = (x + x_add) W^T + b
        f32x4 areg2[XADD ? RB : 1][2];
        bool aval[DEPTH][RB];
        auto issue_a = [&](auto S, const int32_t* idx, int cb, bool on) {
            const bool in_range = cb * 32 + g * 8 < cin;
            const int col = in_range ? cb * 32 + g * 8 : 0;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                aval[S][rb] = on && in_range && idx[rb] >= 0;
                const int64_t src_row = idx[rb] >= 0 ? idx[rb] : 0;
                if constexpr (IO == 2) {  // 8 bf16 = one 16-B piece
                    areg[S][rb][0] = *reinterpret_cast<const f32x4*>(static_cast<const __bf16*>(x_v) + src_row * cin + col);
                } else {
                    const f32x4* p = reinterpret_cast<const f32x4*>(x + src_row * cin + col);
                    areg[S][rb][0] = p[0];
                    areg[S][rb][1] = p[1];
                    if constexpr (XADD) {  // summed in land_a, when both loads have arrived
                        const f32x4* p2 = reinterpret_cast<const f32x4*>(x_add + src_row * cin + col);
                        areg2[rb][0] = p2[0];
                        areg2[rb][1] = p2[1];
                    }
                }
            }
        };
        bf16x8 a_hi[RB], a_lo[RB];
        auto land_a = [&](auto S) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (IO == 2) {
                    a_hi[rb] = __builtin_bit_cast(bf16x8, aval[S][rb] ? areg[S][rb][0] : z);
                } else {
                    f32x4 r0 = areg[S][rb][0], r1 = areg[S][rb][1];
                    if constexpr (XADD) {
                        r0 = r0 + areg2[rb][0];
                        r1 = r1 + areg2[rb][1];
                    }
                    split8(aval[S][rb] ? r0 : z, aval[S][rb] ? r1 : z, &a_hi[rb], &a_lo[rb]);
                }
            }
        };

        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, DEPTH - 1>;
        // the chunk sequence: 32-channel slice outermost, this tile's active offsets inside
        struct Chunk {
            int k, cb;
            bool on, have;
        };
        uint32_t it_todo = todo_all;
        int it_cb = 0;
        auto next_chunk = [&]() {
            Chunk c;
            c.have = true;
            if (it_todo == 0u) {  // next channel slice, offsets from the start
                it_cb += 1;
                it_todo = todo_all;
                c.have = it_cb < cb_n;
            }
            c.k = __builtin_ctz(it_todo);
            it_todo &= it_todo - 1;
            c.cb = it_cb;
            c.on = (my_mask >> c.k) & 1u;  // wave-uniform: does any of this wave's rows have a neighbour at this offset?
            return c;
        };
        auto issue = [&](auto S, const Chunk& c) {
            int32_t idx[RB];
            fetch_idx(c.k, idx);
            stage_w(S, c.k, c.cb);
            issue_a(S, idx, c.cb, c.on);
        };
        auto multiply = [&](bool on, int buf) {
            if (on) {
                const uint4* slot = wlds + buf * kSlot;
#pragma unroll
                for (int n = 0; n < NBT; ++n) {
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, slot[(n * 2 + 0) * 64 + lane]);
                    const bf16x8 bl = __builtin_bit_cast(bf16x8, slot[(n * 2 + 1) * 64 + lane]);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        if constexpr (IO != 2) acc[rb][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[rb], bh, acc[rb][n], 0, 0, 0);
                        acc[rb][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[rb], bl, acc[rb][n], 0, 0, 0);
                        acc[rb][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[rb], bh, acc[rb][n], 0, 0, 0);
                    }
                }
            }
        };

        Chunk cur = next_chunk();
        issue(S0{}, cur);  // prologue: chunk 0
        land_a(S0{});
        commit_w(S0{}, 0);
        __syncthreads();
        int buf = 0;
        STAMP(7);
        if constexpr (DEPTH == 1) {
            for (;;) {
                const Chunk nxt = next_chunk();
                if (nxt.have) issue(S0{}, nxt);
                // (Issuing these loads between the MFMAs below instead -- one per column block -- is slower: every 1-KiB
                // wave load holds the wave for 55-80 cycles at the CU's address unit, lanes of a quad read four different
                // rows; spread over the MFMAs they stall the matrix pipe instead and land later.  tools/probes/conv_stamps.py)
                STAMP(0);
                multiply(cur.on, buf);
                STAMP(1);
                if (!nxt.have) break;
#ifdef SEG3D_CONV_STAMP
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                STAMP(2);
#endif
                land_a(S0{});
                STAMP(3);
                commit_w(S0{}, buf ^ 1);
#ifdef SEG3D_CONV_STAMP
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                STAMP(4);
#endif
                __syncthreads();
                STAMP(5);
                buf ^= 1;
                cur = nxt;
            }
        } else {
            // Two chunks in flight (the idea: 18 - 36 MFMAs per chunk are a fraction of a gather's round trip through L2, so
            // one look-ahead should leave a narrow layer's wave waiting for rows; measured otherwise, see conv_depth2()).
            // Invariant at the top: `cur` is landed (slot buf); `n1` is in flight in set 1.
            Chunk n1 = next_chunk();
            if (n1.have) issue(S1{}, n1);
            for (;;) {
                Chunk n2 = n1;
                if (n1.have) {
                    n2 = next_chunk();
                    if (n2.have) issue(S0{}, n2);  // set 0 was landed: free
                }
                multiply(cur.on, buf);
                if (!n1.have) break;
                land_a(S1{});
                commit_w(S1{}, buf ^ 1);
                __syncthreads();
                buf ^= 1;
                // `n1` is current now, `n2` in flight in set 0
                Chunk n3 = n2;
                if (n2.have) {
                    n3 = next_chunk();
                    if (n3.have) issue(S1{}, n3);
                }
                multiply(n1.on, buf);
                if (!n2.have) break;
                land_a(S0{});
                commit_w(S0{}, buf ^ 1);
                __syncthreads();
                buf ^= 1;
                cur = n2;
                n1 = n3;
            }
        }
    }

    if constexpr (IO == 5) {
        // The workgroup's NBT column blocks are the whole row (host-checked): a row's NBT * 16 values sit in the 16 lanes of
        // one lane group, NBT per lane.  Two-pass mean / variance in registers as rownorm.hip's ln_fwd_kernel takes them.
        float gm[NBT], bt[NBT];
#pragma unroll
        for (int n = 0; n < NBT; ++n) {
            gm[n] = ln.gamma[n * 16 + c16];
            bt[n] = ln.beta[n * 16 + c16];
        }
        const float inv_c = 1.0f / (float)cout;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sum = 0.f;
#pragma unroll
                for (int n = 0; n < NBT; ++n) sum += acc[rb][n][r];
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) sum += __shfl_xor(sum, d, 64);
                const float mu = sum * inv_c;
                float ss = 0.f;
#pragma unroll
                for (int n = 0; n < NBT; ++n) {
                    const float dlt = acc[rb][n][r] - mu;
                    ss = fmaf(dlt, dlt, ss);
                }
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) ss += __shfl_xor(ss, d, 64);
                const float rs = rsqrtf(ss * inv_c + ln.eps);
                const int64_t opos = row0 + rb * 16 + g * 4 + r;
                if (opos < m_out) {
                    float* yr = y + opos * cout + c16;
                    const float* rr = addend ? addend + opos * cout + c16 : nullptr;
#pragma unroll
                    for (int n = 0; n < NBT; ++n) {
                        float o = (acc[rb][n][r] - mu) * rs * gm[n] + bt[n];
                        if (rr) o += rr[n * 16];
                        yr[n * 16] = o;
                    }
                }
            }
        return;
    }

    // D layout of v_mfma_f32_16x16x*: row = (lane>>4)*4 + r, col = lane & 15
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t opos = row0 + rb * 16 + g * 4 + r;
            if (opos < m_out) {
                const int64_t orow = row_order ? (int64_t)row_order[opos] : opos;
                if constexpr (IO == 1 || IO == 2) {  // bf16 storage of the output (and of the residual it is added to)
                    __bf16* yb = static_cast<__bf16*>(y_v) + orow * cout + nb0 * 16 + c16;
                    const __bf16* ab = addend_v ? static_cast<const __bf16*>(addend_v) + orow * cout + nb0 * 16 + c16 : nullptr;
#pragma unroll
                    for (int n = 0; n < NBT; ++n) {
                        const float v = acc[rb][n][r] + (ab ? (float)ab[n * 16] : 0.0f);
                        yb[n * 16] = (__bf16)(relu ? (v < 0.0f ? 0.0f : v) : v);
                    }
                    continue;
                }
                float* yr = y + orow * cout + nb0 * 16 + c16;
                // y = act(x W^T + b (+ addend)): addend = a second gradient path, or the residual of a conv block whose
                // BatchNorm was folded into W and b (eval); relu = that block's activation
                // relu as torch.relu computes it: a NaN stays a NaN (fmaxf would return the other operand and hide a
                // diverged accumulator behind a 0); without relu the value is stored untouched
                if (addend) {
                    const float* ar = addend + orow * cout + nb0 * 16 + c16;
#pragma unroll
                    for (int n = 0; n < NBT; ++n) {
                        // IO = 4 (Linear only): the "addend" is an elementwise factor -- y = (x W^T) * factor
                        const float v = IO == 4 ? acc[rb][n][r] * ar[n * 16] : acc[rb][n][r] + ar[n * 16];
                        yr[n * 16] = relu ? (v < 0.0f ? 0.0f : v) : v;
                    }
                } else {
#pragma unroll
                    for (int n = 0; n < NBT; ++n) {
                        const float v = acc[rb][n][r];
                        yr[n * 16] = relu ? (v < 0.0f ? 0.0f : v) : v;
                    }
                }
            }
        }
#ifdef SEG3D_CONV_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the epilogue's stores are part of the wave's time
    if (g_stamp_buf && lane == 0) {
        unsigned long long* o = g_stamp_buf + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        st_acc[6] = __builtin_amdgcn_s_memtime() - st_begin;
        for (int i = 0; i < 8; ++i) o[i] = st_acc[i];
    }
#endif
}

// SEG3D_CONV_DEPTH=2 (A/B, off by default): column-block widths up to this one run the sparse kernel with two chunks in
// flight.  Measured on the headline scene: every narrow layer 4 - 14 % SLOWER (48 -> 48: 77 -> 82 us, 96 -> 96: 168 -> 183 us)
// -- the second register set costs a wave of occupancy (NBT 6: 115 -> 148 VGPRs, NBT 3: 90 -> 119) and these layers are
// bound by the rate of gather requests, not by one wave's round trip.
constexpr int kDeepMaxNbt = 6;
static bool conv_depth2() {
    static const bool on = [] {
        const char* e = getenv("SEG3D_CONV_DEPTH");
        return e && atoi(e) == 2;
    }();
    return on;
}

template <int NBT, int RB>
int launch_split(const void* x, const int32_t* nbr, int64_t m_out, const void* wp, const float* bias, const void* addend,
                 const int32_t* row_order, int cin, int cout, void* y, int relu, int io, hipStream_t st,
                 const float* x_add = nullptr, LnEpilogue ln = LnEpilogue{nullptr, nullptr, 0.f}) {
    dim3 grid((unsigned)ceil_div64(m_out, 4 * RB * 16), (unsigned)((cout / 16) / NBT));
    // SEG3D_CONV_XCD_RUN (A/B, sparse tables only): every XCD walks one contiguous run of row tiles
    static const int xcd_run = [] {
        const char* e = getenv("SEG3D_CONV_XCD_RUN");
        return (e && atoi(e) == 1) ? 1 : 0;
    }();
    if (xcd_run && nbr != nullptr && io == 0 && grid.x >= 64) {
        grid.x = (grid.x + 7) / 8 * 8;
        relu |= 2;
    }
    // SEG3D_LINEAR_COLGROUPS (A/B): 0 = the column groups of a dense Linear layer as the grid's y dimension (round 4)
    static const int col_adjacent = [] {
        const char* e = getenv("SEG3D_LINEAR_COLGROUPS");
        return (e && atoi(e) == 0) ? 0 : 1;
    }();
    if (col_adjacent && nbr == nullptr && grid.y > 1 && (io == 0 || io == 3 || io == 4)) {
        grid.x = (grid.x + 7) / 8 * 8 * grid.y;
        grid.y = 1;
        relu |= 4;
    }
    if (io == 5) {  // Linear + LayerNorm + residual: the workgroup's columns must be the whole row
        if (nbr != nullptr || NBT * 16 != cout || !ln.gamma || !ln.beta) return SEG3D_EINVAL;
        hipLaunchKernelGGL((spconv_split_kernel<NBT, RB, true, 5>), grid, dim3(256), 0, st, x, nbr, m_out,
                           reinterpret_cast<const uint4*>(wp), bias, addend, nullptr, cin, cout, y, 0, nullptr, ln);
        SEG3D_CHECK_LAUNCH();
        return SEG3D_OK;
    }
    if (io == 1 || io == 2) {  // bf16-storage variants (sparse convs only)
        if (nbr == nullptr) return SEG3D_EINVAL;
        if (io == 1)
            hipLaunchKernelGGL((spconv_split_kernel<NBT, RB, false, 1>), grid, dim3(256), 0, st, x, nbr, m_out,
                               reinterpret_cast<const uint4*>(wp), bias, addend, row_order, cin, cout, y, relu, nullptr, LnEpilogue{nullptr, nullptr, 0.f});
        else
            hipLaunchKernelGGL((spconv_split_kernel<NBT, RB, false, 2>), grid, dim3(256), 0, st, x, nbr, m_out,
                               reinterpret_cast<const uint4*>(wp), bias, addend, row_order, cin, cout, y, relu, nullptr, LnEpilogue{nullptr, nullptr, 0.f});
        SEG3D_CHECK_LAUNCH();
        return SEG3D_OK;
    }
    if (nbr == nullptr && io == 4)  // Linear layer whose output is multiplied elementwise by `addend` (IO = 4)
        hipLaunchKernelGGL((spconv_split_kernel<NBT, RB, true, 4>), grid, dim3(256), 0, st, x, nbr, m_out,
                           reinterpret_cast<const uint4*>(wp), bias, addend, nullptr, cin, cout, y, relu, nullptr, LnEpilogue{nullptr, nullptr, 0.f});
    else if (nbr == nullptr && x_add)  // Linear layer with a second summand on the A operand (IO = 3)
        hipLaunchKernelGGL((spconv_split_kernel<NBT, RB, true, 3>), grid, dim3(256), 0, st, x, nbr, m_out,
                           reinterpret_cast<const uint4*>(wp), bias, addend, nullptr, cin, cout, y, relu, x_add, LnEpilogue{nullptr, nullptr, 0.f});
    else if (nbr == nullptr)  // Linear layer: own instantiation (own symbol in profiles, no table code)
        hipLaunchKernelGGL((spconv_split_kernel<NBT, RB, true>), grid, dim3(256), 0, st, x, nbr, m_out,
                           reinterpret_cast<const uint4*>(wp), bias, addend, nullptr, cin, cout, y, relu, nullptr, LnEpilogue{nullptr, nullptr, 0.f});
    else if (NBT <= kDeepMaxNbt && conv_depth2())  // narrow sparse layers: two chunks in flight
        hipLaunchKernelGGL((spconv_split_kernel<NBT, RB, false, 0, (NBT <= kDeepMaxNbt ? 2 : 1)>), grid, dim3(256), 0, st, x, nbr,
                           m_out, reinterpret_cast<const uint4*>(wp), bias, addend, row_order, cin, cout, y, relu, nullptr, LnEpilogue{nullptr, nullptr, 0.f});
    else
        hipLaunchKernelGGL((spconv_split_kernel<NBT, RB, false>), grid, dim3(256), 0, st, x, nbr, m_out,
                           reinterpret_cast<const uint4*>(wp), bias, addend, row_order, cin, cout, y, relu, nullptr, LnEpilogue{nullptr, nullptr, 0.f});
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // namespace

// ---- internal entry points used by spconv.hip
size_t spconv_split_packed_bytes(int cin_op, int cout_op, int kk) {
    return (size_t)kk * ((cin_op + 31) / 32) * (cout_op / 16) * 1024 * sizeof(__bf16);
}

int spconv_split_pack(const float* weight, int cin, int cout, int kk, int transpose, int flip, void* w_packed,
                      hipStream_t st) {
    const int cin_op = transpose ? cout : cin, cout_op = transpose ? cin : cout;
    const int64_t items = (int64_t)kk * ((cin_op + 31) / 32) * (cout_op / 16) * 64;
    hipLaunchKernelGGL(pack_weight_split, dim3((unsigned)ceil_div64(items, 256)), dim3(256), 0, st, weight, cin, cout, kk,
                       transpose, flip, items, reinterpret_cast<__bf16*>(w_packed));
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

// Forced column-block width (x16) of the gather-GEMM: SEG3D_CONV_NBT read ONCE at load time, or set by
// seg3d_debug_set_conv_nbt (parity tests pin each instantiation with it); 0 = automatic.
static std::atomic<int> g_forced_nbt{[] {
    const char* e = getenv("SEG3D_CONV_NBT");
    return e ? atoi(e) : 0;
}()};

extern "C" int seg3d_debug_set_conv_nbt(int32_t nbt) {
    if (nbt != 0 && nbt != 1 && nbt != 2 && nbt != 3 && nbt != 4 && nbt != 6 && nbt != 8 && nbt != 12) return SEG3D_EINVAL;
    g_forced_nbt.store(nbt, std::memory_order_relaxed);
    return SEG3D_OK;
}

// linear_stream.hip: the row-streaming schedule of the dense Linear layers with cin <= 192 (1 = not its shape)
int linear_stream_fwd(const float* x, int64_t m, const void* wp, const float* bias, const float* addend, int cin, int cout,
                      float* y, int io, hipStream_t st);

static int split_fwd_impl(const void* x, const int32_t* nbr, int64_t m_out, const void* wp, const float* bias,
                          const void* addend, const int32_t* row_order, int cin, int cout, void* y, int relu, int io,
                          hipStream_t st, const float* x_add, LnEpilogue ln) {
    if (x_add && (nbr || io != 0)) return SEG3D_EINVAL;  // the second summand exists for Linear layers only
    if (io == 4 && (nbr || !addend)) return SEG3D_EINVAL;  // ... and so does the elementwise factor
    if (!nbr && !row_order && !relu && !x_add && (io == 0 || io == 4) && g_forced_nbt.load(std::memory_order_relaxed) == 0) {
        // dense rows, W slice in registers, rows streamed (bit-identical results; SEG3D_LINEAR_STREAM=0 switches it off)
        const int rc = linear_stream_fwd(static_cast<const float*>(x), m_out, wp, bias, static_cast<const float*>(addend), cin,
                                         cout, static_cast<float*>(y), io, st);
        if (rc != 1) return rc;
    }
    // Column blocks per workgroup: 192 columns while the launch has >= 400 row tiles; the deepest level has few rows
    // (19k) and 384+ columns: 128-column workgroups put it on the chip in ONE resident round (153 row tiles x 3 = 459 of
    // 512 slots; 96 columns = 612 = a second, mostly empty round) and gather each row 3 times instead of 4
    // (measured per layer, profiles/README.md); narrower tiles re-gather the rows too often.
    const int nb = cout / 16;
    const int64_t row_tiles = ceil_div64(m_out, 4 * 2 * 16);
    int pick = 1;
    if (nb % 12 == 0) pick = row_tiles >= 400 ? 12 : (nb % 8 == 0 ? 8 : 6);
    else if (nb % 6 == 0) pick = 6;
    else if (nb % 4 == 0) pick = 4;
    else if (nb % 3 == 0) pick = 3;
    else if (nb % 2 == 0) pick = 2;
    if (const int w = g_forced_nbt.load(std::memory_order_relaxed); w > 0 && nb % w == 0) pick = w;
    if (io == 5) {  // one column group: the width is the row
        if (nb != 1 && nb != 2 && nb != 3 && nb != 4 && nb != 6 && nb != 8 && nb != 12) return SEG3D_EINVAL;
        pick = nb;
    }
    switch (pick) {
        case 12: return launch_split<12, 2>(x, nbr, m_out, wp, bias, addend, row_order, cin, cout, y, relu, io, st, x_add, ln);
        case 8: return launch_split<8, 2>(x, nbr, m_out, wp, bias, addend, row_order, cin, cout, y, relu, io, st, x_add, ln);
        case 6: return launch_split<6, 2>(x, nbr, m_out, wp, bias, addend, row_order, cin, cout, y, relu, io, st, x_add, ln);
        case 4: return launch_split<4, 2>(x, nbr, m_out, wp, bias, addend, row_order, cin, cout, y, relu, io, st, x_add, ln);
        case 3: return launch_split<3, 2>(x, nbr, m_out, wp, bias, addend, row_order, cin, cout, y, relu, io, st, x_add, ln);
        case 2: return launch_split<2, 2>(x, nbr, m_out, wp, bias, addend, row_order, cin, cout, y, relu, io, st, x_add, ln);
        default: return launch_split<1, 2>(x, nbr, m_out, wp, bias, addend, row_order, cin, cout, y, relu, io, st, x_add, ln);
    }
}

int spconv_split_fwd_io(const void* x, const int32_t* nbr, int64_t m_out, const void* wp, const float* bias,
                        const void* addend, const int32_t* row_order, int cin, int cout, void* y, int relu, int io,
                        hipStream_t st, const float* x_add) {
    return split_fwd_impl(x, nbr, m_out, wp, bias, addend, row_order, cin, cout, y, relu, io, st, x_add,
                          LnEpilogue{nullptr, nullptr, 0.f});
}

int spconv_split_fwd(const float* x, const int32_t* nbr, int64_t m_out, const void* wp, const float* bias,
                     const float* addend, const int32_t* row_order, int cin, int cout, float* y, int relu, hipStream_t st) {
    return spconv_split_fwd_io(x, nbr, m_out, wp, bias, addend, row_order, cin, cout, y, relu, 0, st, nullptr);
}

// ------------------------------------------------------------------ dense Linear layers through the same kernel
// a6 / a22: y = x W^T + b on [rows, C] activations (segformer.py:21-32,58-76, point_transformer_layer.py:260-276,
// cosine_msa.py:58-63,403).  A Linear layer is the single-offset case of the kernel above (identity neighbour
// table): W fragments staged through LDS once per 128-row tile, rows streamed with 32-B loads, split-bf16 MFMA.
extern "C" {

int seg3d_pack_weights_batched(const void* jobs, int32_t n_jobs, int64_t total_blocks, void* stream) {
    if (n_jobs < 0 || total_blocks < 0 || total_blocks > 0x7FFFFFFF || (n_jobs > 0 && !jobs)) return SEG3D_EINVAL;
    if (n_jobs == 0 || total_blocks == 0) return SEG3D_OK;
    hipLaunchKernelGGL(pack_weights_batched, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream),
                       static_cast<const PackJob*>(jobs), (int)n_jobs);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

size_t seg3d_linear_packed_bytes(int32_t cin, int32_t cout, int32_t transpose) {
    if (cin <= 0 || cout <= 0) return 0;
    return spconv_split_packed_bytes(transpose ? cout : cin, transpose ? cin : cout, 1);
}

int seg3d_linear_pack_weight(const float* weight, int32_t cin, int32_t cout, int32_t transpose, void* w_packed,
                             void* stream) {
    const int cin_op = transpose ? cout : cin, cout_op = transpose ? cin : cout;
    if (!weight || !w_packed || cin <= 0 || cout <= 0 || (cin_op & 7) || (cout_op & 15)) return SEG3D_EINVAL;
    return spconv_split_pack(weight, cin, cout, 1, transpose ? 1 : 0, 0, w_packed, as_stream(stream));
}

int seg3d_linear_fwd(const float* x, int64_t m, const void* w_packed, const float* bias, const float* addend, int32_t cin,
                     int32_t cout, float* y, void* stream) {
    if (m < 0 || cin <= 0 || cout <= 0 || (cin & 7) || (cout & 15) || !w_packed) return SEG3D_EINVAL;
    if (m == 0) return SEG3D_OK;
    if (!x || !y) return SEG3D_EINVAL;
    return spconv_split_fwd(x, nullptr, m, w_packed, bias, addend, nullptr, cin, cout, y, 0, as_stream(stream));
}

// y = res + LayerNorm(x W^T + b) in one launch (inference): the encoder layer's out-projection + norm1 + residual and
// fc2 + norm2 + residual (point_transformer_layer.py:289-298).  cout <= 192 (the row must fit one workgroup's column
// blocks: 16, 32, 48, 64, 96, 128 or 192 columns); SEG3D_EINVAL otherwise -- the caller runs the two passes.
int seg3d_linear_layernorm_fwd(const float* x, int64_t m, const void* w_packed, const float* bias, const float* res,
                               const float* gamma, const float* beta, float eps, int32_t cin, int32_t cout, float* y,
                               void* stream) {
    if (m < 0 || cin <= 0 || cout <= 0 || (cin & 7) || (cout & 15) || !w_packed || !gamma || !beta) return SEG3D_EINVAL;
    const int nb = cout / 16;
    if (nb != 1 && nb != 2 && nb != 3 && nb != 4 && nb != 6 && nb != 8 && nb != 12) return SEG3D_EINVAL;
    if (m == 0) return SEG3D_OK;
    if (!x || !y) return SEG3D_EINVAL;
    return split_fwd_impl(x, nullptr, m, w_packed, bias, res, nullptr, cin, cout, y, 0, 5, as_stream(stream), nullptr,
                          LnEpilogue{gamma, beta, eps});
}

// y = (x W^T) * factor, elementwise: the input gradient of fc2 times the saved GELU derivative (point_transformer_layer.py:
// 260-276 backward) -- the gelu_backward pass over [rows, hidden] (two reads, one write) becomes one read in this epilogue.
int seg3d_linear_fwd_mul(const float* x, int64_t m, const void* w_packed, const float* factor, int32_t cin, int32_t cout,
                         float* y, void* stream) {
    if (m < 0 || cin <= 0 || cout <= 0 || (cin & 7) || (cout & 15) || !w_packed) return SEG3D_EINVAL;
    if (m == 0) return SEG3D_OK;
    if (!x || !factor || !y) return SEG3D_EINVAL;
    return spconv_split_fwd_io(x, nullptr, m, w_packed, nullptr, factor, nullptr, cin, cout, y, 0, 4, as_stream(stream), nullptr);
}

// y = (x + x_add) W^T + b: the cosine attention's q | k in-projection reads x + pos (cosine_msa.py:58-63,
// point_transformer_layer.py:289-291); the sum is taken on the A operand's way into the split, so the [rows, C] tensor
// x + pos is never written (inference; training keeps it -- it is the weight gradient's operand).
int seg3d_linear_fwd_sum(const float* x, const float* x_add, int64_t m, const void* w_packed, const float* bias, int32_t cin,
                         int32_t cout, float* y, void* stream) {
    if (m < 0 || cin <= 0 || cout <= 0 || (cin & 7) || (cout & 15) || !w_packed) return SEG3D_EINVAL;
    if (m == 0) return SEG3D_OK;
    if (!x || !x_add || !y) return SEG3D_EINVAL;
    return spconv_split_fwd_io(x, nullptr, m, w_packed, bias, nullptr, nullptr, cin, cout, y, 0, 0, as_stream(stream), x_add);
}

}  // extern "C"
