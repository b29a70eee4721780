// a9 (submanifold 3x3x3 conv) forward / dgrad, "row image" schedule of the split-bf16 gather-GEMM.
//
// spconv_split.hip gathers the 32-channel piece of a neighbour row from global memory and splits it hi | lo once per
// (output row, offset) PAIR: 6.5 - 17 times per input row.  Here a tile's DISTINCT input rows -- its own rows and
// their halo -- are loaded once per 32-channel slice, split once and parked in an LDS image; the 27 offsets then read
// their A fragments from the image through a per-tile table of image slots.  That only pays when a tile is spatially
// compact (few distinct rows per output row), so output rows are processed in MORTON order of their coordinates, 128
// per tile (headline scene: 1.4 - 2.0 distinct rows per output row against 4.2 - 4.9 in table order), and inside a tile
// the rows are sorted by their 27-bit neighbour mask so that 16-row MFMA blocks with no neighbour at an offset are
// skipped (1.3 executed row products per useful one on the deep levels).  Both orders are scheduling only: every
// output row still accumulates slice by slice, offset by offset, product by product in the order of spconv_split_kernel,
// so the results are bit-identical to it.
//
// Per-rulebook "tile plan" (seg3d_conv_plan_build, built once per site level, used by every layer, forward and dgrad):
//   row_order [tiles][128]      tile position -> output row (-1 = padding)
//   ucount    [tiles]           distinct input rows of the tile; > 512 = "direct" tile (below)
//   tilemask  [tiles]           offsets with any neighbour in the tile
//   blkmask   [tiles][32]       per offset: bit rb = 16-row block rb has a neighbour there
//   lidx      [tiles][27][16][8] uint16: 16-B slot of the neighbour's image row, [offset][row % 16][row / 16]
//   uniq      [tiles][512]      the distinct input rows
// Kernel: workgroup = (tile, column group), 4 waves.  Wave tile = (128 / WR) rows x 48 columns: B fragments come
// straight from the packed weight stream in L2 (every wave owns its columns: nothing to share through LDS), prefetched
// one chunk ahead in registers; there is NO barrier per chunk, only two per 32-channel slice around the image refill.
// A tile whose distinct rows do not fit the image (never on the benchmark scenes; possible on scattered sites) is run
// "direct": the image is refilled per offset with that offset's 128 neighbour rows.
#include <rocprim/device/device_radix_sort.hpp>

#include <cstdlib>
#include <type_traits>

#include "common.hpp"

#ifdef SEG3D_TILE_DBG
#define TILE_DBG(d) (d)
#else
#define TILE_DBG(d) 0
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kTile = 128;       // output rows per tile
constexpr int kUMax = 512;       // rows of the LDS image
constexpr int kZeroSlot = 4096;  // first 16-B slot of the all-zero image row (row 512)
constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// 8 floats -> (hi, lo) bf16 fragments; hi = RNE(x), lo = RNE(x - hi)   (as spconv_split.hip's split8)
__device__ __forceinline__ void split8(const f32x4& p, const f32x4& q, u32x4* hi, u32x4* lo) {
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
    *hi = h;
    *lo = l;
}

// first 16-B slot of image row u: a 256-B bank line holds a row pair; the 16 slots of a pair are XOR-swizzled with the
// pair index so that lanes reading the same quarter of consecutive rows (the common case: neighbours of consecutive
// sites are consecutive rows) hit different banks.  Quarter s (0-3 hi, 4-7 lo) of row u lives at slot image_slot(u) ^ s.
__host__ __device__ __forceinline__ int image_slot(int u) {
    const int pair = u >> 1;
    return pair * 16 + ((((u & 1) << 3)) ^ (pair & 15));
}

struct PlanView {
    int32_t* row_order;
    int32_t* ucount;
    uint32_t* tilemask;
    uint8_t* blkmask;
    uint16_t* lidx;
    int32_t* uniq;
};

size_t plan_bytes(int64_t n_tiles) {
    size_t b = 0;
    b += align_up((size_t)n_tiles * kTile * 4, 256);
    b += align_up((size_t)n_tiles * 4, 256);
    b += align_up((size_t)n_tiles * 4, 256);
    b += align_up((size_t)n_tiles * 32, 256);
    b += align_up((size_t)n_tiles * 27 * kTile * 2, 256);
    b += align_up((size_t)n_tiles * kUMax * 4, 256);
    return b;
}

PlanView plan_view(void* plan, int64_t n_tiles) {
    WsCarver ws(plan);
    PlanView v;
    v.row_order = ws.take<int32_t>((size_t)n_tiles * kTile);
    v.ucount = ws.take<int32_t>((size_t)n_tiles);
    v.tilemask = ws.take<uint32_t>((size_t)n_tiles);
    v.blkmask = ws.take<uint8_t>((size_t)n_tiles * 32);
    v.lidx = ws.take<uint16_t>((size_t)n_tiles * 27 * kTile);
    v.uniq = ws.take<int32_t>((size_t)n_tiles * kUMax);
    return v;
}

// ------------------------------------------------------------------------------------------------ plan build
using SortKey = unsigned long long;
constexpr unsigned kKeyBits = 56;  // batch (8) | 3 x 16 interleaved coordinate bits

__device__ __forceinline__ unsigned long long spread3(uint32_t v) {  // 16 bits -> every third bit
    unsigned long long x = v & 0xFFFFu;
    x = (x | (x << 32)) & 0x001F00000000FFFFull;
    x = (x | (x << 16)) & 0x001F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(kThreads) void morton_keys(const int32_t* __restrict__ coords, int m, SortKey* __restrict__ keys,
                                                        uint32_t* __restrict__ vals) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= m) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];  // (b, z, y, x)
    keys[i] = ((unsigned long long)(c.x & 0xFF) << 48) | spread3((uint32_t)c.w) | (spread3((uint32_t)c.z) << 1) |
              (spread3((uint32_t)c.y) << 2);
    vals[i] = (uint32_t)i;
}

constexpr int kHash = 8192;
__device__ __forceinline__ int hash13(int32_t v) { return (int)(((uint32_t)v * 2654435761u) >> 19); }

__global__ __launch_bounds__(kThreads) void plan_kernel(const uint32_t* __restrict__ order, const int32_t* __restrict__ nbr,
                                                        int64_t m, PlanView pv) {
    __shared__ int32_t tab[kHash];
    __shared__ uint16_t tabid[kHash];
    __shared__ int32_t rows[kTile];
    __shared__ uint32_t masks[kTile];
    __shared__ uint8_t newpos[kTile];
    __shared__ int32_t cnt[kThreads];
    __shared__ uint32_t bm[27];
    const int tid = threadIdx.x;
    const int64_t tile = blockIdx.x;
    for (int i = tid; i < kHash; i += kThreads) tab[i] = -1;
    if (tid < kTile) {
        const int64_t pos = tile * kTile + tid;
        rows[tid] = pos < m ? (int32_t)order[pos] : -1;
        masks[tid] = 0u;
    }
    if (tid < 27) bm[tid] = 0u;
    __syncthreads();
    // the tile's 27 x 128 table entries, 13.5 per thread; distinct input rows collected in an LDS hash set
    constexpr int kEnt = (27 * kTile + kThreads - 1) / kThreads;
    int32_t e[kEnt];
#pragma unroll
    for (int j = 0; j < kEnt; ++j) {
        const int idx = j * kThreads + tid;
        int32_t v = -1;
        if (idx < 27 * kTile) {
            const int k = idx >> 7, p = idx & 127;
            const int32_t r = rows[p];
            if (r >= 0) v = nbr[(int64_t)k * m + r];
            if (v >= 0) {
                atomicOr(&masks[p], 1u << k);
                int h = hash13(v);
                for (;;) {
                    const int32_t old = atomicCAS(&tab[h], -1, v);
                    if (old == -1 || old == v) break;
                    h = (h + 1) & (kHash - 1);
                }
            }
        }
        e[j] = v;
    }
    __syncthreads();
    // position inside the tile: rows sorted by (neighbour mask, position), padding last
    if (tid < kTile) {
        const uint32_t key = rows[tid] >= 0 ? masks[tid] : 0xFFFFFFFFu;
        int rank = 0;
        for (int j = 0; j < kTile; ++j) {
            const uint32_t kj = rows[j] >= 0 ? masks[j] : 0xFFFFFFFFu;
            rank += (kj < key) || (kj == key && j < tid);
        }
        newpos[tid] = (uint8_t)rank;
    }
    // image row of every distinct input row = its rank among the occupied hash slots
    int c = 0;
    for (int s = 0; s < kHash / kThreads; ++s) c += tab[tid * (kHash / kThreads) + s] >= 0;
    cnt[tid] = c;
    __syncthreads();
    for (int d = 1; d < kThreads; d <<= 1) {  // inclusive scan
        const int v = tid >= d ? cnt[tid - d] : 0;
        __syncthreads();
        cnt[tid] += v;
        __syncthreads();
    }
    const int total = cnt[kThreads - 1];
    int id = cnt[tid] - c;
    for (int s = 0; s < kHash / kThreads; ++s) {
        const int slot = tid * (kHash / kThreads) + s;
        const int32_t key = tab[slot];
        if (key >= 0) {
            tabid[slot] = (uint16_t)id;
            if (id < kUMax) pv.uniq[tile * kUMax + id] = key;
            ++id;
        }
    }
    if (tid == 0) pv.ucount[tile] = total;
    if (tid < kTile) pv.row_order[tile * kTile + newpos[tid]] = rows[tid];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kEnt; ++j) {
        const int idx = j * kThreads + tid;
        if (idx < 27 * kTile) {
            const int k = idx >> 7, p = idx & 127;
            const int np = newpos[p], rb = np >> 4, c16 = np & 15;
            int slot = kZeroSlot;
            const int32_t v = e[j];
            if (v >= 0) {
                int h = hash13(v);
                while (tab[h] != v) h = (h + 1) & (kHash - 1);
                slot = image_slot((int)tabid[h]);
                atomicOr(&bm[k], 1u << rb);
            }
            if (total <= kUMax) pv.lidx[tile * (27 * kTile) + (k * 16 + c16) * 8 + rb] = (uint16_t)slot;
        }
    }
    __syncthreads();
    if (tid < 32) pv.blkmask[tile * 32 + tid] = tid < 27 ? (uint8_t)bm[tid] : (uint8_t)0;
    if (tid == 0) {
        uint32_t tm = 0;
        for (int k = 0; k < 27; ++k) tm |= (bm[k] != 0u ? 1u : 0u) << k;
        pv.tilemask[tile] = tm;
    }
}

// ------------------------------------------------------------------------------------------------ the conv
// WR x WC = 4 waves: WR row groups of 128 / WR rows, WC column groups of NBW x 16 columns.
// KS (narrow layers, the workgroup's columns are all of cout <= 64): the four waves share ONE 128-row x cout tile and
// split its chunks (every 4th active offset of a slice each) instead of its rows -- 72 MFMAs per wave and chunk against
// one set of B loads instead of 18, and no wave loads a B fragment another wave loads too; the four partial tiles are
// summed through LDS in a fixed order at the end (so this layout is deterministic but not bit-identical to the per-pair
// gather kernel, whose rows accumulate offset by offset).
// IO: storage of the rows, as in spconv_split_kernel -- 0 = float32 in / out; 1 = float32 in, bf16 out; 2 = bf16 in / out (the
// opt-in bf16-storage mode of the sparse-conv feature maps, BASELINE configs[4]): a bf16 row is its own hi half, so only the
// hi image is staged (half the bytes) and a product takes two MFMAs (a . w_hi + a . w_lo) instead of three.  The residual
// addend has the output's storage type; accumulation, bias and the activation stay float32.
template <int WR, int WC, int NBW, bool KS = false, int IO = 0>
__global__ __launch_bounds__(kThreads, 2) void spconv_tile_kernel(const void* __restrict__ x_v, const int32_t* __restrict__ nbr,
                                                                  PlanView pv, int64_t m_out, int n_tiles, int n_cg, int run,
                                                                  const u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                                  const void* __restrict__ addend_v, int cin, int cout,
                                                                  void* __restrict__ y_v, int relu, int dbg) {
    const float* __restrict__ x = static_cast<const float*>(x_v);
    const float* __restrict__ addend = static_cast<const float*>(addend_v);
    float* __restrict__ y = static_cast<float*>(y_v);
    static_assert(KS ? (WR == 1 && WC == 1 && NBW <= 3) : (WR * WC == 4), "four waves");
    constexpr int RB = 8 / WR;  // 16-row blocks per wave
    __shared__ __attribute__((aligned(16))) u32x4 img[(kUMax + 1) * 8];
    __shared__ __attribute__((aligned(16))) uint16_t lidx_s[27 * kTile];
    __shared__ int32_t uniq_s[kUMax];
    __shared__ int32_t rows_s[kTile];
    __shared__ uint32_t bm_s[32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int wr = KS ? 0 : wave / WC, wc = KS ? 0 : wave % WC;
    // Workgroup -> (tile, column group): blockIdx % 8 is the XCD (its own L2).  An XCD walks RUNS of `run` consecutive
    // tiles, column groups of a tile side by side: consecutive tiles are spatial neighbours (Morton order) whose halo rows
    // overlap, so what one stages the next finds in the same L2 instead of pulling it over the fabric again.
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int cg = seq % n_cg, tl = seq / n_cg;
    const int tile = (tl / run) * (8 * run) + xcd * run + (tl % run);
    if (tile >= n_tiles) return;
    const int cb_n = (cin + 31) >> 5, nb_n = cout >> 4;
    const int nb0 = (cg * WC + wc) * NBW;

    const int U = pv.ucount[tile];
    const uint32_t tmask = pv.tilemask[tile];
    const bool direct = U > kUMax;
    if (tid < kTile) rows_s[tid] = pv.row_order[(int64_t)tile * kTile + tid];
    if (tid < 32) bm_s[tid] = pv.blkmask[(int64_t)tile * 32 + tid];
    if (tid < 8) img[kUMax * 8 + tid] = (u32x4){0u, 0u, 0u, 0u};
    if (!direct) {
        const u32x4* src = reinterpret_cast<const u32x4*>(pv.lidx + (int64_t)tile * (27 * kTile));
        u32x4* dst = reinterpret_cast<u32x4*>(lidx_s);
        for (int i = tid; i < 27 * kTile * 2 / 16; i += kThreads) dst[i] = src[i];
        for (int i = tid; i < U; i += kThreads) uniq_s[i] = pv.uniq[(int64_t)tile * kUMax + i];
    }

    f32x4 acc[RB][NBW];
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
        const float b = bias && (!KS || wave == 0) ? bias[(nb0 + n) * 16 + c16] : 0.0f;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb][n] = (f32x4){b, b, b, b};
    }
    __syncthreads();

    // B fragments of chunk (k, cb): this wave's NBW column blocks, hi and lo, one 16-B load per lane each
    u32x4 bfr[2][NBW][2];
    auto load_b = [&](auto S, int k, int cb) {
        const u32x4* src = wp + (((int64_t)k * cb_n + cb) * nb_n + nb0) * 128 + lane;
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
            bfr[S][n][0] = src[(n * 2 + 0) * 64];
            bfr[S][n][1] = src[(n * 2 + 1) * 64];
        }
    };
    // image <- one 32-channel slice of n_rows rows; row_of(u) = global input row or -1 (zeros)
    auto stage = [&](int cb, int n_rows, auto row_of) {
        const int n_items = n_rows * 4;
        for (int i0 = 0; i0 < n_items; i0 += kThreads * 4) {
            f32x4 r[4][IO == 2 ? 1 : 2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * kThreads + tid;
                const int u = i >> 2, q = i & 3;
                const int col = cb * 32 + q * 8;
                const int32_t row = i < n_items ? row_of(u) : -1;
                r[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (IO != 2) r[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (row >= 0 && col < cin) {
                    if constexpr (IO == 2) {  // 8 bf16 = one 16-B piece
                        r[j][0] = *reinterpret_cast<const f32x4*>(static_cast<const __bf16*>(x_v) + (int64_t)row * cin + col);
                    } else {
                        const f32x4* p = reinterpret_cast<const f32x4*>(x + (int64_t)row * cin + col);
                        r[j][0] = p[0];
                        r[j][1] = p[1];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * kThreads + tid;
                if (i < n_items) {
                    const int u = i >> 2, q = i & 3;
                    const int s0 = image_slot(u) ^ q;
                    if constexpr (IO == 2) {
                        img[s0] = __builtin_bit_cast(u32x4, r[j][0]);
                    } else {
                        u32x4 hi, lo;
                        split8(r[j][0], r[j][1], &hi, &lo);
                        img[s0] = hi;
                        img[s0 ^ 4] = lo;
                    }
                }
            }
        }
    };
    // the MFMAs of one chunk: A fragments from the image (double-buffered: the reads of row block rb + 1 are in flight under
    // the MFMAs of block rb, issued whether or not that block is active -- a read is cheaper than a pipeline bubble), B
    // fragments from register set S
    auto multiply = [&](auto S, uint32_t bmw, const uint32_t* li /* RB image slots of this lane's rows */) {
        if (bmw == 0u) return;
        if (TILE_DBG(dbg) & 8) bmw = (1u << RB) - 1u;
        bf16x8 a_hi[2], a_lo[2];
        {
            const uint32_t s = (TILE_DBG(dbg) & 4) ? (uint32_t)lane : li[0] ^ (uint32_t)g;
            a_hi[0] = __builtin_bit_cast(bf16x8, img[s]);
            if constexpr (IO != 2) a_lo[0] = __builtin_bit_cast(bf16x8, img[s ^ 4u]);
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if (rb + 1 < RB) {
                const uint32_t s = (TILE_DBG(dbg) & 4) ? (uint32_t)(lane + 8 * rb) : li[rb + 1] ^ (uint32_t)g;
                a_hi[(rb + 1) & 1] = __builtin_bit_cast(bf16x8, img[s]);
                if constexpr (IO != 2) a_lo[(rb + 1) & 1] = __builtin_bit_cast(bf16x8, img[s ^ 4u]);
            }
            if ((bmw >> rb) & 1u) {
#pragma unroll
                for (int n = 0; n < NBW; ++n) {
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, bfr[S][n][0]);
                    const bf16x8 bl = __builtin_bit_cast(bf16x8, bfr[S][n][1]);
                    if constexpr (IO != 2) acc[rb][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[rb & 1], bh, acc[rb][n], 0, 0, 0);
                    acc[rb][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[rb & 1], bl, acc[rb][n], 0, 0, 0);
                    acc[rb][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[rb & 1], bh, acc[rb][n], 0, 0, 0);
                }
            }
        }
    };
    auto block_mask = [&](int k) {
        return (uint32_t)__builtin_amdgcn_readfirstlane((bm_s[k] >> (wr * RB)) & ((1u << RB) - 1u));
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    if (tmask != 0u && !direct) {
        // image slots of this lane's rows at offset k: RB uint16, contiguous in the [offset][row % 16][row / 16] table
        auto slots_of = [&](int k, uint32_t* li) {
            const uint16_t* p = lidx_s + (k * 16 + c16) * 8 + wr * RB;
            if constexpr (RB == 8) {
                const u32x4 w = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    li[2 * i] = w[i] & 0xFFFFu;
                    li[2 * i + 1] = w[i] >> 16;
                }
            } else if constexpr (RB == 4) {
                const u32x2 w = *reinterpret_cast<const u32x2*>(p);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    li[2 * i] = w[i] & 0xFFFFu;
                    li[2 * i + 1] = w[i] >> 16;
                }
            } else {
                const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
                li[0] = w & 0xFFFFu;
                li[1] = w >> 16;
            }
        };
        uint32_t li[2][RB];
        uint32_t bmw[2];
        uint32_t itmask = tmask;  // the offsets this wave multiplies
        if constexpr (KS) {       // every 4th active one
            itmask = 0u;
            uint32_t t = tmask;
            for (int j = 0; t != 0u; ++j) {
                const int k = __builtin_ctz(t);
                t &= t - 1;
                if ((j & 3) == wave) itmask |= 1u << k;
            }
            itmask = (uint32_t)__builtin_amdgcn_readfirstlane(itmask);
        }
        if (itmask == 0u) {  // (KS, fewer than four active offsets: nothing to multiply, but the image refills need every wave)
            for (int cb = 0; cb < cb_n; ++cb) {
                __syncthreads();
                stage(cb, U, [&](int u) { return uniq_s[u]; });
                __syncthreads();
            }
        } else {
        struct Chunk {
            int k, cb;
            bool have, first;
        };
        uint32_t it_todo = itmask;
        int it_cb = 0;
        bool it_first = true;
        auto next_chunk = [&]() {
            Chunk c;
            c.have = true;
            c.first = it_first;
            it_first = false;
            if (it_todo == 0u) {
                it_cb += 1;
                it_todo = itmask;
                c.have = it_cb < cb_n;
                c.first = true;
            }
            c.k = __builtin_ctz(it_todo);
            it_todo &= it_todo - 1;
            c.cb = it_cb;
            return c;
        };
        // One chunk: (refill the image at a slice boundary,) request the B fragments and the image slots of the NEXT chunk,
        // multiply the current one.  The next chunk's loads are issued unconditionally (past the end: the last chunk again)
        // so that the count of loads in flight is the same on every path and the compiler waits for exactly the current set.
        auto run = [&](auto SC, auto SN, const Chunk& cur, Chunk& nxt) {
            if (cur.first && !((TILE_DBG(dbg) & 1) && cur.cb > 0)) {  // new 32-channel slice: every wave is done with the old image, refill it
                __syncthreads();
                stage(cur.cb, U, [&](int u) { return uniq_s[u]; });
                __syncthreads();
            }
            nxt = next_chunk();
            const int kn = nxt.have ? nxt.k : cur.k, cbn = nxt.have ? nxt.cb : cur.cb;
            if (!(TILE_DBG(dbg) & 2)) load_b(SN, kn, cbn);
            slots_of(kn, li[SN]);
            bmw[SN] = block_mask(kn);
            multiply(SC, bmw[SC], li[SC]);
        };
        Chunk a = next_chunk(), b;
        load_b(S0{}, a.k, a.cb);
        slots_of(a.k, li[0]);
        bmw[0] = block_mask(a.k);
        for (;;) {
            run(S0{}, S1{}, a, b);
            if (!b.have) break;
            run(S1{}, S0{}, b, a);
            if (!a.have) break;
        }
        }
    } else if (tmask != 0u) {
        // direct tile: the image holds the 128 neighbour rows of ONE offset at a time, image row = tile position
        uint32_t li[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) li[rb] = (uint32_t)image_slot((wr * RB + rb) * 16 + c16);
        for (int cb = 0; cb < cb_n; ++cb) {
            uint32_t todo = tmask;
            for (int j = 0; todo != 0u; ++j) {
                const int k = __builtin_ctz(todo);
                todo &= todo - 1;
                __syncthreads();
                stage(cb, kTile, [&](int u) {
                    const int32_t r = rows_s[u];
                    return r >= 0 ? nbr[(int64_t)k * m_out + r] : -1;
                });
                const bool mine = !KS || (j & 3) == wave;
                if (mine) load_b(S0{}, k, cb);
                __syncthreads();
                if (mine) multiply(S0{}, block_mask(k), li);
            }
        }
    }

    // one output row's NBW x 16 columns of this lane: (+ residual) -> activation -> store in the output's storage type
    auto store_row = [&](int32_t orow, const float* v) {
        if constexpr (IO == 0) {
            float* yr = y + (int64_t)orow * cout + nb0 * 16 + c16;
            const float* ar = addend ? addend + (int64_t)orow * cout + nb0 * 16 + c16 : nullptr;
#pragma unroll
            for (int n = 0; n < NBW; ++n) {
                float o = v[n];
                if (ar) o += ar[n * 16];
                yr[n * 16] = relu ? (o < 0.0f ? 0.0f : o) : o;  // (a NaN stays a NaN, as torch.relu)
            }
        } else {
            __bf16* yb = static_cast<__bf16*>(y_v) + (int64_t)orow * cout + nb0 * 16 + c16;
            const __bf16* ab = addend_v ? static_cast<const __bf16*>(addend_v) + (int64_t)orow * cout + nb0 * 16 + c16 : nullptr;
#pragma unroll
            for (int n = 0; n < NBW; ++n) {
                const float o = v[n] + (ab ? (float)ab[n * 16] : 0.0f);
                yb[n * 16] = (__bf16)(relu ? (o < 0.0f ? 0.0f : o) : o);
            }
        }
    };
    if constexpr (KS) {
        // sum the four waves' partial tiles in a fixed order: half the row blocks at a time through the image's memory
        // ([wave][row block][column block][lane] float4: no bank conflicts), wave w finishing row block 4 * half + w
        f32x4* buf = reinterpret_cast<f32x4*>(img);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < NBW; ++n) buf[((wave * 4 + j) * NBW + n) * 64 + lane] = acc[half * 4 + j][n];
            __syncthreads();
            f32x4 sum[NBW];
#pragma unroll
            for (int n = 0; n < NBW; ++n) {
                sum[n] = buf[((0 * 4 + wave) * NBW + n) * 64 + lane];
#pragma unroll
                for (int w = 1; w < 4; ++w) sum[n] = sum[n] + buf[((w * 4 + wave) * NBW + n) * 64 + lane];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int32_t orow = rows_s[(half * 4 + wave) * 16 + g * 4 + r];
                if (orow >= 0) {
                    float v[NBW];
#pragma unroll
                    for (int n = 0; n < NBW; ++n) v[n] = sum[n][r];
                    store_row(orow, v);
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
            const int32_t orow = rows_s[(wr * RB + rb) * 16 + g * 4 + r];
            if (orow >= 0) {
                float v[NBW];
#pragma unroll
                for (int n = 0; n < NBW; ++n) v[n] = acc[rb][n][r];
                store_row(orow, v);
            }
        }
}

// timing experiments only (SEG3D_TILE_DBG bit mask, results become WRONG): 1 = image staged for the first slice only,
// 2 = B fragments never reloaded, 4 = A fragments read from fixed conflict-free slots, 8 = no block skipping.  Compiled in
// only by a -DSEG3D_TILE_DBG build (tools/r4_tile_dbg.sh); the shipped library ignores the variable and its kernels carry
// no trace of the experiments (TILE_DBG(x) is the constant 0).
#ifdef SEG3D_TILE_DBG
static const int g_tile_dbg = [] {
    const char* e = getenv("SEG3D_TILE_DBG");
    return e ? atoi(e) : 0;
}();
#else
static const int g_tile_dbg = 0;
#endif

// SEG3D_TILE_RUN (A/B): consecutive tiles per XCD run (0 or anything outside 1..64 = automatic)
static const int g_tile_run = [] {
    const char* e = getenv("SEG3D_TILE_RUN");
    const int v = e ? atoi(e) : 0;
    return (v >= 1 && v <= 64) ? v : 0;
}();

template <int WR, int WC, int NBW, bool KS = false>
int launch_tile(const void* x, const int32_t* nbr, const PlanView& pv, int64_t m_out, const void* wp, const float* bias,
                const void* addend, int cin, int cout, void* y, int relu, int io, hipStream_t st) {
    const int n_tiles = (int)ceil_div64(m_out, kTile);
    const int n_cg = cout / (WC * NBW * 16);
    int run = n_tiles / 64;  // ~8 runs per XCD: the tail imbalance stays below an eighth
    run = run < 1 ? 1 : (run > 16 ? 16 : run);
    if (g_tile_run > 0) run = g_tile_run;
    const unsigned grid = (unsigned)(ceil_div64(n_tiles, 8 * run) * 8 * run * n_cg);
    if (io == 1)
        hipLaunchKernelGGL((spconv_tile_kernel<WR, WC, NBW, KS, 1>), dim3(grid), dim3(kThreads), 0, st, x, nbr, pv, m_out, n_tiles,
                           n_cg, run, reinterpret_cast<const u32x4*>(wp), bias, addend, cin, cout, y, relu, g_tile_dbg);
    else if (io == 2)
        hipLaunchKernelGGL((spconv_tile_kernel<WR, WC, NBW, KS, 2>), dim3(grid), dim3(kThreads), 0, st, x, nbr, pv, m_out, n_tiles,
                           n_cg, run, reinterpret_cast<const u32x4*>(wp), bias, addend, cin, cout, y, relu, g_tile_dbg);
    else
        hipLaunchKernelGGL((spconv_tile_kernel<WR, WC, NBW, KS, 0>), dim3(grid), dim3(kThreads), 0, st, x, nbr, pv, m_out, n_tiles,
                           n_cg, run, reinterpret_cast<const u32x4*>(wp), bias, addend, cin, cout, y, relu, g_tile_dbg);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

bool sort_scratch_bytes(int64_t n, size_t* bytes) {
    *bytes = 0;
    if (n == 0) return true;
    rocprim::double_buffer<SortKey> k(nullptr, nullptr);
    rocprim::double_buffer<uint32_t> v(nullptr, nullptr);
    return rocprim::radix_sort_pairs(nullptr, *bytes, k, v, (size_t)n, 0u, kKeyBits) == hipSuccess;
}

}  // namespace

// which (cin, cout) the tile schedule takes: 0 = none, else the wave layout id
static int tile_layout(int cin, int cout) {
    if (cin <= 0 || cout <= 0 || (cin & 7) || (cout & 15)) return 0;
    if (cout % 192 == 0) return 1;  // 1 x 4 waves of 128 rows x 48 columns
    if (cout % 128 == 0) return 5;  // 1 x 4 waves of 128 rows x 32 columns
    if (cout % 96 == 0) return 2;   // 2 x 2 waves of 64 rows x 48 columns
    if (cout == 48) return 6;       // four waves split the chunks of one 128 x 48 tile
    if (cout == 32) return 7;       // ... 128 x 32
    return 0;  // (64 columns and other widths: every layout re-stages the image per 32-column group; the per-pair kernel wins)
}

// SEG3D_TILE_LAYOUT (A/B): force a layout id where it divides cout
static const int g_tile_layout = [] {
    const char* e = getenv("SEG3D_TILE_LAYOUT");
    const int v = e ? atoi(e) : 0;
    return (v >= 1 && v <= 5) ? v : 0;  // anything else: automatic
}();

int spconv_tile_fwd(const void* x, const int32_t* nbr, const void* plan, int64_t m_out, const void* wp, const float* bias,
                    const void* addend, int cin, int cout, void* y, int relu, int io, hipStream_t st) {
    const int64_t n_tiles = ceil_div64(m_out, kTile);
    const PlanView pv = plan_view(const_cast<void*>(plan), n_tiles);
    int layout = tile_layout(cin, cout);
    // Few tiles (the 19 k-row level: 153): 192-column workgroups give 1.2 per CU -- a fifth of the CUs runs two, the rest
    // one and idles -- 128-column workgroups 1.8 per CU at two thirds of the work each.
    if (layout == 1 && cout % 128 == 0 && n_tiles * (cout / 192) < 400) layout = 5;
    if (g_tile_layout == 5 && cout % 128 == 0) layout = 5;
    if (g_tile_layout == 1 && cout % 192 == 0) layout = 1;
    if (g_tile_layout == 2 && cout % 96 == 0) layout = 2;
    if (g_tile_layout == 3 && layout >= 6) layout = cout % 48 == 0 ? 3 : 4;  // narrow layers without the chunk split
    if (g_tile_layout == 4 && cout % 32 == 0) layout = 4;
    switch (layout) {
        case 1: return launch_tile<1, 4, 3>(x, nbr, pv, m_out, wp, bias, addend, cin, cout, y, relu, io, st);
        case 2: return launch_tile<2, 2, 3>(x, nbr, pv, m_out, wp, bias, addend, cin, cout, y, relu, io, st);
        case 3: return launch_tile<4, 1, 3>(x, nbr, pv, m_out, wp, bias, addend, cin, cout, y, relu, io, st);
        case 4: return launch_tile<4, 1, 2>(x, nbr, pv, m_out, wp, bias, addend, cin, cout, y, relu, io, st);
        case 5: return launch_tile<1, 4, 2>(x, nbr, pv, m_out, wp, bias, addend, cin, cout, y, relu, io, st);
        case 6: return launch_tile<1, 1, 3, true>(x, nbr, pv, m_out, wp, bias, addend, cin, cout, y, relu, io, st);
        case 7: return launch_tile<1, 1, 2, true>(x, nbr, pv, m_out, wp, bias, addend, cin, cout, y, relu, io, st);
        default: return SEG3D_EINVAL;
    }
}

extern "C" {

int32_t seg3d_spconv_tiled_supported(int32_t cin, int32_t cout) { return tile_layout(cin, cout) != 0 ? 1 : 0; }

size_t seg3d_conv_plan_bytes(int64_t m_out) {
    if (m_out <= 0 || m_out >= (int64_t)0x7FFFFFF0) return 0;
    return plan_bytes(ceil_div64(m_out, kTile));
}

size_t seg3d_conv_plan_workspace_bytes(int64_t m_out) {
    if (m_out <= 0 || m_out >= (int64_t)0x7FFFFFF0) return 0;
    size_t tmp = 0;
    if (!sort_scratch_bytes(m_out, &tmp)) return 0;
    return 2 * align_up((size_t)m_out * sizeof(SortKey), 256) + 2 * align_up((size_t)m_out * sizeof(uint32_t), 256) +
           align_up(tmp, 256) + 256;
}

int seg3d_conv_plan_build(const int32_t* coords, const int32_t* nbr, int64_t m_out, void* plan, void* workspace,
                          size_t workspace_bytes, void* stream) {
    if (m_out < 0 || m_out >= (int64_t)0x7FFFFFF0) return SEG3D_EINVAL;
    if (m_out == 0) return SEG3D_OK;
    if (!coords || !nbr || !plan) return SEG3D_EINVAL;
    if (!workspace || workspace_bytes < seg3d_conv_plan_workspace_bytes(m_out)) return SEG3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    size_t tmp_bytes = 0;
    if (!sort_scratch_bytes(m_out, &tmp_bytes)) return SEG3D_EWORKSPACE;
    WsCarver ws(workspace);
    SortKey* k0 = ws.take<SortKey>((size_t)m_out);
    SortKey* k1 = ws.take<SortKey>((size_t)m_out);
    uint32_t* v0 = ws.take<uint32_t>((size_t)m_out);
    uint32_t* v1 = ws.take<uint32_t>((size_t)m_out);
    void* tmp = ws.take<char>(tmp_bytes);
    hipLaunchKernelGGL(morton_keys, dim3((unsigned)ceil_div64(m_out, kThreads)), dim3(kThreads), 0, st, coords, (int)m_out, k0, v0);
    SEG3D_CHECK_LAUNCH();
    rocprim::double_buffer<SortKey> kb(k0, k1);
    rocprim::double_buffer<uint32_t> vb(v0, v1);
    SEG3D_CHECK_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, kb, vb, (size_t)m_out, 0u, kKeyBits, st));
    const int64_t n_tiles = ceil_div64(m_out, kTile);
    hipLaunchKernelGGL(plan_kernel, dim3((unsigned)n_tiles), dim3(kThreads), 0, st, vb.current(), nbr, m_out,
                       plan_view(plan, n_tiles));
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_spconv_fwd_tiled(const float* x, const int32_t* nbr, const void* plan, int64_t m_out, int64_t m_in,
                           const void* w_packed, int32_t pack_flags, const float* bias, const float* addend, int32_t relu,
                           int32_t cin, int32_t cout, float* y, void* stream) {
    if (m_out < 0 || m_in < 0 || !w_packed || !(pack_flags & 4) || tile_layout(cin, cout) == 0) return SEG3D_EINVAL;
    if (m_out == 0) return SEG3D_OK;
    if (!x || !nbr || !plan || !y || m_out >= (int64_t)0x7FFFFFF0) return SEG3D_EINVAL;
    return spconv_tile_fwd(x, nbr, plan, m_out, w_packed, bias, addend, cin, cout, y, relu ? 1 : 0, 0, as_stream(stream));
}

/* The tiled schedule with the feature maps STORED in bf16 (seg3d_spconv_fwd_act_bf16's contract: x float32 (x_bf16 = 0) or
 * bf16 (x_bf16 = 1) rows, y and the residual addend bf16 [m_out, cout]; accumulation, bias and activation float32). */
int seg3d_spconv_fwd_tiled_bf16(const void* x, int32_t x_bf16, const int32_t* nbr, const void* plan, int64_t m_out, int64_t m_in,
                                const void* w_packed, int32_t pack_flags, const float* bias, const void* addend_bf16,
                                int32_t relu, int32_t cin, int32_t cout, void* y_bf16, void* stream) {
    if (m_out < 0 || m_in < 0 || !w_packed || !(pack_flags & 4) || tile_layout(cin, cout) == 0) return SEG3D_EINVAL;
    if (m_out == 0) return SEG3D_OK;
    if (!x || !nbr || !plan || !y_bf16 || m_out >= (int64_t)0x7FFFFFF0) return SEG3D_EINVAL;
    return spconv_tile_fwd(x, nbr, plan, m_out, w_packed, bias, addend_bf16, cin, cout, y_bf16, relu ? 1 : 0, x_bf16 ? 2 : 1,
                           as_stream(stream));
}

}  // extern "C"
