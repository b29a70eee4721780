// a6 / a22: dense Linear layers y = x W^T + b on [rows, C] activations with C_in <= 192 -- the "row streaming" schedule
// (round 5).  Reference: the nn.Linear calls of seg3d/models/layers/point_transformer_layer.py:260-298 (MLP, norms'
// producers) and cosine_msa.py:58-63,403 (in- / out-projection), forward and input gradient.
//
// Why another schedule.  A Linear layer used to run as the single-offset case of the gather-GEMM (spconv_split.hip): per
// 32-channel chunk a workgroup stages W fragments through LDS, every wave loads ITS rows' 32 channels, one barrier, 72 MFMAs.
// With K = 192 a workgroup lives for 6 chunks, and each chunk waits for a full memory round trip with one chunk (16 KB of
// rows) in flight: 34.7 us per launch at 58 k rows x 192 -> 192 where the bytes (45 MB in, 45 MB out) take 15 us -- the
// kernel is bound by latency x bytes in flight, not by the matrix pipe (29 % busy) and not by HBM.
//
// Here a PERSISTENT workgroup parks its column block of W (K <= 192 input channels x up to 192 columns, hi and lo
// fragments in the packed stream's own order: up to 144 KB of the CU's 160 KB LDS) once, and every wave streams its OWN 16-row
// tiles through it: the 6 chunk loads of a wave's NEXT tile are issued one by one as the current tile's chunks are converted
// (a rolling prefetch: a full tile of rows, 12 KB per wave and 96 KB per CU of DISTINCT rows, is always in flight), B
// fragments come from LDS with one conflict-free ds_read_b128 per fragment, no barrier after the prologue, 216 MFMAs per
// tile.  The product is taken transposed (W fragment as the MFMA's first operand), so that a lane's accumulator is 16
// contiguous bytes of y: 12 instead of 48 vector-memory instructions per tile for the stores, and as few for the addend.
// (Tried first: the W slice of a 48-column wave in 144 VGPRs, rows shared by four waves -- 256 VGPRs, spills as soon as the
// prefetch is pinned in place, and the four waves of a workgroup keep the SAME rows in flight.)
// Products, their order and the accumulator's start (the bias) are those of the old kernel: results are bit-identical.
#include <atomic>
#include <cstdlib>

#include "common.hpp"

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

// 8 floats -> (hi, lo) bf16 fragments; hi = RNE(x), lo = RNE(x - hi)   (the arithmetic of spconv_split.hip's split8)
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

// KC: 32-channel chunks of the input (ceil(cin / 32) == KC); NB: 16-column blocks of the workgroup's column block.
// IO 0: y = x W^T + b (+ addend); IO 4: y = (x W^T) * factor (the "addend" pointer is the elementwise factor: fc2's input
// gradient times gelu').
template <int KC, int NB, int IO>
__global__ __launch_bounds__(512) void linear_stream_kernel(const float* __restrict__ x, int64_t m, const uint4* __restrict__ wp,
                                                            const float* __restrict__ bias, const float* __restrict__ addend,
                                                            int cin, int cout, float* __restrict__ y, int64_t n_tiles) {
    extern __shared__ __attribute__((aligned(16))) uint4 wlds[];  // [KC][NB][hi | lo][64 lanes], then the block's bias [NB * 16]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int nb_n = cout >> 4;
    const int nb0 = (int)blockIdx.y * NB;
    // ---- the workgroup's column block of W, once
    for (int i = threadIdx.x; i < KC * NB * 128; i += blockDim.x) {
        const int c = i / (NB * 128), rem = i - c * (NB * 128);
        wlds[i] = wp[((int64_t)c * nb_n + nb0) * 128 + rem];
    }
    float* blds = reinterpret_cast<float*>(wlds + KC * NB * 128);
    for (int i = threadIdx.x; i < NB * 16; i += blockDim.x) blds[i] = bias ? bias[nb0 * 16 + i] : 0.0f;
    __syncthreads();

    const int64_t last_row = m - 1;
    f32x4 raw[KC][2];
    // chunk c of tile t: this lane's row (tile row c16), channels 32 c + 8 g .. + 7 (clamped in bounds where the last
    // chunk is partial: those lanes' values are replaced by zeros before the split)
    auto issue = [&](int c, int64_t t) {
        const int64_t row = t * 16 + c16 < m ? t * 16 + c16 : last_row;
        const int col = c * 32 + g * 8 < cin ? c * 32 + g * 8 : 0;
        const f32x4* p = reinterpret_cast<const f32x4*>(x + row * cin + col);
        raw[c][0] = p[0];
        raw[c][1] = p[1];
    };
    const int64_t stride = (int64_t)gridDim.x * n_waves;
    int64_t tile = (int64_t)blockIdx.x * n_waves + wave;
    if (tile < n_tiles) {
#pragma unroll
        for (int c = 0; c < KC; ++c) issue(c, tile);
    }
    for (; tile < n_tiles; tile += stride) {
        // the next tile's rows are requested unconditionally (past the end: this tile's again), so that the number of
        // loads in flight is the same on every path and the compiler's waits are exact
        const int64_t nt = tile + stride < n_tiles ? tile + stride : tile;
        // The product is taken TRANSPOSED -- W fragment as the MFMA's first operand, the row fragment as its second (both
        // have the same lane layout, and a * b = b * a exactly, so the sums are the old kernel's bit for bit): the accumulator
        // of lane (g, c16) then holds row c16 of the tile, columns 4 g .. 4 g + 3 of a column block = 16 contiguous bytes of
        // y: one store (and one addend load) per lane and block instead of four.  Accumulators start at the bias.
        f32x4 acc[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[n] = *reinterpret_cast<const f32x4*>(blds + n * 16 + 4 * g);
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
            const bool in_range = c * 32 + g * 8 < cin;
            bf16x8 a_hi, a_lo;
            split8(in_range ? raw[c][0] : z, in_range ? raw[c][1] : z, &a_hi, &a_lo);
            issue(c, nt);
            // the request for the next tile's chunk stays HERE, in front of this chunk's MFMAs (left to itself the scheduler
            // sinks the loads behind the tile's last MFMA, where nothing is left to cover their latency)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const bf16x8 bh = __builtin_bit_cast(bf16x8, wlds[((c * NB + n) * 2 + 0) * 64 + lane]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, wlds[((c * NB + n) * 2 + 1) * 64 + lane]);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, a_lo, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, a_hi, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, a_hi, acc[n], 0, 0, 0);
            }
        }
        const int64_t opos = tile * 16 + c16;
        if (opos < m) {
            float* yr = y + opos * cout + nb0 * 16 + 4 * g;
            const float* ar = addend ? addend + opos * cout + nb0 * 16 + 4 * g : nullptr;
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                f32x4 v = acc[n];
                if (ar) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(ar + n * 16);
                    v = IO == 4 ? v * a : v + a;
                }
                *reinterpret_cast<f32x4*>(yr + n * 16) = v;
            }
        }
    }
}

template <int KC, int NB>
int launch_stream(const float* x, int64_t m, const void* wp, const float* bias, const float* addend, int cin, int cout, float* y,
                  int io, hipStream_t st) {
    const size_t lds = (size_t)KC * NB * 2048 + (size_t)NB * 64;  // W block + its bias
    const int gy = (cout / 16) / NB;
    const int waves = lds > 72 * 1024 ? 8 : 4;          // one big workgroup per CU, or several small ones
    const int per_cu = lds > 72 * 1024 ? 1 : (lds > 36 * 1024 ? 2 : 4);
    const int64_t n_tiles = ceil_div64(m, 16);
    int64_t gx = (256 * per_cu) / gy;                   // one resident round, shared among the column blocks
    if (gx < 1) gx = 1;
    const int64_t need = ceil_div64(n_tiles, waves);
    if (gx > need) gx = need;
    const dim3 grid((unsigned)gx, (unsigned)gy), block((unsigned)(64 * waves));
    auto go = [&](auto kernel) {
        static bool raised = false;  // per instantiation: dynamic LDS beyond the 64 KB default needs the attribute once
        if (!raised && lds > 64 * 1024) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return 1;
            raised = true;
        }
        hipLaunchKernelGGL(kernel, grid, block, lds, st, x, m, reinterpret_cast<const uint4*>(wp), bias, addend, cin, cout, y, n_tiles);
        return 0;
    };
    const int rc = io == 4 ? go(linear_stream_kernel<KC, NB, 4>) : go(linear_stream_kernel<KC, NB, 0>);
    if (rc) return 1;
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

template <int KC>
int launch_stream_nb(const float* x, int64_t m, const void* wp, const float* bias, const float* addend, int cin, int cout,
                     float* y, int io, hipStream_t st) {
    if (cout % 192 == 0) return launch_stream<KC, 12>(x, m, wp, bias, addend, cin, cout, y, io, st);
    if (cout % 96 == 0) return launch_stream<KC, 6>(x, m, wp, bias, addend, cin, cout, y, io, st);
    return launch_stream<KC, 3>(x, m, wp, bias, addend, cin, cout, y, io, st);
}

}  // namespace

// SEG3D_LINEAR_STREAM=1 (A/B, OFF by default): measured on the headline scene's shapes against the gather-GEMM's single-offset
// case it was meant to replace (tools/linear_bench.py, one box): 48 -> 48 10.1 -> 9.2 us, 96 -> 96 17.3 -> 19.8, 192 -> 96
// 24.9 -> 28.5, 192 -> 192 20.1 -> 28.0, 192 -> 384 40.0 -> 36.8; training step 42.4 -> 43.3 ms, forward 12.73 -> 13.27 ms.
// The premise was wrong: ALONE the old kernel moves 192 -> 192 at 4.5 TB/s (its 34.7 us in the step's profile is what sharing
// the chip with the weight-gradient stream costs), and this schedule's 147 KB W prologue per workgroup (256 workgroups pulling
// 38 MB through L2 before the first row is read) costs what the rolling prefetch saves.  Kept as a parity-tested experiment.
static const bool g_linear_stream = [] {
    const char* e = getenv("SEG3D_LINEAR_STREAM");
    return e && atoi(e) == 1;
}();

// seg3d_debug_set_linear_stream (parity tests pin the opt-in schedule with it): 1 = on, 0 = off, -1 = the environment's choice
static std::atomic<int> g_linear_stream_forced{-1};
extern "C" int seg3d_debug_set_linear_stream(int32_t on) {
    if (on < -1 || on > 1) return SEG3D_EINVAL;
    g_linear_stream_forced.store(on, std::memory_order_relaxed);
    return SEG3D_OK;
}

// Used by spconv_split.hip's dense dispatch.  Returns 1 when the shape is not this schedule's (the caller runs the old one):
// cin a multiple of 8 with ceil(cin / 32) in {2, 3, 4, 6} (48 .. 192 channels), cout a multiple of 48, io 0 or 4.
int linear_stream_fwd(const float* x, int64_t m, const void* wp, const float* bias, const float* addend, int cin, int cout,
                      float* y, int io, hipStream_t st) {
    const int forced = g_linear_stream_forced.load(std::memory_order_relaxed);
    if (!(forced < 0 ? g_linear_stream : forced == 1) || (io != 0 && io != 4) || (cin & 7) || cout % 48 != 0 || m <= 0) return 1;
    if (io == 4 && !addend) return 1;
    switch ((cin + 31) / 32) {
        case 2: return launch_stream_nb<2>(x, m, wp, bias, addend, cin, cout, y, io, st);
        case 3: return launch_stream_nb<3>(x, m, wp, bias, addend, cin, cout, y, io, st);
        case 4: return launch_stream_nb<4>(x, m, wp, bias, addend, cin, cout, y, io, st);
        case 6: return launch_stream_nb<6>(x, m, wp, bias, addend, cin, cout, y, io, st);
        default: return 1;
    }
}
