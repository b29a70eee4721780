// a9-a11 forward / dgrad for the WIDE layers (Cin >= 192: SWFormer levels 3-4 and the decoder's bottlenecks), split-bf16.
//
// spconv_split_kernel converts every gathered fp32 row to bf16 hi + lo in registers, once per (kernel offset, 32 channels,
// column block): at C >= 192 that conversion (48 vector instructions per chunk per wave), the register staging of W and
// the waits on both hold the matrix cores to a third of their rate (profiles/r02_pmc_conv.txt: 3.4 vector instructions
// per MFMA, 41 % of the wave cycles waiting).  Here the conversion happens ONCE per layer, in a separate streaming pass
// (seg3d_spconv_presplit: rows -> bf16 hi plane | lo plane, plus one all-zero row that every inactive table entry points
// to), and the gather-GEMM moves both operands global -> LDS by LDS-DMA (global_load_lds_dwordx4: the per-lane SOURCE
// address makes it a row gather, the destination is lane-linear), so the main loop is index arithmetic, LDS fragment reads
// and MFMAs only -- no operand touches a vector register before it is an MFMA fragment.
//   tile: 4 waves x 32 output rows x (NBT x 16) columns, output-stationary (no atomics), chunk = (offset k, 32 channels);
//   per chunk a wave issues 4 DMA pieces for ITS OWN rows (2 row blocks x {hi, lo}: 16 rows x 64 B = 1 KiB each) and its
//   share of the NBT x 2 W pieces; chunk c + 1 is in flight (other LDS buffer) while chunk c is multiplied; the DMAs are
//   inline asm, so the compiler does not drain them in front of the fragment reads -- one hand-placed vmcnt(0) + barrier
//   per chunk;  tile-level skipping of offsets no row of the tile uses, optional processing order of the rows (parity-
//   grouped strided / inverse tables), bias / addend / ReLU epilogue exactly as spconv_split_kernel.
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

// x [m, c] fp32 -> xs: hi plane [m + 1, c] bf16 | lo plane [m + 1, c] bf16, row m of both = 0.  One thread = 8 channels.
__global__ __launch_bounds__(256) void presplit_kernel(const float* __restrict__ x, int64_t m, int c, __bf16* __restrict__ xs) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (m + 1) * (c / 8);
    if (t >= total) return;
    const int64_t plane = (m + 1) * (int64_t)c;
    u32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
    if (t < m * (c / 8)) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(x + t * 8), q = *reinterpret_cast<const f32x4*>(x + t * 8 + 4);
        const float v[8] = {p[0], p[1], p[2], p[3], q[0], q[1], q[2], q[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w = pack_bf16(v[2 * i], v[2 * i + 1]);
            hi[i] = w;
            lo[i] = pack_bf16(v[2 * i] - __builtin_bit_cast(float, w << 16), v[2 * i + 1] - __builtin_bit_cast(float, w & 0xFFFF0000u));
        }
    }
    *reinterpret_cast<u32x4*>(xs + t * 8) = hi;
    *reinterpret_cast<u32x4*>(xs + plane + t * 8) = lo;
}

// 16 bytes per lane global -> LDS (lane l lands at lds_base + 16 l); lds_base must be wave-uniform (an SGPR).
// M0 carries the LDS base and is compiler-reserved: saved and restored inside the statement (cdna_hip_programming 5.7).
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_base) {
    uint32_t keep;
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);  // uniform by construction (wave id, buffer), now provably so
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base)
                 : "memory");
}

template <int NBT>
__global__ __launch_bounds__(256, 2) void spconv_dma_kernel(const __bf16* __restrict__ xs, int64_t m_in,
                                                            const int32_t* __restrict__ nbr, int64_t m_out,
                                                            const uint4* __restrict__ wp, const float* __restrict__ bias,
                                                            const float* __restrict__ addend,
                                                            const int32_t* __restrict__ row_order, int cin, int cout,
                                                            float* __restrict__ y, int relu) {
    constexpr int kW = 4, RB = 2;
    constexpr int kWPieces = NBT * 2;                 // 1-KiB pieces of a W chunk (NBT x {hi, lo})
    constexpr int kWMine = (kWPieces + kW - 1) / kW;  // per wave
    constexpr int kABytes = kW * RB * 2 * 1024;       // A image of a chunk: [wave][rb][hi, lo][16 rows x 64 B]
    constexpr int kBuf = kABytes + kWPieces * 1024;
    __shared__ __attribute__((aligned(16))) char lds[2 * kBuf];  // NBT 12: exactly 80 KiB -> two workgroups per CU
    uint32_t* wave_mask = reinterpret_cast<uint32_t*>(lds);      // 16 B of the first buffer, used before the first DMA

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * (kW * RB * 16) + wave * (RB * 16);
    const int nb0 = blockIdx.y * NBT;
    const int cb_n = cin >> 5, nb_n = cout >> 4;
    const int64_t plane = (m_in + 1) * (int64_t)cin;  // elements between the hi and the lo plane
    // byte offset of the buffers inside the workgroup's LDS allocation (what M0 takes), made provably wave-uniform
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds);

    // ---- which offsets does this tile touch?
    uint32_t my_mask = 0;
    {
        const int64_t pos = row0 + (lane & (RB * 16 - 1));
        const bool ok = pos < m_out;
        const int64_t rc = ok ? (row_order ? (int64_t)row_order[pos] : pos) : m_out - 1;
        int32_t v[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) v[k] = nbr[(int64_t)k * m_out + rc];
#pragma unroll
        for (int k = 0; k < 27; ++k)
            if (__ballot(ok && v[k] >= 0) != 0ull) my_mask |= 1u << k;
    }
    if (lane == 0) wave_mask[wave] = my_mask;
    __syncthreads();
    uint32_t todo = wave_mask[0] | wave_mask[1] | wave_mask[2] | wave_mask[3];
    __syncthreads();  // the masks are read before the first chunk's DMA overwrites them

    f32x4 acc[RB][NBT];
#pragma unroll
    for (int n = 0; n < NBT; ++n) {
        const float b = bias ? bias[(nb0 + n) * 16 + c16] : 0.0f;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb][n] = (f32x4){b, b, b, b};
    }

    if (todo != 0u) {
        // DMA lane role: row r = lane >> 2 of a 16-row block, 16-byte part p = lane & 3 of its 64 B (32 channels)
        const int dr = lane >> 2, dp = lane & 3;
        int64_t grow[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int64_t pos = row0 + rb * 16 + dr;
            grow[rb] = pos < m_out ? (row_order ? (int64_t)row_order[pos] : pos) : -1;
        }
        // neighbour rows of offset k for the lane's DMA rows; none / past the end -> the all-zero row m_in
        auto fetch_idx = [&](int k, int64_t* idx) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int32_t v = grow[rb] >= 0 ? nbr[(int64_t)k * m_out + grow[rb]] : -1;
                idx[rb] = v >= 0 ? (int64_t)v : m_in;
            }
        };
        auto issue = [&](int k, int cb, const int64_t* idx, int buf) {
            const uint32_t base = lds0 + buf * kBuf;
            // A: this wave's two 16-row blocks, hi and lo planes
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const __bf16* src = xs + idx[rb] * cin + cb * 32 + dp * 8;
                dma16(src, base + ((wave * RB + rb) * 2 + 0) * 1024);
                dma16(src + plane, base + ((wave * RB + rb) * 2 + 1) * 1024);
            }
            // W: pieces wave, wave + 4, ... of the chunk's NBT x {hi, lo} (contiguous in the packed stream)
            const uint4* wsrc = wp + (((int64_t)k * cb_n + cb) * nb_n + nb0) * 128;
#pragma unroll
            for (int j = 0; j < kWMine; ++j) {
                const int piece = j * kW + wave;
                if (kWPieces % kW == 0 || piece < kWPieces) dma16(wsrc + piece * 64 + lane, base + kABytes + piece * 1024);
            }
        };

        int k_cur = __builtin_ctz(todo);
        todo &= todo - 1;
        int cb_cur = 0;
        int64_t idx_cur[RB], idx_pre[RB];
        fetch_idx(k_cur, idx_cur);
        if (todo != 0u) fetch_idx(__builtin_ctz(todo), idx_pre);
        issue(k_cur, 0, idx_cur, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        int buf = 0;
        for (;;) {
            int k_nxt = k_cur, cb_nxt = cb_cur + 1;
            bool have_next = true;
            if (cb_nxt == cb_n) {
                cb_nxt = 0;
                if (todo == 0u) {
                    have_next = false;
                } else {
                    k_nxt = __builtin_ctz(todo);
                    todo &= todo - 1;
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) idx_cur[rb] = idx_pre[rb];
                    if (todo != 0u) fetch_idx(__builtin_ctz(todo), idx_pre);
                }
            }
            if (have_next) issue(k_nxt, cb_nxt, idx_cur, buf ^ 1);  // lands in the other buffer while this one is multiplied
            {
                const char* a_img = lds + buf * kBuf + (wave * RB) * 2048;
                const char* w_img = lds + buf * kBuf + kABytes;
                bf16x8 a_hi[RB], a_lo[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {  // A fragment: row c16, channels 8 g .. 8 g + 7
                    a_hi[rb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(a_img + rb * 2048 + c16 * 64 + g * 16));
                    a_lo[rb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(a_img + rb * 2048 + 1024 + c16 * 64 + g * 16));
                }
#pragma unroll
                for (int n = 0; n < NBT; ++n) {
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(w_img + (n * 2 + 0) * 1024 + lane * 16));
                    const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(w_img + (n * 2 + 1) * 1024 + lane * 16));
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        acc[rb][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[rb], bh, acc[rb][n], 0, 0, 0);
                        acc[rb][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[rb], bl, acc[rb][n], 0, 0, 0);
                        acc[rb][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[rb], bh, acc[rb][n], 0, 0, 0);
                    }
                }
            }
            if (!have_next) break;
            // every DMA of the next chunk has landed (they had this chunk's MFMAs to do so), every wave is done reading
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            buf ^= 1;
            k_cur = k_nxt;
            cb_cur = cb_nxt;
        }
    }

    // D layout of v_mfma_f32_16x16x*: row = (lane>>4)*4 + r, col = lane & 15
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t opos = row0 + rb * 16 + g * 4 + r;
            if (opos < m_out) {
                const int64_t orow = row_order ? (int64_t)row_order[opos] : opos;
                float* yr = y + orow * cout + nb0 * 16 + c16;
                if (addend) {
                    const float* ar = addend + orow * cout + nb0 * 16 + c16;
#pragma unroll
                    for (int n = 0; n < NBT; ++n) {
                        const float v = acc[rb][n][r] + ar[n * 16];
                        yr[n * 16] = relu ? (v < 0.0f ? 0.0f : v) : v;  // NaN stays NaN, as torch.relu
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
}

template <int NBT>
int launch_dma(const __bf16* xs, int64_t m_in, const int32_t* nbr, int64_t m_out, const void* wp, const float* bias,
               const float* addend, const int32_t* row_order, int cin, int cout, float* y, int relu, hipStream_t st) {
    dim3 grid((unsigned)ceil_div64(m_out, 128), (unsigned)((cout / 16) / NBT));
    hipLaunchKernelGGL(spconv_dma_kernel<NBT>, grid, dim3(256), 0, st, xs, m_in, nbr, m_out, reinterpret_cast<const uint4*>(wp),
                       bias, addend, row_order, cin, cout, y, relu);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // namespace

extern "C" {

size_t seg3d_spconv_presplit_bytes(int64_t m_in, int32_t cin) {
    if (m_in < 0 || cin <= 0 || (cin & 31)) return 0;
    return (size_t)(m_in + 1) * cin * 2 * sizeof(__bf16);
}

int seg3d_spconv_presplit(const float* x, int64_t m_in, int32_t cin, void* xs, void* stream) {
    if (m_in < 0 || cin <= 0 || (cin & 31) || !xs || (m_in > 0 && !x)) return SEG3D_EINVAL;
    const int64_t items = (m_in + 1) * (cin / 8);
    hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)ceil_div64(items, 256)), dim3(256), 0, as_stream(stream), x, m_in, cin,
                       static_cast<__bf16*>(xs));
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_spconv_fwd_presplit(const void* xs, const int32_t* nbr, int64_t m_out, int64_t m_in, const void* w_packed,
                              int32_t pack_flags, const float* bias, const float* addend, int32_t relu, int32_t cin,
                              int32_t cout, float* y, const int32_t* row_order, void* stream) {
    if (m_out < 0 || m_in < 0 || cin <= 0 || cout <= 0 || (cin & 31) || (cout % 96) || !w_packed || !(pack_flags & 4))
        return SEG3D_EINVAL;
    if (m_out == 0) return SEG3D_OK;
    if (!xs || !nbr || !y) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    const __bf16* x2 = static_cast<const __bf16*>(xs);
    // 192-column tiles while the launch has >= 400 row tiles, 96-column tiles on the small deep levels (more workgroups)
    static const int nbt_env = getenv("SEG3D_CONV_DMA_NBT") ? atoi(getenv("SEG3D_CONV_DMA_NBT")) : 0;
    const int nb = cout / 16;
    int pick = (nb % 12 == 0 && ceil_div64(m_out, 128) >= 400) ? 12 : 6;
    if (nbt_env == 6 || (nbt_env == 12 && nb % 12 == 0)) pick = nbt_env;
    if (pick == 12) return launch_dma<12>(x2, m_in, nbr, m_out, w_packed, bias, addend, row_order, cin, cout, y, relu ? 1 : 0, st);
    return launch_dma<6>(x2, m_in, nbr, m_out, w_packed, bias, addend, row_order, cin, cout, y, relu ? 1 : 0, st);
}

}  // extern "C"
