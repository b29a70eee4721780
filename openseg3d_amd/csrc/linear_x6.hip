// a6 per-point MLPs (point encoder, fusion encoder, classifier trunk): y = x W^T (+ b) at fp32 GRADE on the bf16 matrix
// pipe.  The logits come straight out of these layers at |logit| ~ 200, where the 2^-16 of the three-product split
// (bf16 hi | lo) is the whole 1e-3 budget: rounds 1 - 4 therefore ran them on v_mfma_f32_16x16x4_f32,
// the vector-rate fp32 MFMA (1/16 of the bf16 rate: 0.76 ms of a 12.6 ms forward in seven launches at 39 % of that pipe).
//
// Here every fp32 operand is split THREE ways -- x = h + m + l, each a bf16, 8 + 8 + 8 = the 24 significant bits of a float:
// the split is exact -- and the product keeps every term down to 2^-16 of the leading one:
//     x w  =  h h + (h m + m h) + (m m + h l + l h)  +  [m l + l m + l l: <= 2^-24 |x w|, dropped]
// six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block instead of eight v_mfma_f32_16x16x4_f32 passes per 16 x 16 x 32 at a
// sixteenth of the rate = 6 / 16 of the matrix time.  Each bf16 x bf16 product is exact in the fp32 accumulator, the sums
// are fp32 sums as in the fp32 MFMA: the result differs from the exact-fp32 kernel's by accumulation order only (measured
// against float64 in tests/test_gpu_dense.py: the same 1e-7-grade error as the fp32 kernel's).
//
// Schedule: a workgroup of 4 waves owns 256 rows x 64 columns; a wave 64 rows x 64 columns = 16 accumulator tiles.  The
// product is taken transposed (W fragment first, as linear_stream.hip does): a lane ends up with 4 consecutive columns of one
// row per tile, the four lane groups of a row with one whole 64-byte sector per store instruction.  W's 64-column slab of a 32-channel step (three
// planes, 12 KB) goes through LDS once per workgroup, double-buffered, one barrier per step; the rows come straight from
// global memory into fragments (lane (c16, g): row c16 of a 16-row tile, channels 4 g .. + 3 and 16 + 4 g .. + 3 of the step:
// each 16-byte load instruction reads whole sectors; W is packed in the same channel order), the next step's requested before this step's 96 MFMAs.  Column groups of one row block are neighbours on one XCD
// (they re-read the same rows from its L2).  Optional epilogue: y * scale + shift per column and ReLU (the eval form of the
// BatchNorm1d + ReLU that follows every one of these layers; two roundings, exactly the separate pass's arithmetic).
#include <cstdlib>
#include <type_traits>

#include "attn_common.hpp"

namespace {

using namespace attn;

constexpr int kThreads = 256;
constexpr int kSlab = 4 * 3 * 64;  // uint4 per (32-channel step, 64-column group): [tile j][plane][lane]

// 8 fp32 -> three bf16 fragments with h + m + l == x exactly (round-to-nearest at every level; |m| <= 2^-9 |x|, |l| <= 2^-18 |x|)
__device__ __forceinline__ void split3(const float* v, bf16x8* h, bf16x8* m, bf16x8* l) {
    u32x4 H, M, L;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t wh = pack_bf16(v[2 * i], v[2 * i + 1]);
        const float r0 = v[2 * i] - __builtin_bit_cast(float, wh << 16);
        const float r1 = v[2 * i + 1] - __builtin_bit_cast(float, wh & 0xFFFF0000u);
        const uint32_t wm = pack_bf16(r0, r1);
        const float s0 = r0 - __builtin_bit_cast(float, wm << 16);
        const float s1 = r1 - __builtin_bit_cast(float, wm & 0xFFFF0000u);
        H[i] = wh;
        M[i] = wm;
        L[i] = pack_bf16(s0, s1);
    }
    *h = __builtin_bit_cast(bf16x8, H);
    *m = __builtin_bit_cast(bf16x8, M);
    *l = __builtin_bit_cast(bf16x8, L);
}

// W [cout][cin] (or its transpose [cin][cout] read as the operand of the input gradient) -> fragment stream
// [step kb][group cg][tile j][plane h | m | l][lane][8 bf16]: lane (mi = lane % 16, g = lane / 16) of tile j holds
// column cg * 64 + 16 j + mi, channels kb * 32 + {4 g .. 4 g + 3, 16 + 4 g .. 16 + 4 g + 3} (the order in which the row
// fragments take them: one 16-byte load of a row lane covers a quarter of a 64-byte sector, the four lane groups the whole).
__global__ __launch_bounds__(256) void pack_x6_kernel(const float* __restrict__ w, int cin, int cout, int transpose,
                                                      uint4* __restrict__ wp) {
    const int ncg = cout >> 6;
    const int64_t total = (int64_t)(cin >> 5) * ncg * 4 * 64;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int lane = (int)(t & 63), j = (int)((t >> 6) & 3);
    const int64_t r = t >> 8;
    const int cg = (int)(r % ncg), kb = (int)(r / ncg);
    const int mi = lane & 15, g = lane >> 4;
    const int co = cg * 64 + 16 * j + mi;
    const int ci0 = kb * 32 + 4 * g;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = ci0 + (e < 4 ? e : 12 + e);
        v[e] = transpose ? w[(int64_t)ci * cout + co] : w[(int64_t)co * cin + ci];
    }
    bf16x8 h, m, l;
    split3(v, &h, &m, &l);
    uint4* dst = wp + ((((int64_t)kb * ncg + cg) * 4 + j) * 3) * 64 + lane;
    dst[0] = __builtin_bit_cast(uint4, h);
    dst[64] = __builtin_bit_cast(uint4, m);
    dst[128] = __builtin_bit_cast(uint4, l);
}

// RT = 16-row tiles per wave (4: 256-row workgroups, two per CU; 2: 128-row workgroups, four per CU and the rows requested
// two steps ahead), WPS = resident workgroups per CU the register allocator leaves room for.
template <int RT, int WPS>
__global__ __launch_bounds__(kThreads, WPS) void linear_x6_kernel(const float* __restrict__ x, int64_t m_rows,
                                                                  const uint4* __restrict__ wp, const float* __restrict__ bias,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  int relu, int cin, int cout, float* __restrict__ y, int ncg,
                                                                  int64_t nrb) {
    constexpr int DEPTH = 2;  // row sets (and W slabs) in flight
    __shared__ __attribute__((aligned(16))) uint4 wl[2][kSlab];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    // block -> (row block, column group): blocks are dealt round-robin over the 8 XCDs, so id % 8 labels an L2; the column
    // groups of one row block are consecutive on ONE of them
    const int64_t id = blockIdx.x;
    const int64_t k = id >> 3;
    const int64_t rb = (k / ncg) * 8 + (id & 7);
    const int cg = (int)(k % ncg);
    if (rb >= nrb) return;  // (whole workgroup)
    const int64_t row0 = rb * (64 * RT) + wave * (16 * RT);
    const float* xp[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        int64_t r = row0 + 16 * i + c16;
        r = r < m_rows ? r : m_rows - 1;  // clamped rows are computed and not stored
        xp[i] = x + r * cin + 4 * g;
    }
    const int nkb = cin >> 5;
    const uint4* wsrc = wp + (int64_t)cg * kSlab + tid;  // + kb * ncg * kSlab
    f32x4 raw[DEPTH][RT][2];
    // (slots are compile-time tags: a run-time index would send the register arrays to scratch memory)
    auto issue_rows = [&](int kb, auto slot_tag) {
        constexpr int slot = decltype(slot_tag)::value;
        kb = kb < nkb ? kb : nkb - 1;  // past the end: the last step's again (the count of loads in flight is the same on every path)
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const f32x4* p = reinterpret_cast<const f32x4*>(xp[i] + kb * 32);
            raw[slot][i][0] = p[0];
            raw[slot][i][1] = p[4];  // channels 16 + 4 g ..
        }
    };
    issue_rows(0, std::integral_constant<int, 0>{});
    issue_rows(1, std::integral_constant<int, 1>{});
    u32x4 wreg[2][3];  // W slab of step kb + 1 (slot (kb + 1) & 1), requested a whole step before it is parked in LDS
    auto issue_w = [&](int kb, auto slot_tag) {
        constexpr int slot = decltype(slot_tag)::value;
        kb = kb < nkb ? kb : nkb - 1;
        const u32x4* s = reinterpret_cast<const u32x4*>(wsrc + (int64_t)kb * ncg * kSlab);
        wreg[slot][0] = s[0];
        wreg[slot][1] = s[256];
        wreg[slot][2] = s[512];
    };
    {
        const uint4* s = wsrc;
        wl[0][tid] = s[0];
        wl[0][tid + 256] = s[256];
        wl[0][tid + 512] = s[512];
    }
    issue_w(1, std::integral_constant<int, 1>{});
    f32x4 acc[RT][4];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    auto step = [&](int kb, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        const int cur = kb & 1;
        const bool more = kb + 1 < nkb;  // (uniform)
        bf16x8 xh[RT], xm[RT], xl[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const float v[8] = {raw[SLOT][i][0][0], raw[SLOT][i][0][1], raw[SLOT][i][0][2], raw[SLOT][i][0][3],
                                raw[SLOT][i][1][0], raw[SLOT][i][1][1], raw[SLOT][i][1][2], raw[SLOT][i][1][3]};
            split3(v, &xh[i], &xm[i], &xl[i]);
        }
        // the rows and the W slab two steps ahead are requested in front of this step's MFMAs (nothing in the last two steps:
        // the kernel runs at the L2 -> L1 rate, a repeated request is not free)
        if (kb + 2 < nkb) {  // (uniform)
            issue_rows(kb + DEPTH, slot_tag);
            issue_w(kb + 2, slot_tag);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16x8 wh = __builtin_bit_cast(bf16x8, wl[cur][(j * 3 + 0) * 64 + lane]);
            const bf16x8 wm = __builtin_bit_cast(bf16x8, wl[cur][(j * 3 + 1) * 64 + lane]);
            const bf16x8 wlo = __builtin_bit_cast(bf16x8, wl[cur][(j * 3 + 2) * 64 + lane]);
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                f32x4 a = acc[i][j];
                // smallest terms first
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh[i], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[i], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm[i], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh[i], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm[i], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[i], a, 0, 0, 0);
                acc[i][j] = a;
            }
        }
        if (more) {  // W of step kb + 1, requested during step kb - 1
            u32x4* dst = reinterpret_cast<u32x4*>(&wl[cur ^ 1][tid]);
            dst[0] = wreg[SLOT ^ 1][0];
            dst[256] = wreg[SLOT ^ 1][1];
            dst[512] = wreg[SLOT ^ 1][2];
        }
        __syncthreads();
    };
    if constexpr (DEPTH == 2) {
        int kb = 0;
        for (; kb + 1 < nkb; kb += 2) {
            step(kb, std::integral_constant<int, 0>{});
            step(kb + 1, std::integral_constant<int, 1>{});
        }
        if (kb < nkb) step(kb, std::integral_constant<int, 0>{});
    } else {
        for (int kb = 0; kb < nkb; ++kb) step(kb, std::integral_constant<int, 0>{});
    }

    // lane (c16, g): row 16 i + c16 of the wave's rows, columns cg * 64 + 16 j + 4 g .. + 3 (the four lane groups of a row write
    // one whole 64-byte sector per store instruction)
    const int col0 = cg * 64 + 4 * g;
    f32x4 b4[4], s4[4], t4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b4[j] = bias ? *reinterpret_cast<const f32x4*>(bias + col0 + 16 * j) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if (scale) {
            s4[j] = *reinterpret_cast<const f32x4*>(scale + col0 + 16 * j);
            t4[j] = *reinterpret_cast<const f32x4*>(shift + col0 + 16 * j);
        }
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int64_t r = row0 + 16 * i + c16;
        if (r >= m_rows) continue;
        float* yr = y + r * cout + col0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = acc[i][j];
            if (bias) v = v + b4[j];
            if (scale) v = v * s4[j] + t4[j];  // (-ffp-contract=off: a multiply and an add, as the separate pass rounds)
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            *reinterpret_cast<f32x4*>(yr + 16 * j) = v;
        }
    }
}

// SEG3D_X6_RT (A/B): 16-row tiles per wave, 4 or 2 (default 2); SEG3D_X6_WPS: resident workgroups per CU of the 2-tile form, 3 or 4
static const int g_x6_rt = [] {
    const char* e = getenv("SEG3D_X6_RT");
    const int v = e ? atoi(e) : 2;
    return v == 4 ? 4 : 2;
}();
static const int g_x6_wps = [] {
    const char* e = getenv("SEG3D_X6_WPS");
    const int v = e ? atoi(e) : 3;
    return v == 4 ? 4 : 3;
}();

}  // namespace

extern "C" {

/* Bytes of the three-plane fragment stream of a [cout, cin] Linear weight (cin % 32 == 0, cout % 64 == 0; 0 otherwise). */
size_t seg3d_linear_packed_bytes_x6(int32_t cin, int32_t cout) {
    if (cin <= 0 || cout <= 0 || (cin & 31) || (cout & 63)) return 0;
    return (size_t)(cin >> 5) * (size_t)(cout >> 6) * kSlab * sizeof(uint4);
}

/* weight [cout, cin] fp32 -> w_packed (seg3d_linear_packed_bytes_x6 bytes); transpose != 0: weight is [cin, cout] as stored
 * for the OTHER direction, i.e. the operand of x W instead of x W^T. */
int seg3d_linear_pack_weight_x6(const float* weight, int32_t cin, int32_t cout, int32_t transpose, void* w_packed, void* stream) {
    if (!weight || !w_packed || seg3d_linear_packed_bytes_x6(cin, cout) == 0) return SEG3D_EINVAL;
    const int64_t total = (int64_t)(cin >> 5) * (cout >> 6) * 256;
    hipLaunchKernelGGL(pack_x6_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), weight, cin, cout,
                       transpose ? 1 : 0, static_cast<uint4*>(w_packed));
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

/* y[m, cout] = x[m, cin] W^T (+ bias) (* scale + shift per column) (ReLU) with three-way split operands, six bf16 MFMAs
 * per product: fp32-grade results (see the head of linear_x6.hip).  bias, scale / shift (both or neither) may be NULL. */
int seg3d_linear_fwd_x6(const float* x, int64_t m, const void* w_packed, const float* bias, const float* scale,
                        const float* shift, int32_t relu, int32_t cin, int32_t cout, float* y, void* stream) {
    if (m < 0 || seg3d_linear_packed_bytes_x6(cin, cout) == 0 || (scale == nullptr) != (shift == nullptr)) return SEG3D_EINVAL;
    if (m == 0) return SEG3D_OK;
    if (!x || !w_packed || !y) return SEG3D_EINVAL;
    const int ncg = cout >> 6;
    const int rows = 64 * g_x6_rt;
    const int64_t nrb = (m + rows - 1) / rows;
    const int64_t blocks = (nrb + 7) / 8 * 8 * ncg;
    if (blocks > 0x7FFFFFFFll) return SEG3D_EINVAL;
    if (g_x6_rt == 4)
        hipLaunchKernelGGL((linear_x6_kernel<4, 2>), dim3((unsigned)blocks), dim3(kThreads), 0, as_stream(stream), x, m,
                           static_cast<const uint4*>(w_packed), bias, scale, shift, relu ? 1 : 0, cin, cout, y, ncg, nrb);
    else if (g_x6_wps == 4)
        hipLaunchKernelGGL((linear_x6_kernel<2, 4>), dim3((unsigned)blocks), dim3(kThreads), 0, as_stream(stream), x, m,
                           static_cast<const uint4*>(w_packed), bias, scale, shift, relu ? 1 : 0, cin, cout, y, ncg, nrb);
    else
        hipLaunchKernelGGL((linear_x6_kernel<2, 3>), dim3((unsigned)blocks), dim3(kThreads), 0, as_stream(stream), x, m,
                           static_cast<const uint4*>(w_packed), bias, scale, shift, relu ? 1 : 0, cin, cout, y, ncg, nrb);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // extern "C"
