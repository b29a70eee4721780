// a9-a11: sparse 3x3x3 convolution (submanifold / strided / inverse): C entry points, the exact-fp32
// forward/dgrad kernel, the weight packs and wgrad.  (The split-bf16 forward lives in spconv_split.hip.)
// Replaces spconv's implicit-GEMM kernels behind SubMConv3d / SparseConv3d / SparseInverseConv3d
// (call sites: seg3d/utils/spconv_utils.py:13-32, seg3d/models/backbones/pointtransformer.py:26-34,69-81).
//
// Output-stationary gather-GEMM:   y[r] = bias + sum_k x[nbr[k][r]] . W_k
// Exact-fp32 variant: one wave owns 16 output rows; per kernel offset it gathers the 16 neighbour rows
// straight into A fragments (16 B per lane, 64 B per row per instruction, no atomics, no scatter) and
// streams the pre-packed W_k fragments (1 KiB coalesced per instruction, L2-resident) into
// v_mfma_f32_16x16x4_f32 -- bit-for-bit an fmaf chain.  Offsets with no active neighbour among the
// wave's rows are skipped with a ballot.
//
// Algorithmic bytes per launch: P*(Cin+Cout)*4 + 27*Cin*Cout*4 + P*8, P = active pairs.
#include "common.hpp"

size_t spconv_split_packed_bytes(int cin_op, int cout_op, int kk);
int spconv_split_pack(const float* weight, int cin, int cout, int kk, int transpose, int flip, void* w_packed,
                      hipStream_t st);
int spconv_split_fwd_io(const void* x, const int32_t* nbr, int64_t m_out, const void* wp, const float* bias,
                        const void* addend, const int32_t* row_order, int cin, int cout, void* y, int relu, int io,
                        hipStream_t st, const float* x_add);
int spconv_split_fwd(const float* x, const int32_t* nbr, int64_t m_out, const void* wp, const float* bias,
                     const float* addend, const int32_t* row_order, int cin, int cout, float* y, int relu, hipStream_t st);
size_t wgrad_split_sparse_workspace_bytes(int64_t m_out, int cin, int cout);  // wgrad_split.hip
int wgrad_split_sparse(const void* x, const float* dy, const int32_t* nbr, int64_t m_out, int cin, int cout, float* dw,
                       void* workspace, size_t workspace_bytes, hipStream_t st, int32_t* chunks_out, bool x_bf16);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kRowsPerWave = 16;
constexpr int kRowsPerBlock = kWaves * kRowsPerWave;

// ------------------------------------------------------------------ fp32 weight pack
// src weight[co][k][ci]; operand B_k[ci'][co'] with (ci', co') = (ci, co) or swapped (transpose),
// k' = k or 26-k (flip).  Packed as [k'][ci'/16][co'/16][lane 64][j 4]:
//   lane = ((ci' % 16) / 4) * 16 + (co' % 16),  j = ci' % 4
// so that one wave-wide float4 load is the B fragment set of four consecutive 16x16x4 MFMAs.
__global__ __launch_bounds__(256) void pack_weight(const float* __restrict__ w, int cin_src, int cout_src, int kk,
                                                   int transpose, int flip, float* __restrict__ wp) {
    const int64_t total = (int64_t)kk * cin_src * cout_src;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int cin_op = transpose ? cout_src : cin_src;
    const int cout_op = transpose ? cin_src : cout_src;
    const int cb_n = cin_op / 16, nb_n = cout_op / 16;
    int64_t r = t;
    const int j = (int)(r & 3); r >>= 2;
    const int lane = (int)(r & 63); r >>= 6;
    const int nb = (int)(r % nb_n); r /= nb_n;
    const int cb = (int)(r % cb_n); r /= cb_n;
    const int kp = (int)r;
    const int ci_op = cb * 16 + (lane >> 4) * 4 + j;
    const int co_op = nb * 16 + (lane & 15);
    const int k = flip ? kk - 1 - kp : kp;
    const int ci = transpose ? co_op : ci_op;
    const int co = transpose ? ci_op : co_op;
    wp[t] = w[((int64_t)co * kk + k) * cin_src + ci];
}

// ------------------------------------------------------------------ fp32 forward / dgrad
// DENSE: a Linear layer = one offset whose neighbour of row r is row r (exact-fp32 per-point MLPs, a6)
template <int NBT, bool DENSE>
__global__ __launch_bounds__(kThreads) void spconv_fwd_kernel(const float* __restrict__ x, const int32_t* __restrict__ nbr,
                                                              int64_t m_out, const float* __restrict__ wp,
                                                              const float* __restrict__ bias, int cin, int cout,
                                                              float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * kRowsPerBlock + wave * kRowsPerWave;
    if (row0 >= m_out) return;  // whole wave out of range (wave-uniform)
    const int nb0 = blockIdx.y * NBT;  // first 16-column block of this wave's tile
    const int cb_n = cin >> 4, nb_n = cout >> 4;
    const int64_t my_row = row0 + (lane & 15);
    const bool row_ok = my_row < m_out;
    const int64_t my_row_c = row_ok ? my_row : m_out - 1;

    f32x4 acc[NBT];
#pragma unroll
    for (int n = 0; n < NBT; ++n) {
        const float b = bias ? bias[(nb0 + n) * 16 + (lane & 15)] : 0.0f;
        acc[n] = (f32x4){b, b, b, b};
    }

    for (int k = 0; k < (DENSE ? 1 : 27); ++k) {
        const int32_t raw = DENSE ? (int32_t)my_row_c : nbr[(int64_t)k * m_out + my_row_c];
        const int32_t idx = row_ok ? raw : -1;
        if (__ballot(idx >= 0) == 0ull) continue;
        const float* xrow = x + (int64_t)(idx >= 0 ? idx : 0) * cin + (lane >> 4) * 4;
        const float* wk = wp + ((int64_t)k * cb_n * nb_n + nb0) * 256 + lane * 4;
        for (int cb = 0; cb < cb_n; ++cb) {
            const f32x4 ld = *reinterpret_cast<const f32x4*>(xrow + cb * 16);
            const f32x4 a = idx >= 0 ? ld : (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 b[NBT];
#pragma unroll
            for (int n = 0; n < NBT; ++n)
                b[n] = *reinterpret_cast<const f32x4*>(wk + ((int64_t)cb * nb_n + n) * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int n = 0; n < NBT; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[n][j], acc[n], 0, 0, 0);
            }
        }
    }

    // D layout: row = (lane>>4)*4 + r, col = lane & 15
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t orow = row0 + (lane >> 4) * 4 + r;
        if (orow < m_out) {
            float* yr = y + orow * cout + nb0 * 16 + (lane & 15);
#pragma unroll
            for (int n = 0; n < NBT; ++n) yr[n * 16] = acc[n][r];
        }
    }
}

template <int NBT>
int launch_fwd(const float* x, const int32_t* nbr, int64_t m_out, const float* wp, const float* bias, int cin, int cout,
               float* y, hipStream_t st) {
    dim3 grid((unsigned)ceil_div64(m_out, kRowsPerBlock), (unsigned)((cout / 16) / NBT));
    if (nbr == nullptr)
        hipLaunchKernelGGL((spconv_fwd_kernel<NBT, true>), grid, dim3(kThreads), 0, st, x, nbr, m_out, wp, bias, cin, cout, y);
    else
        hipLaunchKernelGGL((spconv_fwd_kernel<NBT, false>), grid, dim3(kThreads), 0, st, x, nbr, m_out, wp, bias, cin, cout, y);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

// ------------------------------------------------------------------ wgrad
// dw[co][k][ci] += sum_r x[nbr[k][r]][ci] * dy[r][co]   (GEMM with the row index as K dimension)
// One wave owns a (16*JA) x (16*JB) channel block of one offset k over a chunk of output rows.
// Active (row, input) pairs of each 64-row batch are compacted through LDS so MFMA steps only see
// real pairs; a lane loads JA (JB) consecutive channels of its pair's x (dy) row, which assigns
// channel ci = JA*(lane&15)+j to row (lane&15) of the j-th A fragment -- any bijection works since
// the tile is written back through the same map.  Partial blocks are combined with float atomics,
// shaped as whole contiguous rows (see the epilogue).
template <int J>
struct VecLoad;
template <>
struct VecLoad<1> {
    static __device__ __forceinline__ void load(const float* p, float* v) { v[0] = p[0]; }
};
template <>
struct VecLoad<2> {
    static __device__ __forceinline__ void load(const float* p, float* v) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x; v[1] = t.y;
    }
};
template <>
struct VecLoad<3> {
    static __device__ __forceinline__ void load(const float* p, float* v) { v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; }
};
template <>
struct VecLoad<4> {
    static __device__ __forceinline__ void load(const float* p, float* v) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
};

constexpr int kWgradRowsPerWave = 512;

template <int JA, int JB>
__global__ __launch_bounds__(kThreads) void spconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                const int32_t* __restrict__ nbr, int64_t m_out, int cin,
                                                                int cout, float* __restrict__ dw) {
    __shared__ int32_t pair_row[kWaves][64];
    __shared__ int32_t pair_in[kWaves][64];
    __shared__ float stage[kWaves][16 * JB * (16 * JA + 1)];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int k = blockIdx.y;
    const int ca_n = cin / (16 * JA);
    const int ca = blockIdx.z % ca_n, cbk = blockIdx.z / ca_n;
    const int ci0 = ca * 16 * JA, co0 = cbk * 16 * JB;
    const int64_t r_begin = ((int64_t)blockIdx.x * kWaves + wave) * kWgradRowsPerWave;
    if (r_begin >= m_out) return;
    const int64_t r_end = r_begin + kWgradRowsPerWave < m_out ? r_begin + kWgradRowsPerWave : m_out;

    f32x4 acc[JA][JB];
#pragma unroll
    for (int a = 0; a < JA; ++a)
#pragma unroll
        for (int b = 0; b < JB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int32_t* nk = nbr + (int64_t)k * m_out;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 64) {
        const int64_t r = r0 + lane;
        const int32_t idx = r < r_end ? nk[r] : -1;
        const unsigned long long mask = __ballot(idx >= 0);
        if (mask == 0ull) continue;
        const int cnt = __popcll(mask);
        if (idx >= 0) {
            const int p = __popcll(mask & ((1ull << lane) - 1ull));
            pair_row[wave][p] = (int32_t)(r - r_begin);
            pair_in[wave][p] = idx;
        }
        // one wave per LDS slice: LDS ops retire in order, only the compiler must not reorder them
        __builtin_amdgcn_wave_barrier();
        for (int g = 0; g < cnt; g += 4) {
            const int e = g + (lane >> 4);
            float av[JA], bv[JB];
#pragma unroll
            for (int j = 0; j < JA; ++j) av[j] = 0.f;
#pragma unroll
            for (int j = 0; j < JB; ++j) bv[j] = 0.f;
            if (e < cnt) {
                const int32_t in = pair_in[wave][e];
                const int64_t orow = r_begin + pair_row[wave][e];
                VecLoad<JA>::load(x + (int64_t)in * cin + ci0 + JA * (lane & 15), av);
                VecLoad<JB>::load(dy + orow * cout + co0 + JB * (lane & 15), bv);
            }
#pragma unroll
            for (int a = 0; a < JA; ++a)
#pragma unroll
                for (int b = 0; b < JB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // Epilogue.  D[a][b]: row (lane>>4)*4 + rr -> ci_local = JA*row + a ; col lane&15 -> co_local = JB*col + b.
    // Stage the wave's (16*JB) x (16*JA) block through LDS as [co_local][ci_local] and add it to dw in
    // whole rows: one atomic wave-instruction covers 16*JA contiguous floats of one dw row, the shape
    // that runs at the full float-atomic rate (64 scattered lines per instruction run ~17x slower).
    constexpr int CA = 16 * JA, CB = 16 * JB;
    float* st = stage[wave];
#pragma unroll
    for (int a = 0; a < JA; ++a)
#pragma unroll
        for (int b = 0; b < JB; ++b)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int cil = JA * ((lane >> 4) * 4 + rr) + a;
                const int col = JB * (lane & 15) + b;
                st[col * (CA + 1) + cil] = acc[a][b][rr];
            }
    __builtin_amdgcn_wave_barrier();
    for (int col = 0; col < CB; ++col) {
        if (lane < CA) {
            const float v = st[col * (CA + 1) + lane];
            if (v != 0.0f) atomicAdd(&dw[((int64_t)(co0 + col) * 27 + k) * cin + ci0 + lane], v);
        }
    }
}

inline int pick_j(int c) {
    if (c % 64 == 0) return 4;
    if (c % 48 == 0) return 3;
    if (c % 32 == 0) return 2;
    return 1;
}

template <int JA, int JB>
int launch_wgrad(const float* x, const float* dy, const int32_t* nbr, int64_t m_out, int cin, int cout, float* dw,
                 hipStream_t st) {
    dim3 grid((unsigned)ceil_div64(m_out, (int64_t)kWaves * kWgradRowsPerWave), 27,
              (unsigned)((cin / (16 * JA)) * (cout / (16 * JB))));
    hipLaunchKernelGGL((spconv_wgrad_kernel<JA, JB>), grid, dim3(kThreads), 0, st, x, dy, nbr, m_out, cin, cout, dw);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

template <int JA>
int launch_wgrad_b(int jb, const float* x, const float* dy, const int32_t* nbr, int64_t m_out, int cin, int cout,
                   float* dw, hipStream_t st) {
    switch (jb) {
        case 4: return launch_wgrad<JA, 4>(x, dy, nbr, m_out, cin, cout, dw, st);
        case 3: return launch_wgrad<JA, 3>(x, dy, nbr, m_out, cin, cout, dw, st);
        case 2: return launch_wgrad<JA, 2>(x, dy, nbr, m_out, cin, cout, dw, st);
        default: return launch_wgrad<JA, 1>(x, dy, nbr, m_out, cin, cout, dw, st);
    }
}

int dispatch_fwd_f32(const float* x, const int32_t* nbr, int64_t m_out, const float* w_packed, const float* bias, int cin,
                     int cout, float* y, hipStream_t st) {
    const int nb = cout / 16;
    if (nb % 12 == 0) return launch_fwd<12>(x, nbr, m_out, w_packed, bias, cin, cout, y, st);
    if (nb % 8 == 0) return launch_fwd<8>(x, nbr, m_out, w_packed, bias, cin, cout, y, st);
    if (nb % 6 == 0) return launch_fwd<6>(x, nbr, m_out, w_packed, bias, cin, cout, y, st);
    if (nb % 4 == 0) return launch_fwd<4>(x, nbr, m_out, w_packed, bias, cin, cout, y, st);
    if (nb % 3 == 0) return launch_fwd<3>(x, nbr, m_out, w_packed, bias, cin, cout, y, st);
    if (nb % 2 == 0) return launch_fwd<2>(x, nbr, m_out, w_packed, bias, cin, cout, y, st);
    return launch_fwd<1>(x, nbr, m_out, w_packed, bias, cin, cout, y, st);
}

}  // namespace

extern "C" {

size_t seg3d_spconv_packed_bytes(int32_t cin, int32_t cout, int32_t flags) {
    if (cin <= 0 || cout <= 0) return 0;
    const int cin_op = (flags & 1) ? cout : cin, cout_op = (flags & 1) ? cin : cout;
    if (flags & 4) return spconv_split_packed_bytes(cin_op, cout_op, 27);
    return (size_t)27 * cin * cout * sizeof(float);
}

int seg3d_spconv_pack_weight(const float* weight, int32_t cin, int32_t cout, int32_t flags, void* w_packed,
                             void* stream) {
    if (!weight || !w_packed || cin <= 0 || cout <= 0 || (cin & 15) || (cout & 15)) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    if (flags & 4) return spconv_split_pack(weight, cin, cout, 27, flags & 1, (flags >> 1) & 1, w_packed, st);
    const int64_t total = (int64_t)27 * cin * cout;
    hipLaunchKernelGGL(pack_weight, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st, weight, cin, cout, 27,
                       flags & 1, (flags >> 1) & 1, static_cast<float*>(w_packed));
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_spconv_fwd(const float* x, const int32_t* nbr, int64_t m_out, int64_t m_in, const void* w_packed_v,
                     int32_t pack_flags, const float* bias, int32_t cin, int32_t cout, float* y, const int32_t* row_order,
                     void* stream) {
    if (m_out < 0 || m_in < 0 || cin <= 0 || cout <= 0 || (cin & 15) || (cout & 15) || !w_packed_v) return SEG3D_EINVAL;
    if (m_out == 0) return SEG3D_OK;
    if (!x || !nbr || !y) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    if (pack_flags & 4) return spconv_split_fwd(x, nbr, m_out, w_packed_v, bias, nullptr, row_order, cin, cout, y, 0, st);
    return dispatch_fwd_f32(x, nbr, m_out, static_cast<const float*>(w_packed_v), bias, cin, cout, y, st);
}

/* Inference form of a conv block (conv -> BatchNorm(eval) -> (+ residual) -> ReLU, spconv_utils.py:13-32,
 * pointtransformer.py:47-66): with the BatchNorm affine folded into the packed weights and the bias by the caller, the
 * block is one launch -- y = act(conv(x) + bias (+ addend)).  Split-bf16 packs only. */
int seg3d_spconv_fwd_act(const float* x, const int32_t* nbr, int64_t m_out, int64_t m_in, const void* w_packed_v,
                         int32_t pack_flags, const float* bias, const float* addend, int32_t relu, int32_t cin,
                         int32_t cout, float* y, const int32_t* row_order, void* stream) {
    if (m_out < 0 || m_in < 0 || cin <= 0 || cout <= 0 || (cin & 15) || (cout & 15) || !w_packed_v || !(pack_flags & 4))
        return SEG3D_EINVAL;
    if (m_out == 0) return SEG3D_OK;
    if (!x || !nbr || !y) return SEG3D_EINVAL;
    return spconv_split_fwd(x, nbr, m_out, w_packed_v, bias, addend, row_order, cin, cout, y, relu ? 1 : 0, as_stream(stream));
}

/* The same block with the feature maps STORED in bf16 (opt-in, BASELINE configs[4]; the reference's only reduced-
 * precision hook is voxel_pooling.py:12): x is float32 (x_bf16 = 0) or bf16 (x_bf16 = 1) rows, y and the residual addend
 * are bf16; accumulation, bias and activation stay float32.  A bf16 row is its own hi part, so its products take two
 * MFMAs instead of three and its gather moves half the bytes. */
int seg3d_spconv_fwd_act_bf16(const void* x, int32_t x_bf16, const int32_t* nbr, int64_t m_out, int64_t m_in,
                              const void* w_packed_v, int32_t pack_flags, const float* bias, const void* addend_bf16,
                              int32_t relu, int32_t cin, int32_t cout, void* y_bf16, const int32_t* row_order,
                              void* stream) {
    if (m_out < 0 || m_in < 0 || cin <= 0 || cout <= 0 || (cin & 15) || (cout & 15) || !w_packed_v || !(pack_flags & 4))
        return SEG3D_EINVAL;
    if (m_out == 0) return SEG3D_OK;
    if (!x || !nbr || !y_bf16) return SEG3D_EINVAL;
    return spconv_split_fwd_io(x, nbr, m_out, w_packed_v, bias, addend_bf16, row_order, cin, cout, y_bf16, relu ? 1 : 0,
                               x_bf16 ? 2 : 1, as_stream(stream), nullptr);
}

/* a6  exact-fp32 Linear (per-point MLPs): y[m, cout] = x[m, cin] . W^T + bias on v_mfma_f32_16x16x4_f32 -- the
 * single-offset case of the exact-fp32 gather-GEMM (rocBLAS picks 16x64 / 32x32 macro tiles for these tall-skinny
 * shapes: 340-470 us per layer at m = 175k). */
size_t seg3d_linear_packed_bytes_f32(int32_t cin, int32_t cout) {
    return cin > 0 && cout > 0 ? (size_t)cin * cout * sizeof(float) : 0;
}

int seg3d_linear_pack_weight_f32(const float* weight, int32_t cin, int32_t cout, int32_t transpose, void* w_packed,
                                 void* stream) {
    if (!weight || !w_packed || cin <= 0 || cout <= 0 || (cin & 15) || (cout & 15)) return SEG3D_EINVAL;
    const int64_t total = (int64_t)cin * cout;
    hipLaunchKernelGGL(pack_weight, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, as_stream(stream), weight, cin,
                       cout, 1, transpose ? 1 : 0, 0, static_cast<float*>(w_packed));
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_linear_fwd_f32(const float* x, int64_t m, const void* w_packed, const float* bias, int32_t cin, int32_t cout,
                         float* y, void* stream) {
    if (m < 0 || cin <= 0 || cout <= 0 || (cin & 15) || (cout & 15) || !w_packed) return SEG3D_EINVAL;
    if (m == 0) return SEG3D_OK;
    if (!x || !y) return SEG3D_EINVAL;
    return dispatch_fwd_f32(x, nullptr, m, static_cast<const float*>(w_packed), bias, cin, cout, y, as_stream(stream));
}

size_t seg3d_spconv_wgrad_workspace_bytes(int64_t m_out, int32_t cin, int32_t cout) {
    if (m_out < 0 || cin <= 0 || cout <= 0) return 0;
    return wgrad_split_sparse_workspace_bytes(m_out, cin, cout);  // the exact-fp32 path needs none
}

int seg3d_spconv_wgrad(const float* x, const float* dy, const int32_t* nbr, int64_t m_out, int64_t m_in, int32_t cin,
                       int32_t cout, int32_t flags, float* dw, void* workspace, size_t workspace_bytes, void* stream) {
    if (m_out < 0 || m_in < 0 || cin <= 0 || cout <= 0 || (cin & 15) || (cout & 15) || !dw) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    if (m_out > 0 && (!x || !dy || !nbr)) return SEG3D_EINVAL;
    // split-bf16: partial blocks per row chunk in the workspace, summed in a fixed order (writes all of dw)
    if (m_out > 0 && (flags & 4)) return wgrad_split_sparse(x, dy, nbr, m_out, cin, cout, dw, workspace, workspace_bytes, st, nullptr, false);
    SEG3D_CHECK_HIP(hipMemsetAsync(dw, 0, (size_t)27 * cin * cout * sizeof(float), st));
    if (m_out == 0) return SEG3D_OK;
    const int ja = pick_j(cin), jb = pick_j(cout);
    switch (ja) {
        case 4: return launch_wgrad_b<4>(jb, x, dy, nbr, m_out, cin, cout, dw, st);
        case 3: return launch_wgrad_b<3>(jb, x, dy, nbr, m_out, cin, cout, dw, st);
        case 2: return launch_wgrad_b<2>(jb, x, dy, nbr, m_out, cin, cout, dw, st);
        default: return launch_wgrad_b<1>(jb, x, dy, nbr, m_out, cin, cout, dw, st);
    }
}

/* The first half of seg3d_spconv_wgrad (split-bf16 packs only): the per-chunk partial blocks part[chunks][27 * cin * cout] are
 * left in the workspace, *chunks (host memory) receives their count; the fixed-order sum is the caller's to queue
 * (seg3d_reduce_partials / seg3d_reduce_partials_batched with n = nw = 27 * cin * cout), so that a backward pass sums ALL its
 * parameter-gradient partials in one launch. */
int seg3d_spconv_wgrad_partials(const float* x, const float* dy, const int32_t* nbr, int64_t m_out, int64_t m_in, int32_t cin,
                                int32_t cout, void* workspace, size_t workspace_bytes, int32_t* chunks, void* stream) {
    if (m_out < 0 || m_in < 0 || cin <= 0 || cout <= 0 || (cin & 15) || (cout & 15) || !chunks) return SEG3D_EINVAL;
    *chunks = 0;
    if (m_out == 0) return SEG3D_OK;
    if (!x || !dy || !nbr) return SEG3D_EINVAL;
    return wgrad_split_sparse(x, dy, nbr, m_out, cin, cout, nullptr, workspace, workspace_bytes, as_stream(stream), chunks, false);
}

/* seg3d_spconv_wgrad_partials with the x rows stored as bf16 [m_in, cin] (the opt-in copies a training forward saves for its
 * backward, SEG3D_TRAIN_STORAGE=bf16): a bf16 row is its own high half -- 8-byte gathers, no split of x, two MFMAs per product. */
int seg3d_spconv_wgrad_partials_xbf16(const uint16_t* x_bf16, const float* dy, const int32_t* nbr, int64_t m_out, int64_t m_in,
                                      int32_t cin, int32_t cout, void* workspace, size_t workspace_bytes, int32_t* chunks,
                                      void* stream) {
    if (m_out < 0 || m_in < 0 || cin <= 0 || cout <= 0 || (cin & 15) || (cout & 15) || !chunks) return SEG3D_EINVAL;
    *chunks = 0;
    if (m_out == 0) return SEG3D_OK;
    if (!x_bf16 || !dy || !nbr) return SEG3D_EINVAL;
    return wgrad_split_sparse(x_bf16, dy, nbr, m_out, cin, cout, nullptr, workspace, workspace_bytes, as_stream(stream), chunks, true);
}

}  // extern "C"
