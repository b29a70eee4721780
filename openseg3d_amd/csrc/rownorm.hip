// a12 / a22 row-wise normalisation layers on [rows, C] voxel / point features:
//   * post-norm residual LayerNorm of the SWFormer encoder layer (point_transformer_layer.py:289-298):
//         y = res + rowscale * (LN(x) * gamma + beta)                      (forward + backward; rowscale =
//         the per-row stochastic-depth factor of drop.py:6-19, optional)
//   * BatchNorm1d (+ residual) (+ ReLU) of the sparse-conv blocks and point MLPs (spconv_utils.py:13-32,
//     pointtransformer.py:47-66, segformer.py:21-76): batch statistics, affine + activation, backward.
// torch's generic kernels need 75 us (LayerNorm) / 130-170 us (BatchNorm statistics, backward reduce) per call
// on a [121k, 96] tensor that streams in ~15 us; these are plain HBM-bound passes: 16-B accesses, one wave per
// 64/P rows for the row reductions, fixed lane -> channel-quad mapping with register partial sums for the
// column reductions (one float atomic per channel per workgroup at the end).
#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;
constexpr int kLnBwdMaxBlocks = 1024;

// A wave covers R = 64 / P rows; a row is spread over P lanes (P = pow2 >= quads / ITEMS), every lane owning
// ITEMS float4 "quads" of its row: quad q = lane_in_row + i * P.
struct RowMap {
    int quads;  // C / 4
    int p;      // lanes per row (8, 16, 32, 64)
    int items;  // quads per lane (1 or 2)
};

inline RowMap row_map(int c) {
    RowMap m;
    m.quads = c / 4;
    m.items = (m.quads + 63) / 64;
    int need = (m.quads + m.items - 1) / m.items;
    m.p = 8;
    while (m.p < need) m.p <<= 1;
    return m;
}

__device__ __forceinline__ float group_sum(float v, int p) {
    for (int off = 1; off < p; off <<= 1) v += __shfl_xor(v, off, SEG3D_WAVE);
    return v;
}

// ------------------------------------------------------------------ LayerNorm
template <int ITEMS>
__global__ __launch_bounds__(kThreads) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ rowscale, float eps, int64_t m, int c,
                                                          RowMap rm, float* __restrict__ y, float* __restrict__ mean,
                                                          float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows_per_wave = 64 / rm.p;
    const int lr = lane / rm.p, lq = lane % rm.p;
    const float inv_c = 1.0f / (float)c;
    for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * rows_per_wave; base < m;
         base += (int64_t)gridDim.x * 4 * rows_per_wave) {
        const int64_t row = base + lr;
        const bool rok = row < m;
        float4 v[ITEMS];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int q = lq + i * rm.p;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rok && q < rm.quads) v[i] = *reinterpret_cast<const float4*>(x + row * c + 4 * q);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mu = group_sum(s, rm.p) * inv_c;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int q = lq + i * rm.p;
            if (q < rm.quads) {
                const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
                ss += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float rs = rsqrtf(group_sum(ss, rm.p) * inv_c + eps);
        const float sc = rowscale && rok ? rowscale[row] : 1.0f;  // per-row stochastic-depth factor
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int q = lq + i * rm.p;
            if (rok && q < rm.quads) {
                const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * q);
                const float4 b = *reinterpret_cast<const float4*>(beta + 4 * q);
                float4 o;
                o.x = ((v[i].x - mu) * rs * g.x + b.x) * sc;
                o.y = ((v[i].y - mu) * rs * g.y + b.y) * sc;
                o.z = ((v[i].z - mu) * rs * g.z + b.z) * sc;
                o.w = ((v[i].w - mu) * rs * g.w + b.w) * sc;
                if (res) {
                    const float4 r4 = *reinterpret_cast<const float4*>(res + row * c + 4 * q);
                    o.x += r4.x; o.y += r4.y; o.z += r4.z; o.w += r4.w;
                }
                *reinterpret_cast<float4*>(y + row * c + 4 * q) = o;
            }
        }
        if (rok && lq == 0 && mean) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// dx = rstd * (dy*gamma - mean_c(dy*gamma) - xhat * mean_c(dy*gamma*xhat)); dgamma += dy*xhat; dbeta += dy
template <int ITEMS>
__global__ __launch_bounds__(kThreads) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ rowscale,
                                                          int64_t m, int c, RowMap rm, float* __restrict__ dx,
                                                          float* __restrict__ part /*[gridDim.x][2][c]*/) {
    extern __shared__ float red[];  // [4 waves x rows per wave][2][c]: one slot per row lane of the block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows_per_wave = 64 / rm.p;
    const int lr = lane / rm.p, lq = lane % rm.p;
    const float inv_c = 1.0f / (float)c;
    float4 ag[ITEMS], ab[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * rows_per_wave; base < m;
         base += (int64_t)gridDim.x * 4 * rows_per_wave) {
        const int64_t row = base + lr;
        const bool rok = row < m;
        const float mu = rok ? mean[row] : 0.f, rs = rok ? rstd[row] : 0.f;
        const float sc = rowscale && rok ? rowscale[row] : 1.0f;
        float4 xh[ITEMS], gy[ITEMS];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int q = lq + i * rm.p;
            xh[i] = gy[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rok && q < rm.quads) {
                const float4 xv = *reinterpret_cast<const float4*>(x + row * c + 4 * q);
                float4 dv = *reinterpret_cast<const float4*>(dy + row * c + 4 * q);
                dv.x *= sc; dv.y *= sc; dv.z *= sc; dv.w *= sc;
                const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * q);
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                gy[i] = make_float4(dv.x * g.x, dv.y * g.y, dv.z * g.z, dv.w * g.w);
                ag[i].x += dv.x * xh[i].x; ag[i].y += dv.y * xh[i].y; ag[i].z += dv.z * xh[i].z; ag[i].w += dv.w * xh[i].w;
                ab[i].x += dv.x; ab[i].y += dv.y; ab[i].z += dv.z; ab[i].w += dv.w;
                s1 += (gy[i].x + gy[i].y) + (gy[i].z + gy[i].w);
                s2 += (gy[i].x * xh[i].x + gy[i].y * xh[i].y) + (gy[i].z * xh[i].z + gy[i].w * xh[i].w);
            }
        }
        const float c1 = group_sum(s1, rm.p) * inv_c, c2 = group_sum(s2, rm.p) * inv_c;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int q = lq + i * rm.p;
            if (rok && q < rm.quads) {
                float4 o;
                o.x = rs * (gy[i].x - c1 - xh[i].x * c2);
                o.y = rs * (gy[i].y - c1 - xh[i].y * c2);
                o.z = rs * (gy[i].z - c1 - xh[i].z * c2);
                o.w = rs * (gy[i].w - c1 - xh[i].w * c2);
                *reinterpret_cast<float4*>(dx + row * c + 4 * q) = o;
            }
        }
    }
    // block partials: every row lane parks its sums in its own LDS slot, the slots are added in a fixed order and each
    // channel gets one plain store per block; ln_bwd_reduce then sums the blocks in a fixed order (no atomics of any
    // kind, no memset: bit-identical from run to run)
    const int slots = 4 * rows_per_wave;
    float* mine = red + (size_t)(wave * rows_per_wave + lr) * 2 * c;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int q = lq + i * rm.p;
        if (q < rm.quads) {
            *reinterpret_cast<float4*>(mine + 4 * q) = ag[i];
            *reinterpret_cast<float4*>(mine + c + 4 * q) = ab[i];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * c; i += kThreads) {
        float t = red[i];
        for (int l = 1; l < slots; ++l) t += red[(size_t)l * 2 * c + i];
        part[(int64_t)blockIdx.x * 2 * c + i] = t;
    }
}

// dgamma[i] = sum_b part[b][i], dbeta[i] = sum_b part[b][c + i]; 32 block lanes per column, combined in lane order
__global__ __launch_bounds__(1024) void ln_bwd_reduce(const float* __restrict__ part, int nblocks, int c,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[32][33];
    const int col = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + col;
    float s = 0.f;
    if (i < 2 * c) {  // four independent chains (see col_partial_sum)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = q;
        for (; b + 96 < nblocks; b += 128) {
            s0 += part[(int64_t)b * 2 * c + i];
            s1 += part[(int64_t)(b + 32) * 2 * c + i];
            s2 += part[(int64_t)(b + 64) * 2 * c + i];
            s3 += part[(int64_t)(b + 96) * 2 * c + i];
        }
        for (; b < nblocks; b += 32) s0 += part[(int64_t)b * 2 * c + i];
        s = (s0 + s1) + (s2 + s3);
    }
    red[q][col] = s;
    __syncthreads();
    if (q == 0 && i < 2 * c) {
        float t = red[0][col];
#pragma unroll
        for (int k = 1; k < 32; ++k) t += red[k][col];
        if (i < c) dgamma[i] = t;
        else dbeta[i - c] = t;
    }
}

// ------------------------------------------------------------------ column reductions / affine passes (BatchNorm)
// thread -> (channel quad q = t % quads, row lane = t / quads); rows advance by the number of row lanes.
// MODE 0: sums of (x - x0), (x - x0)^2 with x0 = row 0 (shifted sums: no cancellation in the variance)
// MODE 1: sums of g, g * xhat with g = dy masked by (y > 0) when relu, xhat = (x - mean) * rstd
template <int MODE>
__global__ __launch_bounds__(kThreads) void col_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              const float* __restrict__ y, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, int relu, int64_t m, int c,
                                                              float* __restrict__ part /*[gridDim.x][2][c]*/) {
    extern __shared__ float red[];  // [row lanes][2][c] (<= 8 KiB)
    const int quads = c / 4;
    const int lanes = kThreads / quads;  // row lanes per block
    const int q = threadIdx.x % quads, rl = threadIdx.x / quads;
    if (rl < lanes) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        float4 p0 = a, p1 = a;
        if (MODE == 0) p0 = *reinterpret_cast<const float4*>(x + 4 * q);  // shift
        if (MODE == 1) {
            p0 = *reinterpret_cast<const float4*>(mean + 4 * q);
            p1 = *reinterpret_cast<const float4*>(rstd + 4 * q);
        }
        for (int64_t row = (int64_t)blockIdx.x * lanes + rl; row < m; row += (int64_t)gridDim.x * lanes) {
            const float4 xv = *reinterpret_cast<const float4*>(x + row * c + 4 * q);
            if (MODE == 0) {
                const float dx_ = xv.x - p0.x, dy_ = xv.y - p0.y, dz_ = xv.z - p0.z, dw_ = xv.w - p0.w;
                a.x += dx_; a.y += dy_; a.z += dz_; a.w += dw_;
                b.x += dx_ * dx_; b.y += dy_ * dy_; b.z += dz_ * dz_; b.w += dw_ * dw_;
            } else {
                float4 g = *reinterpret_cast<const float4*>(dy + row * c + 4 * q);
                if (relu) {
                    const float4 yv = *reinterpret_cast<const float4*>(y + row * c + 4 * q);
                    g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                    g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
                }
                a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
                b.x += g.x * (xv.x - p0.x) * p1.x; b.y += g.y * (xv.y - p0.y) * p1.y;
                b.z += g.z * (xv.z - p0.z) * p1.z; b.w += g.w * (xv.w - p0.w) * p1.w;
            }
        }
        *reinterpret_cast<float4*>(red + (size_t)rl * 2 * c + 4 * q) = a;
        *reinterpret_cast<float4*>(red + (size_t)rl * 2 * c + c + 4 * q) = b;
    }
    __syncthreads();
    // fixed-order sum over the row lanes, one plain store per channel per block: no atomics anywhere
    for (int i = threadIdx.x; i < 2 * c; i += kThreads) {
        float t = red[i];
        for (int l = 1; l < lanes; ++l) t += red[(size_t)l * 2 * c + i];
        part[(size_t)blockIdx.x * 2 * c + i] = t;
    }
}

// sums[i] = sum over blocks of part[b][i], i < 2c, in a fixed order (32 channels x 32 block lanes per workgroup)
__device__ __forceinline__ float col_partial_sum(const float* __restrict__ part, int nblocks, int c2, int i, float (*red)[33]) {
    const int col = threadIdx.x & 31, q = threadIdx.x >> 5;
    float s = 0.f;
    if (i < c2) {  // four independent chains: the loads overlap instead of queueing behind one accumulator
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = q;
        for (; b + 96 < nblocks; b += 128) {
            s0 += part[(int64_t)b * c2 + i];
            s1 += part[(int64_t)(b + 32) * c2 + i];
            s2 += part[(int64_t)(b + 64) * c2 + i];
            s3 += part[(int64_t)(b + 96) * c2 + i];
        }
        for (; b < nblocks; b += 32) s0 += part[(int64_t)b * c2 + i];
        s = (s0 + s1) + (s2 + s3);
    }
    red[q][col] = s;
    __syncthreads();
    float t = red[0][col];
#pragma unroll
    for (int k = 1; k < 32; ++k) t += red[k][col];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(1024) void col_sums_finish(const float* __restrict__ part, int nblocks, int c,
                                                        float* __restrict__ sums /*[2][c]*/) {
    __shared__ float red[32][33];
    const int i = blockIdx.x * 32 + (threadIdx.x & 31);
    const float t = col_partial_sum(part, nblocks, 2 * c, i, red);
    if ((threadIdx.x >> 5) == 0 && i < 2 * c) sums[i] = t;
}

// GELU (erf form, torch.nn.GELU's default: point_transformer_layer.py:265) and, for the training forward, its derivative
// in the same pass: g = h Phi(h), gp = Phi(h) + h phi(h).  The backward then needs no pass of its own -- d h = gp * (dm W2)
// is taken in the epilogue of that GEMM (seg3d_linear_fwd_mul).
__global__ __launch_bounds__(kThreads) void gelu_fwd_kernel(const float* __restrict__ h, int64_t quads, float* __restrict__ g,
                                                            float* __restrict__ gp) {
    for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < quads; t += (int64_t)gridDim.x * kThreads) {
        const float4 v = reinterpret_cast<const float4*>(h)[t];
        const float in[4] = {v.x, v.y, v.z, v.w};
        float o[4], d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float cdf = 0.5f * (1.0f + erff(in[i] * 0.70710678118654752440f));
            o[i] = in[i] * cdf;
            d[i] = cdf + in[i] * (0.39894228040143267794f * __expf(-0.5f * in[i] * in[i]));
        }
        reinterpret_cast<float4*>(g)[t] = make_float4(o[0], o[1], o[2], o[3]);
        if (gp) reinterpret_cast<float4*>(gp)[t] = make_float4(d[0], d[1], d[2], d[3]);
    }
}

// y = act(x * scale + shift (+ res))
__global__ __launch_bounds__(kThreads) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              int relu, int64_t total_quads, int quads, float* __restrict__ y) {
    for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < total_quads; t += (int64_t)gridDim.x * kThreads) {
        const int q = (int)(t % quads);
        const float4 xv = reinterpret_cast<const float4*>(x)[t];
        const float4 s = *reinterpret_cast<const float4*>(scale + 4 * q);
        const float4 b = *reinterpret_cast<const float4*>(shift + 4 * q);
        float4 o = make_float4(xv.x * s.x + b.x, xv.y * s.y + b.y, xv.z * s.z + b.z, xv.w * s.w + b.w);
        if (res) {
            const float4 r = reinterpret_cast<const float4*>(res)[t];
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (relu) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        reinterpret_cast<float4*>(y)[t] = o;
    }
}

// BatchNorm backward apply: g = dy masked by (y > 0); dx = gamma*rstd * (g - c1 - xhat*c2), dres = g
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                const float* __restrict__ x, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ sums /*[2][c]*/, int relu,
                                                                float inv_m, const float* __restrict__ inv_m_dev,
                                                                int64_t total_quads, int quads,
                                                                float* __restrict__ dx, float* __restrict__ dres) {
    const int c = quads * 4;
    if (inv_m_dev) inv_m = inv_m_dev[0];  // SyncBatchNorm: 1 / (rows of all ranks), known on the device only
    for (int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x; t < total_quads; t += (int64_t)gridDim.x * kThreads) {
        const int q = (int)(t % quads);
        float4 g = reinterpret_cast<const float4*>(dy)[t];
        if (relu) {
            const float4 yv = reinterpret_cast<const float4*>(y)[t];
            g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
            g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
        }
        if (dres) reinterpret_cast<float4*>(dres)[t] = g;
        const float4 xv = reinterpret_cast<const float4*>(x)[t];
        const float4 mu = *reinterpret_cast<const float4*>(mean + 4 * q);
        const float4 rs = *reinterpret_cast<const float4*>(rstd + 4 * q);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * q);
        const float4 s1 = *reinterpret_cast<const float4*>(sums + 4 * q);
        const float4 s2 = *reinterpret_cast<const float4*>(sums + c + 4 * q);
        float4 o;
        o.x = ga.x * rs.x * (g.x - s1.x * inv_m - (xv.x - mu.x) * rs.x * s2.x * inv_m);
        o.y = ga.y * rs.y * (g.y - s1.y * inv_m - (xv.y - mu.y) * rs.y * s2.y * inv_m);
        o.z = ga.z * rs.z * (g.z - s1.z * inv_m - (xv.z - mu.z) * rs.z * s2.z * inv_m);
        o.w = ga.w * rs.w * (g.w - s1.w * inv_m - (xv.w - mu.w) * rs.w * s2.w * inv_m);
        reinterpret_cast<float4*>(dx)[t] = o;
    }
}

// batch statistics -> everything the forward / backward passes and the running buffers need: the block partials are
// summed in a fixed order (32 channels x 32 block lanes per workgroup), then one thread per channel finishes
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                           int nblocks, int64_t m, int c, float eps,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float momentum, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float* __restrict__ stats) {
    __shared__ float red[32][33];
    const int i = blockIdx.x * 32 + (threadIdx.x & 31);
    const float s1 = col_partial_sum(part, nblocks, 2 * c, i < c ? i : 2 * c, red);
    const float s2 = col_partial_sum(part, nblocks, 2 * c, i < c ? c + i : 2 * c, red);
    if ((threadIdx.x >> 5) != 0 || i >= c) return;
    stats[i] = s1;
    stats[c + i] = s2;
    const float inv_m = 1.0f / (float)m;
    const float d = s1 * inv_m;              // mean of (x - x[0])
    const float mean = x[i] + d;
    const float var = fmaxf(s2 * inv_m - d * d, 0.0f);  // biased
    const float rstd = 1.0f / sqrtf(var + eps);
    const float scale = gamma[i] * rstd;
    stats[2 * c + i] = mean;
    stats[3 * c + i] = rstd;
    stats[4 * c + i] = scale;
    stats[5 * c + i] = beta[i] - mean * scale;
    if (running_mean) {
        running_mean[i] = running_mean[i] * (1.0f - momentum) + mean * momentum;
        const float unbiased = m > 1 ? var * ((float)m / (float)(m - 1)) : var;
        running_var[i] = running_var[i] * (1.0f - momentum) + unbiased * momentum;
    }
}

constexpr int kColMaxBlocks = 1024;
constexpr size_t kColSmem = 8192;

inline unsigned blocks_for(int64_t work_items, int per_block) {
    int64_t b = ceil_div64(work_items, per_block);
    if (b > kMaxBlocks) b = kMaxBlocks;
    if (b < 1) b = 1;
    return (unsigned)b;
}

inline bool bad_c(int c) { return c <= 0 || (c & 3) || c > 1024; }

}  // namespace

extern "C" {

int seg3d_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, const float* rowscale,
                        float eps, int64_t m, int32_t c, float* y, float* mean, float* rstd, void* stream) {
    if (m < 0 || bad_c(c) || c > 512) return SEG3D_EINVAL;
    if (m == 0) return SEG3D_OK;
    if (!x || !gamma || !beta || !y) return SEG3D_EINVAL;
    const RowMap rm = row_map(c);
    const unsigned nb = blocks_for(m, 4 * (64 / rm.p));
    if (rm.items == 1)
        hipLaunchKernelGGL(ln_fwd_kernel<1>, dim3(nb), dim3(kThreads), 0, as_stream(stream), x, res, gamma, beta, rowscale,
                           eps, m, c, rm, y, mean, rstd);
    else
        hipLaunchKernelGGL(ln_fwd_kernel<2>, dim3(nb), dim3(kThreads), 0, as_stream(stream), x, res, gamma, beta, rowscale,
                           eps, m, c, rm, y, mean, rstd);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

size_t seg3d_layernorm_bwd_workspace_bytes(int64_t m, int32_t c) {
    if (m < 0 || bad_c(c)) return 0;
    return (size_t)kLnBwdMaxBlocks * 2 * c * sizeof(float);
}

// dx now, the per-block partial sums of dgamma / dbeta left in the workspace as part[nblocks][2][c]: their fixed-order
// sum is a seg3d_reduce_partials job (n = 2 c, nw = c, dw = dgamma, db = dbeta) the caller runs alone or batched.
int seg3d_layernorm_bwd_partials(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                 const float* rowscale, int64_t m, int32_t c, float* dx, void* workspace,
                                 size_t workspace_bytes, int32_t* nblocks, void* stream) {
    if (m < 0 || bad_c(c) || c > 512 || !nblocks) return SEG3D_EINVAL;
    if (workspace_bytes < seg3d_layernorm_bwd_workspace_bytes(m, c) || !workspace) return SEG3D_EWORKSPACE;
    *nblocks = 0;
    if (m == 0) return SEG3D_OK;
    if (!dy || !x || !mean || !rstd || !gamma || !dx) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    const RowMap rm = row_map(c);
    unsigned nb = blocks_for(m, 4 * (64 / rm.p) * 4);
    if (nb > (unsigned)kLnBwdMaxBlocks) nb = kLnBwdMaxBlocks;
    const size_t smem = (size_t)4 * (64 / rm.p) * 2 * c * sizeof(float);
    if (rm.items == 1)
        hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(nb), dim3(kThreads), smem, st, dy, x, mean, rstd, gamma, rowscale, m, c, rm,
                           dx, part);
    else
        hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(nb), dim3(kThreads), smem, st, dy, x, mean, rstd, gamma, rowscale, m, c, rm,
                           dx, part);
    SEG3D_CHECK_LAUNCH();
    *nblocks = (int32_t)nb;
    return SEG3D_OK;
}

int seg3d_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        const float* rowscale, int64_t m, int32_t c, float* dx, float* dgamma, float* dbeta,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (m < 0 || bad_c(c) || c > 512 || !dgamma || !dbeta) return SEG3D_EINVAL;
    if (workspace_bytes < seg3d_layernorm_bwd_workspace_bytes(m, c) || !workspace) return SEG3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    unsigned nb = 0;
    if (m > 0) {
        if (!dy || !x || !mean || !rstd || !gamma || !dx) return SEG3D_EINVAL;
        const RowMap rm = row_map(c);
        nb = blocks_for(m, 4 * (64 / rm.p) * 4);  // ~4 row batches per wave
        if (nb > (unsigned)kLnBwdMaxBlocks) nb = kLnBwdMaxBlocks;
        const size_t smem = (size_t)4 * (64 / rm.p) * 2 * c * sizeof(float);
        if (rm.items == 1)
            hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(nb), dim3(kThreads), smem, st, dy, x, mean, rstd, gamma, rowscale, m, c,
                               rm, dx, part);
        else
            hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(nb), dim3(kThreads), smem, st, dy, x, mean, rstd, gamma, rowscale, m, c,
                               rm, dx, part);
        SEG3D_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(ln_bwd_reduce, dim3((unsigned)((2 * c + 31) / 32)), dim3(1024), 0, st, part, (int)nb, c, dgamma,
                       dbeta);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

size_t seg3d_batchnorm_workspace_bytes(int64_t m, int32_t c) {
    if (m < 0 || bad_c(c)) return 0;
    return (size_t)kColMaxBlocks * 2 * c * sizeof(float);
}

static unsigned col_blocks(int64_t m, int c) {
    const int lanes = kThreads / (c / 4);
    unsigned nb = blocks_for(m, lanes * 16);
    return nb > (unsigned)kColMaxBlocks ? (unsigned)kColMaxBlocks : nb;
}

int seg3d_colstats(const float* x, int64_t m, int32_t c, float* sums /*[2][c]: sum(x-x0), sum((x-x0)^2)*/,
                   void* workspace, size_t workspace_bytes, void* stream) {
    if (m <= 0 || bad_c(c) || !sums || !x) return SEG3D_EINVAL;
    if (!workspace || workspace_bytes < seg3d_batchnorm_workspace_bytes(m, c)) return SEG3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    const unsigned nb = col_blocks(m, c);
    hipLaunchKernelGGL(col_reduce_kernel<0>, dim3(nb), dim3(kThreads), kColSmem, st, x, nullptr, nullptr, nullptr, nullptr, 0, m,
                       c, part);
    SEG3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(col_sums_finish, dim3((unsigned)((2 * c + 31) / 32)), dim3(1024), 0, st, part, (int)nb, c, sums);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_batchnorm_stats(const float* x, int64_t m, int32_t c, float eps, const float* gamma, const float* beta,
                          float momentum, float* running_mean, float* running_var, float* stats /*[6][c]*/,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (m <= 0 || bad_c(c) || !x || !gamma || !beta || !stats || (running_mean && !running_var)) return SEG3D_EINVAL;
    if (!workspace || workspace_bytes < seg3d_batchnorm_workspace_bytes(m, c)) return SEG3D_EWORKSPACE;
    hipStream_t st = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    const unsigned nb = col_blocks(m, c);
    hipLaunchKernelGGL(col_reduce_kernel<0>, dim3(nb), dim3(kThreads), kColSmem, st, x, nullptr, nullptr, nullptr, nullptr, 0, m,
                       c, part);
    SEG3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((c + 31) / 32)), dim3(1024), 0, st, x, part, (int)nb, m, c, eps, gamma,
                       beta, momentum, running_mean, running_var, stats);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_gelu_fwd(const float* h, int64_t n, float* g, float* gp, void* stream) {
    if (n < 0 || (n & 3)) return SEG3D_EINVAL;
    if (n == 0) return SEG3D_OK;
    if (!h || !g) return SEG3D_EINVAL;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(blocks_for(n / 4, kThreads * 4)), dim3(kThreads), 0, as_stream(stream), h, n / 4, g,
                       gp);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_affine_act(const float* x, const float* res, const float* scale, const float* shift, int32_t relu, int64_t m,
                     int32_t c, float* y, void* stream) {
    if (m < 0 || bad_c(c)) return SEG3D_EINVAL;
    if (m == 0) return SEG3D_OK;
    if (!x || !scale || !shift || !y) return SEG3D_EINVAL;
    const int64_t tq = m * (c / 4);
    hipLaunchKernelGGL(affine_act_kernel, dim3(blocks_for(tq, kThreads * 4)), dim3(kThreads), 0, as_stream(stream), x, res,
                       scale, shift, relu, tq, c / 4, y);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_batchnorm_bwd(const float* dy, const float* y, const float* x, const float* mean, const float* rstd,
                        const float* gamma, int32_t relu, int64_t m, int32_t c, float* dx, float* dres,
                        float* sums /*[2][c] out: dbeta, dgamma*/, void* workspace, size_t workspace_bytes, void* stream) {
    if (m <= 0 || bad_c(c) || !sums) return SEG3D_EINVAL;
    if (!workspace || workspace_bytes < seg3d_batchnorm_workspace_bytes(m, c)) return SEG3D_EWORKSPACE;
    if (!dy || !x || !mean || !rstd || !gamma || !dx || (relu && !y)) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    const unsigned nb = col_blocks(m, c);
    hipLaunchKernelGGL(col_reduce_kernel<1>, dim3(nb), dim3(kThreads), kColSmem, st, x, dy, y, mean, rstd, relu, m, c, part);
    SEG3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(col_sums_finish, dim3((unsigned)((2 * c + 31) / 32)), dim3(1024), 0, st, part, (int)nb, c, sums);
    SEG3D_CHECK_LAUNCH();
    const int64_t tq = m * (c / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for(tq, kThreads * 4)), dim3(kThreads), 0, st, dy, y, x, mean, rstd,
                       gamma, sums, relu, 1.0f / (float)m, nullptr, tq, c / 4, dx, dres);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

// The two halves of seg3d_batchnorm_bwd as separate entries, for torch.nn.SyncBatchNorm (tools/train.py:246-247): the
// caller all-reduces `sums` over the ranks between them and hands the apply pass 1 / (total rows) as a device scalar.
int seg3d_batchnorm_bwd_reduce(const float* dy, const float* y, const float* x, const float* mean, const float* rstd,
                               int32_t relu, int64_t m, int32_t c, float* sums, void* workspace, size_t workspace_bytes,
                               void* stream) {
    if (m <= 0 || bad_c(c) || !sums) return SEG3D_EINVAL;
    if (!workspace || workspace_bytes < seg3d_batchnorm_workspace_bytes(m, c)) return SEG3D_EWORKSPACE;
    if (!dy || !x || !mean || !rstd || (relu && !y)) return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    float* part = static_cast<float*>(workspace);
    const unsigned nb = col_blocks(m, c);
    hipLaunchKernelGGL(col_reduce_kernel<1>, dim3(nb), dim3(kThreads), kColSmem, st, x, dy, y, mean, rstd, relu, m, c, part);
    SEG3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(col_sums_finish, dim3((unsigned)((2 * c + 31) / 32)), dim3(1024), 0, st, part, (int)nb, c, sums);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

int seg3d_batchnorm_bwd_apply(const float* dy, const float* y, const float* x, const float* mean, const float* rstd,
                              const float* gamma, const float* sums, const float* inv_count, int32_t relu, int64_t m,
                              int32_t c, float* dx, float* dres, void* stream) {
    if (m <= 0 || bad_c(c) || !sums || !inv_count) return SEG3D_EINVAL;
    if (!dy || !x || !mean || !rstd || !gamma || !dx || (relu && !y)) return SEG3D_EINVAL;
    const int64_t tq = m * (c / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for(tq, kThreads * 4)), dim3(kThreads), 0, as_stream(stream), dy, y, x,
                       mean, rstd, gamma, sums, relu, 0.0f, inv_count, tq, c / 4, dx, dres);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // extern "C"
