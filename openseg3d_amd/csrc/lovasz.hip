// Lovasz-softmax on device (seg3d/models/losses/lovasz_loss.py:13-26 lovasz_grad, :118-158
// lovasz_softmax_flat, :268-290 LovaszLoss.forward with the defaults build_criterion uses: multi_class, per_image
// False, ignore_index 255).  The reference loops over the C classes and for each one sorts all N errors, gathers,
// runs two cumsums and a dot (C x ~12 launches, three times per training step).  Here every class is handled at once:
//
//   lovasz_keys   one thread per row: softmax, then for every class c the pair
//                   key = c << 32 | bits(|[label == c] - p_c|)      (errors are in [0, 1]: bit order == value order)
//                   val = row | [label == c] << 31                  (rows with an ignored label: error 0, fg 0 -- they
//                                                                    sort behind every positive error and weigh 0)
//   rocprim::radix_sort_pairs_desc over the low 32 + log2(C) key bits: class C-1-c occupies [c*n, (c+1)*n), errors
//                 descending inside it (one full-chip sort instead of C single-segment sorts)
//   lovasz_count / lovasz_offsets   foreground count of every 2048-element chunk, exclusive scan per class, totals
//   lovasz_grad   per chunk: running foreground count -> jaccard(k) - jaccard(k-1) in the reference's float32
//                 arithmetic, partial dot with the sorted errors, and coef[row][c] = d loss_c / d p_c scattered back
//   lovasz_finalize  fixed-order sum of the chunk partials, class selection ('present' | 'all' | mask), weights, mean
// Backward is one pass: dlogits = p * (dp - <p, dp>) with dp_c = coef[row][c] * cscale[c] * g.
// No atomics, no memsets: every workspace word that is read was written by an earlier kernel of the same call.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kItems = 8;
constexpr int kChunk = kThreads * kItems;  // 2048 sorted elements per workgroup
constexpr int kMaxClasses = 64;

// softmax(row)[j] == expf(row[j] - m) / s; the row is re-read from L1 instead of being held in a runtime-indexed
// register array (which the compiler would put in scratch)
__device__ __forceinline__ void softmax_stats(const float* __restrict__ row, int c, float* m_out, float* s_out) {
    float m = row[0];
    for (int j = 1; j < c; ++j) m = fmaxf(m, row[j]);
    float s = 0.f;
    for (int j = 0; j < c; ++j) s += expf(row[j] - m);
    *m_out = m;
    *s_out = s;
}

__global__ __launch_bounds__(kThreads) void lovasz_keys(const float* __restrict__ x, const int64_t* __restrict__ label,
                                                        int n, int c, int64_t ignore_index,
                                                        unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int r = blockIdx.x * kThreads + threadIdx.x;
    if (r >= n) return;
    const float* row = x + (int64_t)r * c;
    float m, s;
    softmax_stats(row, c, &m, &s);
    const int64_t y = label[r];
    const bool valid = y != ignore_index;
    for (int j = 0; j < c; ++j) {
        const bool fg = valid && y == j;
        const float err = valid ? fabsf((fg ? 1.f : 0.f) - expf(row[j] - m) / s) : 0.f;
        const int64_t at = (int64_t)j * n + r;
        keys[at] = ((unsigned long long)j << 32) | __float_as_uint(err);
        vals[at] = (uint32_t)r | (fg ? 0x80000000u : 0u);
    }
}

__global__ __launch_bounds__(kThreads) void lovasz_count(const uint32_t* __restrict__ vals, int n, int nchunks,
                                                         int* __restrict__ chunk_cnt /*[c][nchunks]*/) {
    const int seg = blockIdx.y, chunk = blockIdx.x;
    const uint32_t* v = vals + (int64_t)seg * n;
    int cnt = 0;
    const int base = chunk * kChunk + threadIdx.x * kItems;
    for (int j = 0; j < kItems; ++j)
        if (base + j < n) cnt += (int)(v[base + j] >> 31);
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, SEG3D_WAVE);
    __shared__ int red[kThreads / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < kThreads / 64; ++w) t += red[w];
        chunk_cnt[seg * nchunks + chunk] = t;
    }
}

// one wave per sorted segment: exclusive scan of its chunk counts, total foreground count
__global__ __launch_bounds__(64) void lovasz_offsets(const int* __restrict__ chunk_cnt, int nchunks,
                                                     int* __restrict__ chunk_off, int* __restrict__ total) {
    const int seg = blockIdx.x, lane = threadIdx.x;
    int carry = 0;
    for (int t0 = 0; t0 < nchunks; t0 += 64) {
        const int i = t0 + lane;
        const int v = i < nchunks ? chunk_cnt[seg * nchunks + i] : 0;
        int inc = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off, SEG3D_WAVE);
            if (lane >= off) inc += o;
        }
        if (i < nchunks) chunk_off[seg * nchunks + i] = carry + inc - v;
        carry += __shfl(inc, 63, SEG3D_WAVE);
    }
    if (lane == 0) total[seg] = carry;
}

__global__ __launch_bounds__(kThreads) void lovasz_grad(const unsigned long long* __restrict__ keys,
                                                        const uint32_t* __restrict__ vals, int n, int c, int nchunks,
                                                        const int* __restrict__ chunk_off, const int* __restrict__ total,
                                                        float* __restrict__ coef /*[n][c]*/,
                                                        float* __restrict__ part /*[c][nchunks]*/) {
    const int seg = blockIdx.y, chunk = blockIdx.x;
    const int cls = c - 1 - seg;  // descending sort on the class bits
    const int64_t seg0 = (int64_t)seg * n;
    const int base = chunk * kChunk + threadIdx.x * kItems;
    uint32_t v[kItems];
    float err[kItems];
    int mine = 0;
    for (int j = 0; j < kItems; ++j) {
        const bool in = base + j < n;
        v[j] = in ? vals[seg0 + base + j] : 0u;
        err[j] = in ? __uint_as_float((uint32_t)keys[seg0 + base + j]) : 0.f;
        mine += (int)(v[j] >> 31);
    }
    // exclusive scan of the per-thread foreground counts over the workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, SEG3D_WAVE);
        if (lane >= off) inc += o;
    }
    __shared__ int wsum[kThreads / 64];
    __shared__ float wdot[kThreads / 64];
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = chunk_off[seg * nchunks + chunk] + inc - mine;
    for (int w = 0; w < wave; ++w) before += wsum[w];

    const float gts = (float)total[seg];
    float dot = 0.f;
    int cum = before;
    for (int j = 0; j < kItems; ++j) {
        const int k = base + j;
        if (k >= n) break;
        const int fg = (int)(v[j] >> 31);
        const int cum_prev = cum;
        cum += fg;
        // lovasz_grad, lovasz_loss.py:17-25: jaccard = 1 - (gts - cumsum(gt)) / (gts + cumsum(1 - gt)), then differences
        float g = 1.f - (gts - (float)cum) / (gts + (float)(k + 1 - cum));
        if (k > 0) g -= 1.f - (gts - (float)cum_prev) / (gts + (float)(k - cum_prev));
        dot += err[j] * g;
        // d|fg - p| / dp: -1 on foreground rows, +1 on background rows, 0 where the error is exactly 0 (torch abs')
        const float sgn = err[j] == 0.f ? 0.f : (fg ? -1.f : 1.f);
        coef[(int64_t)(v[j] & 0x7FFFFFFFu) * c + cls] = sgn * g;
    }
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, SEG3D_WAVE);
    if (lane == 0) wdot[wave] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < kThreads / 64; ++w) t += wdot[w];
        part[seg * nchunks + chunk] = t;
    }
}

// stats[0] = loss, stats[1] = number of classes averaged, stats[2 + cls] = d loss / d loss_cls
__global__ __launch_bounds__(kMaxClasses) void lovasz_finalize(const float* __restrict__ part, const int* __restrict__ total,
                                                               int c, int nchunks, int mode,
                                                               const int32_t* __restrict__ include,
                                                               const float* __restrict__ class_weight,
                                                               float* __restrict__ stats) {
    __shared__ float loss_c[kMaxClasses];
    __shared__ float w_c[kMaxClasses];
    const int cls = threadIdx.x;
    if (cls < c) {
        const int seg = c - 1 - cls;
        float s = 0.f;
        for (int i = 0; i < nchunks; ++i) s += part[seg * nchunks + i];
        bool use = mode == 0 ? total[seg] > 0 : true;  // 'present' (lovasz_loss.py:143-144) | 'all'
        if (include) use = include[cls] != 0;          // explicit class list: no presence test (lovasz_loss.py:140)
        const float w = use ? (class_weight ? class_weight[cls] : 1.f) : 0.f;
        loss_c[cls] = use ? s * w : 0.f;
        w_c[cls] = use ? w : -1.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sum = 0.f;
        int used = 0;
        for (int j = 0; j < c; ++j)
            if (w_c[j] >= 0.f) {
                sum += loss_c[j];
                ++used;
            }
        stats[0] = used ? sum / (float)used : 0.f;
        stats[1] = (float)used;
        for (int j = 0; j < c; ++j) stats[2 + j] = (w_c[j] >= 0.f && used) ? w_c[j] / (float)used : 0.f;
    }
}

__global__ __launch_bounds__(kThreads) void lovasz_bwd(const float* __restrict__ x, const float* __restrict__ coef,
                                                       const float* __restrict__ stats, const float* __restrict__ gout,
                                                       int n, int c, float* __restrict__ dx) {
    const int r = blockIdx.x * kThreads + threadIdx.x;
    if (r >= n) return;
    const float* row = x + (int64_t)r * c;
    float m, s;
    softmax_stats(row, c, &m, &s);
    const float g = gout[0];
    const float* cf = coef + (int64_t)r * c;
    float dot = 0.f;
    for (int j = 0; j < c; ++j) dot += expf(row[j] - m) / s * (cf[j] * stats[2 + j] * g);
    float* out = dx + (int64_t)r * c;
    for (int j = 0; j < c; ++j) out[j] = expf(row[j] - m) / s * (cf[j] * stats[2 + j] * g - dot);
}

using Key = unsigned long long;

int class_bits(int c) {
    int b = 0;
    while ((1 << b) < c) ++b;
    return b;
}

struct Plan {
    int64_t total;
    int nchunks;
    size_t sort_bytes;
};

bool make_plan(int64_t n, int c, Plan* pl) {
    pl->total = n * c;
    pl->nchunks = (int)ceil_div64(n, kChunk);
    pl->sort_bytes = 0;
    if (pl->total == 0) return true;
    rocprim::double_buffer<Key> k(nullptr, nullptr);
    rocprim::double_buffer<uint32_t> v(nullptr, nullptr);
    size_t bytes = 0;
    if (rocprim::radix_sort_pairs_desc(nullptr, bytes, k, v, (size_t)pl->total, 0u, 32u + class_bits(c)) != hipSuccess)
        return false;
    pl->sort_bytes = bytes;
    return true;
}

}  // namespace

// 0 when the arguments are invalid or no device is visible (the sort's scratch size is asked of rocPRIM, which sizes
// it for the current device).
extern "C" size_t seg3d_lovasz_workspace_bytes(int64_t n, int32_t c) {
    if (n < 0 || c <= 0 || c > kMaxClasses || n * c >= (int64_t)1 << 31) return 0;
    Plan pl;
    if (!make_plan(n, c, &pl)) return 0;
    const size_t t = (size_t)pl.total, cc = (size_t)c * pl.nchunks;
    return 2 * align_up(t * sizeof(Key), 256) + 2 * align_up(t * sizeof(uint32_t), 256) + align_up(pl.sort_bytes, 256) +
           2 * align_up(cc * sizeof(int), 256) + align_up(cc * sizeof(float), 256) + align_up(c * sizeof(int), 256) + 256;
}

extern "C" int seg3d_lovasz_softmax_fwd(const float* logits, const int64_t* labels, int64_t n, int32_t c,
                                        int64_t ignore_index, int32_t classes_mode, const int32_t* include,
                                        const float* class_weight, float* coef, float* stats, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    if (n < 0 || c <= 0 || c > kMaxClasses || n * c >= (int64_t)1 << 31 || !stats || (classes_mode != 0 && classes_mode != 1))
        return SEG3D_EINVAL;
    hipStream_t st = as_stream(stream);
    Plan pl;
    if (!make_plan(n, c, &pl)) return SEG3D_ELAUNCH;
    const size_t need = seg3d_lovasz_workspace_bytes(n, c);
    if (!workspace || need == 0 || workspace_bytes < need) return SEG3D_EWORKSPACE;
    if (n > 0 && (!logits || !labels || !coef)) return SEG3D_EINVAL;
    const size_t t = (size_t)pl.total, cc = (size_t)c * (size_t)(pl.nchunks > 0 ? pl.nchunks : 0);
    WsCarver ws(workspace);
    Key* k0 = ws.take<Key>(t);
    Key* k1 = ws.take<Key>(t);
    uint32_t* v0 = ws.take<uint32_t>(t);
    uint32_t* v1 = ws.take<uint32_t>(t);
    void* sort_tmp = ws.take<char>(pl.sort_bytes);
    int* chunk_cnt = ws.take<int>(cc);
    int* chunk_off = ws.take<int>(cc);
    float* part = ws.take<float>(cc);
    int* total = ws.take<int>((size_t)c);
    const int ni = (int)n;
    if (n > 0) {
        hipLaunchKernelGGL(lovasz_keys, dim3((unsigned)ceil_div64(n, kThreads)), dim3(kThreads), 0, st, logits, labels, ni, c,
                           ignore_index, k0, v0);
        SEG3D_CHECK_LAUNCH();
        rocprim::double_buffer<Key> kb(k0, k1);
        rocprim::double_buffer<uint32_t> vb(v0, v1);
        size_t bytes = pl.sort_bytes;
        SEG3D_CHECK_HIP(rocprim::radix_sort_pairs_desc(sort_tmp, bytes, kb, vb, t, 0u, 32u + class_bits(c), st));
        const dim3 grid((unsigned)pl.nchunks, (unsigned)c);
        hipLaunchKernelGGL(lovasz_count, grid, dim3(kThreads), 0, st, vb.current(), ni, pl.nchunks, chunk_cnt);
        SEG3D_CHECK_LAUNCH();
        hipLaunchKernelGGL(lovasz_offsets, dim3((unsigned)c), dim3(64), 0, st, chunk_cnt, pl.nchunks, chunk_off, total);
        SEG3D_CHECK_LAUNCH();
        hipLaunchKernelGGL(lovasz_grad, grid, dim3(kThreads), 0, st, kb.current(), vb.current(), ni, c, pl.nchunks, chunk_off,
                           total, coef, part);
        SEG3D_CHECK_LAUNCH();
    } else {
        hipLaunchKernelGGL(lovasz_offsets, dim3((unsigned)c), dim3(64), 0, st, chunk_cnt, 0, chunk_off, total);
        SEG3D_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(lovasz_finalize, dim3(1), dim3(kMaxClasses), 0, st, part, total, c, pl.nchunks, classes_mode, include,
                       class_weight, stats);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

extern "C" int seg3d_lovasz_softmax_bwd(const float* logits, const float* coef, const float* stats, const float* grad_out,
                                        int64_t n, int32_t c, float* dlogits, void* stream) {
    if (n < 0 || c <= 0 || c > kMaxClasses || n * c >= (int64_t)1 << 31 || !stats || !grad_out) return SEG3D_EINVAL;
    if (n == 0) return SEG3D_OK;
    if (!logits || !coef || !dlogits) return SEG3D_EINVAL;
    hipLaunchKernelGGL(lovasz_bwd, dim3((unsigned)ceil_div64(n, kThreads)), dim3(kThreads), 0, as_stream(stream), logits, coef,
                       stats, grad_out, (int)n, c, dlogits);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}
