// Device-wide exclusive prefix sums (uint32 and 4 x uint32), two launches:
//   1. per-chunk totals (2048 elements per 256-thread workgroup)
//   2. every workgroup re-derives its base from the chunk totals (<= a few thousand values, L2-resident)
//      and scans its own chunk.
// Used by the voxelizer (first-seen ranks), the group index (CSR offsets), the window partition
// (compact window ids per batching level) and the strided-conv site generation (bitmap popcounts).
#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kItems = 8;
constexpr int kChunk = kThreads * kItems;

__device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint4 add(uint4 a, uint4 b) {
    return make_uint4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ uint32_t zero_of(uint32_t) { return 0u; }
__device__ __forceinline__ uint4 zero_of(uint4) { return make_uint4(0, 0, 0, 0); }

__device__ __forceinline__ uint32_t shfl_up_t(uint32_t v, int d) { return __shfl_up(v, d, SEG3D_WAVE); }
__device__ __forceinline__ uint4 shfl_up_t(uint4 v, int d) {
    return make_uint4(__shfl_up(v.x, d, SEG3D_WAVE), __shfl_up(v.y, d, SEG3D_WAVE),
                      __shfl_up(v.z, d, SEG3D_WAVE), __shfl_up(v.w, d, SEG3D_WAVE));
}

// inclusive scan across the 256 threads of a workgroup; returns this thread's inclusive value and
// the workgroup total through *total.
template <typename T>
__device__ T block_inclusive(T v, T* lds /*[4]*/, T* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) {
        T o = shfl_up_t(v, d);
        if (lane >= d) v = add(v, o);
    }
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    T base = zero_of(v), tot = zero_of(v);
    for (int w = 0; w < kThreads / 64; ++w) {
        if (w < wave) base = add(base, lds[w]);
        tot = add(tot, lds[w]);
    }
    __syncthreads();
    *total = tot;
    return add(v, base);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void chunk_totals(const T* __restrict__ in, int64_t n, T* __restrict__ sums) {
    __shared__ T lds[4];
    const int64_t base = (int64_t)blockIdx.x * kChunk + (int64_t)threadIdx.x * kItems;
    T acc = zero_of(T());
#pragma unroll
    for (int i = 0; i < kItems; ++i)
        if (base + i < n) acc = add(acc, in[base + i]);
    T tot;
    block_inclusive(acc, lds, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void chunk_scan(const T* __restrict__ in, T* __restrict__ out, int64_t n,
                                                       const T* __restrict__ sums, T* __restrict__ total) {
    __shared__ T lds[4];
    // base of this chunk = sum of the totals of all earlier chunks
    T part = zero_of(T());
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += kThreads) part = add(part, sums[j]);
    T chunk_base;
    block_inclusive(part, lds, &chunk_base);

    const int64_t base = (int64_t)blockIdx.x * kChunk + (int64_t)threadIdx.x * kItems;
    T v[kItems];
    T acc = zero_of(T());
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        v[i] = (base + i < n) ? in[base + i] : zero_of(T());
        acc = add(acc, v[i]);
    }
    T tot;
    T incl = block_inclusive(acc, lds, &tot);
    // exclusive prefix of this thread = incl - acc, rebuilt additively (no subtraction for uint4)
    // -> recompute from the neighbour: shuffle is per wave only, so derive it by replaying.
    // incl includes acc; exclusive = incl with acc removed.  uint32 wrap-around arithmetic is exact.
    T run;
    if constexpr (sizeof(T) == 4) {
        run = add(chunk_base, incl - acc);
    } else {
        run = add(chunk_base, make_uint4(incl.x - acc.x, incl.y - acc.y, incl.z - acc.z, incl.w - acc.w));
    }
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        if (base + i < n) out[base + i] = run;
        run = add(run, v[i]);
    }
    if (total != nullptr && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = add(chunk_base, tot);
}

template <typename T>
int scan_impl(const T* in, T* out, int64_t n, T* total, T* tmp, hipStream_t st) {
    if (n < 0) return SEG3D_EINVAL;
    const int64_t nb = ceil_div64(n > 0 ? n : 1, kChunk);
    if (nb > 1) {
        hipLaunchKernelGGL(chunk_totals<T>, dim3((unsigned)nb), dim3(kThreads), 0, st, in, n, tmp);
        SEG3D_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(chunk_scan<T>, dim3((unsigned)nb), dim3(kThreads), 0, st, in, out, n, tmp, total);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // namespace

int scan_exclusive_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total, uint32_t* tmp,
                       hipStream_t st) {
    return scan_impl<uint32_t>(in, out, n, total, tmp, st);
}

int scan_exclusive_u32x4(const uint4* in, uint4* out, int64_t n, uint4* total, uint4* tmp, hipStream_t st) {
    return scan_impl<uint4>(in, out, n, total, tmp, st);
}
