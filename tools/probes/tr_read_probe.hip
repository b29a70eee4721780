// Semantics probe of ds_read_b64_tr_b16 (gfx950) as used by the fused attention kernels:
// per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a 4 x 16 block of 16-bit elements;
// lane i receives column i of the 4 rows (row q in element q).  Prints PASS/FAIL.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int kStride = 144;  // shorts per row (288 B), as the attention image uses
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * kStride];
    for (int i = threadIdx.x; i < 64 * kStride; i += 64) lds[i] = (short)((i / kStride) * 256 + (i % kStride));
    __syncthreads();
    const int lane = threadIdx.x;
    const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int col0 = 32;  // block columns 32..47
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(lds + (4 * grp + q) * kStride + col0 + 4 * p));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = r[j];
}
int main() {
    short* d;
    hipMalloc(&d, 64 * 4 * sizeof(short));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    std::vector<short> h(256);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j) {
            const int grp = lane >> 4, i = lane & 15;
            const short want = (short)((4 * grp + j) * 256 + 32 + i);
            if (h[lane * 4 + j] != want) {
                if (bad < 8) printf("lane %d elem %d: got row %d col %d, want row %d col %d\n", lane, j, h[lane * 4 + j] >> 8,
                                    h[lane * 4 + j] & 255, want >> 8, want & 255);
                ++bad;
            }
        }
    printf(bad ? "FAIL %d\n" : "PASS\n", bad);
    return bad != 0;
}
