// Issue rate of packed fp32 vector instructions on gfx950 against their scalar forms (one question: does a v_pk_fma_f32
// cost one issue slot or two?).  hipcc --offload-arch=gfx950 -O3 tools/probes/pk_rate_probe.hip -o tools/probes/pk_rate_probe
// Prints ns per wave-instruction for chains of independent v_fma_f32, v_pk_fma_f32, v_pk_add_f32, v_pk_mul_f32 and v_exp_f32
// at full occupancy (8 waves per SIMD), i.e. the throughput figure.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed) {
    float a[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = seed + i + threadIdx.x;
        p[i] = (f32x2){seed + i, seed - i};
    }
    const float m = 1.0000001f;
    const f32x2 m2 = {m, m};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, 1e-9f);
            if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], m2, (f32x2){1e-9f, 1e-9f});
            if (MODE == 2) p[i] = p[i] + m2;
            if (MODE == 3) p[i] = p[i] * m2;
            if (MODE == 4) a[i] = __builtin_amdgcn_exp2f(a[i]) * 0.5f;  // (one v_exp + one v_mul per element)
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
double run(const char* name, int per_iter_instr) {
    float* out;
    hipMalloc(&out, 4);
    const int blocks = 256 * 8, iters = 20000;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions issued per SIMD: 8 waves x iters x per_iter_instr
    const double per_simd = 8.0 * iters * per_iter_instr;
    const double ns = ms * 1e6 / per_simd;
    printf("%-14s %8.3f ms  %6.3f ns per wave-instruction per SIMD  (= %.2f cycles at 2.1 GHz)\n", name, ms, ns, ns * 2.1);
    hipFree(out);
    return ns;
}

int main() {
    run<0>("v_fma_f32", 8);
    run<1>("v_pk_fma_f32", 8);
    run<2>("v_pk_add_f32", 8);
    run<3>("v_pk_mul_f32", 8);
    run<4>("v_exp+v_mul", 16);
    return 0;
}
