// Probe harness for wgrad_dense.hip: build with -DSEG3D_PROBE_* switches to see what bounds the kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I openseg3d_amd/csrc tools/wgrad_probe.hip -o /tmp/probe
#include "../openseg3d_amd/csrc/wgrad_dense.hip"

#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
    struct S { int64_t m; int cin, cout; };
    std::vector<S> shapes = {{58453, 192, 192}, {58453, 192, 384}, {121168, 96, 192}, {121168, 96, 96}, {19483, 384, 768}, {6943, 768, 768}};
    for (auto sh : shapes) {
        float *x, *dy, *dw, *db;
        void* ws;
        size_t nb = seg3d_linear_wgrad_workspace_bytes(sh.m, sh.cin, sh.cout);
        hipMalloc(&x, sh.m * sh.cin * 4);
        hipMalloc(&dy, sh.m * sh.cout * 4);
        hipMalloc(&dw, sh.cin * sh.cout * 4);
        hipMalloc(&db, sh.cout * 4);
        hipMalloc(&ws, nb);
        hipMemset(x, 0x3c, sh.m * sh.cin * 4);
        hipMemset(dy, 0x3c, sh.m * sh.cout * 4);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) seg3d_linear_wgrad(x, dy, sh.m, sh.cin, sh.cout, dw, db, ws, nb, nullptr);
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < 20; ++i) seg3d_linear_wgrad(x, dy, sh.m, sh.cin, sh.cout, dw, db, ws, nb, nullptr);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        Plan p = plan(sh.m, sh.cin, sh.cout);
        printf("m=%ld %d->%d  tiles %d chunks %d rows %ld: %.1f us\n", (long)sh.m, sh.cin, sh.cout, p.nbo * p.nbi, p.chunks,
               (long)p.rows, ms / 20 * 1e3);
        hipFree(x); hipFree(dy); hipFree(dw); hipFree(db); hipFree(ws);
    }
    return 0;
}
