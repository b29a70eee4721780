# LDS-shared dense weight gradient: parity tests, then layer bench (full = kernel + reduce; --partials = kernel alone) off / on
mkdir -p gpurun_out/r5w
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py -x -q -k "wgrad or partial" > gpurun_out/r5w/t.log 2>&1; rc=$?; tail -n 12 gpurun_out/r5w/t.log
[ $rc = 0 ] || exit $rc
for v in 0 1; do echo "== SEG3D_WGRAD_LDS=$v (kernel + reduce)"; SEG3D_WGRAD_LDS=$v python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids; echo "== SEG3D_WGRAD_LDS=$v (partials only)"; SEG3D_WGRAD_LDS=$v python tools/wgrad_bench.py --partials 2>&1 | grep -v amdgpu.ids; done
for t in 256 1024; do echo "== LDS target $t (kernel + reduce)"; SEG3D_WGRAD_LDS_TARGET=$t python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids; done
