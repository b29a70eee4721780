# evidence files for the round's last experiments (not rocprof products): profiles/r05_x6_bench.txt, r05_wgrad_dense_ab.txt
mkdir -p gpurun_out/r5e
python tools/x6_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5e/x6_bench.txt
{ for v in 0 1; do echo "== SEG3D_WGRAD_LDS=$v (kernel + fixed-order sum)"; SEG3D_WGRAD_LDS=$v python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids; echo "== SEG3D_WGRAD_LDS=$v (partial blocks only)"; SEG3D_WGRAD_LDS=$v python tools/wgrad_bench.py --partials 2>&1 | grep -v amdgpu.ids; done
  echo "== SEG3D_WGRAD_DENSE_DEPTH=3 (partial blocks only)"; SEG3D_WGRAD_DENSE_DEPTH=3 python tools/wgrad_bench.py --partials 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r5e/wgrad_dense_ab.txt
python tools/host_call_overhead.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5e/host_call_overhead.txt
tail -n 2 gpurun_out/r5e/x6_bench.txt; grep sum gpurun_out/r5e/wgrad_dense_ab.txt
