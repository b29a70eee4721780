# dense weight gradient: three steps of a wave in flight at one wave per SIMD
mkdir -p gpurun_out/r5w
SEG3D_WGRAD_DENSE_DEPTH=3 timeout -k 10 600 python -m pytest tests/test_gpu_dense.py -x -q -k "wgrad_matches or reproducible" > gpurun_out/r5w/t3.log 2>&1; rc=$?; tail -n 4 gpurun_out/r5w/t3.log
[ $rc = 0 ] || exit $rc
for v in 1 3; do echo "== depth $v target auto (partials only)"; SEG3D_WGRAD_DENSE_DEPTH=$v python tools/wgrad_bench.py --partials 2>&1 | grep -E "58453|19483|121168|sum"; for t in 256 512; do echo "== depth $v target $t (partials only)"; SEG3D_WGRAD_DENSE_DEPTH=$v SEG3D_WGRAD_TARGET=$t python tools/wgrad_bench.py --partials 2>&1 | grep -E "58453|19483|121168|sum"; done; done
echo "== depth 3 auto, kernel + reduce"; SEG3D_WGRAD_DENSE_DEPTH=3 python tools/wgrad_bench.py 2>&1 | grep -E "sum"
echo "== depth 1 auto, kernel + reduce"; python tools/wgrad_bench.py 2>&1 | grep -E "sum"
