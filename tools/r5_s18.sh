# x6 kernel variants: 2 tiles per wave at 3 / 4 workgroups per CU, 4 tiles per wave
mkdir -p gpurun_out/r5x
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py -x -q -k "six_product or eval_point_mlp" > gpurun_out/r5x/t2.log 2>&1; rc=$?; tail -n 5 gpurun_out/r5x/t2.log
[ $rc = 0 ] || exit $rc
for v in "2 3" "4 3"; do set -- $v; echo "== RT $1 WPS $2"; SEG3D_X6_RT=$1 SEG3D_X6_WPS=$2 python tools/x6_bench.py 2>&1 | grep -v amdgpu.ids || exit 1; done
