cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3dd
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -x -k "conv or segformer" > gpurun_out/r3dd/tests.txt 2>&1; rc=$?; tail -2 gpurun_out/r3dd/tests.txt; [ $rc = 0 ] || exit $rc
for rep in 1 2; do for d in 1 2; do
  echo "== SEG3D_CONV_DEPTH=$d"
  SEG3D_CONV_DEPTH=$d python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids
done; done > gpurun_out/r3dd/conv.txt 2>&1
grep "==\|total" gpurun_out/r3dd/conv.txt
