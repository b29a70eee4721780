cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3ln
timeout -k 10 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_parity.py -q -x > gpurun_out/r3ln/tests.txt 2>&1; rc=$?; tail -2 gpurun_out/r3ln/tests.txt; [ $rc = 0 ] || exit $rc
for rep in 1 2 3; do for v in 1 0; do
  SEG3D_LINEAR_LN=$v python bench.py --mode fwd --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('linear_ln=$v', d['ms_per_step'])"
done; done
