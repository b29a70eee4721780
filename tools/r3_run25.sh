cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3g
timeout -k 10 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_parity.py tests/test_gpu_training.py -q -x > gpurun_out/r3g/tests.txt 2>&1; rc=$?; tail -3 gpurun_out/r3g/tests.txt; [ $rc = 0 ] || exit $rc
for rep in 1 2; do for v in 1 0; do
  SEG3D_GELU_SAVED_GRAD=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gelu_saved=$v', d['ms_per_step'], d['trained_weights_l1'])"
done; done
