cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3x
timeout -k 10 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_parity.py -q -x > gpurun_out/r3x/tests.txt 2>&1; rc=$?; tail -3 gpurun_out/r3x/tests.txt; [ $rc = 0 ] || exit $rc
python bench.py --mode fwd --steps 10 --warmup 3 > gpurun_out/r3x/fwd.json 2> gpurun_out/r3x/fwd.err; python -c "
import json; d=json.loads(open('gpurun_out/r3x/fwd.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['parity']['max_abs_logit_diff'])"
