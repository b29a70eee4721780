cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3l
timeout -k 10 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_norms.py tests/test_gpu_training.py -q -x > gpurun_out/r3l/tests.txt 2>&1; rc=$?; tail -5 gpurun_out/r3l/tests.txt; [ $rc = 0 ] || exit $rc
python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3l/bench.json 2> gpurun_out/r3l/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r3l/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['fwd_only']['ms_per_step'], d['trained_weights_l1'])"
SEG3D_WGRAD_DEFER=0 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3l/bench_nodefer.json 2> gpurun_out/r3l/bench_nodefer.err; python -c "
import json; d=json.loads(open('gpurun_out/r3l/bench_nodefer.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['fwd_only']['ms_per_step'], d['trained_weights_l1'])"
