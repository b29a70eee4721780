cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/r3_fulltests.sh || exit 1
mkdir -p gpurun_out/r3f2
python bench.py > gpurun_out/r3f2/default.json 2> gpurun_out/r3f2/default.err; echo default rc=$?
python -c "
import json; d=json.loads(open('gpurun_out/r3f2/default.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['fwd_only']['ms_per_step'], d['parity']['max_abs_logit_diff'], d['roofline']['frac'], d['cpu_baseline']['value'])"
