cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3x
for rep in 1 2 3; do for v in 1 0; do
  SEG3D_INPROJ_SUM=$v python bench.py --mode fwd --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sum=$v', d['ms_per_step'])"
done; done
