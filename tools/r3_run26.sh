cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do for v in 0 1; do
  SEG3D_CONV_DMA=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dma=$v', d['ms_per_step'], d['fwd_only']['ms_per_step'])"
done; done
