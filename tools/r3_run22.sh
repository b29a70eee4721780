cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3m
for b in 0 512 2048 8192 -1; do
  echo "== SEG3D_SUBM_ORDER=$b"
  SEG3D_SUBM_ORDER=$b python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r3m/conv.txt 2>&1
grep "==\|total" gpurun_out/r3m/conv.txt
