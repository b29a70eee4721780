cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d16
for v in 0 1 0 1; do echo "== DH6 $v"; SEG3D_ATTN_BWD_DH6=$v python tools/attn_bench.py --bwd --drop 0.1 --stages 0 2>&1 | grep -v amdgpu; done > gpurun_out/r3d16/dh6.txt
SEG3D_ATTN_BWD_DH6=1 timeout -k 10 600 python -m pytest tests/test_gpu_attention.py -x -q > gpurun_out/r3d16/tests.txt 2>&1; echo tests rc=$?
