cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d15
timeout -k 10 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_layer.py -x -q > gpurun_out/r3d15/tests.txt 2>&1; echo tests rc=$?
bash tools/ab_lib.sh "python tools/attn_bench.py --bwd --drop 0.1" 2 > gpurun_out/r3d15/ab_bwd.txt 2>&1
