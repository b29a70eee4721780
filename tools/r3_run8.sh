cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d8
timeout -k 10 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_layer.py -x -q > gpurun_out/r3d8/tests_attn.txt 2>&1; echo attn tests rc=$?
bash tools/ab_lib.sh "python tools/attn_bench.py --bwd --drop 0.1" 2 > gpurun_out/r3d8/ab_bwd.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_training.py -x -q > gpurun_out/r3d8/tests_train.txt 2>&1; echo train tests rc=$?
