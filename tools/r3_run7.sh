cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d7
timeout -k 10 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_layer.py -x -q > gpurun_out/r3d7/tests_attn.txt 2>&1; echo attn tests rc=$?
bash tools/ab_lib.sh "python tools/attn_bench.py" 2 > gpurun_out/r3d7/ab.txt 2>&1
bash tools/ab_lib.sh "python tools/attn_bench.py --drop 0.1" 1 > gpurun_out/r3d7/ab_drop.txt 2>&1
bash tools/ab_lib.sh "python tools/attn_bench.py --workload dense2m --iters 5" 1 > gpurun_out/r3d7/ab_dense.txt 2>&1
