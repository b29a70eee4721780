cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d11
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py -x -q -s -k "knn_attention or hip_graph or every_parameter or multi_sweep or segformer_ms or fusion" > gpurun_out/r3d11/tests.txt 2>&1; echo tests rc=$?
python bench.py --workload multi_sweeps --batch 2 --scenes 2 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3d11/ms.json 2> gpurun_out/r3d11/ms.err; echo ms rc=$?
python tools/bf16_storage_probe.py > gpurun_out/r3d11/bf16_headline.txt 2>&1; echo probe rc=$?
python tools/bf16_storage_probe.py --dense > gpurun_out/r3d11/bf16_dense.txt 2>&1; echo probe dense rc=$?
