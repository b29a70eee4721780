cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d10
timeout -k 10 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_attention.py -x -q -s -k "hip_graph or every_parameter or unsupported_head" > gpurun_out/r3d10/tests.txt 2>&1; echo tests rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d10/ms -- python3 bench.py --workload multi_sweeps --batch 2 --scenes 2 --mode fwd --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3d10/ms.json 2> gpurun_out/r3d10/ms.err; echo ms rc=$?
