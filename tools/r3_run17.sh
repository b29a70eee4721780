cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3q
timeout -k 10 900 python -m pytest tests/test_gpu_training.py -q -x -k "hip_graph or probe or bench_streams" > gpurun_out/r3q/three.txt 2>&1; rc=$?; tail -5 gpurun_out/r3q/three.txt; [ $rc = 0 ] || exit $rc
bash tools/r3_fulltests.sh
