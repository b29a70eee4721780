cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3full
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r3full/gpu_tests.txt 2>&1; echo gpu tests rc=$?
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3full/smoke.txt 2>&1; echo smoke rc=$?
