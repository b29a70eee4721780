cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d6
timeout -k 10 600 python -m pytest tests/test_gpu_attention.py -x -q > gpurun_out/r3d6/tests_attn.txt 2>&1; echo attn tests rc=$?
bash tools/ab_lib.sh "python tools/attn_bench.py" 2 > gpurun_out/r3d6/ab.txt 2>&1
python tools/attn_bench.py --drop 0.1 > gpurun_out/r3d6/drop.txt 2>&1
cp openseg3d_amd/csrc/libS.so openseg3d_amd/csrc/libseg3d_hip.so
python tools/probes/attn_stamps.py > gpurun_out/r3d6/stamps.txt 2>&1; echo stamps rc=$?
