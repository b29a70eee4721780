cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d2
python -m pytest tests/test_gpu_training.py -x -q -k "probe or plain_ddp or two_gpus or side_stream or rehearsal or contract_line" > gpurun_out/r3d2/tests.txt 2>&1; echo tests rc=$?
cp openseg3d_amd/csrc/libseg3d_hip.so /tmp/libgood.so
cp openseg3d_amd/csrc/libS.so openseg3d_amd/csrc/libseg3d_hip.so
python tools/probes/attn_stamps.py > gpurun_out/r3d2/stamps.txt 2>&1; echo stamps rc=$?
cp /tmp/libgood.so openseg3d_amd/csrc/libseg3d_hip.so
python bench.py --workload dense2m --steps 4 --warmup 2 > gpurun_out/r3d2/dense.json 2> gpurun_out/r3d2/dense.err; echo dense rc=$?
