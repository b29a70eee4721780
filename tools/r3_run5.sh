cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d5
for w in 1 2 3; do echo "== WGS_PER_CU $w"; SEG3D_ATTN_WGS_PER_CU=$w python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r3d5/wgs.txt
cp openseg3d_amd/csrc/libS.so openseg3d_amd/csrc/libseg3d_hip.so
python tools/probes/attn_stamps.py > gpurun_out/r3d5/stamps.txt 2>&1; echo stamps rc=$?
