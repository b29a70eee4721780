out=gpurun_out/r5f; mkdir -p $out
timeout -k 10 300 python tools/sparse_wgrad_bench.py --partials > $out/swg_f32.log 2>&1 || exit 1
timeout -k 10 300 python tools/sparse_wgrad_bench.py --xbf16 > $out/swg_xb.log 2>&1 || exit 1
paste <(grep -v amdgpu $out/swg_f32.log | awk '{print $1,$2,$4,$5,$6,$7,$8}') <(grep -v amdgpu $out/swg_xb.log | awk '{print $8}')
cp openseg3d_amd/csrc/libseg3d_hip.so /tmp/lib_keep.so && cp openseg3d_amd/csrc/libW.so openseg3d_amd/csrc/libseg3d_hip.so
timeout -k 10 300 python tools/probes/wgrad_stamps.py > $out/stamps.log 2>&1; rc=$?
cp /tmp/lib_keep.so openseg3d_amd/csrc/libseg3d_hip.so
grep -v amdgpu $out/stamps.log
exit $rc
