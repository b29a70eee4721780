out=gpurun_out/r5n; mkdir -p $out
SEG3D_WGRAD_SB=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sparse_conv_forward_and_backward or wide_tiles or inverse_conv" > $out/parity.log 2>&1 || { tail -n 30 $out/parity.log; exit 1; }
tail -n 1 $out/parity.log
for rep in 1 2; do for v in 0 1; do SEG3D_WGRAD_SB=$v timeout -k 10 300 python tools/sparse_wgrad_bench.py --partials > $out/swg_sb$v.log 2>&1 || exit 1; echo "sb=$v $(grep sum $out/swg_sb$v.log)"; done; done
paste <(grep -v amdgpu $out/swg_sb0.log | awk '{print $1,$2,$4,$5,$6,$7,$8}') <(grep -v amdgpu $out/swg_sb1.log | awk '{print $8}')
