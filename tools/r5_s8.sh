out=gpurun_out/r5j; mkdir -p $out
for b in 1 2 4 13 26; do SEG3D_WGRAD_XCD_BLOCK=$b timeout -k 10 300 python tools/sparse_wgrad_bench.py --partials > $out/swg_xb$b.log 2>&1 || exit 1; echo "xcd block $b: $(grep sum $out/swg_xb$b.log)"; done
paste <(grep -v amdgpu $out/swg_xb1.log | awk '{print $1,$2,$4,$5,$6,$7,$8}') <(grep -v amdgpu $out/swg_xb2.log | awk '{print $8}') <(grep -v amdgpu $out/swg_xb4.log | awk '{print $8}') <(grep -v amdgpu $out/swg_xb13.log | awk '{print $8}') <(grep -v amdgpu $out/swg_xb26.log | awk '{print $8}')
SEG3D_WGRAD_XCD_BLOCK=13 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sparse_conv_forward_and_backward or wide_tiles or inverse_conv" > $out/parity.log 2>&1 || { tail -n 30 $out/parity.log; exit 1; }
tail -n 1 $out/parity.log
timeout -k 10 300 python tools/attn_bench.py --bwd --drop 0.1 > $out/attn.txt 2>&1; grep -v amdgpu $out/attn.txt | tail -n 3
