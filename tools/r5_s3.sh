# round 5, GPU session 3: bf16-copy weight-gradient kernels alone; centre-first unit order of the sparse weight gradient
out=gpurun_out/r5e; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py -x -q -k "row_streaming" > $out/dense_tests.log 2>&1 || { tail -n 40 $out/dense_tests.log; exit 1; }
tail -n 1 $out/dense_tests.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sparse_conv_forward_and_backward or wide_tiles or inverse_conv" > $out/parity.log 2>&1 || { tail -n 30 $out/parity.log; exit 1; }
tail -n 1 $out/parity.log
for v in 0 1 0 1; do SEG3D_WGRAD_CENTER_FIRST=$v timeout -k 10 300 python tools/sparse_wgrad_bench.py --partials > $out/swg_cf$v.log 2>&1 || exit 1; grep sum $out/swg_cf$v.log; cp $out/swg_cf$v.log $out/swg_cf${v}_last.log; done
timeout -k 10 300 python tools/sparse_wgrad_bench.py --xbf16 > $out/swg_xb.log 2>&1 || exit 1
paste <(grep -v amdgpu $out/swg_cf0.log | awk '{print $1,$2,$4,$5,$6,$7,$8}') <(grep -v amdgpu $out/swg_cf1.log | awk '{print $8}') <(grep -v amdgpu $out/swg_xb.log | awk '{print $8}')
timeout -k 10 300 python tools/wgrad_bench.py --partials > $out/dwg_f32.log 2>&1 || exit 1
timeout -k 10 300 python tools/wgrad_bench.py --xbf16 > $out/dwg_xb.log 2>&1 || exit 1
paste <(grep -v amdgpu $out/dwg_f32.log | awk '{print $1,$2,$3,$4,$5}') <(grep -v amdgpu $out/dwg_xb.log | awk '{print $3,$4,$5}')
