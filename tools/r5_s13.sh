out=gpurun_out/r5q; mkdir -p $out
SEG3D_WGRAD_LEAN=1 timeout -k 10 600 python -m pytest tests/test_gpu_dense.py -x -q -k "wgrad" > $out/dense_tests.log 2>&1 || { tail -n 40 $out/dense_tests.log; exit 1; }
tail -n 1 $out/dense_tests.log
for rep in 1 2; do for v in 0 1; do SEG3D_WGRAD_LEAN=$v timeout -k 10 300 python tools/wgrad_bench.py > $out/dwg_$v.log 2>&1 || exit 1; done; done
paste <(grep -v amdgpu $out/dwg_0.log | awk '{print $1,$2,$3,$4,$5}') <(grep -v amdgpu $out/dwg_1.log | awk '{print $4,$5}')
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sparse_conv_forward_and_backward or wide_tiles or inverse_conv" > $out/parity.log 2>&1 || { tail -n 30 $out/parity.log; exit 1; }
tail -n 1 $out/parity.log
