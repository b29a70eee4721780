out=gpurun_out/r5g; mkdir -p $out
for v in 0 1 0 1; do SEG3D_CONV_XCD_RUN=$v timeout -k 10 300 python tools/conv_bench.py > $out/conv_x$v.log 2>&1 || exit 1; done
paste <(grep -v amdgpu $out/conv_x0.log | awk '{print $2,$3,$4,$5,$6}') <(grep -v amdgpu $out/conv_x1.log | awk '{print $6}')
for r in 2048 3072 4096 6144; do SEG3D_WGRAD_SPARSE_ROWS=$r timeout -k 10 300 python tools/sparse_wgrad_bench.py > $out/swg_r$r.log 2>&1 || exit 1; echo rows $r $(grep sum $out/swg_r$r.log); done
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conv_tiled.py -x -q -k "sparse_conv or inverse_conv or tiled" > $out/parity.log 2>&1 || { tail -n 30 $out/parity.log; exit 1; }
tail -n 1 $out/parity.log
SEG3D_CONV_XCD_RUN=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sparse_conv_forward_and_backward or inverse_conv or wide_tiles" > $out/parity_x1.log 2>&1 || { tail -n 30 $out/parity_x1.log; exit 1; }
tail -n 1 $out/parity_x1.log
timeout -k 10 900 python bench.py --workload dense2m --storage bf16 --steps 6 --warmup 2 --no-cpu-baseline > $out/dense2m_bf16.json 2> $out/dense2m_bf16.err || { tail -n 20 $out/dense2m_bf16.err; exit 1; }
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5g/dense2m_bf16.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["fwd_only"]["ms_per_step"], {k: v for k, v in d["train_storage"].items() if "ms" in k or "peak" in k})
PY
