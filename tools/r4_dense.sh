out="gpurun_out/r4e"; mkdir -p "$out"
timeout -k 10 600 python -m pytest tests/test_gpu_conv_tiled.py tests/test_gpu_parity.py -x -q -k "bf16" 2>&1 | tail -3
python bench.py --workload dense2m --mode fwd --steps 6 --warmup 2 --scenes 1 --no-cpu-baseline --storage bf16 > $out/dense_fwd_bf16.json 2> $out/dense_fwd_bf16.err; echo rc=$?
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r4e/dense_fwd_bf16.json") if l.startswith("{")][-1])
print("dense2m fwd bf16 storage ms", j["ms_per_step"], "fp32 storage ms", j["storage"]["fwd_ms_per_step_fp32_storage"], j["storage"]["vs_own_fp32_storage"], j["roofline"]["us_per_launch"])
for l in j["conv_layers"]: print(l["rows"], l["cin"], l["cout"], l["us"], l["frac"])
PY
python bench.py --mode fwd --steps 10 --warmup 3 --no-cpu-baseline --storage bf16 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('headline fwd bf16', j['ms_per_step'], j['storage']['fwd_ms_per_step_fp32_storage'], j['storage']['vs_own_fp32_storage'])"
