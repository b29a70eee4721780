# int4 work-item descriptors: attention + window tests, attention bench fwd/bwd with dropout
mkdir -p gpurun_out/r5z
timeout -k 10 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_layer.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5z/t.log 2>&1; rc=$?; tail -n 6 gpurun_out/r5z/t.log
[ $rc = 0 ] || exit $rc
python tools/attn_bench.py --bwd --drop 0.1 > gpurun_out/r5z/attn.txt 2>&1; tail -n 10 gpurun_out/r5z/attn.txt
python tools/attn_bench.py --bwd --drop 0.1 > gpurun_out/r5z/attn2.txt 2>&1; tail -n 1 gpurun_out/r5z/attn2.txt
python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5z/default.json 2> gpurun_out/r5z/default.err || exit 1
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5z/default.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["fwd_only"]["ms_per_step"], d["attention_roofline"]["frac"])
PY
