# six-product per-point MLP kernel: parity test, layer bench, then the headline bench line with and without it
mkdir -p gpurun_out/r5x
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py -x -q -k six_product > gpurun_out/r5x/t.log 2>&1; rc=$?; tail -n 15 gpurun_out/r5x/t.log
[ $rc = 0 ] || exit $rc
python tools/x6_bench.py > gpurun_out/r5x/bench.log 2>&1 || { tail -n 20 gpurun_out/r5x/bench.log; exit 1; }
cat gpurun_out/r5x/bench.log
python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5x/x6.json 2> gpurun_out/r5x/x6.err || { tail -n 20 gpurun_out/r5x/x6.err; exit 1; }
SEG3D_POINT_MLP=fp32 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5x/f32.json 2> gpurun_out/r5x/f32.err || exit 1
python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5x/x6b.json 2> gpurun_out/r5x/x6b.err || exit 1
python - <<'PY'
import json
for n in ("x6","f32","x6b"):
    d=json.loads([l for l in open(f"gpurun_out/r5x/{n}.json") if l.startswith("{")][-1])
    p=d["parity"]
    print(n, d["ms_per_step"], d["fwd_only"]["ms_per_step"], p["max_abs_logit_diff"], p.get("vs_fp64_oracle",{}).get("gpu_max_abs_logit_diff"), p["after_training"]["max_abs_logit_diff"])
PY
