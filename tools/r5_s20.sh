mkdir -p gpurun_out/r5y
timeout -k 10 300 python -m pytest tests/test_gpu_dense.py -x -q -k "six_product or eval_point_mlp" > gpurun_out/r5y/t3.log 2>&1 || { tail -n 20 gpurun_out/r5y/t3.log; exit 1; }
python tools/x6_bench.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests/test_gpu_training.py -m gpu -x -q --durations=5 > gpurun_out/r5y/tests2.log 2>&1; rc=$?; tail -n 12 gpurun_out/r5y/tests2.log
[ $rc = 0 ] || exit $rc
python bench.py > gpurun_out/r5y/default.json 2> gpurun_out/r5y/default.err || { tail -n 20 gpurun_out/r5y/default.err; exit 1; }
SEG3D_POINT_MLP=fp32 python bench.py --no-fp32-exact > gpurun_out/r5y/default_f32mlp.json 2> gpurun_out/r5y/default_f32mlp.err || exit 1
python bench.py --mode fwd --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5y/fwd.json 2> gpurun_out/r5y/fwd.err || exit 1
SEG3D_POINT_MLP=fp32 python bench.py --mode fwd --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5y/fwd_f32mlp.json 2> gpurun_out/r5y/fwd_f32mlp.err || exit 1
python bench.py --segmentor spnet --steps 10 --warmup 3 > gpurun_out/r5y/spnet.json 2> gpurun_out/r5y/spnet.err || exit 1
python - <<'PY'
import json
for n in ("default","default_f32mlp","fwd","fwd_f32mlp","spnet"):
    d=json.loads([l for l in open(f"gpurun_out/r5y/{n}.json") if l.startswith("{")][-1])
    p=d.get("parity") or {}
    print(n, d["ms_per_step"], (d.get("fwd_only") or {}).get("ms_per_step"), p.get("max_abs_logit_diff"), (p.get("vs_fp64_oracle") or {}).get("gpu_max_abs_logit_diff"), (p.get("after_training") or {}).get("max_abs_logit_diff"))
PY
