mkdir -p gpurun_out/r5q
for v in end fwd end fwd; do SEG3D_BENCH_PREFETCH=$v SEG3D_BENCH_IDLE_PROBE=0 python bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5q/pf_$v.json 2> gpurun_out/r5q/pf_$v.err || exit 1
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r5q/pf_$v.json") if l.startswith("{")][-1])
print("prefetch $v: step", d["ms_per_step"], "fwd", d["fwd_only"]["ms_per_step"])
PY
done
