out=gpurun_out/r5o; mkdir -p $out
for rep in 1 2 3; do for v in 0 1; do
  SEG3D_WGRAD_SB=$v timeout -k 10 600 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-fp32-exact > $out/bench_$v.json 2> $out/bench_$v.err || exit 1
  python - <<PY
import json
d = json.loads([l for l in open("$out/bench_$v.json") if l.startswith("{")][-1])
print("sb=$v", d["ms_per_step"], d["fwd_only"]["ms_per_step"], d["trained_weights_l1"])
PY
done; done
