out=gpurun_out/r5m; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py -x -q > $out/dense_tests.log 2>&1 || { tail -n 40 $out/dense_tests.log; exit 1; }
tail -n 1 $out/dense_tests.log
for rep in 1 2; do for v in 0 1; do SEG3D_WGRAD_GROUP=$v timeout -k 10 300 python tools/wgrad_bench.py > $out/dwg_$v.log 2>&1 || exit 1; done; done
paste <(grep -v amdgpu $out/dwg_0.log | awk '{print $1,$2,$3,$4,$5}') <(grep -v amdgpu $out/dwg_1.log | awk '{print $4,$5}')
for rep in 1 2; do for v in 0 1; do
  SEG3D_WGRAD_GROUP=$v timeout -k 10 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-fp32-exact > $out/bench_$v.json 2> $out/bench_$v.err || exit 1
  python - <<PY
import json
d = json.loads([l for l in open("$out/bench_$v.json") if l.startswith("{")][-1])
print("group=$v", d["ms_per_step"], d["fwd_only"]["ms_per_step"], d["trained_weights_l1"])
PY
done; done
