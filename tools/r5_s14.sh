out=gpurun_out/r5r; mkdir -p $out
for rep in 1 2 3; do for v in 0 1; do
  SEG3D_BENCH_PIPE_THREAD=$v timeout -k 10 600 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-fp32-exact > $out/bench_$v.json 2> $out/bench_$v.err || { tail -n 20 $out/bench_$v.err; exit 1; }
  python - <<PY
import json
d = json.loads([l for l in open("$out/bench_$v.json") if l.startswith("{")][-1])
i = d["idle"]
print("thread=$v", d["ms_per_step"], d["fwd_only"]["ms_per_step"], d["trained_weights_l1"], "idle", i["gpu_idle_ms"], i["gpu_segments_ms_steady"], i["gpu_segments_ms_fed"], i["host_in_prefetch_ms"])
PY
done; done
timeout -k 10 900 python -m pytest tests/test_gpu_training.py -x -q -k "bench" > $out/bench_tests.log 2>&1 || { tail -n 30 $out/bench_tests.log; exit 1; }
tail -n 1 $out/bench_tests.log
