mkdir -p gpurun_out/r5q
for v in 1 0 1 0; do SEG3D_SLOW_STREAM_QUERY=$v python bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5q/sq_$v.json 2> gpurun_out/r5q/sq_$v.err || exit 1
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r5q/sq_$v.json") if l.startswith("{")][-1]); i=d["idle"]
print("slow stream query $v: step", d["ms_per_step"], "fwd", d["fwd_only"]["ms_per_step"], "host enqueue", i["host_enqueue_ms"], "idle", i["gpu_idle_ms"], "steady", i["gpu_step_ms_steady"], "fed", i["gpu_step_ms_fed"])
PY
done
