# input pipeline stream at high priority: host time blocked in its read-backs, GPU idle, step
mkdir -p gpurun_out/r5q
for v in 0 -1 0 -1; do SEG3D_PIPE_PRIORITY=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5q/p$v.json 2> gpurun_out/r5q/p$v.err || exit 1
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r5q/p$v.json") if l.startswith("{")][-1]); i=d["idle"]
print("prio $v: step", d["ms_per_step"], "fwd", d["fwd_only"]["ms_per_step"], "idle", i["gpu_idle_ms"], "steady", i["gpu_step_ms_steady"], "fed", i["gpu_step_ms_fed"], "prefetch blocked", i["host_in_prefetch_ms"], "between", i["gpu_between_steps_ms"], "segments", i["gpu_segments_ms_steady"])
PY
done
python -c "import torch; print(torch.cuda.Stream.priority_range())"
