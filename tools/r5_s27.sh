# training forward: x + pos summed on the in-projection's A operand, recomputed in the backward for the q | k weight gradient
mkdir -p gpurun_out/r5q
timeout -k 10 900 python -m pytest tests/test_gpu_layer.py tests/test_gpu_attention.py -m gpu -x -q > gpurun_out/r5q/t_inproj.log 2>&1; rc=$?; tail -n 3 gpurun_out/r5q/t_inproj.log; [ $rc = 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_training.py -m gpu -x -q -k "reproducible or every_parameter or reentrant or side_stream or bf16_training" > gpurun_out/r5q/t_inproj2.log 2>&1; rc=$?; tail -n 3 gpurun_out/r5q/t_inproj2.log; [ $rc = 0 ] || exit $rc
for v in 0 1 0 1; do SEG3D_INPROJ_SUM_TRAIN=$v SEG3D_BENCH_IDLE_PROBE=0 python bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5q/ip_$v.json 2> gpurun_out/r5q/ip_$v.err || exit 1
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r5q/ip_$v.json") if l.startswith("{")][-1])
print("inproj sum train $v: step", d["ms_per_step"], "fwd", d["fwd_only"]["ms_per_step"], "mem", d["peak_memory_gb"], "l1", d.get("trained_weights_l1"))
PY
done
