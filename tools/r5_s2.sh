# round 5, GPU session 2: row-streaming Linear kernel (bit identity, A/B timing), bf16 copies on the side stream
out=gpurun_out/r5d; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py -x -q > $out/dense_tests.log 2>&1 || { tail -n 40 $out/dense_tests.log; exit 1; }
tail -n 2 $out/dense_tests.log
SEG3D_LINEAR_STREAM=0 timeout -k 10 300 python tools/linear_bench.py > $out/lin_old.log 2>&1 || exit 1
timeout -k 10 300 python tools/linear_bench.py > $out/lin_new.log 2>&1 || exit 1
paste <(grep -v amdgpu $out/lin_old.log | awk '{print $2,$3,$4,$5,$7}') <(grep -v amdgpu $out/lin_new.log | awk '{print $7}')
timeout -k 10 900 python -m pytest tests/test_gpu_layer.py tests/test_gpu_training.py -x -q -k "not bench and not ddp and not scene_parallel and not full_size and not two_gpus and not spnet_error" > $out/train_tests.log 2>&1 || { tail -n 40 $out/train_tests.log; exit 1; }
tail -n 2 $out/train_tests.log
SEG3D_LINEAR_STREAM=0 timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-exact > $out/bench_old.json 2> $out/bench_old.err || exit 1
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-exact > $out/bench_new.json 2> $out/bench_new.err || exit 1
SEG3D_LINEAR_STREAM=0 timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-exact > $out/bench_old2.json 2> $out/bench_old2.err || exit 1
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-exact > $out/bench_new2.json 2> $out/bench_new2.err || exit 1
timeout -k 10 900 python bench.py --workload dense2m --storage bf16 --steps 6 --warmup 2 --no-cpu-baseline > $out/dense2m_bf16.json 2> $out/dense2m_bf16.err || { tail -n 20 $out/dense2m_bf16.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_old", "bench_new", "bench_old2", "bench_new2", "dense2m_bf16"):
    d = json.loads([l for l in open(f"gpurun_out/r5d/{f}.json") if l.startswith("{")][-1])
    print(f, d["ms_per_step"], d["fwd_only"]["ms_per_step"], d["trained_weights_l1"], d.get("train_storage"), d.get("peak_memory_gb"))
PY
