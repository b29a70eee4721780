# round 5, GPU session 1: depth-2 sparse weight gradient (parity + A/B), bf16 training copies (tolerance + dense scene), default bench
out=gpurun_out/r5c; mkdir -p $out
SEG3D_WGRAD_DEPTH=2 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sparse_conv_forward_and_backward or wide_tiles or inverse_conv" > $out/parity_depth2.log 2>&1 || { tail -n 30 $out/parity_depth2.log; exit 1; }
tail -n 2 $out/parity_depth2.log
timeout -k 10 300 python tools/sparse_wgrad_bench.py > $out/swg_d1.log 2>&1 || exit 1
SEG3D_WGRAD_DEPTH=2 timeout -k 10 300 python tools/sparse_wgrad_bench.py > $out/swg_d2.log 2>&1 || exit 1
timeout -k 10 300 python tools/sparse_wgrad_bench.py > $out/swg_d1b.log 2>&1 || exit 1
SEG3D_WGRAD_DEPTH=2 timeout -k 10 300 python tools/sparse_wgrad_bench.py > $out/swg_d2b.log 2>&1 || exit 1
paste <(grep -v amdgpu $out/swg_d1.log | awk '{print $1,$2,$4,$5,$6,$7,$8}') <(grep -v amdgpu $out/swg_d2.log | awk '{print $8}') <(grep -v amdgpu $out/swg_d1b.log | awk '{print $8}') <(grep -v amdgpu $out/swg_d2b.log | awk '{print $8}')
timeout -k 10 600 python -m pytest tests/test_gpu_training.py -x -q -s -k "bf16_training_copies" > $out/bf16_copies.log 2>&1 || { tail -n 30 $out/bf16_copies.log; exit 1; }
grep -i "worst\|passed\|failed" $out/bf16_copies.log
timeout -k 10 900 python bench.py --steps 10 --warmup 3 > $out/bench_default.json 2> $out/bench_default.err || { tail -n 20 $out/bench_default.err; exit 1; }
timeout -k 10 900 python bench.py --workload dense2m --storage bf16 --steps 6 --warmup 2 --no-cpu-baseline > $out/dense2m_bf16.json 2> $out/dense2m_bf16.err || { tail -n 20 $out/dense2m_bf16.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_default", "dense2m_bf16"):
    d = json.loads([l for l in open(f"gpurun_out/r5c/{f}.json") if l.startswith("{")][-1])
    print(f, d["ms_per_step"], d["fwd_only"]["ms_per_step"], d.get("idle"), d.get("train_storage"), d.get("peak_memory_gb"))
    if "parity" in d:
        print({k: v for k, v in d["parity"].items() if k != "pinned_by"})
PY
