out=gpurun_out/r5i; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_layer.py -x -q > $out/attn_tests.log 2>&1 || { tail -n 30 $out/attn_tests.log; exit 1; }
tail -n 1 $out/attn_tests.log
for rep in 1 2; do for b in 1 0; do
  SEG3D_ATTN_XCD_BLOCK=$b timeout -k 10 300 python tools/attn_bench.py --bwd --drop 0.1 > $out/attn_b${b}_$rep.txt 2>&1 || exit 1
  echo "block env=$b: $(grep -v amdgpu $out/attn_b${b}_$rep.txt | tail -n 2 | tr '\n' ' ')"
done; done
for rep in 1 2; do for b in 1 0; do
  SEG3D_ATTN_XCD_BLOCK=$b timeout -k 10 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-fp32-exact > $out/bench_b${b}_$rep.json 2> $out/bench_b${b}_$rep.err || exit 1
  python - <<PY
import json
d = json.loads([l for l in open("$out/bench_b${b}_$rep.json") if l.startswith("{")][-1])
print("block env=$b", d["ms_per_step"], d["fwd_only"]["ms_per_step"], d["attention_roofline"]["ms_per_forward"], d["attention_roofline"]["frac"])
PY
done; done
