# one-rank rehearsal of the data-parallel wrappers against the plain step, alternating on ONE box
O=gpurun_out/r4ddp; mkdir -p $O
B="bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-fp32-exact"
for rep in 1 2; do
python $B > $O/plain_$rep.json 2> $O/plain_$rep.err; echo plain rc=$?
SEG3D_BENCH_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $B > $O/native_$rep.json 2> $O/native_$rep.err; echo native rc=$?
SEG3D_DDP=torch SEG3D_BENCH_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 $B > $O/torch_$rep.json 2> $O/torch_$rep.err; echo torch rc=$?
done
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
try:
    l = json.loads([x for x in open(sys.argv[1]) if x.strip().startswith("{")][-1])
    print(sys.argv[1], l["ms_per_step"], l.get("trained_weights_l1"), l["config"].get("collective"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
