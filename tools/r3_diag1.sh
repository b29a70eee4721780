# round 3, diagnostic 1: is the narrow-head attention time tau-dependent?  
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d1
for tau in 1 0.2 0.03 0.01; do
  echo "== one_sweep tau $tau"; python tools/attn_bench.py --tau $tau 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r3d1/tau_one_sweep.txt
for tau in 1 0.03; do
  echo "== dense2m tau $tau"; python tools/attn_bench.py --workload dense2m --iters 5 --tau $tau 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r3d1/tau_dense.txt
echo "== dropout fwd+bwd baseline" > gpurun_out/r3d1/fb.txt
python tools/attn_bench.py --bwd --drop 0.1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3d1/fb.txt
python tools/attn_bench.py --bwd 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3d1/fb.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d1/dense -- python3 bench.py --workload dense2m --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3d1/dense.json 2> gpurun_out/r3d1/dense.err
echo dense rc=$?
python bench.py > gpurun_out/r3d1/default.json 2> gpurun_out/r3d1/default.err
echo default rc=$?
