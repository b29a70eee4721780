# The bench lines of DESIGN.md section 6, one file per workload under gpurun_out/r2f/ (run on the GPU box:
#   gpurun --timeout 1150 -- 'bash tools/final_benches.sh').  Every line goes to a file: a silent run is taken for hung.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2f
python bench.py > gpurun_out/r2f/default.json 2> gpurun_out/r2f/default.err; echo default rc=$?
python bench.py --workload dense2m --steps 8 --warmup 3 > gpurun_out/r2f/dense2m.json 2> gpurun_out/r2f/dense2m.err; echo dense rc=$?
python bench.py --workload cylinder --batch 4 --steps 8 --warmup 3 --scenes 2 > gpurun_out/r2f/cylinder.json 2> gpurun_out/r2f/cylinder.err; echo cyl rc=$?
python bench.py --workload multi_sweeps --batch 2 --steps 8 --warmup 3 --scenes 2 > gpurun_out/r2f/multi.json 2> gpurun_out/r2f/multi.err; echo ms rc=$?
python bench.py --segmentor spnet --steps 10 --warmup 3 > gpurun_out/r2f/spnet.json 2> gpurun_out/r2f/spnet.err; echo spnet rc=$?
SEG3D_CONV_PRECISION=fp32 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2f/fp32.json 2> gpurun_out/r2f/fp32.err; echo fp32 rc=$?
