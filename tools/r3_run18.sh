cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3pv2
bash tools/ab_lib.sh "python bench.py --mode fwd --steps 8 --warmup 3 --no-cpu-baseline | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({k:d.get(k) for k in ('ms_per_step','parity','attention_roofline')}))\"" 2 > gpurun_out/r3pv2/ab.txt 2>&1
cp openseg3d_amd/csrc/libA.so openseg3d_amd/csrc/libseg3d_hip.so
tail -4 gpurun_out/r3pv2/ab.txt | cut -c1-600
python tools/linear_bench.py > gpurun_out/r3pv2/linear.txt 2>&1; tail -14 gpurun_out/r3pv2/linear.txt
