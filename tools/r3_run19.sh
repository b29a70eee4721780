cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3pv2
bash tools/ab_lib.sh "python bench.py --mode fwd --steps 3 --warmup 2 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d.get('parity')))\"" 1 > gpurun_out/r3pv2/ab_parity.txt 2>&1
cp openseg3d_amd/csrc/libA.so openseg3d_amd/csrc/libseg3d_hip.so
cat gpurun_out/r3pv2/ab_parity.txt | cut -c1-900
