cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3d4
timeout -k 10 600 python -m pytest tests/test_gpu_attention.py -x -q > gpurun_out/r3d4/tests_attn.txt 2>&1; echo attn tests rc=$?
for rep in 1 2; do
for x in 1 8; do echo "== XCD $x"; SEG3D_ATTN_XCD=$x python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done
done > gpurun_out/r3d4/xcd.txt
for x in 1 8; do
SEG3D_ATTN_XCD=$x rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r3d4/fetch$x -- python3 tools/attn_bench.py --iters 2 > /dev/null 2>&1
python tools/pmc_summary.py $(ls gpurun_out/r3d4/fetch$x/*/*counter_collection.csv | head -1) attn_fused > gpurun_out/r3d4/fetch$x.txt 2>&1
done
echo done
