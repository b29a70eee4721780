# tiled-conv development loop on one box: parity tests of the tile schedule, then the per-layer conv bench with the
# schedule off and on.  usage (gpurun): bash tools/r4_conv_ab.sh <tag>
tag="${1:-x}"
out="gpurun_out/r4a"
mkdir -p "$out"
timeout -k 10 400 python -m pytest tests/test_gpu_conv_tiled.py -x -q > "$out/tiled_tests_$tag.log" 2>&1
echo "tests rc=$?"; tail -3 "$out/tiled_tests_$tag.log"
SEG3D_CONV_TILED=0 timeout -k 10 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids > "$out/conv_old_$tag.txt" || exit 1
timeout -k 10 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids > "$out/conv_tiled_$tag.txt" || exit 1
paste "$out/conv_old_$tag.txt" "$out/conv_tiled_$tag.txt" | cut -c1-200
