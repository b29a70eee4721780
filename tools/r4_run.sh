out="gpurun_out/r4a"; mkdir -p "$out"
timeout -k 10 400 python -m pytest tests/test_gpu_conv_tiled.py -x -q 2>&1 | tail -2
for d in 1 0 4 16; do
  SEG3D_TILE_RUN=$d timeout -k 10 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | awk '{print $3,$5,$6}' > "$out/run_$d.txt" || exit 1
done
paste "$out"/run_1.txt "$out"/run_0.txt "$out"/run_4.txt "$out"/run_16.txt | awk '{print $1,$2,"|",$3,$6,$9,$12}'
