# timing experiments of the tile kernel (SEG3D_TILE_DBG: wrong results, timing only)
out="gpurun_out/r4a"; mkdir -p "$out"
for d in 0 1 2 4 8 7 15; do
  SEG3D_TILE_DBG=$d timeout -k 10 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | awk '{print $3,$5,$6}' > "$out/dbg_$d.txt" || exit 1
done
paste "$out"/dbg_0.txt "$out"/dbg_1.txt "$out"/dbg_2.txt "$out"/dbg_4.txt "$out"/dbg_8.txt "$out"/dbg_7.txt "$out"/dbg_15.txt | awk '{print $1,$2,"|",$3,$6,$9,$12,$15,$18,$21}'
