# timing experiments of the tile kernel (SEG3D_TILE_DBG: wrong results, timing only).  The experiments are compiled in only
# with -DSEG3D_TILE_DBG: rebuild spconv_tile.o with that define (and relink) before running this on the GPU box, e.g.
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
#         -DSEG3D_TILE_DBG -c openseg3d_amd/csrc/spconv_tile.hip -o openseg3d_amd/csrc/spconv_tile.o
out="gpurun_out/r4a"; mkdir -p "$out"
for d in 0 1 2 4 8 7 15; do
  SEG3D_TILE_DBG=$d timeout -k 10 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | awk '{print $3,$5,$6}' > "$out/dbg_$d.txt" || exit 1
done
paste "$out"/dbg_0.txt "$out"/dbg_1.txt "$out"/dbg_2.txt "$out"/dbg_4.txt "$out"/dbg_8.txt "$out"/dbg_7.txt "$out"/dbg_15.txt | awk '{print $1,$2,"|",$3,$6,$9,$12,$15,$18,$21}'
