out="gpurun_out/r4a"; mkdir -p "$out"
for d in 1 5 2; do
  SEG3D_TILE_LAYOUT=$d timeout -k 10 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | awk '{print $3,$5,$6}' > "$out/lay_$d.txt" || exit 1
done
paste "$out"/lay_1.txt "$out"/lay_5.txt "$out"/lay_2.txt | awk '{print $1,$2,"|",$3,$6,$9}'
