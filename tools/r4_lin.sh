out="gpurun_out/r4a"; mkdir -p "$out"
for n in 0 6 4 3; do
  SEG3D_CONV_NBT=$n timeout -k 10 200 python tools/linear_bench.py 2>&1 | grep -v amdgpu.ids > "$out/lin_$n.txt" || exit 1
done
paste "$out"/lin_0.txt "$out"/lin_6.txt "$out"/lin_4.txt "$out"/lin_3.txt | awk '{print $2,$3,$5,"| fwd",$7,$(7+20),$(7+40),$(7+60),"| wgrad", $16}'
