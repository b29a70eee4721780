# attention forward fabric traffic with blocks of consecutive items per XCD (SEG3D_ATTN_XCD_BLOCK = 1 / 4 / 8)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/r5h; mkdir -p $P
for b in 1 4 8; do
  export SEG3D_ATTN_XCD_BLOCK=$b
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch$b -- python3 bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline > $P/fetch$b.json 2> $P/fetch$b.err && \
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write$b -- python3 bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline > $P/write$b.json 2> $P/write$b.err && \
  python3 tools/pmc_kernel_traffic.py $P/fetch$b/*/*_counter_collection.csv $P/write$b/*/*_counter_collection.csv 'attn_fused_fwd' > $P/attention_traffic_block$b.txt 2>&1
  echo "block=$b rc=$?"
  rm -rf $P/fetch$b $P/write$b
  python3 tools/attn_bench.py > $P/attn_time_block$b.txt 2>&1
  cat $P/attention_traffic_block$b.txt; grep -v amdgpu $P/attn_time_block$b.txt | tail -n 8
done
