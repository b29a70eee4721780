# Fabric traffic of the attention forward per launch, XCD-aware unit order (default) against the flat order:
#   gpurun -- 'bash tools/collect_attn_traffic.sh'  ->  gpurun_out/r3a/attention_traffic_xcd{8,1}.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/r3a; mkdir -p $P
for xcd in 8 1; do
  export SEG3D_ATTN_XCD=$xcd
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch$xcd -- python3 bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline > $P/fetch$xcd.json 2> $P/fetch$xcd.err && \
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write$xcd -- python3 bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline > $P/write$xcd.json 2> $P/write$xcd.err && \
  python3 tools/pmc_kernel_traffic.py $P/fetch$xcd/*/*_counter_collection.csv $P/write$xcd/*/*_counter_collection.csv 'attn_fused_fwd' > $P/attention_traffic_xcd$xcd.txt 2>&1
  echo "xcd=$xcd rc=$?"
  rm -rf $P/fetch$xcd $P/write$xcd
done
cat $P/attention_traffic_xcd8.txt $P/attention_traffic_xcd1.txt
