# HBM traffic of the sparse-conv launches for one bench workload (two --pmc passes, merged on the box):
#   bash tools/collect_traffic.sh <tag> <bench.py arguments...>   ->  gpurun_out/r5t/<tag>_pmc_conv_traffic.json
tag="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/r5t/$tag; mkdir -p $P
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -- python3 bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-exact "$@" > $P/fetch.json 2> $P/fetch.err && cp gpurun_out/bench_layers.json $P/layers.json && \
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -- python3 bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-exact "$@" > $P/write.json 2> $P/write.err && \
python3 tools/pmc_conv_traffic.py $P/fetch/*/*_counter_collection.csv $P/write/*/*_counter_collection.csv $P/layers.json gpurun_out/r5t/${tag}_pmc_conv_traffic.json
rc=$?
rm -rf $P/fetch $P/write
echo "$tag rc=$rc"
exit $rc
