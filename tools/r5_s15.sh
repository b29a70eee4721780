# per-workload fabric traffic of the conv launches (two --pmc passes each), then the default bench line with the final library
for spec in "dense2m --workload dense2m" "cylinder --workload cylinder --batch 4" "multi_sweeps --workload multi_sweeps --batch 2" "spnet --segmentor spnet"; do
  set -- $spec
  bash tools/collect_traffic.sh "$@" || exit 1
done
ls -la gpurun_out/r5t/*.json
