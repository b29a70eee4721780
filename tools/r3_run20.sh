bash tools/collect_traffic.sh dense2m --workload dense2m && \
bash tools/collect_traffic.sh cylinder --workload cylinder --batch 4 && \
bash tools/collect_traffic.sh multi_sweeps --workload multi_sweeps --batch 2 && \
bash tools/collect_traffic.sh spnet --segmentor spnet
