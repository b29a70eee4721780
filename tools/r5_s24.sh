SEG3D_WGRAD_DENSE_EARLY=1 timeout -k 10 600 python -m pytest tests/test_gpu_dense.py -x -q -k "wgrad_matches or reproducible" 2>&1 | tail -n 2
for v in 0 1 0 1; do echo "== early $v (partials only)"; SEG3D_WGRAD_DENSE_EARLY=$v python tools/wgrad_bench.py --partials 2>&1 | grep -E "58453|19483|121168|sum"; done
