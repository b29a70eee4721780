for t in 0 1; do
SEG3D_CONV_TILED=$t python bench.py --segmentor spnet --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-exact 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('spnet tiled=$t', j['ms_per_step'], j['fwd_only']['ms_per_step'], j['roofline']['us_per_launch']); [print('   ',l) for l in j['conv_layers']] if $t else None"
done
timeout -k 10 600 python -m pytest tests/test_gpu_conv_tiled.py tests/test_gpu_parity.py -x -q -k "tiled or spnet or sparse_conv" 2>&1 | tail -3
