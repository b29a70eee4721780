# A/B of two builds of the library on one box (attention core): expects csrc/libA.so, csrc/libB.so
cd $GRAFT_REPO_ROOT/openseg3d_amd/csrc
for rep in 1 2; do
for v in A B; do
    cp lib$v.so libseg3d_hip.so
    echo "== $v fwd"; (cd ../..; python tools/attn_bench.py 2>&1 | tail -1)
    echo "== $v fwd+bwd drop"; (cd ../..; python tools/attn_bench.py --bwd --drop 0.1 2>&1 | tail -1)
done
done
cp libA.so libseg3d_hip.so
