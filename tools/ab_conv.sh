# A/B of two builds of the library on one box: tools/ab_conv.sh  (expects csrc/libA.so, csrc/libB.so)
cd $GRAFT_REPO_ROOT/openseg3d_amd/csrc
for rep in 1 2 3; do
  for v in A B; do
    cp lib$v.so libseg3d_hip.so
    echo "== $v $rep"; (cd ../..; python tools/conv_bench.py 2>&1 | grep -E "19483  384 ->  384|19483  768|58453  192 ->  192|121168   96 ->   96|total" | sort -u)
  done
done
cp libA.so libseg3d_hip.so
