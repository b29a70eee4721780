# A/B/C... of several builds of libseg3d_hip.so on ONE box:  csrc/lib<NAME>.so for every name in the list
#   gpurun -- 'bash tools/ab_libs.sh "A B C" "python tools/attn_bench.py --bwd" 2'      (names, command, repetitions)
# Leaves the LAST named library installed.  The copies are build artefacts (git-ignored).
names="$1"; cmd="$2"; reps="${3:-2}"
cd "$GRAFT_REPO_ROOT/openseg3d_amd/csrc" || exit 1
for rep in $(seq "$reps"); do
  for v in $names; do
    cp lib$v.so libseg3d_hip.so
    echo "== $v $rep"; (cd ../..; eval "$cmd" 2>&1 | grep -v amdgpu.ids)
  done
done
