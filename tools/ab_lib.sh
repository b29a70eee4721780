# A/B of two builds of libseg3d_hip.so on ONE box (boxes of the pool differ by +-5 %, more than most kernel changes):
#   build the baseline, cp csrc/libseg3d_hip.so csrc/libA.so; build the candidate, cp ... csrc/libB.so; then
#   gpurun -- 'bash tools/ab_lib.sh "python tools/conv_bench.py" 3'      (command, repetitions)
# Leaves libB.so installed.  The copies are build artefacts (git-ignored).
cmd="$1"; reps="${2:-2}"
cd "$GRAFT_REPO_ROOT/openseg3d_amd/csrc" || exit 1
for rep in $(seq "$reps"); do
  for v in A B; do
    cp lib$v.so libseg3d_hip.so
    echo "== $v $rep"; (cd ../..; eval "$cmd" 2>&1 | grep -v amdgpu.ids)
  done
done
cp libB.so libseg3d_hip.so
