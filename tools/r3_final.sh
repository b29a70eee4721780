cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/r3_fulltests.sh && bash tools/final_benches.sh
