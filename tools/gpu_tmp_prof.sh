cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/prof_round.sh 5 2>&1 | tail -5
ls gpurun_out/prof_r5 | head -30
