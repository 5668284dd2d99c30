cd $GRAFT_REPO_ROOT
bash tools/prof_round.sh r05f > gpurun_out/prof_round.log 2>&1
tail -3 gpurun_out/prof_round.log
