cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
bash tools/prof_round.sh r05h > gpurun_out/prof_round.log 2>&1
tail -1 gpurun_out/prof_round.log
