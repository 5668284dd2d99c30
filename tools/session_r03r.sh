cd $GRAFT_REPO_ROOT
for noise in 1e-5 1e-4 1e-3; do
for th in 0 6 10 14 30; do
  echo "noise $noise dense threshold $th: $(EZHIP_DEC_DEBUG=$((th * 256)) python3 tools/probe_decode1.py $noise 2>&1 | tail -2 | tr '\n' ' ')"
done; done
