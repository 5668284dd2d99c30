cd $GRAFT_REPO_ROOT
for noise in 1e-5 5e-4; do
for th in 0 8 40 80; do
  echo "noise $noise dense threshold $th: $(EZHIP_DEC_DEBUG=$((th * 256)) python3 tools/probe_decode1.py $noise 2>&1 | tail -1)"
done; done
