cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for noise in 1e-5 5e-4; do
  rm -rf gpurun_out/dec_$noise
  rocprofv3 --kernel-trace --stats -d gpurun_out/dec_$noise -o t -- python3 tools/probe_decode1.py $noise > /dev/null 2>&1
  echo "== noise $noise"
  f=$(ls gpurun_out/dec_$noise/*/t_kernel_stats.csv gpurun_out/dec_$noise/t_kernel_stats.csv 2>/dev/null | head -1)
  head -12 $f | cut -c1-200
done
