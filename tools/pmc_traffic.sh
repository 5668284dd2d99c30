cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest $R/tests/test_gpu_interp.py -x -q 2>&1 | tail -2
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  PROBE_POLAR=yes timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o probe -- python3 $R/tools/perf_probe.py cubic > $R/gpurun_out/pmc_$c.log 2>&1
  rm -rf $R/gpurun_out/pmcu_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcu_$c -o ub -- $R/tools/ubench > $R/gpurun_out/pmcu_$c.log 2>&1
done
ls $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmcu_FETCH_SIZE
