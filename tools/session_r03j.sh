R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O; rm -f $O/cfg3.txt
cd $R
for v in "" "EZHIP_POLAR_WIND_SIDE=1" ""; do
  echo "variant [$v]" >> $O/cfg3.txt
  env $v python3 tools/probe_cfg3.py >> $O/cfg3.txt 2>&1
done
timeout 1500 python3 -m pytest tests/test_gpu_interp.py tests/test_gpu_vs_reference_build.py -x -q -m gpu > $O/pytest.txt 2>&1
grep -v amdgpu.ids $O/cfg3.txt; tail -n 3 $O/pytest.txt
bash tools/trace_timeline.sh r03j_tl tools/probe_cfg3.py | tail -8
