R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
for v in "" "EZHIP_PTS_NOTILE=1" "EZHIP_WIND_NEWTON_LITERAL=1" "EZHIP_PTS_NOTILE=1 EZHIP_WIND_NEWTON_LITERAL=1" "EZHIP_PTS_XCD=1" ""; do
  echo "variant [$v]" >> $O/cfg3.txt
  env $v python3 tools/probe_cfg3.py >> $O/cfg3.txt 2>&1
done
bash tools/prof_cmd.sh r03e_cfg3trace tools/probe_cfg3.py > /dev/null 2>&1
timeout 1500 python3 -m pytest tests/test_gpu_interp.py tests/test_gpu_vs_reference_build.py -x -q -m gpu > $O/pytest.txt 2>&1
grep -v amdgpu.ids $O/cfg3.txt; head -4 $R/gpurun_out/r03e_cfg3trace/summary.txt; tail -n 3 $O/pytest.txt
