cd $GRAFT_REPO_ROOT
for v in "EZHIP_NOP=1" "EZHIP_PTS_NT=1" "EZHIP_NOP=1" "EZHIP_PTS_NT=1"; do echo "[$v] $(env $v python3 tools/probe_cfg3.py 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ')"; done
