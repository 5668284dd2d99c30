cd $GRAFT_REPO_ROOT
timeout 600 python3 tools/probe_fst_host.py 2>&1 | grep -v amdgpu.ids
