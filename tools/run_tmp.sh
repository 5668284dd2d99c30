set -x
cd $GRAFT_REPO_ROOT
EZHIP_TRACE_FIRST=1 python tools/probe_cfg3_first.py
