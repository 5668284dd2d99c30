cd $GRAFT_REPO_ROOT
for l in pf10w4 pf0w4; do echo $l; EZHIP_LIBRARY=$GRAFT_REPO_ROOT/devlibs/$l.so python tools/probe_cfg3_batch.py 2 4 8 16 2>&1 | grep -v amdgpu; done
echo product; python tools/probe_cfg3_batch.py 2 4 8 16 2>&1 | grep -v amdgpu
