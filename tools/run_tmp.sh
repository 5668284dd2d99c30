python tools/sweep_cfg3.py "" "EZHIP_UVT_READ2=1" 2>&1 | grep -v amdgpu.ids
