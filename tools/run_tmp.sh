for k in positive mixed stripes; do python tools/probe_a32_enc.py 7200 3601 $k 2>&1 | grep -v amdgpu.ids | tail -2; done
EZHIP_A32_RLE_ENC_HOST=1 python tools/probe_a32_enc.py 7200 3601 mixed 2>&1 | grep -v amdgpu.ids | tail -2
