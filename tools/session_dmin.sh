O=gpurun_out/$1; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_packers.py -x -q -k "uncompress" 2>&1 | tail -6) > $O/pytest.txt
python tools/probe_decode_minimum.py > $O/probe.txt 2>&1
EZHIP_DEC_NO_DMIN=1 python tools/probe_decode_minimum.py > $O/probe_nodmin.txt 2>&1
python tools/probe_decode_minimum.py 1e-3 > $O/probe_noisy.txt 2>&1
cat $O/pytest.txt; grep -v amdgpu $O/probe.txt $O/probe_nodmin.txt $O/probe_noisy.txt | cut -c1-220
