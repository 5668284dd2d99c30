python -m pytest tests/test_gpu_packers.py -x -q -m gpu -k "armn32 or a32 or uncompress32 or compress32" 2>&1 | tail -2
python tools/probe_a32.py 2>&1 | grep -v amdgpu.ids | grep "zlng\|host walk \|default"
python tools/fuzz_armn32.py 80 13 2>&1 | tail -1
