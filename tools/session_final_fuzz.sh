# end-of-round campaign with fresh seeds: the wind path (the (a, b) rotation form), the decoders (2048-bit windows, the plane walk), the rest unchanged code
cd $GRAFT_REPO_ROOT
B=${1:-93}      # seed prefix: bash tools/session_final_fuzz.sh 94
for t in "fuzz_vs_ref.py 400 ${B}01" "fuzz_vs_ref2.py 400 ${B}02" "fuzz_vs_ref3.py 300 ${B}03" "fuzz_vs_ref4.py 150 ${B}04" "fuzz_average.py 100 ${B}05" "fuzz_armn.py 2000 ${B}06" "fuzz_armn32.py 300 ${B}07" "fuzz_shapes.py 200 ${B}08"; do
  echo "== $t"; timeout 600 python3 tools/$t 2>&1 | grep -v amdgpu.ids | tail -3
done
EZHIP_A32_DEVICE_WALK=1 timeout 300 python3 tools/fuzz_armn32.py 200 ${B}09 2>&1 | grep -v amdgpu.ids | tail -2
