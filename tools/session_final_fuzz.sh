# end-of-round campaign with fresh seeds: the wind path (the (a, b) rotation form), the decoders (2048-bit windows, the plane walk), the rest unchanged code
cd $GRAFT_REPO_ROOT
for t in "fuzz_vs_ref.py 400 9301" "fuzz_vs_ref2.py 400 9302" "fuzz_vs_ref3.py 300 9303" "fuzz_vs_ref4.py 150 9304" "fuzz_average.py 100 9305" "fuzz_armn.py 2000 9306" "fuzz_armn32.py 300 9307" "fuzz_shapes.py 200 9308"; do
  echo "== $t"; timeout 600 python3 tools/$t 2>&1 | grep -v amdgpu.ids | tail -3
done
EZHIP_A32_DEVICE_WALK=1 timeout 300 python3 tools/fuzz_armn32.py 200 9309 2>&1 | grep -v amdgpu.ids | tail -2
