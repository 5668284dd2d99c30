cd $GRAFT_REPO_ROOT
for t in fuzz_vs_ref.py fuzz_vs_ref2.py fuzz_vs_ref3.py fuzz_vs_ref4.py fuzz_average.py; do
  FUZZ_PRODUCT_ONLY=1 timeout 600 python3 tools/$t 60 4001 > /tmp/po.txt 2>&1; echo "$t product-only: exit $?"; grep -v amdgpu.ids /tmp/po.txt | tail -n 2 | cut -c1-300
done
timeout 1500 python3 -m pytest tests/test_gpu_vs_reference_build.py -x -q -m gpu 2>&1 | tail -3
# larger regression campaign (new seeds)
O=gpurun_out/r03q; mkdir -p $O
( timeout 1800 python3 tools/fuzz_vs_ref.py 8000 5101 > $O/f1.txt 2>&1; tail -n 1 $O/f1.txt )
( timeout 1800 python3 tools/fuzz_vs_ref2.py 6000 5103 > $O/f2.txt 2>&1; tail -n 1 $O/f2.txt )
( FUZZ_HEMI=1 timeout 1800 python3 tools/fuzz_vs_ref2.py 3000 5113 > $O/f2h.txt 2>&1; tail -n 1 $O/f2h.txt )
( timeout 1200 python3 tools/fuzz_vs_ref3.py 3000 5104 > $O/f3.txt 2>&1; tail -n 1 $O/f3.txt )
( timeout 1200 python3 tools/fuzz_vs_ref4.py 1200 5105 > $O/f4.txt 2>&1; tail -n 1 $O/f4.txt )
( timeout 600 python3 tools/fuzz_armn.py 2000 5107 > $O/f6.txt 2>&1; tail -n 1 $O/f6.txt )
( timeout 600 python3 tools/fuzz_armn32.py 500 5108 > $O/f8.txt 2>&1; tail -n 1 $O/f8.txt )
( timeout 600 python3 tools/fuzz_interpv.py 500 5109 > $O/f9.txt 2>&1; tail -n 1 $O/f9.txt )
