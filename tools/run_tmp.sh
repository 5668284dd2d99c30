cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5fuzz
for spec in "fuzz_vs_ref2 250 501" "fuzz_vs_ref2 250 502" "fuzz_vs_ref 200 503" "fuzz_vs_ref3 200 504" "fuzz_vs_ref4 60 505" "fuzz_armn32 200 506" "fuzz_armn_wide 150 507" "fuzz_armn 508 60" "fuzz_average 60 509" "fuzz_interpv 60 510"; do
  set -- $spec
  timeout 900 python tools/$1.py $2 $3 > gpurun_out/r5fuzz/$1_$3.txt 2>&1
  echo "$spec: $(grep -v amdgpu gpurun_out/r5fuzz/$1_$3.txt | tail -1)"
done
