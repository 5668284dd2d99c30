cd $GRAFT_REPO_ROOT
# the encoder at seven (product) against eight waves per SIMD (devlibs/librmn_ez_hip_w8.so), kernel times from the trace
bash tools/prof_list.sh r03t_w7 "k_armn_enc1" tools/probe_cfg5_fused.py 32 | grep "k_armn_enc1" | awk '{print $(NF-1)}' | sort -n | awk '{a[NR]=$1} END {print "seven waves: k_armn_enc1 n", NR, "min", a[1], "median", a[int((NR+1)/2)]}'
grep "two kernels" gpurun_out/r03t_w7/under_rocprof.txt | head -2
EZHIP_LIBRARY=$GRAFT_REPO_ROOT/devlibs/librmn_ez_hip_w8.so bash tools/prof_list.sh r03t_w8 "k_armn_enc1" tools/probe_cfg5_fused.py 32 | grep "k_armn_enc1" | awk '{print $(NF-1)}' | sort -n | awk '{a[NR]=$1} END {print "eight waves: k_armn_enc1 n", NR, "min", a[1], "median", a[int((NR+1)/2)]}'
grep "two kernels" gpurun_out/r03t_w8/under_rocprof.txt | head -2
bash tools/prof_list.sh r03t_w7b "k_armn_enc1" tools/probe_cfg5_fused.py 32 | grep "k_armn_enc1" | awk '{print $(NF-1)}' | sort -n | awk '{a[NR]=$1} END {print "seven waves: k_armn_enc1 n", NR, "min", a[1], "median", a[int((NR+1)/2)]}'
