# The round's profile artefacts (copy what is wanted from gpurun_out/<tag>/ to profiles/):
#   bench.json                    plain bench line
#   bench_under_rocprof.json      bench line of the profiled run (HIP-event launch time to compare with the trace)
#   kernel_stats.csv              rocprofv3 --kernel-trace --stats of the same command
#   dispatches.csv                every k_sepx / k_armn / k_cf dispatch of the profiled run: id, kernel, grid, start, end, duration
#   pmc_FETCH_SIZE.csv, pmc_WRITE_SIZE.csv   separate --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE x2 on gfx950, WRITE_SIZE exact), per dispatch
#   pmc_SQ_INSTS_VALU.csv         a third pass: VALU wave-instructions per dispatch (the issue-bound rooflines of bench.py read them from profiles/)
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rm -rf $O/trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$T" "$O" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_sepx", "k_armn", "k_cf_", "k_cond", "k_stats", "k_bb_", "k_pts", "k_uvt", "k_st<", "k_polar", "k_interpv", "k_pg_", "k_a32_", "k_dsc", "k_dmin", "k_drg", "k_rle", "k_armn_dec"))]
with open(sys.argv[2] + "/dispatches.csv", "w") as f:
    f.write("dispatch_id,kernel,grid_x,workgroup_x,start_ns,end_ns,duration_us\n")
    for r in keep:
        f.write("%s,\"%s\",%s,%s,%s,%s,%.3f\n" % (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][:60], r["Grid_Size_X"], r["Workgroup_Size_X"],
                r["Start_Timestamp"], r["End_Timestamp"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf $O/trace
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  rm -rf $O/pmc_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/pmc_$c.log 2>&1
  C=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$C" "$O/pmc_$c.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    f.write("dispatch_id,kernel,grid,counter,value\n")
    for r in rows:
        if any(k in r["Kernel_Name"] for k in ("k_sepx", "k_armn", "k_cf_", "k_cond", "k_stats", "k_bb_", "k_pts", "k_uvt", "k_st<", "k_polar")):
            f.write("%s,\"%s\",%s,%s,%s\n" % (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][:60], r["Grid_Size"], r["Counter_Name"], r["Counter_Value"]))
PY
  rm -rf $O/pmc_$c
done
ls -la $O
