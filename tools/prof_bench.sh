# bench.py + rocprofv3 --kernel-trace --stats of the same command; outputs under gpurun_out/<tag>/ (copy summaries to profiles/)
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
python3 $R/bench.py > $R/gpurun_out/$TAG/bench.json 2> $R/gpurun_out/$TAG/bench.err
rm -rf $R/gpurun_out/$TAG/trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -o b -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/$TAG/bench_under_rocprof.json 2> $R/gpurun_out/$TAG/rocprof.err
find $R/gpurun_out/$TAG/trace -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$TAG/kernel_stats.csv \;
# the trace itself is large: keep the k_sepx / k_armn rows only (dispatch id, kernel, grid, start, end)
T=$(find $R/gpurun_out/$TAG/trace -name "*kernel_trace.csv" | head -1)
python3 - "$T" "$R/gpurun_out/$TAG" <<'PY'
import csv, sys
src, out = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
keep = [r for r in rows if "k_sepx" in r["Kernel_Name"] or "k_armn" in r["Kernel_Name"] or "k_cf_" in r["Kernel_Name"] or "k_cond" in r["Kernel_Name"]]
with open(out + "/dispatches.csv", "w") as f:
    f.write("dispatch_id,kernel,grid_x,workgroup_x,lds_bytes,vgpr,start_ns,end_ns,duration_us\n")
    for r in keep:
        f.write("%s,\"%s\",%s,%s,%s,%s,%s,%s,%.3f\n" % (r["Dispatch_Id"], r["Kernel_Name"][:60], r["Grid_Size_X"], r["Workgroup_Size_X"], r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""),
                r["Start_Timestamp"], r["End_Timestamp"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf $R/gpurun_out/$TAG/trace
ls -la $R/gpurun_out/$TAG
