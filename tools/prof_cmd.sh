# rocprofv3 kernel trace of an arbitrary python tool: bash tools/prof_cmd.sh <tag> <script> [args]; summary by (kernel, grid) -> gpurun_out/<tag>/summary.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
python3 $R/$@ > $R/gpurun_out/$TAG/plain.txt 2>&1
rm -rf $R/gpurun_out/$TAG/trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/trace -o b -- python3 $R/$@ > $R/gpurun_out/$TAG/under_rocprof.txt 2> $R/gpurun_out/$TAG/rocprof.err
T=$(find $R/gpurun_out/$TAG/trace -name "*kernel_trace.csv" | head -1)
python3 - "$T" > $R/gpurun_out/$TAG/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"][:70], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:72s} grid={k[1]:>10s} n={len(v):5d} avg={sum(v)/len(v):9.1f} min={min(v):9.1f} total_ms={sum(v)/1e3:8.2f}")
PY
rm -rf $R/gpurun_out/$TAG/trace
cat $R/gpurun_out/$TAG/plain.txt; head -25 $R/gpurun_out/$TAG/summary.txt
