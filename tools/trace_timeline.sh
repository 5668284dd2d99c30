# kernel timeline (start offsets and durations, us) of the LAST iterations of a probe: bash tools/trace_timeline.sh <tag> <script> [args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG; rm -rf $R/gpurun_out/$TAG/trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/trace -o b -- python3 $R/$@ > $R/gpurun_out/$TAG/run.txt 2>&1
T=$(find $R/gpurun_out/$TAG/trace -name "*kernel_trace.csv" | head -1)
python3 - "$T" > $R/gpurun_out/$TAG/timeline.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:60]))
    prev_end = max(prev_end, e)
PY
rm -rf $R/gpurun_out/$TAG/trace
cat $R/gpurun_out/$TAG/timeline.txt
