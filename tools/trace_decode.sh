# kernel timeline of one single-stream decode: bash tools/trace_decode.sh <tag>
TAG=${1:-dectl}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
rm -rf $R/gpurun_out/$TAG/trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/trace -o b -- python3 $R/tools/probe_decode_timeline.py > $R/gpurun_out/$TAG/out.txt 2> $R/gpurun_out/$TAG/rocprof.err
T=$(find $R/gpurun_out/$TAG/trace -name "*kernel_trace.csv" | head -1)
python3 - "$T" > $R/gpurun_out/$TAG/timeline.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# clusters separated by > 5 ms of idle
cl = [[rows[0]]]
for a, b in zip(rows, rows[1:]):
    if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 5_000_000: cl.append([])
    cl[-1].append(b)
last = cl[-1]
t0 = int(last[0]["Start_Timestamp"]); prev = None
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {((s - prev) / 1e3 if prev else 0):7.1f}  grid {r['Grid_Size_X']:>9}  {r['Kernel_Name'][:70]}")
    prev = e
print(f"total {(prev - t0) / 1e3:.1f} us, {len(last)} kernels, busy {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last) / 1e3:.1f} us")
PY
rm -rf $R/gpurun_out/$TAG/trace
cat $R/gpurun_out/$TAG/out.txt | tail -2; cat $R/gpurun_out/$TAG/timeline.txt
