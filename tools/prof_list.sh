# rocprofv3 kernel trace of a python tool; the launches of kernels matching PATTERN in launch order: bash tools/prof_list.sh <tag> <pattern> <script> [args]
TAG=$1; PAT=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
rm -rf $R/gpurun_out/$TAG/trace
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/trace -o b -- python3 $R/$@ > $R/gpurun_out/$TAG/under_rocprof.txt 2> $R/gpurun_out/$TAG/rocprof.err
T=$(find $R/gpurun_out/$TAG/trace -name "*kernel_trace.csv" | head -1)
python3 - "$T" "$PAT" > $R/gpurun_out/$TAG/list.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = re.compile(sys.argv[2])
for r in rows:
    if pat.search(r["Kernel_Name"]):
        print(f'{r["Kernel_Name"][:60]:62s} grid={r["Grid_Size_X"]:>10s} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:10.1f} us')
PY
rm -rf $R/gpurun_out/$TAG/trace
grep -v amdgpu.ids $R/gpurun_out/$TAG/under_rocprof.txt | tail -12; cat $R/gpurun_out/$TAG/list.txt
