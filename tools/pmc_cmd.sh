# HBM traffic of the kernels matching PATTERN in a python tool: separate --pmc passes for FETCH_SIZE (x2 on gfx950, MI355X_MICROARCH.md) and WRITE_SIZE;
# bash tools/pmc_cmd.sh <tag> <pattern> <script> [args] -> per (kernel, grid): dispatches, median KiB per dispatch
TAG=$1; PAT=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 $R/$@ > $O/pmc_$c.log 2>&1
  C=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$C" "$PAT" "$c" <<'PY' | tee $O/$c.txt
import csv, sys, re, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2]); agg = collections.defaultdict(list)
for r in rows:
    if pat.search(r["Kernel_Name"]): agg[(r["Kernel_Name"].split("(")[0][:50], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    mult = 2.0 if sys.argv[3] == "FETCH_SIZE" else 1.0
    print(f"{sys.argv[3]:10s} {k[0]:52s} grid={k[1]:>10s} n={len(v):4d} median {statistics.median(v) * mult * 1024 / 1e6:10.2f} MB per dispatch (corrected)")
PY
  rm -rf $O/pmc_$c
done
