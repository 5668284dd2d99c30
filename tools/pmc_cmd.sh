# rocprofv3 --pmc pass(es) over a python tool: bash tools/pmc_cmd.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- <script> [args]
TAG=$1; shift
PASSES=()
while [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
k=0
for P in "${PASSES[@]}"; do
  k=$((k+1))
  rm -rf $R/gpurun_out/$TAG/pmc$k
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/pmc$k -o p -- python3 $R/$@ > $R/gpurun_out/$TAG/pmc$k.log 2>&1
  C=$(find $R/gpurun_out/$TAG/pmc$k -name "*counter_collection.csv" | head -1)
  python3 - "$C" >> $R/gpurun_out/$TAG/pmc_summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = (r["Kernel_Name"][:50], r["Grid_Size"], r["Counter_Name"])
    agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    import os
    if any(f in k[0] for f in os.environ.get("KFILTER", "k_armn,k_sepx").split(",")):
        print(f"{k[0]:52s} grid={k[1]:>10s} {k[2]:28s} dispatches={v[0]:4d} mean={v[1]/v[0]:16.1f}")
PY
  rm -rf $R/gpurun_out/$TAG/pmc$k
done
cat $R/gpurun_out/$TAG/pmc_summary.txt
