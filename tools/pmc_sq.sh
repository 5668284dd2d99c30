# SQ counters of the kernels matching PATTERN in a python tool (one pass per counter group): bash tools/pmc_sq.sh <tag> <pattern> <script> [args]
TAG=$1; PAT=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
mkdir -p $O; rm -f $O/sq.txt
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  rm -rf $O/p
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p -o p -- python3 $R/$@ > $O/log.txt 2>&1
  C=$(find $O/p -name "*counter_collection.csv" | head -1)
  python3 - "$C" "$PAT" >> $O/sq.txt <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2]); agg = collections.defaultdict(list)
for r in rows:
    if pat.search(r["Kernel_Name"]): agg[(r["Kernel_Name"].split("(")[0][:40], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print("%-42s grid=%10s %-24s n=%3d mean %.4g" % (k[0], k[1], k[2], len(v), sum(v) / len(v)))
PY
  rm -rf $O/p
done
cat $O/sq.txt
