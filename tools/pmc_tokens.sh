# WRITE_SIZE / time of the token pass (k_sepx<3,16,3>) with plain vs nontemporal stores (EZHIP_DEBUG=256 = nontemporal)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dbg in 0 256; do
  export EZHIP_DEBUG=$dbg
  rm -rf $R/gpurun_out/pmctok_$dbg
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmctok_$dbg -o p -- python3 $R/tools/probe_cfg5.py 32 > $R/gpurun_out/pmctok_$dbg.log 2>&1
  C=$(find $R/gpurun_out/pmctok_$dbg -name "*counter_collection.csv" | head -1)
  python3 - "$C" "$dbg" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "k_sepx" in r["Kernel_Name"] and r["Grid_Size"] == "7380992": agg[r["Kernel_Name"][:30]].append(float(r["Counter_Value"]))
for k, v in agg.items(): print("EZHIP_DEBUG=%s %s WRITE_SIZE per field: %.2f MB (%d launches)" % (sys.argv[2], k, sum(v) / len(v) * 1024 / 1e6 / 32, len(v)))
PY
  rm -rf $R/gpurun_out/pmctok_$dbg
  python3 $R/tools/probe_cfg5.py 32 2>/dev/null | head -1
done
