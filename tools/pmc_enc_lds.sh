# LDS-array cycles of k_armn_enc1 per phase: SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS of the 32-field launch under the EZHIP_ENC_DEBUG knock-outs
# (develop build: EZHIP_LIBRARY=devlibs/librmn_ez_hip_dev.so; 32 staging only; 3 = no emission + no copy-out; 1 no emission; 2 no copy-out; 0 everything)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/enclds; mkdir -p $O; rm -f $O/summary.txt
for dbg in 0 32 3 1 2; do
  export EZHIP_ENC_DEBUG=$dbg
  rm -rf $O/p
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/p -o p -- python3 $R/tools/probe_enc_batch.py $dbg > $O/log_$dbg.txt 2>&1
  C=$(find $O/p -name "*counter_collection.csv" | head -1)
  python3 - "$C" "$dbg" >> $O/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "k_armn_enc1" in r["Kernel_Name"] and int(r["Grid_Size"]) > 20000000: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(agg["SQ_WAVES"]) / max(1, len(agg["SQ_WAVES"]))
print("EZHIP_ENC_DEBUG=%-3s per wave: LDS instructions %.1f  LDS-array cycles %.0f  of them bank conflicts %.0f   (waves %.0f)" % (sys.argv[2], *(sum(agg[k]) / len(agg[k]) / w for k in ("SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT")), w))
PY
  rm -rf $O/p
done
cat $O/summary.txt
