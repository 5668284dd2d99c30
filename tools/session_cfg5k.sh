# kernel times of the cfg5 pipeline's passes under their knock-outs (develop build): what of B and E is memory, what is instruction issue
export EZHIP_LIBRARY=$GRAFT_REPO_ROOT/devlibs/librmn_ez_hip_dev.so
O=gpurun_out/$1; mkdir -p $O
for v in "EZHIP_ENC_DEBUG=0" "EZHIP_ENC_DEBUG=32" "EZHIP_ENC_DEBUG=1" "EZHIP_ENC_DEBUG=3" "EZHIP_DEBUG=1" "EZHIP_DEBUG=16" "EZHIP_DEBUG=8"; do
  env $v bash tools/prof_cmd.sh $1/run tools/probe_enc_batch.py ${v#EZHIP_ENC_DEBUG=} > /dev/null 2>&1
  echo "== $v" >> $O/kernel_times.txt
  grep "k_sepx\|k_armn_enc1\|k_bb_" gpurun_out/$1/run/summary.txt | head -6 | cut -c1-40,70- >> $O/kernel_times.txt
done
cat $O/kernel_times.txt
