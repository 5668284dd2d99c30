cd $GRAFT_REPO_ROOT
bash tools/prof_list.sh r03m_w8 "k_armn_enc1" tools/probe_enc_w8.py 32 | awk '{print}' | tail -50 | python3 -c "
import sys
lines=sys.stdin.read().splitlines()
for l in lines:
    if 'k_armn_enc1' not in l: print(l)
import collections
agg=collections.defaultdict(list)
for l in lines:
    if 'k_armn_enc1' in l:
        name=l.split('(')[0].strip(); agg[name].append(float(l.split()[-2]))
for k,v in agg.items(): print(k, 'n', len(v), 'min', min(v), 'median', sorted(v)[len(v)//2])
"
