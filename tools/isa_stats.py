"""Instruction mix of one kernel in a device assembly listing (hipcc --cuda-device-only -S): python tools/isa_stats.py file.s <kernel name substring> [top]"""
import sys, re, collections
src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]; top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
cur = None; cnt = collections.defaultdict(collections.Counter); meta = {}
for ln in src:
    m = re.match(r"^(_Z\w+):", ln)
    if m: cur = m.group(1); continue
    if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"): cur = None if ln.startswith(".Lfunc_end") else cur
    if cur is None: continue
    t = ln.strip()
    if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
        m2 = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size|accum_offset)\s+(\S+)", ln)
        if m2: meta.setdefault(cur, {})[m2.group(1)] = m2.group(2)
        m3 = re.match(r"\s*;\s*(ScratchSize|Occupancy|NumVgprs|NumAgprs|TotalNumVgprs|LDSByteSize|SGPRBlocks|NumSgprs):\s*(\S+)", ln)
        if m3: meta.setdefault(cur, {})[m3.group(1)] = m3.group(2)
        continue
    cnt[cur][t.split()[0]] += 1
for k in cnt:
    if pat in k:
        c = cnt[k]; tot = sum(c.values())
        grp = collections.Counter()
        for op, n in c.items():
            g = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
            grp[g] += n
        print(k[:110]); print("  total", tot, dict(grp), meta.get(k, {}))
        print("  " + ", ".join(f"{op} {n}" for op, n in c.most_common(top)))
