"""Turns what tools/prof_round.sh left in gpurun_out/<tag>/ into the round's tracked artefacts under profiles/:

    python tools/make_profiles.py <tag> <prefix>          e.g.  r02g r02

    <prefix>_bench.json, <prefix>_bench_under_rocprof.json, <prefix>_bench_kernel_stats.csv      copied
    <prefix>_timed_launches.csv          the 60 timed 32-field k_sepx<3, 16, 0> launches of the profiled run
    <prefix>_kernel_stats_by_grid.csv    the kernel trace split by (kernel, grid)
    <prefix>_pmc_FETCH_SIZE_per_dispatch.csv / _WRITE_SIZE_   copied
    <prefix>_pmc_traffic.json            HBM traffic per field of k_sepx and of the fused cfg5 pipeline (FETCH_SIZE x 2 on gfx950, WRITE_SIZE exact, KiB)
"""
import collections, csv, json, os, shutil, statistics as st, sys

tag, prefix = sys.argv[1], sys.argv[2]
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
for a, b in (("bench.json", "bench.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"), ("kernel_stats.csv", "bench_kernel_stats.csv"),
             ("pmc_FETCH_SIZE.csv", "pmc_FETCH_SIZE_per_dispatch.csv"), ("pmc_WRITE_SIZE.csv", "pmc_WRITE_SIZE_per_dispatch.csv")):
    shutil.copyfile(os.path.join(src, a), os.path.join(dst, f"{prefix}_{b}"))

bench = json.loads(open(os.path.join(src, "bench_under_rocprof.json")).readline())
F = bench["roofline"]["fields_per_launch"]
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
rows = list(csv.DictReader(open(os.path.join(src, "dispatches.csv"))))
by_grid = collections.Counter(r["grid_x"] for r in rows if "k_sepx<3, 16, 0>" in r["kernel"])
big_grid = max(by_grid, key=lambda g: int(g))
big = sorted((r for r in rows if "k_sepx<3, 16, 0>" in r["kernel"] and r["grid_x"] == big_grid), key=lambda r: int(r["start_ns"]))
W, K = bench["warmup"], bench["steps"]
timed = big[W:W + K]
mean_us = st.mean(float(r["duration_us"]) for r in timed)
with open(os.path.join(dst, f"{prefix}_timed_launches.csv"), "w") as f:
    f.write(f"# the {K} TIMED k_sepx<3, 16, 0> launches ({F} fields each, grid {big_grid}) of the profiled bench run (rocprofv3 --kernel-trace; "
            f"launches {W + 1}..{W + K} of that kernel: {W} warm-up steps precede them); mean {mean_us:.2f} us -> "
            f"{alg / mean_us / 1e3:.1f} GB/s = {alg / mean_us / 1e3 / 8000:.4f} of 8 TB/s (bench line of the same run: {bench['roofline']['frac']:.4f})\n")
    f.write("dispatch_id,kernel,grid_x,start_ns,end_ns,duration_us\n")
    for r in timed:
        f.write(f"{r['dispatch_id']},\"{r['kernel']}\",{r['grid_x']},{r['start_ns']},{r['end_ns']},{r['duration_us']}\n")

agg = collections.defaultdict(list)
for r in rows:
    agg[(r["kernel"], r["grid_x"])].append(float(r["duration_us"]))
with open(os.path.join(dst, f"{prefix}_kernel_stats_by_grid.csv"), "w") as f:
    f.write("kernel,grid_x,calls,total_us,avg_us,min_us,max_us\n")
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        f.write(f"\"{k}\",{g},{len(v)},{sum(v):.1f},{st.mean(v):.2f},{min(v):.2f},{max(v):.2f}\n")


def pmc(name):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(src, f"pmc_{name}.csv"))):
        out[(r["kernel"], r["grid"])].append(float(r["value"]))
    return out


fetch, write = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")


def per_field(kernel_part, grid=None, nfields=F):
    def pick(tab):
        cands = [(k, v) for k, v in tab.items() if kernel_part in k[0] and (grid is None or k[1] == grid)]
        if not cands:
            return None, None
        k, v = max(cands, key=lambda kv: int(kv[0][1]))
        return k, st.median(v)
    kf, rd = pick(fetch); kw, wr = pick(write)
    if rd is None or wr is None:
        return None
    return {"grid": kf[1], "read_MB": round(rd * 2 * 1024 / nfields / 1e6, 2), "write_MB": round(wr * 1024 / nfields / 1e6, 2)}


main = per_field("k_sepx<3, 16, 0>")
A = per_field("k_sepx<3, 16, 2>"); B = per_field("k_sepx<3, 16, 3>"); E = per_field("k_armn_enc1")
# round 3: pass A is the bound pass (no interpolation): k_bb_bounds + k_bb_select + k_bb_eval + k_bb_special
BB = [per_field(k) for k in ("k_bb_bounds", "k_bb_select", "k_bb_eval", "k_bb_special")]
if all(BB):
    A = {"grid": "+".join(x["grid"] for x in BB), "read_MB": round(sum(x["read_MB"] for x in BB), 2), "write_MB": round(sum(x["write_MB"] for x in BB), 2),
         "parts": dict(zip(("k_bb_bounds", "k_bb_select", "k_bb_eval", "k_bb_special"), BB))}
P3 = per_field("k_uvt<32, 32, false, false>", nfields=1) or per_field("k_uvt<32, 32>", nfields=1) or per_field("k_pts2_irgd3w", nfields=1) or per_field("k_pts2", nfields=1)      # round 4: k_uvt from the second call of a grid set on
KB3 = bench.get("extras", {}).get("cfg3_uvint_batch", {}).get("workload", "")
KB3 = int(KB3.split(",")[1].split()[0]) if "pairs per call" in KB3 else 0                                             # c_ezuvint_batch_dev's pairs per call in this bench
P3B = per_field("k_uvt<32, 32, false, true>", nfields=KB3) if KB3 else None
S3 = per_field("k_pts_special2c", nfields=1) or per_field("k_pts_special2", nfields=1); W3 = per_field("k_polar_wind", nfields=1)
zl = bench["pack"]["cfg5"]["zlng_bytes"] if "cfg5" in bench["pack"] else bench["pack"]["zlng_bytes"]      # (round 6: the pack object is nested)
out = {
    "workload": "python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 under rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes, tools/prof_round.sh); "
                f"per-dispatch values in {prefix}_pmc_FETCH_SIZE_per_dispatch.csv / {prefix}_pmc_WRITE_SIZE_per_dispatch.csv (KiB); medians over the dispatches of a (kernel, grid)",
    "FETCH_SIZE_correction": "x2 (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE exact",
    "fields_per_launch": F,
    "kernel": f"k_sepx<3, 16, 0> (grid {main['grid']} = {F} fields)",
    "read_MB_per_field": main["read_MB"], "write_MB_per_field": main["write_MB"],
    "algorithmic_MB_per_field": round(alg / F / 1e6, 2),
    "traffic_MB_per_field": round(main["read_MB"] + main["write_MB"], 2),
    "traffic_over_algorithmic": round((main["read_MB"] + main["write_MB"]) / (alg / F / 1e6), 3),
}
if A and B and E:
    tot = sum(x["read_MB"] + x["write_MB"] for x in (A, B, E))
    calg = (alg / F - 4 * bench["config"]["points_per_field"] + zl) / 1e6          # source field in, compressed record out
    out["cfg5_fused_pipeline_MB_per_field"] = {("A k_bb_* (extrema from bounds of the source windows)" if all(BB) else "A k_sepx<3,16,2> (min/max only)"): A,
                                               "B k_sepx<3,16,3> (tokens)": B, "E k_armn_enc1 (one launch per batch)": E}
    out["cfg5_fused_pipeline_total_MB_per_field"] = round(tot, 1)
    out["cfg5_algorithmic_MB_per_field"] = round(calg, 2)
    out["cfg5_traffic_over_algorithmic"] = round(tot / calg, 2)
    out["cfg5_notes"] = ("floor of this structure: 2 x 38.7 (source read twice: bounds, tokens) + 51.9 + 51.9 (tokens out and in) + 25.4 (stream) = 206.6 MB = 3.22 x; "
                         "measured above it: halo rows of the source and of the token rows")
if P3:
    parts = {"k_uvt (k_pts2 before round 4)": P3, "k_pts_special2c": S3, "k_polar_wind": W3}
    tot3 = sum(x["read_MB"] + x["write_MB"] for x in parts.values() if x)
    out["cfg3_kernels_MB_per_pair"] = parts
    out["cfg3_traffic_MB_per_pair"] = round(tot3, 1)
    out["cfg3_algorithmic_MB_per_pair"] = 90.21
    out["cfg3_traffic_over_algorithmic"] = round(tot3 / 90.21, 2)
    if P3B:
        out["cfg3_batch_kernel_MB_per_pair"] = {"k_uvt<32, 32, false, true> (%d pairs per launch)" % KB3: P3B}
        out["cfg3_batch_traffic_MB_per_pair"] = round(P3B["read_MB"] + P3B["write_MB"] + (S3["read_MB"] + S3["write_MB"] if S3 else 0.0), 1)
    out["cfg3_notes"] = ("x, y of the located points and the per-point wind rotation (one packed word) -- 96 MB, read from the set's tile-ordered copy -- are inputs of every call next to "
                         "the staged source windows (~0.94 cells of 8 bytes per point) and 64 MB of results")
# round 5: the figures bench.py's other roofline objects read (every one labelled with this file in the line)
lone = [(k, st.median(v)) for k, v in fetch.items() if "k_sepx<3, 16, 0>" in k[0]]
if lone:
    kmin = min(lone, key=lambda kv: abs(int(kv[0][1]) - int(big_grid) / F))[0]      # the one-field launch of the headline's grid pair (the batch's grid / F, + its own pole blocks)
    if kmin in write:
        out["single_field_traffic_MB"] = round((st.median(fetch[kmin]) * 2 + st.median(write[kmin])) * 1024 / 1e6, 2)
        out["single_field_grid"] = kmin[1]
S1 = per_field("k_st<32, 32", nfields=1); S1s = per_field("k_pts_special(", nfields=1) or per_field("k_pts_special", nfields=1)
if S1:
    out["cfg3_sint_kernels_MB_per_field"] = {"k_st": S1, "k_pts_special": S1s}
    out["cfg3_sint_traffic_MB_per_field"] = round(sum(x["read_MB"] + x["write_MB"] for x in (S1, S1s) if x), 1)
CF = [per_field(k, nfields=1) for k in ("k_stats", "k_cf_header", "k_cf_pack16")]
if all(CF):
    out["compact_float_traffic_bytes_per_value"] = round(sum(x["read_MB"] + x["write_MB"] for x in CF) * 1e6 / bench["config"]["points_per_field"], 2)
try:
    valu = pmc("SQ_INSTS_VALU")

    def instr(kernel_part, nfields):
        c = [(k, v) for k, v in valu.items() if kernel_part in k[0]]
        if not c:
            return None
        k, v = max(c, key=lambda kv: int(kv[0][1]))
        return st.median(v) / nfields
    parts5 = {k: instr(k, F) for k in ("k_bb_bounds", "k_bb_select", "k_bb_eval", "k_bb_special", "k_sepx<3, 16, 3>", "k_armn_enc1")}
    if all(v is not None for v in parts5.values()):
        out["cfg5_valu_wave_instructions_per_field_by_kernel"] = {k: round(v) for k, v in parts5.items()}
        out["cfg5_valu_wave_instructions_per_field"] = round(sum(parts5.values()))
    i_st = instr("k_st<32, 32", 1)
    if i_st is not None:
        out["cfg3_sint_valu_wave_instructions_per_field"] = round(i_st + (instr("k_pts_special", 1) or 0))
    i_uvt = instr("k_uvt<32, 32, false, false>", 1) or instr("k_uvt<32, 32>", 1)
    if i_uvt is not None:
        out["cfg3_uvint_valu_wave_instructions_per_pair"] = round(i_uvt)
    shutil.copyfile(os.path.join(src, "pmc_SQ_INSTS_VALU.csv"), os.path.join(dst, f"{prefix}_pmc_SQ_INSTS_VALU_per_dispatch.csv"))
except FileNotFoundError:
    pass
json.dump(out, open(os.path.join(dst, f"{prefix}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
print(f"timed launches: mean {mean_us:.2f} us, frac {alg / mean_us / 1e3 / 8000:.4f}; bench line {bench['roofline']['avg_launch_us']:.2f} us / {bench['roofline']['frac']:.4f}")
