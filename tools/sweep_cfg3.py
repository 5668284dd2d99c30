"""cfg3 (c_ezuvint_dev, rotated Z-on-E 2560x1280 -> L 4000x2000, bicubic, polar correction on) under development switches, INTERLEAVED on one device:
us per wind pair and the results compared with the first variant's.  python tools/sweep_cfg3.py "VAR=val" ...  ("" = defaults)"""
import os, sys, time
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from librmn_amd import ezscint as ez
import ezcases as ec
ni, nj, no, mo = 2560, 1280, 4000, 2000
ax, ay = ec.ze_axes(ni, nj)
g_in = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); g_out = ez.ezqkdef(no, mo, "L", 9, 9, 0, 0)
assert ez.ezdefset(g_out, g_in) == 1
uu, vv = ec.synth_wind(ni, nj, seed=3)
d_u = torch.from_numpy(uu).cuda(); d_v = torch.from_numpy(vv).cuda()
ez.use_stream(torch.cuda.current_stream().cuda_stream)
variants = sys.argv[1:] or [""]
outs = {v: (torch.empty(no * mo, device="cuda"), torch.empty(no * mo, device="cuda")) for v in variants}
res = {v: [] for v in variants}; touched = set()
def run(v, reps):
    for k in touched: os.environ.pop(k, None)
    for kv in filter(None, v.split(",")):
        k, val = kv.split("="); os.environ[k] = val; touched.add(k)
    o_u, o_v = outs[v]
    for _ in range(3): assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) >= 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ez.ezuvint_dev(o_u, o_v, d_u, d_v)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for v in variants: run(v, 20)
for rnd in range(7):
    for v in variants: res[v].append(run(v, 20))
v0 = variants[0]
for v in variants:
    same = bool(torch.equal(outs[v][0], outs[v0][0]) and torch.equal(outs[v][1], outs[v0][1]))
    r = sorted(res[v]); print(f"[{v or 'defaults'}] median {r[3]:.1f} min {r[0]:.1f} max {r[-1]:.1f} us per pair; results equal to the first variant's: {same}", flush=True)
