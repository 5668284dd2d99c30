"""c_ezuvint_dev on cfg2's grid pair (G 4400x2200 -> L 7200x3601, bicubic, polar correction): us per pair under development switches, interleaved.  python tools/probe_cfg2_winds.py "VAR=1" ..."""
import os, sys, time
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from librmn_amd import ezscint as ez
import ezcases as ec
ni, nj, no, mo = 4400, 2200, 7200, 3601
gi = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); go = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
assert ez.ezdefset(go, gi) == 1
uu, vv = ec.synth_wind(ni, nj, seed=3)
d_u = torch.from_numpy(uu).cuda(); d_v = torch.from_numpy(vv).cuda()
ez.use_stream(torch.cuda.current_stream().cuda_stream)
variants = sys.argv[1:] or [""]
outs = {v: (torch.empty(no * mo, device="cuda"), torch.empty(no * mo, device="cuda")) for v in variants}
res = {v: [] for v in variants}
def run(v, reps):
    for kv in filter(None, v.split(",")): k, val = kv.split("="); os.environ[k] = val
    o_u, o_v = outs[v]
    for _ in range(3): assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) >= 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ez.ezuvint_dev(o_u, o_v, d_u, d_v)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps * 1e6
    for kv in filter(None, v.split(",")): os.environ.pop(kv.split("=")[0], None)
    return dt
for rnd in range(5):
    for v in variants: res[v].append(run(v, 20))
for v in variants:
    r = sorted(res[v]); same = bool(torch.equal(outs[v][0], outs[variants[0]][0]) and torch.equal(outs[v][1], outs[variants[0]][1]))
    print(f"[{v or 'defaults'}] median {r[2]:.1f} min {r[0]:.1f} us per pair; equal to the first variant: {same}", flush=True)
