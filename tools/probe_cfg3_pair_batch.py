"""cfg3 wind pairs through c_ezuvint_batch_dev: us per pair by batch size against single c_ezuvint_dev calls; results compared bit for bit.  python tools/probe_cfg3_pair_batch.py [K ...]"""
import os, sys, time
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from librmn_amd import ezscint as ez
import ezcases as ec
ni, nj, no, mo = 2560, 1280, 4000, 2000
Ks = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]
KM = max(Ks)
ax, ay = ec.ze_axes(ni, nj)
g_in = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); g_out = ez.ezqkdef(no, mo, "L", 9, 9, 0, 0)
assert ez.ezdefset(g_out, g_in) == 1
ez.use_stream(torch.cuda.current_stream().cuda_stream)
uu, vv = ec.synth_wind(ni, nj, seed=3)
base_u = torch.from_numpy(uu).cuda(); base_v = torch.from_numpy(vv).cuda()
d_u = torch.stack([base_u * (1.0 + 0.01 * f) for f in range(KM)]).contiguous(); d_v = torch.stack([base_v * (1.0 - 0.01 * f) for f in range(KM)]).contiguous()
o_u = torch.empty((KM, no * mo), device="cuda"); o_v = torch.empty_like(o_u)
r_u = torch.empty_like(o_u); r_v = torch.empty_like(o_u)
for rep in range(2):
    for f in range(KM):
        assert ez.ezuvint_dev(r_u[f], r_v[f], d_u[f], d_v[f]) >= 0
torch.cuda.synchronize()
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
t1 = timed(lambda: [ez.ezuvint_dev(o_u[f], o_v[f], d_u[f], d_v[f]) for f in range(KM)]) / KM
print(f"single calls: {t1:.1f} us per pair")
for K in Ks:
    if K < 2: continue
    o_u.zero_(); o_v.zero_()
    t = timed(lambda: ez.ezuvint_batch_dev(o_u, o_v, d_u, d_v, K)) / K
    torch.cuda.synchronize()
    print(f"batch of {K:2d}: {t:.1f} us per pair; equal to the single calls: {bool(torch.equal(o_u[:K], r_u[:K]) and torch.equal(o_v[:K], r_v[:K]))}")
