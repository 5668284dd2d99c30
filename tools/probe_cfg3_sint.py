"""c_ezsint_dev on cfg3's grid pair (k_st): us per field, median of five bursts of 20"""
import os, sys, time
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from librmn_amd import ezscint as ez
import ezcases as ec
ni, nj, no, mo = 2560, 1280, 4000, 2000
ax, ay = ec.ze_axes(ni, nj)
g_in = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); g_out = ez.ezqkdef(no, mo, "L", 9, 9, 0, 0)
assert ez.ezdefset(g_out, g_in) == 1
ez.use_stream(torch.cuda.current_stream().cuda_stream)
f = ec.synth_field(ni, nj, seed=3); f.reshape(nj, ni)[:, -1] = f.reshape(nj, ni)[:, 0]
d_f = torch.from_numpy(f).cuda(); o = torch.empty(no * mo, device="cuda")
for _ in range(3): assert ez.ezsint_dev(o, d_f) >= 0
torch.cuda.synchronize()
ts = []
for b in range(5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ez.ezsint_dev(o, d_f)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / 20)
print(f"c_ezsint_dev cfg3: median {sorted(ts)[2]:.1f} min {min(ts):.1f} us per field; checksum {float(o.double().sum()):.6f}")
