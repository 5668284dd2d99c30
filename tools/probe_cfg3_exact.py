"""cfg3 winds with ezhip_set_wind_exact(1) in a loop: us per pair (kernel traces: bash tools/prof_cmd.sh <tag> tools/probe_cfg3_exact.py)"""
import os, sys, time
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from librmn_amd import ezscint as ez
import ezcases as ec
ni, nj, no, mo = 2560, 1280, 4000, 2000
ax, ay = ec.ze_axes(ni, nj)
g_in = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); g_out = ez.ezqkdef(no, mo, "L", 9, 9, 0, 0)
assert ez.ezdefset(g_out, g_in) == 1
uu, vv = ec.synth_wind(ni, nj, seed=3)
d_u = torch.from_numpy(uu).cuda(); d_v = torch.from_numpy(vv).cuda()
o_u = torch.empty(no * mo, device="cuda"); o_v = torch.empty(no * mo, device="cuda")
ez.use_stream(torch.cuda.current_stream().cuda_stream)
ez.set_wind_exact(1)
for _ in range(3): assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) >= 0
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ez.ezuvint_dev(o_u, o_v, d_u, d_v)
torch.cuda.synchronize()
print(f"cfg3 exact winds: {(time.perf_counter() - t0) / 10 * 1e6:.1f} us per pair")
