"""c_ezsint_dev on cfg3's grid pair (rotated Z-on-E 2560x1280 -> L 4000x2000): us per scalar field (k_pts, k_st from the second call on); python tools/probe_cfg3_scalar.py [cubic|linear|nearest]"""
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
deg = sys.argv[1] if len(sys.argv) > 1 else "cubic"
assert ez.ezsetopt("interp_degree", deg) == 0
uu, vv = ec.synth_wind(ni, nj, seed=3)
d_u = torch.from_numpy(uu).cuda()
o_u = torch.empty(no * mo, device="cuda")
ez.use_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    assert ez.ezsint_dev(o_u, d_u) >= 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ez.ezsint_dev(o_u, d_u)
torch.cuda.synchronize()
print(f"cfg3 grid pair, c_ezsint_dev (scalar, {deg}): {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per field")
