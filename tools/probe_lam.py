"""a regional rotated source (Z-on-E without wrap, 2000 x 1500) -> lat-lon 3600 x 2400 inside it: c_ezsint_dev and c_ezuvint_dev, us per call (staged kernels from the second
call of the set on; EZHIP_NO_ST=1 / EZHIP_NO_UVT=1: the gathering kernels)"""
import os, sys, time
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from librmn_amd import ezscint as ez
import ezcases as ec, oraclelib as ol
ni, nj, no, mo = 2000, 1500, 3600, 2400
ax, ay = ec.zereg_axes(ni, nj)
g_in = ez.ezgdef_fmem(ni, nj, "Z", "E", *ol.cxgaig("E", 40.0, 270.0, 50.0, 95.0), ax, ay)
g_out = ez.ezqkdef(no, mo, "L", *ol.cxgaig("L", 28.0, 255.0, 24.0 / (mo - 1), 36.0 / (no - 1)))
assert ez.ezdefset(g_out, g_in) == 1
uu, vv = ec.synth_wind(ni, nj, seed=3)
d_u = torch.from_numpy(uu).cuda(); d_v = torch.from_numpy(vv).cuda()
o_u = torch.empty(no * mo, device="cuda"); o_v = torch.empty(no * mo, device="cuda")
ez.use_stream(torch.cuda.current_stream().cuda_stream)
def t(fn):
    for _ in range(3): assert fn() >= 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 20 * 1e6
print(f"regional source {ni} x {nj} -> {no} x {mo}: c_ezsint_dev {t(lambda: ez.ezsint_dev(o_u, d_u)):.1f} us per field, c_ezuvint_dev {t(lambda: ez.ezuvint_dev(o_u, o_v, d_u, d_v)):.1f} us per pair")
