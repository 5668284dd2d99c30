"""cfg3's first-call cost by part: c_ezdefset, the set's located x, y (ezhip_prepare_set: the device since round 5, EZHIP_HOST_LOCATE=1: host threads), the first
and the second c_ezuvint_dev (wind matrix, special points' list, tile table, tile-ordered copy).  ms, wall clock."""
import os, sys, time
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from librmn_amd import ezscint as ez
import ezcases as ec
ni, nj, no, mo = 2560, 1280, 4000, 2000
ax, ay = ec.ze_axes(ni, nj)
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t0 = time.perf_counter()
g_in = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); g_out = ez.ezqkdef(no, mo, "L", 9, 9, 0, 0)
assert ez.ezdefset(g_out, g_in) == 1
t1 = time.perf_counter()
ez.use_stream(torch.cuda.current_stream().cuda_stream)
assert ez.prepare_set() >= 0
torch.cuda.synchronize(); t2 = time.perf_counter()
uu, vv = ec.synth_wind(ni, nj, seed=3)
d_u = torch.from_numpy(uu).cuda(); d_v = torch.from_numpy(vv).cuda()
o_u = torch.empty(no * mo, device="cuda"); o_v = torch.empty(no * mo, device="cuda")
torch.cuda.synchronize(); t3 = time.perf_counter()
assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) >= 0
torch.cuda.synchronize(); t4 = time.perf_counter()
assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) >= 0
torch.cuda.synchronize(); t5 = time.perf_counter()
print(f"define grids + set {1e3 * (t1 - t0):.2f} | prepare_set (locate x, y of {no * mo} points) {1e3 * (t2 - t1):.2f} | first c_ezuvint_dev {1e3 * (t4 - t3):.2f} | second {1e3 * (t5 - t4):.2f} ms; host threads available: {os.cpu_count()}")
# a second, different set in the same process: its first-call cost without the process's one-time work (code object load, first allocations)
ta = time.perf_counter()
g_in2 = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); g_out2 = ez.ezqkdef(no - 8, mo - 4, "L", 9, 9, 0, 0)
assert ez.ezdefset(g_out2, g_in2) == 1
tb = time.perf_counter()
assert ez.prepare_set() >= 0
torch.cuda.synchronize(); tc = time.perf_counter()
o_u2 = torch.empty((no - 8) * (mo - 4), device="cuda"); o_v2 = torch.empty_like(o_u2)
torch.cuda.synchronize(); td = time.perf_counter()
assert ez.ezuvint_dev(o_u2, o_v2, d_u, d_v) >= 0
torch.cuda.synchronize(); te = time.perf_counter()
print(f"second set in the same process: define {1e3 * (tb - ta):.2f} | prepare_set {1e3 * (tc - tb):.2f} | first c_ezuvint_dev {1e3 * (te - td):.2f} ms")
ax3, ay3 = ec.ze_axes(2400, 1200)
ta = time.perf_counter()
g_in3 = ez.ezgdef_fmem(2400, 1200, "Z", "E", *ec.E_IG, ax3, ay3)
assert ez.ezdefset(g_out, g_in3) == 1
tb = time.perf_counter()
assert ez.prepare_set() >= 0
torch.cuda.synchronize(); tc = time.perf_counter()
print(f"third set (another source grid): define {1e3 * (tb - ta):.2f} | prepare_set {1e3 * (tc - tb):.2f} ms")
