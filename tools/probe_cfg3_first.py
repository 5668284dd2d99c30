"""cfg3's first-call cost by part: c_ezdefset, the set's located x, y (ezhip_prepare_set: host threads for a rotated source, EZHIP_DEVICE_LOCATE=1: the device), the first
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
