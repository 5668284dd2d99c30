"""Timings of the secondary paths at realistic sizes (development tool): PS source, masked interpolation, hemispheric source."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import ezcases as ec
from librmn_amd import ezscint as ez

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

ez.use_stream(torch.cuda.current_stream().cuda_stream)
no, mo = 7200, 3601
go = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
d_out = torch.empty(no * mo, dtype=torch.float32, device="cuda")
# polar-stereographic source 4001 x 4001, 5 km at 60N, pole in the middle
gn = ez.ezqkdef(4001, 4001, "N", 20005, 20005, 50, 1000)       # pi = pj = 2000.5, d60 = 5000 m, dgrw = 10
assert gn >= 0 and ez.ezdefset(go, gn) == 1
ez.ezsetopt("interp_degree", "cubic")
t0 = time.perf_counter(); assert ez.prepare_set() == 0; torch.cuda.synchronize(); tp = (time.perf_counter() - t0) * 1e3
d_in = torch.rand(4001 * 4001, dtype=torch.float32, device="cuda")
print(f"N 4001x4001 -> L 7200x3601 cubic: first call {tp:.0f} ms, {bench(lambda: ez.ezsint_dev(d_out, d_in)):.0f} us per field")
# hemispheric A source 1440 x 360 north
ga = ez.ezqkdef(1440, 360, "A", 1, 0, 0, 0)
assert ez.ezdefset(go, ga) == 1 and ez.prepare_set() == 0
d_in2 = torch.rand(1440 * 360, dtype=torch.float32, device="cuda")
print(f"A north 1440x360 -> L 7200x3601 cubic: {bench(lambda: ez.ezsint_dev(d_out, d_in2)):.0f} us per field")
# masked interpolation at cfg2 size
gg = ez.ezqkdef(4400, 2200, "G", 0, 0, 0, 0)
assert ez.ezdefset(go, gg) == 1 and ez.prepare_set() == 0
d_in3 = torch.rand(4400 * 2200, dtype=torch.float32, device="cuda")
d_mi = (torch.rand(4400 * 2200, device="cuda") > 0.2).to(torch.int32); d_mo = torch.empty(no * mo, dtype=torch.int32, device="cuda")
L = ez._lib()
import ctypes
L.c_ezsint_mdm_dev.argtypes = [ctypes.c_void_p] * 4
print(f"c_ezsint_mdm G 4400x2200 -> L 7200x3601 cubic: {bench(lambda: L.c_ezsint_mdm_dev(d_out.data_ptr(), d_mo.data_ptr(), d_in3.data_ptr(), d_mi.data_ptr())):.0f} us per field (c_ezsint alone: {bench(lambda: ez.ezsint_dev(d_out, d_in3)):.0f})")
