"""cfg2's first-call cost by part (EZHIP_TRACE_FIRST=1 prints the library's own split): grid definitions (Gaussian latitudes), c_ezdefset, ezhip_prepare_set, the first and second c_ezsint_dev"""
import os, sys, time
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from librmn_amd import ezscint as ez
import ezcases as ec
ni, nj, no, mo = 4400, 2200, 7200, 3601
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t0 = time.perf_counter()
g_in = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0)
ta = time.perf_counter()
g_out = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
tb = time.perf_counter()
assert ez.ezdefset(g_out, g_in) == 1
t1 = time.perf_counter()
ez.use_stream(torch.cuda.current_stream().cuda_stream)
assert ez.prepare_set() >= 0
torch.cuda.synchronize(); t2 = time.perf_counter()
d_in = torch.from_numpy(ec.synth_field(ni, nj, seed=2)).cuda(); d_out = torch.empty(no * mo, device="cuda")
torch.cuda.synchronize(); t3 = time.perf_counter()
assert ez.ezsint_dev(d_out, d_in) >= 0
torch.cuda.synchronize(); t4 = time.perf_counter()
assert ez.ezsint_dev(d_out, d_in) >= 0
torch.cuda.synchronize(); t5 = time.perf_counter()
print(f"define G {1e3 * (ta - t0):.2f} | define L {1e3 * (tb - ta):.2f} | c_ezdefset {1e3 * (t1 - tb):.2f} | prepare_set {1e3 * (t2 - t1):.2f} | first c_ezsint_dev {1e3 * (t4 - t3):.2f} | second {1e3 * (t5 - t4):.2f} ms")
# a second separable set in the same process (another target): without the process's one-time work
ta = time.perf_counter()
g_out2 = ez.ezqkdef(no - 16, mo - 7, "L", 5, 5, 0, 0)
assert ez.ezdefset(g_out2, g_in) == 1
tb = time.perf_counter()
assert ez.prepare_set() >= 0
torch.cuda.synchronize(); tc = time.perf_counter()
print(f"second set in the same process: define + set {1e3 * (tb - ta):.2f} | prepare_set {1e3 * (tc - tb):.2f} ms")
