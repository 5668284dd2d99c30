"""cfg3 probe (development tool): c_ezuvint, Z-on-E 2560x1280 -> L 4000x2000, device-resident pairs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from librmn_amd import ezscint as ez
import ezcases as ec

NI, NJ, NO, MO = 2560, 1280, 4000, 2000
ax, ay = ec.ze_axes(NI, NJ)
gdin = ez.ezgdef_fmem(NI, NJ, "Z", "E", *ec.E_IG, ax, ay)
gdout = ez.ezqkdef(NO, MO, "L", 9, 9, 0, 0)
assert ez.ezdefset(gdout, gdin) == 1
stream = torch.cuda.current_stream(); ez.use_stream(stream.cuda_stream)
uu, vv = ec.synth_wind(NI, NJ, seed=3)
for a in (uu, vv):
    a2 = a.reshape(NJ, NI); a2[:, -1] = a2[:, 0]
d_u = torch.from_numpy(uu).cuda(); d_v = torch.from_numpy(vv).cuda()
o_u = torch.empty(NO * MO, dtype=torch.float32, device="cuda"); o_v = torch.empty_like(o_u)
for polar in ("yes", "no"):
    ez.ezsetopt("interp_degree", "cubic"); ez.ezsetopt("polar_correction", polar)
    t0 = time.perf_counter(); rc = ez.ezuvint_dev(o_u, o_v, d_u, d_v); torch.cuda.synchronize()
    first = time.perf_counter() - t0
    assert rc in (0, 2), rc
    for _ in range(5):
        ez.ezuvint_dev(o_u, o_v, d_u, d_v)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 20
    t0 = time.perf_counter(); e0.record(stream)
    for _ in range(reps):
        ez.ezuvint_dev(o_u, o_v, d_u, d_v)
    e1.record(stream); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e6
    us = e0.elapsed_time(e1) * 1e3 / reps
    alg = 2 * 4 * NI * NJ + 2 * 4 * NO * MO
    print(f"cfg3 c_ezuvint_dev cubic polar={polar}: first call {first:.2f} s, {us:.1f} us/pair on the stream ({wall:.1f} us wall), "
          f"{NO*MO/us:.1f} Mpoint-pairs/s, {alg/us/1e3:.0f} GB/s algorithmic (90.21 MB per pair)")
