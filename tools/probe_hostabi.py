"""host-pointer ABI probe (development tool): c_ezsint(zout, zin) with pageable numpy buffers, cfg2 shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from librmn_amd import ezscint as ez
import ezcases as ec

NI_S, NJ_S, NI_D, NJ_D = 4400, 2200, 7200, 3601
gdin = ez.ezqkdef(NI_S, NJ_S, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(NI_D, NJ_D, "L", 5, 5, 0, 0)
ez.ezdefset(gdout, gdin)
zin = ec.synth_field(NI_S, NJ_S, seed=2)
t0 = time.perf_counter(); rc, z = ez.ezsint(zin, NI_D * NJ_D); first = time.perf_counter() - t0
ts = []
for _ in range(8):
    t0 = time.perf_counter(); rc, z = ez.ezsint(zin, NI_D * NJ_D); ts.append(time.perf_counter() - t0)
print(f"c_ezsint host-pointer ABI cfg2: first call (plan + locate) {first*1e3:.1f} ms, steady {np.median(ts)*1e3:.2f} ms per field "
      f"({NI_D*NJ_D/np.median(ts)/1e6:.0f} Mpoints/s, {142.4288/np.median(ts)/1e3:.1f} GB/s of PCIe payload)")
