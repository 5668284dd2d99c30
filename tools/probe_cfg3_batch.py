"""cfg3's grid pair, c_ezsint_batch_dev of F fields from the rotated source (k_st over the batch): us per field; EZHIP_NO_ST_BATCH=1: field by field"""
import os, sys, time
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from librmn_amd import ezscint as ez
import ezcases as ec
ni, nj, no, mo = 2560, 1280, 4000, 2000
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ax, ay = ec.ze_axes(ni, nj)
g_in = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); g_out = ez.ezqkdef(no, mo, "L", 9, 9, 0, 0)
assert ez.ezdefset(g_out, g_in) == 1
base = ec.synth_field(ni, nj, seed=3); base.reshape(nj, ni)[:, -1] = base.reshape(nj, ni)[:, 0]
d_in = torch.stack([torch.from_numpy(base).cuda() + 0.01 * f for f in range(F)]).contiguous()
d_out = torch.empty((F, no * mo), dtype=torch.float32, device="cuda")
ez.use_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(3): assert ez.ezsint_batch_dev(d_out, d_in, F) >= 0
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ez.ezsint_batch_dev(d_out, d_in, F)
torch.cuda.synchronize()
print(f"cfg3 grid pair, c_ezsint_batch_dev of {F} fields: {(time.perf_counter() - t0) / 10 / F * 1e6:.1f} us per field")
