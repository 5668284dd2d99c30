"""c_armn_compress32_dev (field and record in HBM) of a ni x nj REAL field: ms per field; python tools/probe_a32_enc.py ni nj [kind]"""
import os, sys, time
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from librmn_amd import packers as pk
import test_oracle_armn32 as ta32
ni, nj = int(sys.argv[1]), int(sys.argv[2]); kind = sys.argv[3] if len(sys.argv) > 3 else "positive"
f = ta32.field32(ni, nj, kind, seed=5)
d_f = torch.from_numpy(f).cuda(); d_z = torch.zeros(ni * nj + 64, dtype=torch.int32, device="cuda")
best = 1e9
for k in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    zl = pk.armn_compress32_dev(d_z, d_f, ni, nj, 32)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"c_armn_compress32_dev {ni} x {nj} {kind}: {zl} bytes (ratio {zl / (4.0 * ni * nj):.3f}), {best * 1e3:.2f} ms")
