"""c_armn_uncompress32 with record and field in HBM (c_armn_compress32_dev -> c_armn_uncompress32_zdev): ms per field; python tools/probe_a32_dev.py ni nj [kind]
EZHIP_DEC_SCAN=2 prints the verdict of each decoder form per plane"""
import os, sys, time
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from librmn_amd import packers as pk
import test_oracle_armn32 as ta32
ni, nj = int(sys.argv[1]), int(sys.argv[2]); kind = sys.argv[3] if len(sys.argv) > 3 else "positive"
f = ta32.field32(ni, nj, kind, seed=5)
d_f = torch.from_numpy(f).cuda(); d_z = torch.zeros(ni * nj + 64, dtype=torch.int32, device="cuda"); d_b = torch.empty(ni * nj, dtype=torch.float32, device="cuda")
zl = pk.armn_compress32_dev(d_z, d_f, ni, nj, 32)
print(f"{ni} x {nj} {kind}: record {zl} bytes (ratio {zl / (4.0 * ni * nj):.3f})")
best = 1e9
for k in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = pk.armn_uncompress32_zdev(d_b, d_z, 4 * ((zl + 3) // 4), ni, nj, 32)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"c_armn_uncompress32_zdev: rc {rc}, {best * 1e3:.2f} ms, bit-identical {bool(torch.equal(d_b.view(torch.int32), d_f.view(torch.int32)))}")
