"""armn UNCOMPRESS of a batch of full-size cfg5 records at ratio ~0.5 (for kernel traces: tools/prof_cmd.sh <tag> tools/probe_decode_batch.py [F] [noise])"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import ezcases as ec
from librmn_amd import packers as pk
no, mo = 7200, 3601
n = no * mo
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 5e-4
stride = 4 + n // 2 + 64
base = torch.from_numpy(ec.synth_field(no, mo, seed=40, noise=noise)).cuda()
d_f = torch.stack([base + 0.01 * f for f in range(F)]).contiguous()
recs = torch.zeros(F * stride, dtype=torch.int32, device="cuda")
rc, zl = pk.pack16_compress_batch_dev(recs, stride, d_f, n, F, no, mo, 16)
print("zlng ratio", float(zl[0]) / (2.0 * n))
toks = torch.zeros((F, 1 + n // 2), dtype=torch.int32, device="cuda")
cap = int(max(zl) + 3) // 4 + 1
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = pk.armn_uncompress_batch_dev(toks, 1 + n // 2, recs[4:], stride, cap, no, mo, 16, F)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"decode batch of {F}: {dt * 1e3:.2f} ms = {dt * 1e3 / F:.2f} ms per field (rc {r})")
