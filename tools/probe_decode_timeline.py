"""ONE armn UNCOMPRESS of a full-size cfg5 record, six times 20 ms apart (for tools/trace_decode.sh: the kernel timeline of the last one)"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import ezcases as ec
from librmn_amd import packers as pk
no, mo = 7200, 3601
n = no * mo
stride = 4 + n // 2 + 64
f = ec.synth_field(no, mo, seed=40, noise=1e-5).reshape(1, n)
d_f = torch.from_numpy(f).cuda()
recs = torch.zeros(stride, dtype=torch.int32, device="cuda")
rc, zl = pk.pack16_compress_batch_dev(recs, stride, d_f, n, 1, no, mo, 16)
toks = torch.zeros((1, 1 + n // 2), dtype=torch.int32, device="cuda")
cap = int(zl[0] + 3) // 4 + 1
for rep in range(6):
    time.sleep(0.02)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = pk.armn_uncompress_batch_dev(toks, 1 + n // 2, recs[4:], stride, cap, no, mo, 16, 1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"decode: {dt * 1e3:.3f} ms (rc {r})")
