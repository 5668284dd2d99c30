"""armn UNCOMPRESS of ONE full-size cfg5 record, a few times (for kernel traces: tools/prof_cmd.sh <tag> tools/probe_decode1.py [noise])"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import ezcases as ec
from librmn_amd import packers as pk
no, mo = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (7200, 3601)
n = no * mo
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-5
stride = 4 + n // 2 + 64
f = ec.synth_field(no, mo, seed=40, noise=noise).reshape(1, n)
d_f = torch.from_numpy(f).cuda()
recs = torch.zeros(stride, dtype=torch.int32, device="cuda")
rc, zl = pk.pack16_compress_batch_dev(recs, stride, d_f, n, 1, no, mo, 16)
print("zlng ratio", float(zl[0]) / (2.0 * n))
toks = torch.zeros((1, 1 + n // 2), dtype=torch.int32, device="cuda")
cap = int(zl[0] + 3) // 4 + 1
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = pk.armn_uncompress_batch_dev(toks, 1 + n // 2, recs[4:], stride, cap, no, mo, 16, 1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"decode: {dt * 1e3:.2f} ms (rc {r})")
ref = toks.clone()
for mode in ("0", "2"):
    os.environ["EZHIP_DEC_SCAN"] = mode
    toks.zero_()
    for rep in range(1 if mode == "2" else 5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = pk.armn_uncompress_batch_dev(toks, 1 + n // 2, recs[4:], stride, cap, no, mo, 16, 1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"EZHIP_DEC_SCAN={mode}: {dt * 1e3:.2f} ms (rc {r}); tokens equal to the default form's: {bool(torch.equal(toks, ref))}")
os.environ.pop("EZHIP_DEC_SCAN")
