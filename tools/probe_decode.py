"""Times armn UNCOMPRESS on full-size cfg5 records: one stream, and a batch decoded concurrently (development tool)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import ezcases as ec
from librmn_amd import packers as pk

no, mo = 7200, 3601
n = no * mo
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
stride = 4 + n // 2 + 64
fields = np.stack([ec.synth_field(no, mo, seed=40 + f, noise=1e-5 * (1 + f % 3)) for f in range(F)]).reshape(F, n)
d_f = torch.from_numpy(fields).cuda()
recs = torch.zeros(F * stride, dtype=torch.int32, device="cuda")
rc, zl = pk.pack16_compress_batch_dev(recs, stride, d_f, n, F, no, mo, 16)
print("zlng ratio", (zl / (2.0 * n)).round(3))
toks = torch.zeros((F, 1 + n // 2), dtype=torch.int32, device="cuda")
for cap, what in ((int(zl.max() + 3) // 4 + 1, "record length known"), (stride - 4, "capacity = plain record")):
    for nb in (1, F):
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = pk.armn_uncompress_batch_dev(toks, 1 + n // 2, recs[4:], stride, cap, no, mo, 16, nb)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"decode batch of {nb} ({what}): {dt * 1e3:.2f} ms total, {dt * 1e3 / nb:.2f} ms per field (rc {r})")
out = torch.empty(n, dtype=torch.float32, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
pk.uncompress_unpack16_dev(out, recs, no, mo, 16, 1)
torch.cuda.synchronize(); print(f"record -> floats: {(time.perf_counter() - t0) * 1e3:.2f} ms")
