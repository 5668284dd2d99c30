"""One configuration of the vertical interpolation, a few launches (for rocprofv3 --pmc passes): python tools/probe_iv_one.py [f32|f64] [fused|posn|findpos] [algo]"""
import sys, os, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from librmn_amd import interpv as V
dt = torch.float64 if "f64" in sys.argv else torch.float32
mode = "posn" if "posn" in sys.argv else "findpos" if "findpos" in sys.argv else "fused"
algo = {"nearest": V.NEAREST, "lagrange": V.CUBIC_LAGRANGE, "derivs": V.CUBIC_DERIVS}.get(next((a for a in sys.argv[1:] if a in ("nearest", "lagrange", "derivs")), ""), V.LINEAR)
n, ns, nd = 7200 * 3601 // 4, 80, 60
ps = 1.0 + 0.05 * torch.sin(torch.arange(n, device="cuda", dtype=dt) * 1e-3)
vls = torch.linspace(1, ns, ns, device="cuda", dtype=dt)[:, None] * ps[None, :]
vld = (torch.linspace(1.5, ns - 0.5, nd, device="cuda", dtype=dt)[:, None] + torch.zeros((1, n), device="cuda", dtype=dt)).contiguous()
ss = torch.sin(vls * 0.1); sds = torch.cos(vls * 0.1) * 0.1
sd = torch.empty((nd, n), device="cuda", dtype=dt); sdd = torch.empty_like(sd)
posn = torch.empty((nd, n), device="cuda", dtype=torch.int32)
V.findpos_dev(n, vls, posn, vld)
f = {"fused": lambda: V.column_dev(algo, V.X_LAPSERATE, n, vls, ss, sds, None, vld, sd, sdd, 1, 1, 0.1, 0.1),
     "posn": lambda: V.interp_dev(algo, n, vls, ss, sds, posn, vld, sd, sdd, 0, 0),
     "findpos": lambda: V.findpos_dev(n, vls, posn, vld)}[mode]
f(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    f()
torch.cuda.synchronize()
print(f"{dt} {mode} algo {algo}: {(time.perf_counter() - t) / 5 * 1e3:.3f} ms")
