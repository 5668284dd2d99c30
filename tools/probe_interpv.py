"""Times the vertical-interpolation kernel at a model-sized problem (device-resident arrays) and prints the achieved
algorithmic bandwidth per entry point.  Development tool: python tools/probe_interpv.py [ncols ns nd]"""
import sys, time
import torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from librmn_amd import interpv as V

n = int(sys.argv[1]) if len(sys.argv) > 1 else 7200 * 3601 // 4
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 80
nd = int(sys.argv[3]) if len(sys.argv) > 3 else 60
coherent = "--random" not in sys.argv       # default: hybrid-like levels varying smoothly along the columns, common destination levels
for dt, es in ((torch.float32, 4), (torch.float64, 8)):
    if coherent:
        ps = 1.0 + 0.05 * torch.sin(torch.arange(n, device="cuda", dtype=dt) * 1e-3)
        vls = torch.linspace(1, ns, ns, device="cuda", dtype=dt)[:, None] * ps[None, :]
        vld = (torch.linspace(1.5, ns - 0.5, nd, device="cuda", dtype=dt)[:, None] + torch.zeros((1, n), device="cuda", dtype=dt)).contiguous()
    else:
        inc = torch.rand((ns, n), device="cuda", dtype=dt) + 0.2
        vls = torch.cumsum(inc, 0); del inc
        vld = vls[0] + torch.rand((nd, n), device="cuda", dtype=dt) * (vls[-1] - vls[0])
    ss = torch.sin(vls * 0.1); sds = torch.cos(vls * 0.1) * 0.1
    sd = torch.empty((nd, n), device="cuda", dtype=dt); sdd = torch.empty_like(sd)
    posn = torch.empty((nd, n), device="cuda", dtype=torch.int32)

    def timed(f, reps=5):
        f(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps

    rows = [("findpos", lambda: V.findpos_dev(n, vls, posn, vld), (ns + nd) * es + 4 * nd)]
    for nm, al, nin, nout in (("nearest", V.NEAREST, 2, 1), ("linear", V.LINEAR, 2, 1), ("lagrange", V.CUBIC_LAGRANGE, 2, 1), ("derivs", V.CUBIC_DERIVS, 3, 2)):
        rows.append((nm + " (from posn)", (lambda al=al: V.interp_dev(al, n, vls, ss, sds, posn, vld, sd, sdd, 0, 0)), (ns * nin + nd * (1 + nout)) * es + 4 * nd))
        rows.append((nm + " fused", (lambda al=al: V.column_dev(al, V.X_LAPSERATE, n, vls, ss, sds, None, vld, sd, sdd, 1, 1, 0.1, 0.1)), (ns * nin + nd * (1 + nout)) * es))
    for nm, f, bpc in rows:
        t = timed(f)
        print(f"{str(dt):14s} {nm:22s} {t * 1e3:8.3f} ms  {bpc * n / t / 1e9:8.1f} GB/s algorithmic ({n} columns, {ns} -> {nd} levels)", flush=True)
    del vls, ss, sds, vld, sd, sdd, posn
