"""The cfg5 pipeline (ezhip_ezsint_pack16_compress_batch_dev, 32 full-size fields) under development switches, INTERLEAVED on one device: us per field (wall clock incl.
the one synchronisation) and the records compared with the first variant's.  python tools/sweep_cfg5.py "VAR=val" "VAR=val,VAR2=val" ...  ("" = defaults)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from librmn_amd import ezscint as ez, packers as pk
import ezcases as ec
F = 32
ni, nj, no, mo = 4400, 2200, 7200, 3601
gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
ez.ezdefset(gdout, gdin)
stream = torch.cuda.current_stream(); ez.use_stream(stream.cuda_stream)
base = torch.from_numpy(ec.synth_field(ni, nj, seed=1000)).cuda()
gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
d_in = torch.empty((F, ni * nj), dtype=torch.float32, device="cuda")
for f in range(F):
    d_in[f] = base * (1.0 + 1e-3 * (torch.rand(ni * nj, device="cuda", generator=gen) - 0.5)) + 0.01 * f
n = no * mo; rs = 4 + n // 2 + 16
variants = sys.argv[1:] or [""]
recs = {v: torch.zeros((F, rs), dtype=torch.int32, device="cuda") for v in variants}
res = {v: [] for v in variants}; zls = {}
touched = set()
def setenv(v):
    for k in touched: os.environ.pop(k, None)
    for kv in filter(None, v.split(",")):
        k, val = kv.split("="); os.environ[k] = val; touched.add(k)
for rnd in range(6):
    for v in variants:
        setenv(v)
        for _ in range(2 if rnd == 0 else 1):
            rc, zl = pk.ezsint_pack16_compress_batch_dev(recs[v], rs, d_in, F, no, mo, 16); assert rc == 0, rc
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rc, zl = pk.ezsint_pack16_compress_batch_dev(recs[v], rs, d_in, F, no, mo, 16)
            ts.append((time.perf_counter() - t0) * 1e6 / F)
        zls[v] = list(zl)
        if rnd: res[v].append(min(ts))
v0 = variants[0]
for v in variants:
    same = zls[v] == zls[v0] and all(bool(torch.equal(recs[v][f][:4 + (int(zls[v][f]) + 3) // 4], recs[v0][f][:4 + (int(zls[v0][f]) + 3) // 4])) for f in range(F))
    r = sorted(res[v]); print(f"[{v or 'defaults'}] median {r[len(r) // 2]:.2f} min {r[0]:.2f} max {r[-1]:.2f} us per field; records equal to the first variant's: {same}", flush=True)
