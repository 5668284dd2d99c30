"""fused cfg5 batch pipeline under EZHIP_ENC_DEBUG knock-outs (timing only; knock-outs produce wrong streams)"""
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
n = no * mo
rs = 4 + n // 2 + 16
recs = torch.zeros((F, rs), dtype=torch.int32, device="cuda")
for dbg in sys.argv[1:] or ["0"]:
    os.environ["EZHIP_ENC_DEBUG"] = dbg
    ts = []
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, zl = pk.ezsint_pack16_compress_batch_dev(recs, rs, d_in, F, no, mo, 16)
        ts.append((time.perf_counter() - t0) * 1e6 / F)
    print(f"EZHIP_ENC_DEBUG={dbg}: fused pipeline {min(ts):.1f} us per field (wall), rc={rc} zlng[0]={zl[0]}", flush=True)
