"""cfg5 pipeline: passes B + E in one launch (k_sepx_enc, EZHIP_CFG5_FUSED=1) against the two-kernel form (default); records compared.
   python tools/probe_cfg5_fused.py [F]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from librmn_amd import ezscint as ez, packers as pk
import ezcases as ec

F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
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
ref = None
for variant in ("", "EZHIP_CFG5_FUSED=1", "", "EZHIP_CFG5_FUSED=1"):
    os.environ.pop("EZHIP_CFG5_FUSED", None)
    if variant: os.environ["EZHIP_CFG5_FUSED"] = "1"
    recs = torch.zeros((F, rs), dtype=torch.int32, device="cuda")
    for _ in range(3):
        rc, zl = pk.ezsint_pack16_compress_batch_dev(recs, rs, d_in, F, no, mo, 16)
        assert rc == 0, rc
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc, zl = pk.ezsint_pack16_compress_batch_dev(recs, rs, d_in, F, no, mo, 16)
        ts.append((time.perf_counter() - t0) * 1e6 / F)
    same = None
    if ref is None:
        ref = (list(zl), recs.clone())
    else:
        same = list(zl) == ref[0] and all(bool(torch.equal(recs[f][:4 + (int(zl[f]) + 3) // 4], ref[1][f][:4 + (int(zl[f]) + 3) // 4])) for f in range(F))
    print(f"[{variant or 'two kernels'}]: {min(ts):.1f} us per field (min of 6; median {sorted(ts)[3]:.1f}), zlng[0]={zl[0]}, records equal to the first variant: {same}", flush=True)
