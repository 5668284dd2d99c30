"""k_sepx_enc knock-outs (EZHIP_ENC_DEBUG): kernel time from hip events around the fused call's device work is not separable from the host side, so this
   only RUNS the variants (results are wrong under a knock-out); time them with tools/prof_cmd.sh.  python tools/probe_sepenc_knock.py [F]"""
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
recs = torch.zeros((F, rs), dtype=torch.int32, device="cuda")
for dbg in [int(a) for a in sys.argv[2:]] or [0, 32, 64, 1, 2, 8, 11, 75]:
    os.environ["EZHIP_ENC_DEBUG"] = str(dbg)
    os.environ["EZHIP_CFG5_ONLY_FUSED"] = "1"; os.environ["EZHIP_CFG5_FUSED"] = "1"
    for _ in range(2):
        rc, zl = pk.ezsint_pack16_compress_batch_dev(recs, rs, d_in, F, no, mo, 16)
    torch.cuda.synchronize()
    # marker launch so that the variants can be told apart in the trace: dbg + 1 tiny fills
    for _ in range(dbg + 1):
        recs[0, :1].fill_(0)
    torch.cuda.synchronize()
    print("debug", dbg, "rc", rc, flush=True)
