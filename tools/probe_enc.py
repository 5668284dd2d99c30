"""one-pass encoder alone on 32 full-size token fields (event-timed); EZHIP_ENC_DEBUG knock-outs give wrong streams (timing only)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, ctypes
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
out = torch.empty((F, n), dtype=torch.float32, device="cuda")
assert pk.ezsint_pack16_batch_dev(recs, rs, out, d_in, F, n, 16) == 0
del out
for dbg in sys.argv[1:] or ["0"]:
    os.environ["EZHIP_ENC_DEBUG"] = dbg
    r2 = recs.clone()
    ts = []
    for it in range(4):
        r2.copy_(recs); torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc, zl = pk.pack16_compress_batch_dev(r2, rs, None, 0, F, no, mo, 16, prepacked=1)
        ts.append((time.perf_counter() - t0) * 1e6 / F)
    print(f"EZHIP_ENC_DEBUG={dbg}: per-field launches (memset + k_armn_enc1 + k_cond_copy): {min(ts):.1f} us per field (wall), zlng[0]={zl[0]}")

os.environ["EZHIP_ENC_DEBUG"] = "16"
import ctypes
from librmn_amd.lib import load_library
L = load_library()
for mode in ("per-field launches", "one fused batch launch"):
    if mode == "per-field launches":
        r2 = recs.clone(); pk.pack16_compress_batch_dev(r2, rs, None, 0, F, no, mo, 16, prepacked=1)
    else:
        pk.ezsint_pack16_compress_batch_dev(r2, rs, d_in, F, no, mo, 16)
    out = (ctypes.c_ulonglong * 8)()
    if L.packhip_armn_enc1_profile(out) == 0 and out[0]:
        nb = out[0]
        print(mode, ": chunk blocks in the last launch:", nb, " mean cycles of thread 0 per phase [staging, tiles+scan, emission, look-back, copy-out]:",
              [round(out[k] / nb) for k in range(1, 6)], " look-back windows per chunk: %.2f, polls of a missing granule per chunk: %.2f" % (out[6] / nb, out[7] / nb))
