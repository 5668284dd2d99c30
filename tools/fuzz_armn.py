"""randomized armn_compress shapes against the CPU oracle, COMPRESS and UNCOMPRESS (development tool)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import packcases as pc
import test_oracle_packers as top
from librmn_amd import packers as pk
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = n = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    ni = int(rng.integers(16, 1500)); nj = int(rng.integers(16, 400)); nbits = int(rng.integers(5, 17))
    kind = ["smooth", "noisy", "constant", "bigdiff"][int(rng.integers(0, 4))]
    try:
        tok = pc.token_field(ni, nj, nbits, kind, seed=it)
    except Exception:
        kind = "smooth"; tok = pc.token_field(ni, nj, nbits, kind, seed=it)
    words = pc.tokens_to_words(tok)
    bw = np.zeros(words.size + 8, np.uint32); bw[:words.size] = words
    bg = bw.copy()
    zw = top.O().orc_armn_compress(bw.ctypes.data, ni, nj, 1, nbits, 1)
    zg = pk.armn_compress(bg, ni, nj, nbits)
    n += 1
    ok = zg == zw and (np.array_equal(bg[:(zw - 1) // 4], bw[:(zw - 1) // 4]) if zw > 0 else np.array_equal(bg[:words.size], words))
    if ok and zw > 0:          # decode the stream in place on the device: the original token words must come back
        dec = np.zeros(1 + ni * nj // 2 + 2, np.uint32); dec[:min(dec.size, bg.size)] = bg[:min(dec.size, bg.size)]
        ok = pk.armn_uncompress(dec, ni, nj, nbits) == ni * nj * 2 and np.array_equal(dec[:words.size], words)
    if ok:                     # and a raw oracle stream (compressible or not) through the device decoder
        import torch
        top.O().orc_armn_compress_setlevel(int(rng.integers(0, 2)))
        z = np.zeros(ni * nj + 64, np.uint32)
        zl = top.O().orc_armn_encode(z.ctypes.data, tok.ctypes.data, ni, nj, nbits)
        top.O().orc_armn_compress_setlevel(-1)
        zwds = (zl - 1) // 4 + 1
        d_z = torch.from_numpy(z[:zwds].view(np.int32).copy()).cuda(); d_o = torch.zeros(1 + ni * nj // 2, dtype=torch.int32, device="cuda")
        ok = pk.armn_uncompress_dev(d_o, d_z, zwds, ni, nj, nbits) == ni * nj * 2 and np.array_equal(d_o.cpu().numpy().view(np.uint32)[:words.size], words)
    if not ok:
        bad += 1; print("MISMATCH", ni, nj, nbits, kind, zg, zw, flush=True)
print("armn cases", n, "bad", bad)
