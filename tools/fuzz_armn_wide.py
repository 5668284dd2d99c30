"""armn_compress UNCOMPRESS on WIDE random shapes (rows of 800 - 2700 tiles: the parallel form's default territory) against the oracle's tokens; oracle streams of
both methods.  python tools/fuzz_armn_wide.py [seed] [ncases]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import packcases as pc
import test_oracle_packers as top
from librmn_amd import packers as pk
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = n = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    ni = int(rng.integers(2400, 8100)); nj = int(rng.integers(16, 160)); nbits = int(rng.choice([16, 16, 12, 9]))
    kind = ["smooth", "noisy", "constant", "bigdiff"][int(rng.integers(0, 4))]
    level = int(rng.integers(0, 2))
    tok = pc.token_field(ni, nj, nbits, kind, seed=it)
    O = top.O(); O.orc_armn_compress_setlevel(level)
    z = np.zeros(ni * nj + 64, np.uint32)
    zl = O.orc_armn_encode(z.ctypes.data, tok.ctypes.data, ni, nj, nbits); O.orc_armn_compress_setlevel(-1)
    zw = (zl - 1) // 4 + 1
    slack = int(rng.integers(0, 40))                                   # the caller's word count may exceed the stream (garbage behind it)
    zz = z[:zw + slack].copy(); zz[zw:] = rng.integers(0, 1 << 32, slack, dtype=np.uint64).astype(np.uint32)
    d_z = torch.from_numpy(zz.view(np.int32).copy()).cuda(); d_o = torch.full((1 + ni * nj // 2 + 2,), -1, dtype=torch.int32, device="cuda")
    words = pc.tokens_to_words(tok)
    ok = pk.armn_uncompress_dev(d_o, d_z, zw + slack, ni, nj, nbits) == ni * nj * 2 and np.array_equal(d_o.cpu().numpy().view(np.uint32)[:words.size], words)
    n += 1
    if not ok:
        bad += 1; print("MISMATCH", ni, nj, nbits, kind, level, slack, flush=True)
print("armn wide cases", n, "bad", bad)
sys.exit(1 if bad else 0)
