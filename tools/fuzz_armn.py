"""randomized armn_compress shapes against the CPU oracle (development tool)"""
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
    if not ok:
        bad += 1; print("MISMATCH", ni, nj, nbits, kind, zg, zw, flush=True)
print("armn cases", n, "bad", bad)
