"""Generates tests/golden/cfg1_full_golden.npz: the reference's own c_ezsint (oracle/_ref/libezref.so)
at BASELINE cfg1's exact size: 'L' 400x200 (lat0 -89.55, lon0 0, 0.9 deg: ig = 90, 90, 45, 0) ->
'L' 800x400 (lat0 -90, lon0 0, 0.45 deg: ig = 45, 45, 0, 0), bilinear, input
tests/ezcases.synth_field(400, 200, seed=1).  The bilinear outputs (polar correction on / off) are stored
whole; nearest and bicubic as float64 sum + bit hash + sampled rows.

    python tests/golden/make_cfg1_full.py
"""
import os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from reflib import ref, fptr          # noqa: E402
import ezcases as ec                  # noqa: E402

NI, NJ, NO, MO = 400, 200, 800, 400
ROWS = np.array([0, 1, 2, 3, 100, 199, 200, 396, 397, 398, 399])
DEG = {0: b"nearest", 1: b"linear", 3: b"cubic"}


def bit_hash(z):
    u = np.ascontiguousarray(z).reshape(-1).view(np.uint32)
    return np.array([int(u.astype(np.uint64).sum()) & 0xFFFFFFFF, int(np.bitwise_xor.reduce(u))], np.uint32)


def main():
    L = ref()
    gdin = L.c_ezqkdef(NI, NJ, b"L", 90, 90, 45, 0, 0)
    gdout = L.c_ezqkdef(NO, MO, b"L", 45, 45, 0, 0, 0)
    assert L.c_ezdefset(gdout, gdin) == 1
    zin = ec.synth_field(NI, NJ, seed=1)
    out = {"rows": ROWS}
    z = np.zeros((MO, NO), np.float32)
    for degree in (1, 3, 0):
        for polar in (1, 0):
            L.c_ezsetopt(b"interp_degree", DEG[degree])
            L.c_ezsetopt(b"polar_correction", b"yes" if polar else b"no")
            rc = L.c_ezsint(fptr(z), fptr(zin))
            key = f"d{degree}_p{polar}"
            out[key + "/rc"] = np.int32(rc)
            out[key + "/sum"] = np.float64(z.astype(np.float64).sum())
            out[key + "/hash"] = bit_hash(z)
            out[key + "/rows"] = z[ROWS].copy()
            if degree == 1:
                out[key + "/z"] = z.copy()
            print(key, rc, out[key + "/sum"], z.ravel()[0], z.ravel()[-1], flush=True)
    L.c_ezsetopt(b"interp_degree", b"cubic"); L.c_ezsetopt(b"polar_correction", b"yes")
    np.savez_compressed(os.path.join(HERE, "cfg1_full_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
