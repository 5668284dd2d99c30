"""Generates tests/golden/cfg3_full_golden.npz: the reference's own c_ezuvint / c_ezsint / c_gdxyfll
(oracle/_ref/libezref.so) run at BASELINE cfg3's FULL size (Z-on-E 2560x1280 rotated global grid ->
L 4000x2000), sampled.  Runs only in the build container (needs
`ulimit -s unlimited`: the reference keeps 24 bytes per point of automatic arrays on the stack, appendix D.2).

Inputs: tests/ezcases.synth_wind(2560, 1280, seed=3) with column ni = column 1 (the grid's duplicate
column).  Per (degree, polar): rows ROWS (all columns) and columns COLS (all rows) of uuout / vvout
and of the scalar c_ezsint of the u field, float64 sums and whole-field bit hashes (uint32 sum and
xor of the float bit patterns of all 8 M points).  Also the located x,y of the set (c_gdxyfll on
c_gdll of the target) on the same rows / columns + whole-array bit hashes: the direct check of the
rotated-source locate (ez_gfxyfll.c:38-57).

    bash -c 'ulimit -s unlimited; python tests/golden/make_cfg3_full.py'
"""
import os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from reflib import ref, fptr          # noqa: E402
import ezcases as ec                  # noqa: E402

NI, NJ, NO, MO = 2560, 1280, 4000, 2000
ROWS = np.array([0, 1, 2, 3, 4, 5, 17, 500, 999, 1000, 1500, 1994, 1995, 1996, 1997, 1998, 1999])
COLS = np.array([0, 1, 2, 3, 100, 1999, 2000, 3996, 3997, 3998, 3999])
DEG = {0: b"nearest", 1: b"linear", 3: b"cubic"}


def bit_hash(z):
    u = np.ascontiguousarray(z).reshape(-1).view(np.uint32)
    return np.array([int(u.astype(np.uint64).sum()) & 0xFFFFFFFF, int(np.bitwise_xor.reduce(u))], np.uint32)


def inputs():
    uu, vv = ec.synth_wind(NI, NJ, seed=3)
    for a in (uu, vv):
        a2 = a.reshape(NJ, NI); a2[:, -1] = a2[:, 0]
    return uu, vv


def main():
    L = ref()
    ax, ay = ec.ze_axes(NI, NJ)
    gdin = L.c_ezgdef_fmem(NI, NJ, b"Z", b"E", *ec.E_IG, fptr(ax), fptr(ay))
    gdout = L.c_ezqkdef(NO, MO, b"L", 9, 9, 0, 0, 0)
    assert gdin >= 0 and gdout >= 0
    assert L.c_ezdefset(gdout, gdin) == 1
    uu, vv = inputs()
    out = {"rows": ROWS, "cols": COLS}
    lat = np.zeros((MO, NO), np.float32); lon = np.zeros((MO, NO), np.float32)
    assert L.c_gdll(gdout, fptr(lat), fptr(lon)) >= 0
    x = np.zeros((MO, NO), np.float32); y = np.zeros((MO, NO), np.float32)
    assert L.c_gdxyfll(gdin, fptr(x), fptr(y), fptr(lat), fptr(lon), NO * MO) >= 0
    for nm, a in (("x", x), ("y", y)):
        out[nm + "/rows"] = a[ROWS].copy(); out[nm + "/cols"] = a[:, COLS].copy(); out[nm + "/hash"] = bit_hash(a)
    u = np.zeros((MO, NO), np.float32); v = np.zeros((MO, NO), np.float32); z = np.zeros((MO, NO), np.float32)
    for degree in (3, 1, 0):
        for polar in (1, 0):
            L.c_ezsetopt(b"interp_degree", DEG[degree])
            L.c_ezsetopt(b"polar_correction", b"yes" if polar else b"no")
            assert L.c_ezuvint(fptr(u), fptr(v), fptr(uu), fptr(vv)) >= 0
            assert L.c_ezsint(fptr(z), fptr(uu)) >= 0
            key = f"d{degree}_p{polar}"
            for nm, a in (("u", u), ("v", v), ("z", z)):
                out[f"{key}/{nm}/rows"] = a[ROWS].copy(); out[f"{key}/{nm}/cols"] = a[:, COLS].copy()
                out[f"{key}/{nm}/sum"] = np.float64(a.astype(np.float64).sum())
                out[f"{key}/{nm}/hash"] = bit_hash(a)
            print(key, out[f"{key}/u/sum"], out[f"{key}/v/sum"], out[f"{key}/z/sum"], flush=True)
    L.c_ezsetopt(b"interp_degree", b"cubic"); L.c_ezsetopt(b"polar_correction", b"yes")
    np.savez_compressed(os.path.join(HERE, "cfg3_full_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
