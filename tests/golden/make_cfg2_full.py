"""Generates tests/golden/cfg2_full_golden.npz: the reference's own c_ezsint (oracle/_ref/libezref.so)
run at BASELINE cfg2's FULL size (G 4400x2200 -> L 7200x3601), sampled.  Runs only in the build
container (needs `ulimit -s unlimited`: the reference keeps full-size temporaries on the stack).

Two inputs:
  probe  -- the survey's drv2 field 280 + 20 sinf(.005 i) cosf(.007 j).  The
            libm vectors sinf(.005f*i), cosf(.007f*j) are stored so the field is rebuilt libm-free;
            the reference's output reproduces the survey's printed anchors (sum 7.26056487e+09,
            z[0] 281.818, z[mid] 276.863, z[last] 278.271), asserted below.
  synth  -- tests/ezcases.synth_field(seed=2).
Outputs per (input, degree, polar): the rows ROWS (all columns) and columns COLS (all rows) of the
reference's output, its float64 sum, and for polar_correction=no a bit hash (uint32 sum and xor of
the float bit patterns) of all 25.9 M points -- usable where the HIP path is bit-exact (nearest, linear).

    ulimit -s unlimited; python tests/golden/make_cfg2_full.py
"""
import ctypes, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from reflib import ref, fptr          # noqa: E402
import ezcases as ec                  # noqa: E402

NI, NJ, NO, MO = 4400, 2200, 7200, 3601
ROWS = np.array([0, 1, 2, 3, 4, 17, 900, 1800, 1801, 2700, 3596, 3597, 3598, 3599, 3600])
COLS = np.array([0, 1, 2, 3, 100, 3599, 3600, 7196, 7197, 7198, 7199])
DEG = {0: b"nearest", 1: b"linear", 3: b"cubic"}


def probe_vectors():
    libm = ctypes.CDLL("libm.so.6")
    libm.sinf.restype = ctypes.c_float; libm.sinf.argtypes = [ctypes.c_float]
    libm.cosf.restype = ctypes.c_float; libm.cosf.argtypes = [ctypes.c_float]
    f32 = np.float32
    s = np.array([libm.sinf(f32(f32(.005) * f32(i))) for i in range(NI)], np.float32)
    c = np.array([libm.cosf(f32(f32(.007) * f32(j))) for j in range(NJ)], np.float32)
    return s, c


def probe_field(s, c):
    return np.ascontiguousarray((np.float32(280) + (np.float32(20) * s)[None, :] * c[:, None]).astype(np.float32))


def bit_hash(z):
    u = z.view(np.uint32)
    return np.array([int(u.astype(np.uint64).sum()) & 0xFFFFFFFF, int(np.bitwise_xor.reduce(u))], np.uint32)


def main():
    L = ref()
    gdin = L.c_ezqkdef(NI, NJ, b"G", 0, 0, 0, 0, 0)
    gdout = L.c_ezqkdef(NO, MO, b"L", 5, 5, 0, 0, 0)
    assert L.c_ezdefset(gdout, gdin) == 1
    s, c = probe_vectors()
    out = {"probe_sin": s, "probe_cos": c, "rows": ROWS, "cols": COLS}
    fields = {"probe": probe_field(s, c), "synth": ec.synth_field(NI, NJ, seed=2)}
    z = np.zeros((MO, NO), np.float32)
    for fname, zin in fields.items():
        for degree in (3, 1, 0):
            for polar in (1, 0):
                L.c_ezsetopt(b"interp_degree", DEG[degree])
                L.c_ezsetopt(b"polar_correction", b"yes" if polar else b"no")
                assert L.c_ezsint(fptr(z), fptr(zin)) == 0
                key = f"{fname}/d{degree}_p{polar}"
                out[key + "/rows"] = z[ROWS].copy(); out[key + "/cols"] = z[:, COLS].copy()
                out[key + "/sum"] = np.float64(z.astype(np.float64).sum())
                if not polar:
                    out[key + "/hash"] = bit_hash(z.ravel())
                if fname == "probe" and degree == 3 and polar:
                    f = z.ravel()
                    got = "%.8e %.3f %.3f %.3f" % (out[key + "/sum"], f[0], f[f.size // 2], f[-1])
                    assert got == "7.26056487e+09 281.818 276.863 278.271", got
                print(key, out[key + "/sum"], flush=True)
    L.c_ezsetopt(b"interp_degree", b"cubic"); L.c_ezsetopt(b"polar_correction", b"yes")
    np.savez_compressed(os.path.join(HERE, "cfg2_full_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
