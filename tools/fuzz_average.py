"""interp_degree = average / sph_average on random grid pairs: the product against the reference build (oracle/_ref/libezref.so), bit for bit.
python tools/fuzz_average.py [ncases] [seed]"""
import ctypes, os, sys
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np
import reflib, ezcases as ec
from librmn_amd import ezscint as ez
R = reflib.ref()
fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0; ran = 0
for k in range(ncases):
    st = str(rng.choice(["G", "A", "B", "L"]))
    nj = int(rng.integers(20, 90)) * 2; ni = 2 * nj if st == "G" else int(rng.integers(40, 260))
    if st == "B": ni += 1 - ni % 2; nj += 1
    if st == "L":                                     # a regional box, the target well inside it
        src = (ni, nj, "L", (50, 50, 6000, 20000))
        no = int(rng.integers(4, 20)); mo = int(rng.integers(4, 14))
        dlat = int(rng.integers(80, 250)); dlon = int(rng.integers(80, 250))
        span_lat = nj * 0.5 - 8; span_lon = ni * 0.5 - 8
        mo = max(3, min(mo, int(span_lat * 100 / dlat) - 1)); no = max(3, min(no, int(span_lon * 100 / dlon) - 1))
        dst = (no, mo, "L", (dlat, dlon, 6000 + 400, 20000 + 400))
    else:
        src = (ni, nj, st, (0, 0, 0, 0))
        mo = int(rng.integers(5, 30)); no = int(rng.integers(8, 60))
        lat0 = int(rng.integers(1500, 4000)); dlat = max(50, (16500 - lat0 - 1500) // mo); dlon = 36000 // no
        dst = (no, mo, "L", (int(dlat), int(dlon), lat0, 0))             # rows between about 75 S and 75 N
    degree = "average" if rng.random() < 0.5 else "sph_average"
    zin = ec.synth_field(ni, nj, seed=int(rng.integers(1 << 30)))
    if st == "B":
        z2 = zin.reshape(nj, ni); z2[:, -1] = z2[:, 0]; zin = np.ascontiguousarray(z2.reshape(-1))
    gr_in = R.c_ezqkdef(ni, nj, src[2].encode(), *src[3], 0); gr_out = R.c_ezqkdef(dst[0], dst[1], b"L", *dst[3], 0)
    R.c_ezsetopt(b"interp_degree", degree.encode()); R.c_ezsetopt(b"polar_correction", b"no")
    if R.c_ezdefset(gr_out, gr_in) != 1: continue
    zr = np.zeros(dst[0] * dst[1], np.float32)
    rc_r = R.c_ezsint(fp(zr), fp(zin))
    R.c_ezsetopt(b"interp_degree", b"cubic"); R.c_ezsetopt(b"polar_correction", b"yes")
    gdin = ez.ezqkdef(ni, nj, src[2], *src[3]); gdout = ez.ezqkdef(dst[0], dst[1], "L", *dst[3])
    ez.ezdefset(gdout, gdin)
    ez.ezsetopt("interp_degree", degree); ez.ezsetopt("polar_correction", "no")
    rc, z = ez.ezsint(zin, dst[0] * dst[1])
    ran += 1
    ez.ezsetopt("interp_degree", "cubic"); ez.ezsetopt("polar_correction", "yes")
    if rc != rc_r or not np.array_equal(z.view(np.uint32), zr.view(np.uint32)):
        bad += 1
        print("MISMATCH", degree, src, dst, rc, rc_r, int(np.count_nonzero(z != zr)), flush=True)
print(f"averaging degrees: {ran} of {ncases} cases ran, {bad} mismatches")
sys.exit(1 if bad else 0)
