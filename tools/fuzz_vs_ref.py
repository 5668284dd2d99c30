"""c_ezsint / c_ezuvint on random grid pairs, the product on the GPU against the reference build (oracle/_ref/libezref.so) directly:
nearest and bilinear bit for bit, bicubic within 1e-5 (the separable kernel is <= 1 ulp), winds within 1e-5 |V|.
python tools/fuzz_vs_ref.py [ncases] [seed]"""
import ctypes, os, sys
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np
import reflib, ezcases as ec
from librmn_amd import ezscint as ez
R = reflib.ref()
fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DEG = {0: "nearest", 1: "linear", 3: "cubic"}
bad = ran = 0
cub_cases = cub_same = wind_cases = wind_same = 0
for k in range(ncases):
    st = str(rng.choice(["G", "A", "B", "L", "Lreg", "Ghem", "Ahem"]))
    nj = int(rng.integers(12, 80)) * 2; ni = 2 * nj if st in ("G", "Ghem") else int(rng.integers(30, 220))
    ig = (0, 0, 0, 0); t = st[0]
    if st == "B": ni += 1 - ni % 2; nj += 1
    if st == "L": ig = (int(round(18000 / nj)), int(round(36000 / ni)), int(round(9000 / nj)), 0)            # global, cells centred
    if st == "Lreg": ig = (50, 50, 6000, 20000)
    if st in ("Ghem", "Ahem"): ig = (2, int(rng.integers(0, 2)), 0, 0); nj //= 2; nj = max(nj, 8)
    no = int(rng.integers(10, 150)); mo = int(rng.integers(6, 90))
    if st == "Lreg":
        dig = (int(rng.integers(40, 120)), int(rng.integers(40, 120)), 6000 - 300, 20000 - 300)              # reaches beyond the box: extrapolation
    else:
        dlat = 18000 // (mo - 1)                         # pole to pole when the spacing divides 180 degrees, else 85 S .. at most 85 N
        if dlat * (mo - 1) != 18000 or rng.random() < 0.5: dlat = 17000 // mo
        dig = (int(dlat), int(36000 // no), 0 if dlat * (mo - 1) == 18000 else 500, 0)
    polar = int(rng.integers(0, 2)); degree = int(rng.choice([0, 1, 3])); extrap = str(rng.choice(["maximum", "value", "nearest", "linear"]))
    zin = ec.synth_field(ni, nj, seed=int(rng.integers(1 << 30)))
    uu, vv = ec.synth_wind(ni, nj, seed=int(rng.integers(1 << 30)))
    if st == "B":
        for a in (zin, uu, vv):
            z2 = a.reshape(nj, ni); z2[:, -1] = z2[:, 0]
    gr_in = R.c_ezqkdef(ni, nj, t.encode(), *ig, 0); gr_out = R.c_ezqkdef(no, mo, b"L", *dig, 0)
    if gr_in < 0 or gr_out < 0: continue
    for name, val in (("interp_degree", DEG[degree]), ("polar_correction", "yes" if polar else "no"), ("extrap_degree", extrap)):
        R.c_ezsetopt(name.encode(), val.encode())
    R.c_ezsetval(b"extrap_value", ctypes.c_float(-3.5))
    if R.c_ezdefset(gr_out, gr_in) != 1: continue
    zr = np.zeros(no * mo, np.float32); ur = zr.copy(); vr = zr.copy()
    rc_r = R.c_ezsint(fp(zr), fp(zin))
    winds = st not in ("Ghem", "Ahem")
    rcv_r = R.c_ezuvint(fp(ur), fp(vr), fp(uu), fp(vv)) if winds else 0
    gdin = ez.ezqkdef(ni, nj, t, *ig); gdout = ez.ezqkdef(no, mo, "L", *dig)
    if gdin < 0 or gdout < 0: print("product refuses", st, ig); bad += 1; continue
    ez.ezdefset(gdout, gdin)
    ez.ezsetopt("interp_degree", DEG[degree]); ez.ezsetopt("polar_correction", "yes" if polar else "no"); ez.ezsetopt("extrap_degree", extrap); ez.ezsetval("extrap_value", -3.5)
    rc, z = ez.ezsint(zin, no * mo)
    ran += 1
    ok = rc == rc_r
    if ok:
        if degree == 3: ok = bool(np.all(np.abs(z - zr) <= 1e-5 * np.maximum(np.abs(zr), 1e-30)))
        else: ok = np.array_equal(z.view(np.uint32), zr.view(np.uint32))
    if ok and degree == 3:
        cub_cases += 1; same_z = np.array_equal(z.view(np.uint32), zr.view(np.uint32)); cub_same += int(same_z)
        if not same_z and os.environ.get("FUZZ_VERBOSE"):
            d = np.flatnonzero(z.view(np.uint32) != zr.view(np.uint32)); rows = np.unique(d // no)
            print(f"CUBIC not bit-identical: {d.size} of {no * mo} points, rows {rows[:12].tolist()}{'...' if rows.size > 12 else ''} of {mo}, max rel {float(np.max(np.abs(z[d] - zr[d]) / np.maximum(np.abs(zr[d]), 1e-30))):.2e}",
                  dict(src=(ni, nj, t, ig), dst=(no, mo, dig), polar=polar, extrap=extrap), flush=True)
    if ok and winds:
        if os.environ.get("FUZZ_WIND_EXACT"): ez.set_wind_exact(1)
        rcv, u, v = ez.ezuvint(uu, vv, no * mo)
        ez.set_wind_exact(0)
        wind_cases += 1; wind_same += int(rcv == rcv_r and np.array_equal(u.view(np.uint32), ur.view(np.uint32)) and np.array_equal(v.view(np.uint32), vr.view(np.uint32)))
        sc = np.maximum(np.hypot(ur.astype(np.float64), vr.astype(np.float64)), 1e-3)
        ok = rcv == rcv_r and bool(np.all(np.abs(u - ur) <= 1e-5 * sc) and np.all(np.abs(v - vr) <= 1e-5 * sc))
    if not ok:
        bad += 1
        print("MISMATCH", dict(src=(ni, nj, t, ig), dst=(no, mo, dig), degree=degree, polar=polar, extrap=extrap, rc=(rc, rc_r)), flush=True)
for name, val in (("interp_degree", "cubic"), ("polar_correction", "yes"), ("extrap_degree", "maximum")):
    R.c_ezsetopt(name.encode(), val.encode())
print(f"product vs reference build: {ran} of {ncases} cases ran, {bad} mismatches")
print(f"  bit for bit: bicubic scalars {cub_same} of {cub_cases} cases; winds{' (exact mode)' if os.environ.get('FUZZ_WIND_EXACT') else ''} {wind_same} of {wind_cases} cases")
sys.exit(1 if bad else 0)
