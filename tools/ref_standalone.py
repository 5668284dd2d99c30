"""The reference build (oracle/_ref/libezref.so) on ONE Yin-Yang case of tools/fuzz_vs_ref4.py in a process of its own: the reference library's answers
depend on what the process did before (grid table, cached sets, option side effects of c_ezyymint), so a disagreement seen in a long fuzzing session is
checked against a fresh process before it counts.       python tools/ref_standalone.py case.npz out.npz"""
import ctypes, os, sys
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np
import reflib, ezcases as ec
R = reflib.ref()
fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
d = np.load(sys.argv[1], allow_pickle=True)
DEG = {0: "nearest", 1: "linear", 3: "cubic"}
mode = str(d["mode"]); uni, unj = int(d["uni"]), int(d["unj"]); uax, uay = np.ascontiguousarray(d["uax"]), np.ascontiguousarray(d["uay"])
ni, nj, t, ig, rf = tuple(d["other"]); oax, oay = np.ascontiguousarray(d["oax"]), np.ascontiguousarray(d["oay"])
for name, val in (("interp_degree", DEG[int(d["degree"])]), ("polar_correction", "yes" if int(d["polar"]) else "no"), ("extrap_degree", str(d["extrap"])), ("cloud_interp_alg", "distance")):
    R.c_ezsetopt(name.encode(), val.encode())
R.c_ezsetval(b"extrap_value", ctypes.c_float(-3.5))
a = R.c_ezgdef_fmem(uni, unj, b"Z", b"E", *ec.YIN_IG, fp(uax), fp(uay)); b = R.c_ezgdef_fmem(uni, unj, b"Z", b"E", *ec.YAN_IG, fp(uax), fp(uay))
gu = R.c_ezgdef_supergrid(uni, 2 * unj, b"U", b"F", 1, 2, (ctypes.c_int * 2)(a, b))
go = R.c_ezqkdef(int(ni), int(nj), t.encode(), *[int(v) for v in ig], 0) if oax.size == 0 else R.c_ezgdef_fmem(int(ni), int(nj), t.encode(), rf.encode(), *[int(v) for v in ig], fp(oax), fp(oay))
gi, gd = (gu, go) if mode == "yy_src" else (go, gu)
n_out = int(ni) * int(nj) if mode == "yy_src" else 2 * uni * unj
assert R.c_ezdefset(gd, gi) == 1
zin, uu, vv = (np.ascontiguousarray(d[k]) for k in ("zin", "uu", "vv"))
zr = np.zeros(n_out, np.float32); ur = zr.copy(); vr = zr.copy()
rc = R.c_ezsint(fp(zr), fp(zin)); rcv = R.c_ezuvint(fp(ur), fp(vr), fp(uu), fp(vv)) if mode == "yy_src" else 0
# the latitudes that condition the wind direction at the target points: the true one and the ones in the two Yin-Yang frames
lat_t = np.zeros(n_out, np.float32); lon_t = lat_t.copy(); R.c_gdll(gd, fp(lat_t), fp(lon_t))
worst = np.abs(lat_t.astype(np.float64))
for g in (a, b):
    xs = np.zeros(n_out, np.float32); ys = xs.copy(); R.c_gdxyfll(g, fp(xs), fp(ys), fp(lat_t), fp(lon_t.copy()), n_out)
    dy = float(uay[1] - uay[0]); worst = np.maximum(worst, np.minimum(np.abs(float(uay[0]) + (ys.astype(np.float64) - 1.0) * dy), 90.0))
np.savez(sys.argv[2], zr=zr, ur=ur, vr=vr, rc=rc, rcv=rcv, worst=worst)
