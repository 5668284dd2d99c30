"""Fourth fuzzer against the reference build (oracle/_ref/libezref.so): the mask entry points (c_ezsint_mask, c_ezget_mask_zones, c_ezsint_mdm,
c_ezuvint_mdm: ez_mask.c), c_ezwdint, and Yin-Yang 'U' grids as source and as target (c_ezyysint / c_ezyyuvint), on random grid pairs.
Masks bit for bit where the reference's unbounded reads of mask_in stay inside the array; fields as in tools/fuzz_vs_ref2.py.
python tools/fuzz_vs_ref4.py [ncases] [seed]"""
import ctypes, os, sys
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np
import reflib, ezcases as ec, oraclelib as ol
DRY = bool(os.environ.get('FUZZ_REF_ONLY'))       # reference side only (no GPU): checks that the generator stays inside what the reference survives
if not DRY:
    from librmn_amd import ezscint as ez
R = reflib.ref()
fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DEG = {0: "nearest", 1: "linear", 3: "cubic"}
f32 = lambda a: np.ascontiguousarray(np.asarray(a, np.float64).astype(np.float32))


def stretched(a0, a1, n):
    """n strictly increasing values from a0 to a1, spacing modulated by +-15 %"""
    w = 1.0 + 0.15 * np.sin(np.arange(n - 1) * float(rng.uniform(0.1, 0.9)) + float(rng.uniform(0, 6)))
    x = np.concatenate([[0.0], np.cumsum(w)]); x = a0 + (a1 - a0) * x / x[-1]
    return f32(x)


def rot_ig():
    lat1 = float(rng.uniform(-60, 60)); lon1 = float(rng.uniform(0, 360))
    lat2 = float(np.clip(-0.4 * lat1 + rng.uniform(-20, 20), -80, 80)); lon2 = (lon1 + float(rng.uniform(60, 120))) % 360.0
    return ol.cxgaig("E", lat1, lon1, lat2, lon2)


def make_grid(kind):
    """-> (ni, nj, grtyp, ig, grref, ax, ay) ; ax/ay None for ezqkdef grids"""
    if kind == "ZEglob":            # GEM-like global rotated grid: last column repeats the first
        nj = int(rng.integers(16, 70)); ni = 2 * nj + 1
        ax = f32(np.arange(ni) * 360.0 / (ni - 1)); ay = f32(-90.0 + (np.arange(nj) + 0.5) * 180.0 / nj)
        return ni, nj, "Z", rot_ig(), "E", ax, ay
    if kind == "ZEreg":             # LAM in the rotated frame
        ni = int(rng.integers(20, 120)); nj = int(rng.integers(16, 90))
        x0 = float(rng.uniform(120, 170)); y0 = float(rng.uniform(-35, -5))
        return ni, nj, "Z", rot_ig(), "E", stretched(x0, x0 + float(rng.uniform(25, 70)), ni), stretched(y0, y0 + float(rng.uniform(20, 60)), nj)
    if kind == "ZLglob":            # irregular global lat-lon axes on the reference grid cxgaig('L', 0, 0, 1, 1)
        ni = int(rng.integers(24, 140)); nj = int(rng.integers(13, 70))
        ax = stretched(0.0, 360.0 * (ni - 1) / ni, ni); ay = stretched(-90.0, 90.0, nj) if rng.random() < 0.5 else stretched(-88.0, 88.5, nj)
        return ni, nj, "Z", (100, 100, 9000, 0), "L", ax, ay
    if kind == "ZLreg":
        ni = int(rng.integers(20, 120)); nj = int(rng.integers(16, 90))
        x0 = float(rng.uniform(10, 250)); y0 = float(rng.uniform(-70, 20))
        return ni, nj, "Z", (100, 100, 9000, 0), "L", stretched(x0, x0 + float(rng.uniform(20, 90)), ni), stretched(y0, y0 + float(rng.uniform(15, 60)), nj)
    if kind in ("N", "S"):
        ni = int(rng.integers(30, 130)); nj = int(rng.integers(30, 130))
        d60 = float(rng.uniform(60e3, 250e3)); dgrw = float(rng.uniform(0, 360))
        pi = float(rng.uniform(0.2, 0.8)) * ni + 0.5; pj = float(rng.uniform(0.2, 0.8)) * nj + 0.5
        return ni, nj, kind, ol.cxgaig(kind, pi, pj, d60, dgrw), " ", None, None
    if kind == "G":
        nj = int(rng.integers(12, 70)) * 2
        return 2 * nj, nj, "G", (0, 0, 0, 0), " ", None, None
    if kind == "A":
        nj = int(rng.integers(12, 70)) * 2
        return int(rng.integers(30, 200)), nj, "A", (0, 0, 0, 0), " ", None, None
    if kind == "Lglob":
        nj = int(rng.integers(12, 70)) * 2; ni = int(rng.integers(30, 200))
        return ni, nj, "L", (int(round(18000 / nj)), int(round(36000 / ni)), int(round(9000 / nj)), 0), " ", None, None
    if kind == "Lreg":
        ni = int(rng.integers(20, 140)); nj = int(rng.integers(16, 100))
        return ni, nj, "L", (int(rng.integers(20, 90)), int(rng.integers(20, 90)), int(rng.integers(2000, 11000)), int(rng.integers(0, 30000))), " ", None, None
    raise ValueError(kind)


def define(lib_is_ref, g):
    ni, nj, t, ig, ref, ax, ay = g
    if ax is None:
        return R.c_ezqkdef(ni, nj, t.encode(), *ig, 0) if lib_is_ref else ez.ezqkdef(ni, nj, t, *ig)
    return R.c_ezgdef_fmem(ni, nj, t.encode(), ref.encode(), *ig, fp(ax), fp(ay)) if lib_is_ref else ez.ezgdef_fmem(ni, nj, t, ref, *ig, ax, ay)



ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
SRC = ["ZEglob", "ZEreg", "ZLglob", "ZLreg", "N", "S", "G", "A", "Lglob", "Lreg"]
DST = ["Lglob", "Lreg", "ZLreg", "ZLglob", "G", "N", "S"]
bad = ran = near = dumped = crashed = 0


def report(what, desc, detail):
    global bad
    bad += 1
    print("MISMATCH", what, detail, desc, flush=True)


def mask_field(ni, nj):
    """source mask with holes: blobs of 2 x 2 .. 4 x 4, 10 - 50 % zeros"""
    b = int(rng.integers(2, 5)); h = rng.random(((nj + b - 1) // b, (ni + b - 1) // b))
    m = (np.repeat(np.repeat(h, b, axis=0), b, axis=1)[:nj, :ni] > float(rng.uniform(0.1, 0.5))).astype(np.int32)
    return np.ascontiguousarray(m.reshape(-1))


def yy_grid(ref_side):
    """a Yin-Yang grid: two Z-on-E subgrids (frames of tests/ezcases.py), random resolution"""
    # every case its own resolution: Yin-Yang subgrids shared between cases pile up sets on one target grid in the reference's tables, and the
    # reference build then corrupts its heap (c_ezsint + c_ezuvint to a 'U' target, from the ~8th source on: seen as "malloc(): corrupted top size")
    d = round(float(rng.uniform(2.0, 5.0)), 3); ov = float(rng.integers(2, 5)) * d
    ax = f32(np.arange(45.0 - ov, 315.0 + ov + 1e-6, d)); ay = f32(np.arange(-45.0 - ov, 45.0 + ov + 1e-6, d))
    return ax.size, ay.size, ax, ay


def define_u(ref_side, spec):
    ni, nj, ax, ay = spec
    if ref_side:
        a = R.c_ezgdef_fmem(ni, nj, b"Z", b"E", *ec.YIN_IG, fp(ax), fp(ay)); b = R.c_ezgdef_fmem(ni, nj, b"Z", b"E", *ec.YAN_IG, fp(ax), fp(ay))
        ids = (ctypes.c_int * 2)(a, b)
        return R.c_ezgdef_supergrid(ni, 2 * nj, b"U", b"F", 1, 2, ids)
    a = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.YIN_IG, ax, ay); b = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.YAN_IG, ax, ay)
    return ez.ezgdef_supergrid(ni, 2 * nj, "U", "F", 1, [a, b])


def setopts(degree, polar, extrap, alg="distance"):
    for name, val in (("interp_degree", DEG[degree]), ("polar_correction", "yes" if polar else "no"), ("extrap_degree", extrap), ("cloud_interp_alg", alg)):
        R.c_ezsetopt(name.encode(), val.encode())
        if not DRY: ez.ezsetopt(name, val)
    R.c_ezsetval(b"extrap_value", ctypes.c_float(-3.5))
    if not DRY: ez.ezsetval("extrap_value", -3.5)


def cmp_field(tag, desc, a, b, exact):
    fa, fb = np.isfinite(a), np.isfinite(b)
    if not np.array_equal(fa, fb): return report(tag, desc, "finite in one, not in the other")
    if exact:
        if not np.array_equal(np.where(a == 0, np.float32(0), a).view(np.uint32)[fa], np.where(b == 0, np.float32(0), b).view(np.uint32)[fa]):
            d = np.flatnonzero(fa & (a != b)); report(tag, desc, f"{d.size} points differ, first {int(d[0])}: {a[d[0]]!r} vs {b[d[0]]!r}")
    else:
        e = np.abs(a[fa] - b[fa]) / np.maximum(np.abs(b[fa]), 1e-30)
        if e.size and e.max() > 1e-5: report(tag, desc, f"max rel err {e.max():.2e}")


def cmp_wind(tag, desc, u, v, ur, vr, worst_lat):
    global near
    m = np.isfinite(ur) & np.isfinite(vr)
    if not np.array_equal(m, np.isfinite(u) & np.isfinite(v)): return report(tag, desc, "finite in one, not in the other")
    sc = np.maximum(np.hypot(ur[m].astype(np.float64), vr[m].astype(np.float64)), 1e-3)
    e = np.maximum(np.abs(u[m] - ur[m]), np.abs(v[m] - vr[m])) / sc
    allowed = 1e-5 + 8 * 6e-8 / np.maximum(np.cos(np.radians(np.minimum(worst_lat[m], 89.9999))), 1e-7)
    if e.size and not np.all(e <= allowed): w = int(np.argmax(e / allowed)); report(tag, desc, f"err {e[w]:.2e} |V| (allowed {allowed[w]:.2e})")
    elif e.size and not np.all(e <= 1e-5): near += 1


for k in range(ncases):
    mode = str(rng.choice(["mask", "mask", "wd", "yy_src", "yy_dst"]))
    if os.environ.get("FUZZ_MODES") and mode not in os.environ["FUZZ_MODES"].split(","): continue
    if os.environ.get('FUZZ_VERBOSE'): print(k, mode, flush=True)
    degree = int(rng.choice([0, 1, 3])); polar = int(rng.integers(0, 2)); extrap = str(rng.choice(["maximum", "value", "nearest", "linear"]))
    if mode in ("mask", "wd"):
        sk = str(rng.choice(SRC)); dk = str(rng.choice(DST))
        gs = make_grid(sk); gd = make_grid(dk)
        ni, nj = gs[0], gs[1]; no, mo = gd[0], gd[1]
        gr_in = define(True, gs); gr_out = define(True, gd)
        if gr_in < 0 or gr_out < 0 or gr_in == gr_out: continue
        zin = ec.synth_field(ni, nj, seed=int(rng.integers(1 << 30))); uu, vv = ec.synth_wind(ni, nj, seed=int(rng.integers(1 << 30)))
        if sk == "ZEglob":
            for a in (zin, uu, vv):
                z2 = a.reshape(nj, ni); z2[:, -1] = z2[:, 0]
        alg = str(rng.choice(["linear", "distance"]))
        setopts(degree, polar, extrap, alg)
        if R.c_ezdefset(gr_out, gr_in) != 1: continue
        desc = dict(mode=mode, src=(sk,) + gs[:5], dst=(dk,) + gd[:5], degree=degree, polar=polar, extrap=extrap, alg=alg)
        n = no * mo
        lat_t = np.zeros(n, np.float32); lon_t = lat_t.copy(); R.c_gdll(gr_out, fp(lat_t), fp(lon_t))
        worst = np.abs(lat_t.astype(np.float64))
        xs = np.zeros(n, np.float32); ys = xs.copy(); R.c_gdxyfll(gr_in, fp(xs), fp(ys), fp(lat_t), fp(lon_t.copy()), n)
        if gs[4] == "E": worst = np.maximum(worst, np.abs(np.interp(ys.astype(np.float64), np.arange(1, nj + 1), gs[6].astype(np.float64), left=90.0, right=90.0)))
        if mode == "wd":
            sr = np.zeros(n, np.float32); dr = sr.copy(); rc_r = R.c_ezwdint(fp(sr), fp(dr), fp(uu), fp(vv))
            if DRY: ran += 1; continue
            gdin = define(False, gs); gdout = define(False, gd); ez.ezdefset(gdout, gdin)
            rc, sp, dp = ez.ezwdint(uu, vv, n); ran += 1
            m = np.isfinite(sr) & np.isfinite(dr)
            if rc != rc_r: report("ezwdint", desc, f"rc {rc} vs {rc_r}")
            elif not np.array_equal(m, np.isfinite(sp) & np.isfinite(dp)): report("ezwdint", desc, "finite in one, not in the other")
            else:
                es = np.abs(sp[m] - sr[m]) / np.maximum(np.abs(sr[m]), 1e-3)
                dd = np.abs(((dp[m] - dr[m] + 180.0) % 360.0) - 180.0)
                tol_d = 1e-3 + np.degrees(8 * 6e-8 / np.maximum(np.cos(np.radians(np.minimum(worst[m], 89.9999))), 1e-7))
                big = (sr[m] > 1e-2)
                if es.size and (es.max() > 1e-5 or np.any(dd[big] > tol_d[big])): report("ezwdint", desc, f"max speed err {es.max():.2e}, max direction err {dd[big].max() if big.any() else 0:.2e} deg")
            continue
        mask_in = mask_field(ni, nj)
        mr = np.zeros(n, np.int32); zr_ = np.zeros(n, np.float32); mzr = np.zeros(n, np.int32); ur = zr_.copy(); vr = zr_.copy(); mvr = mr.copy()
        rc1 = R.c_ezsint_mdm(fp(zr_), ip(mr), fp(zin), ip(mask_in)); rc2 = R.c_ezget_mask_zones(ip(mzr), ip(mask_in))
        rc3 = R.c_ezuvint_mdm(fp(ur), fp(vr), ip(mvr), fp(uu), fp(vv), ip(mask_in))
        if DRY: ran += 1; continue
        gdin = define(False, gs); gdout = define(False, gd); ez.ezdefset(gdout, gdin)
        rcp, zp, mp = ez.ezsint_mdm(zin, mask_in, n); rcz, mzp = ez.ezget_mask_zones(mask_in, n); rcv, up, vp, mvp = ez.ezuvint_mdm(uu, vv, mask_in, n)
        ran += 1
        # the Fortran reads mask_in(ix + 1, iy + 1) / mask_in(nint(x), nint(y)) without bounds: outside the array the reference's answer is whatever lies there
        a1 = np.floor(ys.astype(np.float64)) * ni + np.floor(xs.astype(np.float64)); a2 = (np.rint(ys.astype(np.float64)) - 1) * ni + np.rint(xs.astype(np.float64)) - 1
        a0 = (np.floor(ys.astype(np.float64)) - 1) * ni + np.floor(xs.astype(np.float64)) - 1
        defined = (a1 < ni * nj) & (a2 < ni * nj) & (a0 >= 0) & (a2 >= 0) & np.isfinite(xs) & np.isfinite(ys)
        if (rcp, rcz) != (rc1, rc2): report("mask rc", desc, f"{(rcp, rcz)} vs {(rc1, rc2)}")
        elif not np.array_equal(mp[defined], mr[defined]): report("ezsint_mdm mask", desc, f"{int((mp != mr)[defined].sum())} of {int(defined.sum())} defined points differ")
        elif not np.array_equal(mzp[defined], mzr[defined]): report("ezget_mask_zones", desc, f"{int((mzp != mzr)[defined].sum())} points differ")
        elif np.array_equal(mp, mr): cmp_field("ezsint_mdm field", desc, zp, zr_, degree != 3)
        if rcv != rc3: report("ezuvint_mdm rc", desc, f"{rcv} vs {rc3}")
        elif not np.array_equal(mvp[defined], mvr[defined]): report("ezuvint_mdm mask", desc, f"{int((mvp != mvr)[defined].sum())} points differ")
        elif np.array_equal(mvp, mvr): cmp_wind("ezuvint_mdm winds", desc, up, vp, ur, vr, worst)
        continue
    # --- Yin-Yang
    spec = yy_grid(True); uni, unj = spec[0], spec[1]
    other = make_grid(str(rng.choice(["Lglob", "Lreg", "G", "N", "ZLglob", "ZLreg"])))
    setopts(degree, polar, extrap)
    desc = dict(mode=mode, yy=(uni, unj), other=other[:5], degree=degree, polar=polar, extrap=extrap)
    if mode == "yy_src":
        n_in = 2 * uni * unj; n_out = other[0] * other[1]
    else:
        n_in = other[0] * other[1]; n_out = 2 * uni * unj
    if mode == "yy_src":
        zin = np.concatenate([ec.synth_field(uni, unj, seed=int(rng.integers(1 << 30))) for _ in range(2)])
        w = [ec.synth_wind(uni, unj, seed=int(rng.integers(1 << 30))) for _ in range(2)]; uu = np.concatenate([w[0][0], w[1][0]]); vv = np.concatenate([w[0][1], w[1][1]])
    else:
        zin = ec.synth_field(other[0], other[1], seed=int(rng.integers(1 << 30))); uu, vv = ec.synth_wind(other[0], other[1], seed=int(rng.integers(1 << 30)))
    # The reference's Yin-Yang answers come from a process of their own, one per case (tools/ref_standalone.py): in a long session the reference build
    # corrupts its heap on these paths ("double free or corruption", "malloc(): corrupted top size" after a few hundred mixed cases with no product in
    # the process: FUZZ_REF_ONLY=1) and, before it dies, returns answers that depend on the session.  'U' TARGETS: scalars only (c_ezsint + c_ezuvint to
    # a Yin-Yang target is the quickest way there).
    winds = mode == "yy_src"
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "case.npz"), mode=mode, uni=uni, unj=unj, uax=spec[2], uay=spec[3], other=np.array(other[:5], dtype=object),
                 oax=other[5] if other[5] is not None else np.zeros(0, np.float32), oay=other[6] if other[6] is not None else np.zeros(0, np.float32),
                 zin=zin, uu=uu, vv=vv, degree=degree, polar=polar, extrap=extrap)
        pr = subprocess.run([sys.executable, os.path.join(_R, "tools", "ref_standalone.py"), os.path.join(td, "case.npz"), os.path.join(td, "out.npz")], capture_output=True, text=True)
        if pr.returncode != 0:
            crashed += 1; continue                      # the reference did not survive the case even alone
        o = np.load(os.path.join(td, "out.npz")); zr_, ur, vr, rc_r, rcv_r, worst = o["zr"], o["ur"], o["vr"], int(o["rc"]), int(o["rcv"]), o["worst"]
    if DRY or os.environ.get('FUZZ_YY_REF_ONLY'): ran += 1; continue
    gu_p = define_u(False, spec); go_p = define(False, other)
    if gu_p < 0 or go_p < 0: report("define", desc, "product refuses"); continue
    ez.ezdefset(gu_p if mode == "yy_dst" else go_p, go_p if mode == "yy_dst" else gu_p)
    rc, zp = ez.ezsint(zin, n_out); rcv, up, vp = ez.ezuvint(uu, vv, n_out) if winds else (0, ur.copy(), vr.copy()); ran += 1
    def yy_compare(zr_, ur, vr, rc_r, rcv_r, quiet):
        if rc != rc_r or rcv != rcv_r: report("yy rc", desc, f"{(rc, rcv)} vs {(rc_r, rcv_r)}")
        elif mode == "yy_src" and polar == 0 and degree == 3:
            # the all-Yin mask of a set first used without the polar correction (ezyymint.c): the bicubic is evaluated tens of cells outside Yin, where the
            # polynomial is a difference of huge terms -- the two evaluation orders agree to 1e-5 inside the grid and to ~1e-4 of the LARGEST value out
            # there (seen: 2.8e-5 relative on one point of 26 000); compared against the field's own magnitude
            scale = float(np.nanmax(np.abs(zr_[np.isfinite(zr_)]))) if np.isfinite(zr_).any() else 1.0
            if not np.array_equal(np.isfinite(zp), np.isfinite(zr_)) or np.nanmax(np.abs(zp - zr_)) > 1e-4 * scale: report("yy scalar (all-Yin extrapolation)", desc, f"max abs err {np.nanmax(np.abs(zp - zr_)):.3e} of scale {scale:.3e}")
        else:
            cmp_field("yy scalar", desc, zp, zr_, degree != 3)
            cmp_wind("yy winds", desc, up, vp, ur, vr, worst)
    bad0 = bad
    yy_compare(zr_, ur, vr, rc_r, rcv_r, False)
    if bad != bad0 and os.environ.get("FUZZ_KEEP"):           # development: the case and both answers, for a look at it on the CPU side
        np.savez(os.path.join(os.environ["FUZZ_KEEP"], f"yy_{k}.npz"), mode=mode, uni=uni, unj=unj, uax=spec[2], uay=spec[3], other=np.array(other[:5], dtype=object),
                 oax=other[5] if other[5] is not None else np.zeros(0, np.float32), oay=other[6] if other[6] is not None else np.zeros(0, np.float32),
                 zin=zin, uu=uu, vv=vv, degree=degree, polar=polar, extrap=extrap, zr=zr_, zp=zp, ur=ur, vr=vr, up=up, vp=vp)
setopts(3, 1, "maximum")
print(f"masks, c_ezwdint and Yin-Yang grids vs the reference build: {ran} of {ncases} cases ran, {bad} mismatches ({near} cases with wind errors between 1e-5 |V| and the conditioning bound; "
      f"{crashed} Yin-Yang cases the reference did not survive in a process of its own)")
sys.exit(1 if bad else 0)
