"""Third fuzzer against the reference build (oracle/_ref/libezref.so): the coordinate functions and the point-wise entry points on random grids --
c_gdll, c_gdxyfll, c_gdllfxy, c_gdxysint (= c_gdxysval), c_gdllsval, c_gdllvval, c_gdwdfuv, c_gduvfwd.  Grids: the kinds of tools/fuzz_vs_ref2.py plus
hemispheric / y-inverted A B G.  Bit for bit wherever no rotated frame is involved (coordinates, nearest, bilinear); through a rotated frame
coordinates within 2 ulp-equivalents amplified by 1 / cos(latitude); bicubic within 1e-5; winds within 1e-5 |V| (+ the conditioning bound).
python tools/fuzz_vs_ref3.py [ncases] [seed]"""
import ctypes, os, sys
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np
import reflib, ezcases as ec, oraclelib as ol
DRY = bool(os.environ.get('FUZZ_REF_ONLY'))
STRICT = not os.environ.get('FUZZ_LOOSE_ROTATED')      # coordinates through a rotated frame bit for bit too (the product computes them with the host libm)
if not DRY:
    from librmn_amd import ezscint as ez
R = reflib.ref()
fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
f32 = lambda a: np.ascontiguousarray(np.asarray(a, np.float64).astype(np.float32))
DEG = {0: "nearest", 1: "linear", 3: "cubic"}


def stretched(a0, a1, n):
    w = 1.0 + 0.15 * np.sin(np.arange(n - 1) * float(rng.uniform(0.1, 0.9)) + float(rng.uniform(0, 6)))
    x = np.concatenate([[0.0], np.cumsum(w)]); x = a0 + (a1 - a0) * x / x[-1]
    return f32(x)


def rot_ig():
    lat1 = float(rng.uniform(-60, 60)); lon1 = float(rng.uniform(0, 360))
    lat2 = float(np.clip(-0.4 * lat1 + rng.uniform(-20, 20), -80, 80)); lon2 = (lon1 + float(rng.uniform(60, 120))) % 360.0
    return ol.cxgaig("E", lat1, lon1, lat2, lon2)


def make_grid(kind):
    if kind == "ZEglob":
        nj = int(rng.integers(16, 60)); ni = 2 * nj + 1
        return ni, nj, "Z", rot_ig(), "E", f32(np.arange(ni) * 360.0 / (ni - 1)), f32(-90.0 + (np.arange(nj) + 0.5) * 180.0 / nj)
    if kind == "ZEreg":
        ni = int(rng.integers(20, 100)); nj = int(rng.integers(16, 80)); x0 = float(rng.uniform(120, 170)); y0 = float(rng.uniform(-35, -5))
        return ni, nj, "Z", rot_ig(), "E", stretched(x0, x0 + float(rng.uniform(25, 70)), ni), stretched(y0, y0 + float(rng.uniform(20, 60)), nj)
    if kind == "ZLglob":
        ni = int(rng.integers(24, 120)); nj = int(rng.integers(13, 60))
        return ni, nj, "Z", (100, 100, 9000, 0), "L", stretched(0.0, 360.0 * (ni - 1) / ni, ni), stretched(-90.0, 90.0, nj) if rng.random() < 0.5 else stretched(-88.0, 88.5, nj)
    if kind == "ZLreg":
        ni = int(rng.integers(20, 100)); nj = int(rng.integers(16, 80)); x0 = float(rng.uniform(10, 250)); y0 = float(rng.uniform(-70, 20))
        return ni, nj, "Z", (100, 100, 9000, 0), "L", stretched(x0, x0 + float(rng.uniform(20, 90)), ni), stretched(y0, y0 + float(rng.uniform(15, 60)), nj)
    if kind in ("N", "S"):
        ni = int(rng.integers(30, 110)); nj = int(rng.integers(30, 110))
        return ni, nj, kind, ol.cxgaig(kind, float(rng.uniform(0.2, 0.8)) * ni + 0.5, float(rng.uniform(0.2, 0.8)) * nj + 0.5, float(rng.uniform(60e3, 250e3)), float(rng.uniform(0, 360))), " ", None, None
    if kind in ("G", "Ghem", "Ginv"):
        nj = int(rng.integers(12, 60)) * 2
        ig = (0, 0, 0, 0) if kind == "G" else (0, 1, 0, 0) if kind == "Ginv" else (int(rng.integers(1, 3)), int(rng.integers(0, 2)), 0, 0)
        return 2 * nj, nj // (2 if kind == "Ghem" else 1), "G", ig, " ", None, None
    if kind in ("A", "Ahem", "B", "Binv"):
        nj = int(rng.integers(12, 60)) * 2; ni = int(rng.integers(30, 160)); t = kind[0]
        if t == "B": ni += 1 - ni % 2; nj += 1
        ig = (0, 0, 0, 0) if kind in ("A", "B") else (0, 1, 0, 0) if kind == "Binv" else (int(rng.integers(1, 3)), int(rng.integers(0, 2)), 0, 0)
        if kind == "Ahem": nj //= 2
        return ni, nj, t, ig, " ", None, None
    if kind == "Lglob":
        nj = int(rng.integers(12, 60)) * 2; ni = int(rng.integers(30, 160))
        return ni, nj, "L", (int(round(18000 / nj)), int(round(36000 / ni)), int(round(9000 / nj)), 0), " ", None, None
    if kind == "Lreg":
        ni = int(rng.integers(20, 120)); nj = int(rng.integers(16, 90))
        return ni, nj, "L", (int(rng.integers(20, 90)), int(rng.integers(20, 90)), int(rng.integers(2000, 11000)), int(rng.integers(0, 30000))), " ", None, None
    if kind == "E":
        nj = int(rng.integers(12, 50)); ni = 2 * nj + 1
        return ni, nj, "E", rot_ig(), " ", None, None
    raise ValueError(kind)


def define(ref_side, g):
    ni, nj, t, ig, ref, ax, ay = g
    if ax is None:
        return R.c_ezqkdef(ni, nj, t.encode(), *ig, 0) if ref_side else ez.ezqkdef(ni, nj, t, *ig)
    return R.c_ezgdef_fmem(ni, nj, t.encode(), ref.encode(), *ig, fp(ax), fp(ay)) if ref_side else ez.ezgdef_fmem(ni, nj, t, ref, *ig, ax, ay)


def bits_equal(a, b):
    """bit for bit, except that a zero is a zero: the reference build returns -0.0 for the first longitude of A / B / L grids (flang's AMOD), +0.0 elsewhere"""
    a = np.where(a == 0, np.float32(0), a); b = np.where(b == 0, np.float32(0), b)
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


KINDS = ["ZEglob", "ZEreg", "ZLglob", "ZLreg", "N", "S", "G", "Ghem", "Ginv", "A", "Ahem", "B", "Binv", "Lglob", "Lreg", "E"]
bad = ran = 0
notes = {}


def report(what, desc, detail):
    global bad
    bad += 1
    print("MISMATCH", what, detail, desc, flush=True)


for k in range(ncases):
    kind = str(rng.choice(KINDS))
    if os.environ.get('FUZZ_VERBOSE'): print(k, kind, flush=True)
    g = make_grid(kind)
    ni, nj = g[0], g[1]
    rotated = g[4] == "E" or g[2] == "E"
    hemi_or_inv = kind in ("Ghem", "Ginv", "Ahem", "Binv")
    gr = define(True, g)
    if gr < 0: continue
    gp = -1 if DRY else define(False, g)
    desc = dict(kind=kind, grid=g[:5])
    if not DRY and gp < 0: report("define", desc, "product refuses the grid"); continue
    ran += 1
    n = ni * nj
    # --- the parameter getters: c_ezgprm, c_ezgxprm, c_gdgaxes
    if not DRY:
        L0 = ez._lib()
        def prm(lib, gid):
            gt = ctypes.create_string_buffer(8); v = [ctypes.c_int32() for _ in range(6)]
            rc = lib.c_ezgprm(gid, gt, *[ctypes.byref(q) for q in v])
            gt2 = ctypes.create_string_buffer(8); gref = ctypes.create_string_buffer(8); w = [ctypes.c_int32() for _ in range(10)]
            rc2 = lib.c_ezgxprm(gid, ctypes.byref(w[0]), ctypes.byref(w[1]), gt2, ctypes.byref(w[2]), ctypes.byref(w[3]), ctypes.byref(w[4]), ctypes.byref(w[5]), gref,
                                ctypes.byref(w[6]), ctypes.byref(w[7]), ctypes.byref(w[8]), ctypes.byref(w[9]))
            return (rc, gt.value[:1], tuple(q.value for q in v)), (rc2, gt2.value[:1], gref.value[:1] if g[5] is not None else b"", tuple(q.value for q in (w if g[5] is not None else w[:6])))
        pr, pp_ = prm(R, gr), prm(L0, gp)
        if pr != pp_: report("ezgprm / ezgxprm", desc, f"{pp_} vs {pr}")
        if g[5] is not None:
            ax_r = np.zeros(ni, np.float32); ay_r = np.zeros(nj, np.float32); ax_p = ax_r.copy(); ay_p = ay_r.copy()
            R.c_gdgaxes(gr, fp(ax_r), fp(ay_r)); L0.c_gdgaxes(gp, fp(ax_p), fp(ay_p))
            if not (bits_equal(ax_r, ax_p) and bits_equal(ay_r, ay_p)): report("gdgaxes", desc, "axes differ")
    # --- c_gdll
    lat_r = np.zeros(n, np.float32); lon_r = lat_r.copy(); R.c_gdll(gr, fp(lat_r), fp(lon_r))
    if not DRY:
        rc, lat_p, lon_p = ez.gdll(gp, n)
        if rotated and not STRICT:
            dl = np.abs(lat_p - lat_r); do = np.abs(((lon_p - lon_r + 180.0) % 360.0) - 180.0) * np.cos(np.radians(lat_r.astype(np.float64)))
            if dl.max() > 2e-5 or do.max() > 4e-5: report("gdll", desc, f"max |dlat| {dl.max():.2e} deg, max |dlon| cos(lat) {do.max():.2e} deg")
            elif not (bits_equal(lat_p, lat_r) and bits_equal(lon_p, lon_r)): notes["gdll through a rotation: last-bit differences"] = notes.get("gdll through a rotation: last-bit differences", 0) + 1
        elif not (bits_equal(lat_p, lat_r) and bits_equal(lon_p, lon_r)):
            d = np.flatnonzero((lat_p.view(np.uint32) != lat_r.view(np.uint32)) | (lon_p.view(np.uint32) != lon_r.view(np.uint32))); report("gdll", desc, f"{d.size} points differ, first {int(d[0])}: ({lat_p[d[0]]!r}, {lon_p[d[0]]!r}) vs ({lat_r[d[0]]!r}, {lon_r[d[0]]!r})")
    # --- random geographic points: uniform on the sphere, the grid's own nodes, the poles and the seam
    m = 400
    plat = np.degrees(np.arcsin(rng.uniform(-1, 1, m))); plon = rng.uniform(0, 360, m)
    sel = rng.integers(0, n, m // 4); plat[:m // 4] = lat_r[sel]; plon[:m // 4] = lon_r[sel]
    plat[-6:] = [90.0, -90.0, 89.999, -89.999, 0.0, 45.0]; plon[-6:] = [0.0, 180.0, 359.999, 0.0, 360.0, 0.0]
    plat = f32(plat); plon = f32(plon)
    xr = np.zeros(m, np.float32); yr = xr.copy(); lo = plon.copy(); R.c_gdxyfll(gr, fp(xr), fp(yr), fp(plat), fp(lo), m)
    if not DRY:
        rc, xp, yp = ez.gdxyfll(gp, plat, plon)
        fin = np.isfinite(xr) & np.isfinite(yr)
        if (rotated and not STRICT):
            tol = 2e-4 if rotated else 0.0
            # near a pole of the rotated frame x is ill-conditioned (longitude): compare x scaled by cos of the rotated latitude, approximated from y
            ok = np.array_equal(fin, np.isfinite(xp) & np.isfinite(yp))
            if ok and rotated:
                dy = np.abs(yp[fin] - yr[fin]); dx = np.abs(xp[fin] - xr[fin])
                big = (dy > 1e-3 * np.maximum(1.0, np.abs(yr[fin]) * 1e-3)) | (dx > 0.02 * np.maximum(1.0, np.abs(xr[fin]) * 1e-3))
                if big.any(): ok = False
            elif ok and not (bits_equal(xp, xr) and bits_equal(yp, yr)): ok = False
            if not ok:
                d = np.flatnonzero((xp != xr) | (yp != yr)); w = d[np.argmax(np.abs(xp[d] - xr[d]) + np.abs(yp[d] - yr[d]))]
                report("gdxyfll", desc, f"{d.size} points differ, worst at lat/lon ({plat[w]!r}, {plon[w]!r}): ({xp[w]!r}, {yp[w]!r}) vs ({xr[w]!r}, {yr[w]!r})")
        elif not (bits_equal(xp, xr) and bits_equal(yp, yr)):
            d = np.flatnonzero((xp.view(np.uint32) != xr.view(np.uint32)) | (yp.view(np.uint32) != yr.view(np.uint32)))
            report("gdxyfll", desc, f"{d.size} points differ, first at lat/lon ({plat[d[0]]!r}, {plon[d[0]]!r}): ({xp[d[0]]!r}, {yp[d[0]]!r}) vs ({xr[d[0]]!r}, {yr[d[0]]!r})")
    # --- random x, y -> lat, lon
    px = f32(rng.uniform(0.6, ni + 0.4, m)); py = f32(rng.uniform(0.6, nj + 0.4, m))
    px[:40] = np.round(px[:40]); py[:40] = np.round(py[:40]); px[:40] = np.clip(px[:40], 1, ni); py[:40] = np.clip(py[:40], 1, nj)
    la_r = np.zeros(m, np.float32); lo_r = la_r.copy(); R.c_gdllfxy(gr, fp(la_r), fp(lo_r), fp(px), fp(py), m)
    if not DRY:
        rc, la_p, lo_p = ez.gdllfxy(gp, px, py)
        if rotated and not STRICT:
            dl = np.abs(la_p - la_r); do = np.abs(((lo_p - lo_r + 180.0) % 360.0) - 180.0) * np.cos(np.radians(la_r.astype(np.float64)))
            if dl.max() > 2e-5 or do.max() > 4e-5: report("gdllfxy", desc, f"max |dlat| {dl.max():.2e}, max |dlon| cos(lat) {do.max():.2e}")
        elif not (bits_equal(la_p, la_r) and bits_equal(lo_p, lo_r)):
            d = np.flatnonzero((la_p.view(np.uint32) != la_r.view(np.uint32)) | (lo_p.view(np.uint32) != lo_r.view(np.uint32))); report("gdllfxy", desc, f"{d.size} points differ, first x,y ({px[d[0]]!r}, {py[d[0]]!r}): ({la_p[d[0]]!r}, {lo_p[d[0]]!r}) vs ({la_r[d[0]]!r}, {lo_r[d[0]]!r})")
    # --- interpolation at given x, y (c_gdxysint) and at lat / lon (c_gdllsval)
    zin = ec.synth_field(ni, nj, seed=int(rng.integers(1 << 30)))
    uu, vv = ec.synth_wind(ni, nj, seed=int(rng.integers(1 << 30)))
    if kind in ("ZEglob", "E", "B", "Binv"):
        for a in (zin, uu, vv):
            z2 = a.reshape(nj, ni); z2[:, -1] = z2[:, 0]
    degree = int(rng.choice([0, 1, 3]))
    R.c_ezsetopt(b"interp_degree", DEG[degree].encode())
    if not DRY: ez.ezsetopt("interp_degree", DEG[degree])
    sx = px.copy(); sy = py.copy()
    sx[-20:] = f32(rng.uniform(-3, ni + 4, 20)); sy[-20:] = f32(rng.uniform(-3, nj + 4, 20))          # some points outside: the leaf kernels clamp
    if kind == "E" or hemi_or_inv: sx = px.copy(); sy = py.copy()
    zr = np.zeros(m, np.float32); rc_r = R.c_gdxysval(gr, fp(zr), fp(zin), fp(sx), fp(sy), m)
    zl = np.zeros(m, np.float32); lo2 = plon.copy(); rc_l = R.c_gdllsval(gr, fp(zl), fp(zin), fp(plat), fp(lo2), m)
    if not DRY:
        rc, zp = ez.gdxysint(zin, gp, sx, sy)
        def cmp_scalar(tag, a, b, exact):
            fa, fb = np.isfinite(a), np.isfinite(b)
            if not np.array_equal(fa, fb): return report(tag, dict(desc, degree=degree), "finite in one, not in the other")
            if exact:
                if not bits_equal(a, b):
                    d = np.flatnonzero(a.view(np.uint32) != b.view(np.uint32)); report(tag, dict(desc, degree=degree), f"{d.size} points differ, first {int(d[0])}: {a[d[0]]!r} vs {b[d[0]]!r}")
            else:
                e = np.abs(a[fa] - b[fa]) / np.maximum(np.abs(b[fa]), 1e-30)
                if e.size and e.max() > 1e-5: report(tag, dict(desc, degree=degree), f"max rel err {e.max():.2e}")
        cmp_scalar("gdxysint", zp, zr, degree != 3)
        rc, zlp = ez.gdllsval(gp, zin, plat, plon)
        if rc != rc_l: report("gdllsval", dict(desc, degree=degree), f"rc {rc} vs {rc_l}")
        elif rotated:
            e = np.abs(zlp - zl) / np.maximum(np.abs(zl), 1e-30)
            if np.nanmax(e) > 2e-4: report("gdllsval", dict(desc, degree=degree), f"max rel err {np.nanmax(e):.2e} (through a rotated locate)")
        else: cmp_scalar("gdllsval", zlp, zl, degree != 3)
    # --- winds at lat / lon: c_gdllvval, c_gdwdfuv, c_gduvfwd
    if hemi_or_inv and not DRY:
        # hemispheric / y-inverted sources: c_gdxyvval and c_gdllvval are two scalar interpolations each (gdxyvval.c:89, gdllvval.c:44-47), no wind chain
        L_ = ez._lib()
        a_r = np.zeros(m, np.float32); b_r = a_r.copy(); R.c_gdxyvval(gr, fp(a_r), fp(b_r), fp(uu), fp(vv), fp(px.copy()), fp(py.copy()), m)
        a_p = np.zeros(m, np.float32); b_p = a_p.copy(); rcp = L_.c_gdxyvval(gp, fp(a_p), fp(b_p), fp(uu), fp(vv), fp(px.copy()), fp(py.copy()), m)
        if rcp < 0: report("gdxyvval (hemispheric / inverted)", dict(desc, degree=degree), "refused")
        elif degree != 3 and not (bits_equal(a_p, a_r) and bits_equal(b_p, b_r)): report("gdxyvval (hemispheric / inverted)", dict(desc, degree=degree), f"{int(((a_p != a_r) | (b_p != b_r)).sum())} points differ")
        elif degree == 3 and (np.any(np.abs(a_p - a_r) > 1e-5 * np.maximum(np.abs(a_r), 1e-3)) or np.any(np.abs(b_p - b_r) > 1e-5 * np.maximum(np.abs(b_r), 1e-3))): report("gdxyvval (hemispheric / inverted)", dict(desc, degree=degree), "beyond 1e-5")
        a_r = np.zeros(m, np.float32); b_r = a_r.copy(); R.c_gdllvval(gr, fp(a_r), fp(b_r), fp(uu), fp(vv), fp(plat), fp(plon.copy()), m)
        rcp, a_p, b_p = ez.gdllvval(gp, uu, vv, plat, plon)
        if rcp < 0: report("gdllvval (hemispheric / inverted)", dict(desc, degree=degree), "refused")
        elif degree != 3 and not (bits_equal(a_p, a_r) and bits_equal(b_p, b_r)): report("gdllvval (hemispheric / inverted)", dict(desc, degree=degree), f"{int(((a_p != a_r) | (b_p != b_r)).sum())} points differ")
        elif degree == 3 and (np.any(np.abs(a_p - a_r) > 1e-5 * np.maximum(np.abs(a_r), 1e-3)) or np.any(np.abs(b_p - b_r) > 1e-5 * np.maximum(np.abs(b_r), 1e-3))): report("gdllvval (hemispheric / inverted)", dict(desc, degree=degree), "beyond 1e-5")
    if not hemi_or_inv and kind != "E":
        ur = np.zeros(m, np.float32); vr = ur.copy(); lo3 = plon.copy(); rcv_r = R.c_gdllvval(gr, fp(ur), fp(vr), fp(uu), fp(vv), fp(plat), fp(lo3), m)
        wu = f32(rng.normal(0, 15, m)); wv = f32(rng.normal(0, 15, m)); wu[:5] = 0.0; wv[5:8] = 0.0; wu[8] = wv[8] = 0.0
        sp_r = np.zeros(m, np.float32); wd_r = sp_r.copy(); lo4 = plon.copy(); R.c_gdwdfuv(gr, fp(sp_r), fp(wd_r), fp(wu), fp(wv), fp(plat), fp(lo4), m)
        u2_r = np.zeros(m, np.float32); v2_r = u2_r.copy(); lo5 = plon.copy(); R.c_gduvfwd(gr, fp(u2_r), fp(v2_r), fp(sp_r), fp(wd_r), fp(plat), fp(lo5), m)
        if not DRY:
            worst = np.abs(plat.astype(np.float64))            # the latitudes that condition the wind direction: the true one and the one in the rotated frame
            if g[4] == "E": worst = np.maximum(worst, np.abs(np.interp(yr.astype(np.float64), np.arange(1, nj + 1), g[6].astype(np.float64), left=90.0, right=90.0)))
            cosl = np.maximum(np.cos(np.radians(np.minimum(worst, 89.9999))), 1e-7)
            bound = 1e-5 + (8 * 6e-8 / cosl if rotated else 0.0)
            rcv, up, vp = ez.gdllvval(gp, uu, vv, plat, plon)
            sc = np.maximum(np.hypot(ur.astype(np.float64), vr.astype(np.float64)), 1e-3)
            e = np.maximum(np.abs(up - ur), np.abs(vp - vr)) / sc
            fin = np.isfinite(e)
            if rcv != rcv_r: report("gdllvval", dict(desc, degree=degree), f"rc {rcv} vs {rcv_r}")
            elif np.any(e[fin] > np.broadcast_to(bound, e.shape)[fin] + (2e-4 if rotated else 0.0)): report("gdllvval", dict(desc, degree=degree), f"max err {np.nanmax(e):.2e} |V|")
            # winds at given x, y (c_gdxyvval: the two components as they are on the grid) and as speed / direction (c_gdxywdval, c_gdllwdval)
            L_ = ez._lib()
            vx = px.copy(); vy = py.copy()
            a_r = np.zeros(m, np.float32); b_r = a_r.copy(); R.c_gdxyvval(gr, fp(a_r), fp(b_r), fp(uu), fp(vv), fp(vx), fp(vy), m)
            a_p = np.zeros(m, np.float32); b_p = a_p.copy(); L_.c_gdxyvval(gp, fp(a_p), fp(b_p), fp(uu), fp(vv), fp(vx), fp(vy), m)
            if degree != 3:
                if not (bits_equal(a_p, a_r) and bits_equal(b_p, b_r)): report("gdxyvval", dict(desc, degree=degree), f"{int(((a_p != a_r) | (b_p != b_r)).sum())} points differ")
            elif np.any(np.abs(a_p - a_r) > 1e-5 * np.maximum(np.abs(a_r), 1e-3)) or np.any(np.abs(b_p - b_r) > 1e-5 * np.maximum(np.abs(b_r), 1e-3)): report("gdxyvval", dict(desc, degree=degree), "beyond 1e-5")
            for nm, args_r, args_p in (("gdxywdval", (vx, vy), (vx, vy)), ("gdllwdval", (plat, plon.copy()), (plat, plon.copy()))):
                s_r = np.zeros(m, np.float32); d_r = s_r.copy(); getattr(R, "c_" + nm)(gr, fp(s_r), fp(d_r), fp(uu), fp(vv), fp(args_r[0]), fp(args_r[1]), m)
                s_p = np.zeros(m, np.float32); d_p = s_p.copy(); getattr(L_, "c_" + nm)(gp, fp(s_p), fp(d_p), fp(uu), fp(vv), fp(args_p[0]), fp(args_p[1]), m)
                fin = np.isfinite(s_r) & np.isfinite(d_r)
                dd = np.abs(((d_p[fin] - d_r[fin] + 180.0) % 360.0) - 180.0)
                tol_d = 1e-3 + (np.degrees(8 * 6e-8 / cosl[fin]) if rotated else 0.0) + (2e-2 if rotated and nm == "gdllwdval" else 0.0)
                big = s_r[fin] > 1e-2
                if not np.array_equal(fin, np.isfinite(s_p) & np.isfinite(d_p)): report(nm, dict(desc, degree=degree), "finite in one, not in the other")
                elif np.any(np.abs(s_p[fin] - s_r[fin]) > (2e-4 if rotated and nm == "gdllwdval" else 1e-5) * np.maximum(s_r[fin], 1e-3)) or np.any(dd[big] > np.broadcast_to(tol_d, dd.shape)[big]):
                    report(nm, dict(desc, degree=degree), f"max speed err {np.abs(s_p[fin] - s_r[fin]).max():.2e}, max direction err {dd[big].max() if big.any() else 0:.2e} deg")
            rc, sp_p, wd_p = ez.gdwdfuv(gp, wu, wv, plat, plon)
            dd = np.abs(((wd_p - wd_r + 180.0) % 360.0) - 180.0)
            if np.any(np.abs(sp_p - sp_r) > 1e-5 * np.maximum(sp_r, 1e-3)) or np.any(dd[sp_r > 0] > 1e-3 + (np.degrees(8 * 6e-8 / cosl[sp_r > 0]) if rotated else 0.0)):
                w = int(np.argmax(dd * (sp_r > 0))); report("gdwdfuv", desc, f"max |dspd| {np.abs(sp_p - sp_r).max():.2e}, max |ddir| {(dd[sp_r > 0].max() if (sp_r > 0).any() else 0.0):.2e} deg at lat {plat[w]!r}: {wd_p[w]!r} vs {wd_r[w]!r}")
            rc, u2_p, v2_p = ez.gduvfwd(gp, sp_r, wd_r, plat, plon)
            sc = np.maximum(sp_r.astype(np.float64), 1e-3)
            e = np.maximum(np.abs(u2_p - u2_r), np.abs(v2_p - v2_r)) / sc
            if np.any(e > bound): report("gduvfwd", desc, f"max err {e.max():.2e} |V| at lat {plat[int(np.argmax(e / bound))]!r}")
R.c_ezsetopt(b"interp_degree", b"cubic")
print(f"coordinate and point-wise entry points vs the reference build: {ran} of {ncases} grids, {bad} mismatches; notes: {notes}")
sys.exit(1 if bad else 0)
