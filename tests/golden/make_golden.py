"""Generates tests/golden/ez_golden.npz from the reference's own code (oracle/_ref/libezref.so,
built by oracle/build_ref.sh from /root/reference).  Runs only in the build container.
Fixtures are DATA: for every case of tests/ezcases.py the reference's outputs (zout per degree and
polar_correction setting, target lat/lon, located x/y, Gaussian latitudes, wind outputs).
Inputs are regenerated from tests/ezcases.py (deterministic, libm-free).

    python tests/golden/make_golden.py
"""
import ctypes, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from reflib import ref, fptr          # noqa: E402
import ezcases as ec                  # noqa: E402

DEG = {0: b"nearest", 1: b"linear", 3: b"cubic"}


def define(L, spec):
    ni, nj, grtyp, ig, grref, axes = spec
    if grtyp in ("Z", "Y"):
        ax, ay = axes(ni, nj)
        return L.c_ezgdef_fmem(ni, nj, grtyp.encode(), grref.encode(), ig[0], ig[1], ig[2], ig[3], fptr(ax), fptr(ay))
    return L.c_ezqkdef(ni, nj, grtyp.encode(), ig[0], ig[1], ig[2], ig[3], 0)


def case_inputs(name, case):
    ni, nj = case["src"][:2]
    zin = ec.synth_field(ni, nj, seed=11)
    uu, vv = ec.synth_wind(ni, nj, seed=21)
    if case["src"][2] in ("Z", "B") or name == "Lrepeat_to_L":
        for a in (zin, uu, vv):
            a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
    return zin, uu, vv


def main():
    L = ref()
    out = {}
    for nj in (8, 32, 200, 2200):
        gd = L.c_ezqkdef(2 * nj, nj, b"G", 0, 0, 0, 0, 0)
        ax = np.zeros(2 * nj, np.float32); ay = np.zeros(nj, np.float32)
        L.c_gdgaxes(gd, fptr(ax), fptr(ay))
        out[f"gausslat_{nj}"] = ay
    for name, case in sorted(ec.scalar_cases().items()):
        gdin = define(L, case["src"]); gdout = define(L, ec.dst_spec(case))
        L.c_ezdefset(gdout, gdin)
        no, mo = case["dst"][:2]
        zin, uu, vv = case_inputs(name, case)
        for degree in (0, 1, 3):
            for polar in (1, 0):
                L.c_ezsetopt(b"interp_degree", DEG[degree])
                L.c_ezsetopt(b"polar_correction", b"yes" if polar else b"no")
                z = np.zeros(no * mo, np.float32)
                rc = L.c_ezsint(fptr(z), fptr(zin))
                out[f"{name}/z_d{degree}_p{polar}"] = z
                out[f"{name}/rc_d{degree}_p{polar}"] = np.int32(rc)
                if name != "G_to_G" and name != "L_to_G":
                    u = np.zeros(no * mo, np.float32); v = np.zeros(no * mo, np.float32)
                    rc = L.c_ezuvint(fptr(u), fptr(v), fptr(uu), fptr(vv))
                    out[f"{name}/u_d{degree}_p{polar}"] = u
                    out[f"{name}/v_d{degree}_p{polar}"] = v
                    if degree == 3:          # c_ezwdint: speed / direction on the target grid
                        sp = np.zeros(no * mo, np.float32); wd = np.zeros(no * mo, np.float32)
                        rc = L.c_ezwdint(fptr(sp), fptr(wd), fptr(uu), fptr(vv))
                        out[f"{name}/spd_d{degree}_p{polar}"] = sp
                        out[f"{name}/dir_d{degree}_p{polar}"] = wd
        lat = np.zeros(no * mo, np.float32); lon = np.zeros(no * mo, np.float32)
        L.c_gdll(gdout, fptr(lat), fptr(lon))
        x = np.zeros(no * mo, np.float32); y = np.zeros(no * mo, np.float32)
        lon2 = lon.copy()
        L.c_gdxyfll(gdin, fptr(x), fptr(y), fptr(lat), fptr(lon2), no * mo)
        out[f"{name}/lat"] = lat; out[f"{name}/lon"] = lon; out[f"{name}/x"] = x; out[f"{name}/y"] = y
    # Yin-Yang 'U' source -> L / G / N targets (c_ezyysint / c_ezyyuvint)
    yni, ynj = ec.YY_NI, ec.YY_NJ
    yax, yay = ec.yy_axes(yni, ynj)
    gy = L.c_ezgdef_fmem(yni, ynj, b"Z", b"E", *ec.YIN_IG, fptr(yax), fptr(yay))
    ga = L.c_ezgdef_fmem(yni, ynj, b"Z", b"E", *ec.YAN_IG, fptr(yax), fptr(yay))
    ids = np.array([gy, ga], np.int32)
    gu = L.c_ezgdef_supergrid(yni, 2 * ynj, b"U", b"F", 1, 2, ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    z, uu, vv = ec.yy_fields()
    for tname, (no, mo, gt, ig) in ec.yy_targets().items():
        go = L.c_ezqkdef(no, mo, gt.encode(), *ig, 0)
        assert L.c_ezdefset(go, gu) == 1
        for degree in (0, 1, 3):
            L.c_ezsetopt(b"interp_degree", DEG[degree]); L.c_ezsetopt(b"polar_correction", b"yes")
            zo = np.zeros(no * mo, np.float32); uo = np.zeros(no * mo, np.float32); vo = np.zeros(no * mo, np.float32)
            assert L.c_ezsint(fptr(zo), fptr(z)) == 0 and L.c_ezuvint(fptr(uo), fptr(vo), fptr(uu), fptr(vv)) == 0
            out[f"YY_to_{tname}/z_d{degree}"] = zo; out[f"YY_to_{tname}/u_d{degree}"] = uo; out[f"YY_to_{tname}/v_d{degree}"] = vo
            if degree == 3:
                so = np.zeros(no * mo, np.float32); do_ = np.zeros(no * mo, np.float32)
                assert L.c_ezwdint(fptr(so), fptr(do_), fptr(uu), fptr(vv)) == 0
                out[f"YY_to_{tname}/spd_d3"] = so; out[f"YY_to_{tname}/dir_d3"] = do_
    # regular rotated 'E' grids
    for name, case in sorted(ec.e_cases().items()):
        ni, nj = case["src"][:2]; no, mo = case["dst"][:2]
        gdin = define(L, case["src"]); gdout = define(L, ec.dst_spec(case))
        assert L.c_ezdefset(gdout, gdin) == 1
        zin = ec.synth_field(ni, nj, seed=11)
        if case["src"][2] == "E":
            z2 = zin.reshape(nj, ni); z2[:, -1] = z2[:, 0]; zin = np.ascontiguousarray(z2.reshape(-1))
        for degree in (0, 1, 3):
            for polar in case["polar"]:
                L.c_ezsetopt(b"interp_degree", DEG[degree]); L.c_ezsetopt(b"polar_correction", b"yes" if polar else b"no")
                zo = np.zeros(no * mo, np.float32)
                rc = L.c_ezsint(fptr(zo), fptr(zin))
                out[f"{name}/z_d{degree}_p{polar}"] = zo; out[f"{name}/rc_d{degree}_p{polar}"] = np.int32(rc)
    # hemispheric A / B grids: scalars only
    for name, case in sorted(ec.hemi_cases().items()):
        ni, nj = case["src"][:2]; no, mo = case["dst"][:2]
        gdin = define(L, case["src"]); gdout = define(L, ec.dst_spec(case))
        assert L.c_ezdefset(gdout, gdin) == 1
        zin = ec.synth_field(ni, nj, seed=11)
        if case["src"][2] == "B":
            z2 = zin.reshape(nj, ni); z2[:, -1] = z2[:, 0]; zin = np.ascontiguousarray(z2.reshape(-1))
        for degree in (0, 1, 3):
            for polar in (1, 0):
                L.c_ezsetopt(b"interp_degree", DEG[degree]); L.c_ezsetopt(b"polar_correction", b"yes" if polar else b"no")
                zo = np.zeros(no * mo, np.float32)
                rc = L.c_ezsint(fptr(zo), fptr(zin))
                out[f"{name}/z_d{degree}_p{polar}"] = zo; out[f"{name}/rc_d{degree}_p{polar}"] = np.int32(rc)
        lat = np.zeros(no * mo, np.float32); lon = np.zeros(no * mo, np.float32)
        L.c_gdll(gdout, fptr(lat), fptr(lon))
        out[f"{name}/lat"] = lat; out[f"{name}/lon"] = lon
    # Yin-Yang TARGET: from an ordinary grid (two plain interpolations) and from another Yin-Yang grid (2.5-degree subgrids)
    tni, tnj = ec.YYT_NI, ec.YYT_NJ
    tax, tay = ec.yyt_axes(tni, tnj)
    ty = L.c_ezgdef_fmem(tni, tnj, b"Z", b"E", *ec.YIN_IG, fptr(tax), fptr(tay))
    ta = L.c_ezgdef_fmem(tni, tnj, b"Z", b"E", *ec.YAN_IG, fptr(tax), fptr(tay))
    tids = np.array([ty, ta], np.int32)
    tu = L.c_ezgdef_supergrid(tni, 2 * tnj, b"U", b"F", 1, 2, tids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    gsrc = L.c_ezqkdef(64, 32, b"G", 0, 0, 0, 0, 0)
    zg = ec.synth_field(64, 32, seed=11)
    for degree in (0, 1, 3):
        L.c_ezsetopt(b"interp_degree", DEG[degree]); L.c_ezsetopt(b"polar_correction", b"yes")
        zo = np.zeros(2 * tni * tnj, np.float32)
        uo = np.zeros(2 * tni * tnj, np.float32); vo = np.zeros(2 * tni * tnj, np.float32)
        ug, vg = ec.synth_wind(64, 32, seed=21)
        assert L.c_ezdefset(tu, gsrc) == 1 and L.c_ezsint(fptr(zo), fptr(zg)) >= 0
        out[f"G_to_YY/z_d{degree}"] = zo.copy()
        assert L.c_ezuvint(fptr(uo), fptr(vo), fptr(ug), fptr(vg)) >= 0
        out[f"G_to_YY/u_d{degree}"] = uo.copy(); out[f"G_to_YY/v_d{degree}"] = vo.copy()
        assert L.c_ezdefset(tu, gu) == 1 and L.c_ezsint(fptr(zo), fptr(z)) >= 0
        out[f"YY_to_YY/z_d{degree}"] = zo.copy()
        assert L.c_ezuvint(fptr(uo), fptr(vo), fptr(uu), fptr(vv)) >= 0
        out[f"YY_to_YY/u_d{degree}"] = uo.copy(); out[f"YY_to_YY/v_d{degree}"] = vo.copy()
    L.c_ezsetopt(b"interp_degree", b"cubic"); L.c_ezsetopt(b"polar_correction", b"yes")
    np.savez_compressed(os.path.join(HERE, "ez_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
