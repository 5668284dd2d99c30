"""CPU oracle vs the committed golden vectors (tests/golden/ez_golden.npz, produced from the
reference's own code by tests/golden/make_golden.py).  Bit-exact.  Needs no reference build."""
import ctypes, os
import numpy as np
import pytest

import oraclelib as ol
import ezcases as ec

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ez_golden.npz"))
CASES = ec.scalar_cases()


def orc_define(spec):
    ni, nj, grtyp, ig, grref, axes = spec
    if grtyp in ("Z", "Y"):
        ax, ay = axes(ni, nj)
        return ol.grid_define(ni, nj, grtyp, ig, grref, ax, ay)
    return ol.grid_define(ni, nj, grtyp, ig)


def case_inputs(name, case):
    ni, nj = case["src"][:2]
    zin = ec.synth_field(ni, nj, seed=11)
    uu, vv = ec.synth_wind(ni, nj, seed=21)
    if case["src"][2] in ("Z", "B") or name == "Lrepeat_to_L":
        for a in (zin, uu, vv):
            a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
    return zin, uu, vv


@pytest.mark.parametrize("nj", [8, 32, 200, 2200])
def test_gauss_lat_golden(nj):
    mine = np.zeros(nj, np.float32)
    ol.oracle().orc_gauss_lat(ol.fptr(mine), nj)
    assert np.array_equal(mine.view(np.uint32), GOLD[f"gausslat_{nj}"].view(np.uint32))
    if nj == 2200:   # SURVEY.md section 8c item 1 probe anchors
        assert abs(mine[0] - (-89.9374390)) < 1e-6 and abs(mine[1] - (-89.8559875)) < 1e-6
        assert abs(mine[1099] - (-0.0409012)) < 1e-6 and abs(mine[2199] - 89.9374390) < 1e-6


@pytest.mark.parametrize("name", sorted(CASES))
def test_scalar_golden(name):
    O = ol.oracle()
    case = CASES[name]
    no, mo = case["dst"][:2]
    zin, uu, vv = case_inputs(name, case)
    gi = orc_define(case["src"]); go = orc_define(ec.dst_spec(case))
    gs = O.orc_defset(go, gi)
    for degree in (0, 1, 3):
        for polar in (1, 0):
            opts = ol.default_opts(degre_interp=degree, polar_correction=polar)
            z = np.zeros(no * mo, np.float32)
            rc = O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(z), ol.fptr(zin))
            assert rc == int(GOLD[f"{name}/rc_d{degree}_p{polar}"])
            assert np.array_equal(z.view(np.uint32), GOLD[f"{name}/z_d{degree}_p{polar}"].view(np.uint32)), (name, degree, polar)
    assert np.array_equal(ol.np_from(go.contents.lat, no * mo), GOLD[f"{name}/lat"])
    assert np.array_equal(ol.np_from(go.contents.lon, no * mo), GOLD[f"{name}/lon"])
    x = np.zeros(no * mo, np.float32); y = np.zeros(no * mo, np.float32)
    lon2 = GOLD[f"{name}/lon"].copy()
    O.orc_gdxyfll(gi, ol.fptr(x), ol.fptr(y), ol.fptr(GOLD[f"{name}/lat"].copy()), ol.fptr(lon2), no * mo)
    assert np.array_equal(x.view(np.uint32), GOLD[f"{name}/x"].view(np.uint32))
    assert np.array_equal(y.view(np.uint32), GOLD[f"{name}/y"].view(np.uint32))


@pytest.mark.parametrize("name", [n for n in sorted(CASES) if n not in ("G_to_G", "L_to_G")])
def test_vector_golden(name):
    O = ol.oracle()
    case = CASES[name]
    no, mo = case["dst"][:2]
    zin, uu, vv = case_inputs(name, case)
    gi = orc_define(case["src"]); go = orc_define(ec.dst_spec(case))
    gs = O.orc_defset(go, gi)
    for degree in (0, 1, 3):
        for polar in (1, 0):
            opts = ol.default_opts(degre_interp=degree, polar_correction=polar)
            u = np.zeros(no * mo, np.float32); v = np.zeros(no * mo, np.float32)
            O.orc_ezuvint(gs, ctypes.byref(opts), ol.fptr(u), ol.fptr(v), ol.fptr(uu), ol.fptr(vv))
            assert np.array_equal(u.view(np.uint32), GOLD[f"{name}/u_d{degree}_p{polar}"].view(np.uint32)), (name, degree, polar)
            assert np.array_equal(v.view(np.uint32), GOLD[f"{name}/v_d{degree}_p{polar}"].view(np.uint32)), (name, degree, polar)


@pytest.mark.parametrize("name", [n for n in sorted(CASES) if n not in ("G_to_G", "L_to_G")])
def test_wdint_golden(name):
    """orc_ezwdint (speed / direction) against the reference's c_ezwdint outputs in the fixture, bit-exact"""
    O = ol.oracle()
    case = CASES[name]
    gi = orc_define(case["src"]); go = orc_define(ec.dst_spec(case))
    gs = O.orc_defset(go, gi)
    no, mo = case["dst"][:2]
    _, uu, vv = case_inputs(name, case)
    for polar in (1, 0):
        opts = ol.default_opts(degre_interp=3, polar_correction=polar)
        s = np.zeros(no * mo, np.float32); d = np.zeros(no * mo, np.float32)
        O.orc_ezwdint(gs, ctypes.byref(opts), ol.fptr(s), ol.fptr(d), ol.fptr(uu), ol.fptr(vv))
        assert np.array_equal(s.view(np.uint32), GOLD[f"{name}/spd_d3_p{polar}"].view(np.uint32)), (name, polar)
        assert np.array_equal(d.view(np.uint32), GOLD[f"{name}/dir_d3_p{polar}"].view(np.uint32)), (name, polar)


def test_full_size_cfg2_golden():
    """the oracle at BASELINE cfg2's full size against the reference's own run of it
    (tests/golden/make_cfg2_full.py): sampled rows/columns, float64 sum and whole-field bit hash,
    bit-exact; plus the survey's printed anchors of the same run."""
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "cfg2_full_golden.npz"))
    ni, nj, no, mo = 4400, 2200, 7200, 3601
    zin = np.ascontiguousarray((np.float32(280) + (np.float32(20) * G["probe_sin"])[None, :] * G["probe_cos"][:, None]).astype(np.float32))
    O = ol.oracle()
    gi = ol.grid_define(ni, nj, "G"); go = ol.grid_define(no, mo, "L", (5, 5, 0, 0))
    gs = O.orc_defset(go, gi)
    z = np.zeros((mo, no), np.float32)
    for degree, polar in ((3, 1), (1, 0), (0, 0)):
        opts = ol.default_opts(degre_interp=degree, polar_correction=polar)
        assert O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(z), ol.fptr(zin)) == 0
        key = f"probe/d{degree}_p{polar}"
        assert np.array_equal(z[G["rows"]].view(np.uint32), G[key + "/rows"].view(np.uint32)), key
        assert np.array_equal(z[:, G["cols"]].view(np.uint32), G[key + "/cols"].view(np.uint32)), key
        assert float(z.astype(np.float64).sum()) == float(G[key + "/sum"]), key
        if not polar:
            u = z.ravel().view(np.uint32)
            assert (int(u.astype(np.uint64).sum()) & 0xFFFFFFFF, int(np.bitwise_xor.reduce(u))) == tuple(int(v) for v in G[key + "/hash"]), key
        if degree == 3:
            f = z.ravel()
            assert "%.8e %.3f %.3f %.3f" % (f.astype(np.float64).sum(), f[0], f[f.size // 2], f[-1]) == "7.26056487e+09 281.818 276.863 278.271"


@pytest.mark.parametrize("tname", sorted(ec.yy_targets()))
def test_yinyang_golden(tname):
    """Yin-Yang 'U' source -> L / G / N: the oracle's c_ezyysint / c_ezyyuvint restatement against the reference's outputs"""
    O = ol.oracle()
    ni, nj = ec.YY_NI, ec.YY_NJ
    ax, ay = ec.yy_axes(ni, nj)
    sg = O.orc_supergrid_define(ol.grid_define(ni, nj, "Z", ec.YIN_IG, "E", ax, ay), ol.grid_define(ni, nj, "Z", ec.YAN_IG, "E", ax, ay))
    no, mo, gt, ig = ec.yy_targets()[tname]
    go = ol.grid_define(no, mo, gt, ig)
    z, uu, vv = ec.yy_fields()
    for degree in (0, 1, 3):
        opts = ol.default_opts(degre_interp=degree)
        zo = np.zeros(no * mo, np.float32); uo = np.zeros(no * mo, np.float32); vo = np.zeros(no * mo, np.float32)
        assert O.orc_ezyysint(sg, go, ctypes.byref(opts), ol.fptr(zo), ol.fptr(z)) == 0
        assert O.orc_ezyyuvint(sg, go, ctypes.byref(opts), ol.fptr(uo), ol.fptr(vo), ol.fptr(uu), ol.fptr(vv)) == 0
        for got, key in ((zo, "z"), (uo, "u"), (vo, "v")):
            assert np.array_equal(got.view(np.uint32), GOLD[f"YY_to_{tname}/{key}_d{degree}"].view(np.uint32)), (tname, key, degree)


def test_yinyang_target_golden():
    """a Yin-Yang TARGET is its two Z-on-E subgrids taken one after the other (ezyysint.c:79-86, :162-230): the oracle's
    composition against the reference's outputs, bit for bit"""
    O = ol.oracle()
    ni, nj = ec.YY_NI, ec.YY_NJ
    ax, ay = ec.yy_axes(ni, nj)
    tni, tnj = ec.YYT_NI, ec.YYT_NJ
    tax, tay = ec.yyt_axes(tni, tnj)
    z, _, _ = ec.yy_fields()
    zg = ec.synth_field(64, 32, seed=11)
    nsub = tni * tnj
    for degree in (0, 1, 3):
        opts = ol.default_opts(degre_interp=degree)
        got_g = np.zeros(2 * nsub, np.float32); got_y = np.zeros(2 * nsub, np.float32)
        for k, ig in enumerate((ec.YIN_IG, ec.YAN_IG)):
            tgt = ol.grid_define(tni, tnj, "Z", ig, "E", tax, tay)
            gs = O.orc_defset(tgt, ol.grid_define(64, 32, "G"))
            part = np.zeros(nsub, np.float32)
            O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(part), ol.fptr(zg)); got_g[k * nsub:(k + 1) * nsub] = part
            sg = O.orc_supergrid_define(ol.grid_define(ni, nj, "Z", ec.YIN_IG, "E", ax, ay), ol.grid_define(ni, nj, "Z", ec.YAN_IG, "E", ax, ay))
            tgt2 = ol.grid_define(tni, tnj, "Z", ig, "E", tax, tay)
            O.orc_ezyysint(sg, tgt2, ctypes.byref(opts), ol.fptr(part), ol.fptr(z)); got_y[k * nsub:(k + 1) * nsub] = part
        assert np.array_equal(got_g.view(np.uint32), GOLD[f"G_to_YY/z_d{degree}"].view(np.uint32)), degree
        assert np.array_equal(got_y.view(np.uint32), GOLD[f"YY_to_YY/z_d{degree}"].view(np.uint32)), degree


def _bit_hash(z):
    u = np.ascontiguousarray(z).reshape(-1).view(np.uint32)
    return (int(u.astype(np.uint64).sum()) & 0xFFFFFFFF, int(np.bitwise_xor.reduce(u)))


def test_cfg1_exact_size_golden():
    """BASELINE cfg1 at its exact size ('L' 400x200 -> 'L' 800x400): the oracle equals the reference's own run
    (tests/golden/make_cfg1_full.py) bit for bit, all degrees, polar correction on / off"""
    G1 = np.load(os.path.join(os.path.dirname(__file__), "golden", "cfg1_full_golden.npz"))
    O = ol.oracle()
    gi = ol.grid_define(400, 200, "L", (90, 90, 45, 0)); go = ol.grid_define(800, 400, "L", (45, 45, 0, 0))
    gs = O.orc_defset(go, gi)
    zin = ec.synth_field(400, 200, seed=1)
    for degree in (1, 3, 0):
        for polar in (1, 0):
            opts = ol.default_opts(degre_interp=degree, polar_correction=polar)
            z = np.zeros(800 * 400, np.float32)
            rc = O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(z), ol.fptr(zin))
            key = f"d{degree}_p{polar}"
            assert rc == int(G1[key + "/rc"])
            assert _bit_hash(z) == tuple(int(v) for v in G1[key + "/hash"]), key
            if degree == 1:
                assert np.array_equal(z.view(np.uint32), G1[key + "/z"].reshape(-1).view(np.uint32)), key


def test_cfg3_full_size_golden():
    """BASELINE cfg3 at full size (Z-on-E 2560x1280 -> L 4000x2000): located x,y, c_ezsint and c_ezuvint of the oracle
    equal the reference's own run (tests/golden/make_cfg3_full.py) bit for bit over all 8 M points (bicubic with polar
    correction, nearest without)"""
    G3 = np.load(os.path.join(os.path.dirname(__file__), "golden", "cfg3_full_golden.npz"))
    O = ol.oracle()
    ni, nj, no, mo = 2560, 1280, 4000, 2000
    ax, ay = ec.ze_axes(ni, nj)
    gi = ol.grid_define(ni, nj, "Z", ec.E_IG, "E", ax, ay); go = ol.grid_define(no, mo, "L", (9, 9, 0, 0))
    gs = O.orc_defset(go, gi)
    uu, vv = ec.synth_wind(ni, nj, seed=3)
    for a in (uu, vv):
        a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
    u = np.zeros(no * mo, np.float32); v = np.zeros(no * mo, np.float32); z = np.zeros(no * mo, np.float32)
    for degree, polar in ((3, 1), (0, 0)):
        opts = ol.default_opts(degre_interp=degree, polar_correction=polar)
        assert O.orc_ezuvint(gs, ctypes.byref(opts), ol.fptr(u), ol.fptr(v), ol.fptr(uu), ol.fptr(vv)) >= 0
        assert O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(z), ol.fptr(uu)) >= 0
        key = f"d{degree}_p{polar}"
        for nm, a in (("u", u), ("v", v), ("z", z)):
            assert _bit_hash(a) == tuple(int(t) for t in G3[f"{key}/{nm}/hash"]), (key, nm)
    assert _bit_hash(ol.np_from(gs.contents.x, no * mo)) == tuple(int(t) for t in G3["x/hash"])
    assert _bit_hash(ol.np_from(gs.contents.y, no * mo)) == tuple(int(t) for t in G3["y/hash"])
