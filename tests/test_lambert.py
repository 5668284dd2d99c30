"""Lambert conformal '!' grids: c_ezqkdef('!') decodes its descriptors through the GRIB grid dictionary `gribtable`
(igaxg95.F:63-110 -> qqqgbld :201-268, qgblukig :113-180), coordinates through ez_lambxyfll99 / ez_lambllfxy99 (ez_lambfll.inc, ez_llflamb.inc).
The dictionary reader of the reference needs its file layer (fnom / fclos / getenvc: App.h) and cannot run in oracle/_ref/libezref.so; the two PROJECTION
routines are compiled there from the reference's own .inc files and pin the product's REAL arithmetic directly: c_gdxyfll / c_gdll / c_gdllfxy of a '!'
grid must equal what ez_lambfll / ez_llflamb compose from them (:50-62, :43-58), bit for bit.  The fixture tests/golden/gribtable is synthetic."""
import ctypes, os
import numpy as np
import pytest

import reflib
from librmn_amd import ezscint as ez

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
os.environ.setdefault("CMCCONST", GOLD)          # read once, at the first '!' grid of the process (qqqgbld's `once`)

TABLE = {211: (12.190, -133.459, 81.2705, 81.2705, -95.0, 25.0, 25.0, 93, 65, 57.290, -49.385, 7, 0, 3),
         901: (30.000, -110.000, 25.0, 25.0, -100.0, 40.0, 50.0, 120, 90, 49.5, -78.0, 54, 2, 3),
         902: (-55.000, 20.000, 40.0, 40.0, 35.0, -30.0, -45.0, 80, 60, -25.0, 70.0, 54, 0, 3)}


def _ref_xyfll(lat, lon, latin1, latin2, yaxislat, yaxislon):
    R = reflib.ref()
    n = ctypes.c_int32(lat.size)
    x = np.zeros(lat.size, np.float32); y = np.zeros(lat.size, np.float32)
    f = lambda v: ctypes.byref(ctypes.c_float(v))
    R.ez_lambxyfll99_(reflib.fptr(x), reflib.fptr(y), reflib.fptr(np.ascontiguousarray(lat, np.float32)), reflib.fptr(np.ascontiguousarray(lon, np.float32)),
                      ctypes.byref(n), f(latin1), f(latin2), f(yaxislat), f(yaxislon))
    return x, y


def _ref_llfxy(x, y, latin1, latin2, yaxislat, yaxislon):
    R = reflib.ref()
    n = ctypes.c_int32(x.size)
    lat = np.zeros(x.size, np.float32); lon = np.zeros(x.size, np.float32)
    f = lambda v: ctypes.byref(ctypes.c_float(v))
    R.ez_lambllfxy99_(reflib.fptr(lat), reflib.fptr(lon), reflib.fptr(np.ascontiguousarray(x, np.float32)), reflib.fptr(np.ascontiguousarray(y, np.float32)),
                      ctypes.byref(n), f(latin1), f(latin2), f(yaxislat), f(yaxislon))
    return lat, lon


def _params(code):
    t = TABLE[code]
    f32 = np.float32
    lat11, lon11, dx, dy, yaxislon, latin1, latin2 = (f32(v) for v in t[:7])
    latnn = f32(t[9])
    yaxislat = f32(0.5) * (lat11 + latnn)
    return lat11, lon11, dx * f32(1000.0), dy * f32(1000.0), yaxislon, latin1, latin2, yaxislat, t[7], t[8]


@pytest.mark.skipif(not reflib.have_ref(), reason="oracle/_ref/libezref.so not built")
@pytest.mark.parametrize("code", [211, 901, 902])
def test_lambert_coordinates_against_the_reference_projection(code):
    lat11, lon11, dx, dy, yaxislon, latin1, latin2, yaxislat, ni, nj = _params(code)
    gd = ez.ezqkdef(ni, nj, "!", code, 0, 0, 0)
    assert gd >= 0
    # c_gdll: x = i + 1, y = j + 1 -> projection metres -> ez_lambllfxy99; negative longitudes + 360 (ez_calclatlon.c:229-250)
    x11, y11 = _ref_xyfll(np.array([lat11]), np.array([lon11]), latin1, latin2, yaxislat, yaxislon)
    ii, jj = np.meshgrid(np.arange(1, ni + 1, dtype=np.float32), np.arange(1, nj + 1, dtype=np.float32))
    xm = (x11[0] + dx * (ii.ravel() - np.float32(1.0))).astype(np.float32); ym = (y11[0] + dy * (jj.ravel() - np.float32(1.0))).astype(np.float32)
    wlat, wlon = _ref_llfxy(xm, ym, latin1, latin2, yaxislat, yaxislon)
    wlon = np.where(wlon < 0, wlon + np.float32(360.0), wlon).astype(np.float32)
    rc, lat, lon = ez.gdll(gd, ni * nj)
    assert rc >= 0
    assert np.array_equal(lat.view(np.uint32), wlat.view(np.uint32)) and np.array_equal(lon.view(np.uint32), wlon.view(np.uint32)), (np.abs(lat - wlat).max(), np.abs(lon - wlon).max())
    # c_gdxyfll of random points: ez_lambfll (:50-62)
    rng = np.random.default_rng(code)
    la = rng.uniform(float(lat.min()) - 2, float(lat.max()) + 2, 5000).astype(np.float32); lo = rng.uniform(-180, 360, 5000).astype(np.float32)
    xr, yr = _ref_xyfll(la, lo, latin1, latin2, yaxislat, yaxislon)
    wx = (np.float32(1.0) + (xr - x11[0]) / dx).astype(np.float32); wy = (np.float32(1.0) + (yr - y11[0]) / dy).astype(np.float32)
    rc, x, y = ez.gdxyfll(gd, la, lo)
    assert rc >= 0
    assert np.array_equal(x.view(np.uint32), wx.view(np.uint32)) and np.array_equal(y.view(np.uint32), wy.view(np.uint32)), (np.abs(x - wx).max(), np.abs(y - wy).max())
    # and back: c_gdllfxy (gdllfxy.c:172-174) returns the raw longitude of ez_llflamb
    L = ez._lib()
    L.c_gdllfxy.argtypes = [ctypes.c_int32] + [ctypes.c_void_p] * 4 + [ctypes.c_int32]
    gx = rng.uniform(1, ni, 3000).astype(np.float32); gy = rng.uniform(1, nj, 3000).astype(np.float32)
    blat = np.zeros(3000, np.float32); blon = np.zeros(3000, np.float32)
    assert L.c_gdllfxy(gd, blat.ctypes.data, blon.ctypes.data, gx.ctypes.data, gy.ctypes.data, 3000) >= 0
    w2lat, w2lon = _ref_llfxy((x11[0] + dx * (gx - np.float32(1.0))).astype(np.float32), (y11[0] + dy * (gy - np.float32(1.0))).astype(np.float32), latin1, latin2, yaxislat, yaxislon)
    assert np.array_equal(blat.view(np.uint32), w2lat.view(np.uint32)) and np.array_equal(blon.view(np.uint32), w2lon.view(np.uint32))
    # the round trip closes to REAL accuracy
    rc, x2, y2 = ez.gdxyfll(gd, blat, blon)
    assert np.abs(x2 - gx).max() < 2e-2 and np.abs(y2 - gy).max() < 2e-2


def test_lambert_descriptor_table():
    """the dictionary: first line with the GRIB code, comments skipped, a code that is missing fails at the definition; after the first coordinates the grid's ig3 / ig4
    are 256 * sub-centre + projection code and 0 (igaxg95.F:97-98 writes them through ez_llflamb's arguments)"""
    assert ez.ezqkdef(10, 10, "!", 777, 0, 0, 0) < 0
    gd = ez.ezqkdef(120, 90, "!", 901, 54, 5, 6)
    assert gd >= 0
    rc, lat, lon = ez.gdll(gd, 120 * 90)
    assert abs(lat[0] - 30.0) < 1e-3 and abs(lon[0] - 250.0) < 1e-3        # corner (1,1) is the table's (the table's (ni,nj) corner only serves the latitude of the cone's y axis)
    L = ez._lib()
    L.c_ezgprm.argtypes = [ctypes.c_int32, ctypes.c_char_p] + [ctypes.c_void_p] * 6
    t = ctypes.create_string_buffer(4)
    v = [ctypes.c_int32(0) for _ in range(6)]
    assert L.c_ezgprm(gd, t, *[ctypes.addressof(q) for q in v]) >= 0
    assert t.value[:1] == b"!" and [q.value for q in v] == [120, 90, 901, 54, 256 * 2 + 3, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("degree", ["cubic", "linear", "nearest"])
def test_lambert_scalar_interpolation_is_the_pointwise_one(degree):
    """c_ezsint towards a '!' grid = c_gdllsval of the source at the grid's lat / lon; c_ezsint FROM a '!' grid = c_gdxysval at the located x, y (extrapolation zone
    outside the regional source): both sides are pinned against the reference elsewhere; the coordinates by the test above"""
    import ezcases as ec
    assert ez.ezsetopt("interp_degree", degree) == 0
    try:
        gl = ez.ezqkdef(120, 90, "!", 901, 0, 0, 0)
        gg = ez.ezqkdef(360, 181, "G", 0, 0, 0, 0)
        z = ec.synth_field(360, 181, seed=8)
        assert ez.ezdefset(gl, gg) == 1
        rc, out = ez.ezsint(z, 120 * 90)
        assert rc >= 0
        rc, lat, lon = ez.gdll(gl, 120 * 90)
        rc, want = ez.gdllsval(gg, z, lat, lon)
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
        # '!' as the source: a lat-lon target that lies partly outside it
        zl = ec.synth_field(120, 90, seed=9)
        gt = ez.ezqkdef(200, 120, "L", 25, 25, 11500, 24000)          # 0.25 degrees from 25 N, 240 E: covers the cone grid and more
        assert ez.ezdefset(gt, gl) == 1
        assert ez.ezsetopt("extrap_degree", "value") == 0 and ez.ezsetval("extrap_value", -99.0) == 0
        rc, out2 = ez.ezsint(zl, 200 * 120)
        assert rc in (0, 2)
        rc, tlat, tlon = ez.gdll(gt, 200 * 120)
        rc, x, y = ez.gdxyfll(gl, tlat, tlon)
        inside = (x >= 1.5) & (x <= 118.5) & (y >= 1.5) & (y <= 88.5)
        rc, w2 = ez.gdxysint(zl, gl, x, y)
        assert inside.sum() > 1000 and np.array_equal(out2[inside].view(np.uint32), w2[inside].view(np.uint32))
        outside = (np.rint(x) < 1) | (np.rint(x) > 120) | (np.rint(y) < 1) | (np.rint(y) > 90)
        assert outside.any() and (out2[outside] == np.float32(-99.0)).all()
    finally:
        ez.ezsetopt("interp_degree", "cubic"); ez.ezsetopt("extrap_degree", "maximum")


# ---------------------------------------------------------------------------------------------
# winds on '!' grids: ez_lamb_llwfgdw.inc (c_gdwdfuv), ez_lamb_gdwfllw.inc (c_gduvfwd), c_ezuvint
# ---------------------------------------------------------------------------------------------
def _np_lamb_alpha(code, lon, to_wd):
    """the grid's rotation angle at a longitude, REAL: the direction (45 N, lon) -> (50 N, lon) in grid coordinates from the reference's own projection routine"""
    f32 = np.float32
    lat11, lon11, dx, dy, yaxislon, latin1, latin2, yaxislat, ni, nj = _params(code)
    x11, y11 = _ref_xyfll(np.array([lat11]), np.array([lon11]), latin1, latin2, yaxislat, yaxislon)
    xy = []
    for la in (45.0, 50.0):
        xr, yr = _ref_xyfll(np.full(lon.size, la, f32), lon, latin1, latin2, yaxislat, yaxislon)
        xy.append(((f32(1.0) + (xr - x11[0]) / dx).astype(f32), (f32(1.0) + (yr - y11[0]) / dy).astype(f32)))
    delx = (xy[1][0] - xy[0][0]).astype(f32); dely = (xy[1][1] - xy[0][1]).astype(f32)
    pie = f32(3.1415926535898)
    at = np.arctan2(dely, delx).astype(f32)
    return (pie * f32(0.5) - at).astype(f32) if to_wd else (at - f32(0.5) * pie).astype(f32)


def _np_llwfgdw(code, u, v, lon):
    """ez_lamb_llwfgdw.inc:41-80 in REAL"""
    f32 = np.float32
    a = _np_lamb_alpha(code, lon, True)
    c, s = np.cos(a).astype(f32), np.sin(a).astype(f32)
    z1 = (u * c - v * s).astype(f32); z2 = (u * s + v * c).astype(f32)
    spd = np.sqrt((z1 * z1 + z2 * z2).astype(f32)).astype(f32)
    with np.errstate(invalid="ignore"):
        d = (f32(270.0) - f32(57.295779513082) * np.arctan2(z2, z1).astype(f32)).astype(f32)
    d = np.where(z1 == 0, np.where(z2 >= 0, f32(180.0), f32(0.0)), d)
    d = np.where(spd == 0, f32(0.0), d).astype(f32)
    d = np.fmod(np.fmod(d, f32(360.0)) + f32(360.0), f32(360.0)).astype(f32)
    return spd, d


def _np_gdwfllw(code, spd, wd, lon):
    """ez_lamb_gdwfllw.inc:41-57 in REAL"""
    f32 = np.float32
    a = _np_lamb_alpha(code, lon, False)
    psi = (f32(270.0) - wd).astype(f32)
    uuu = (np.cos((psi * f32(1.7453292519943e-2)).astype(f32)).astype(f32) * spd).astype(f32)
    vvv = (np.sin((psi * f32(1.7453292519943e-2)).astype(f32)).astype(f32) * spd).astype(f32)
    c, s = np.cos(a).astype(f32), np.sin(a).astype(f32)
    return (uuu * c - vvv * s).astype(f32), (uuu * s + vvv * c).astype(f32)


def _ang_diff(a, b):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)) % 360.0
    return np.minimum(d, 360.0 - d)


@pytest.mark.gpu
@pytest.mark.skipif(not reflib.have_ref(), reason="oracle/_ref/libezref.so not built")
@pytest.mark.parametrize("code", [211, 901, 902])
def test_lambert_wind_conversions(code):
    """c_gdwdfuv / c_gduvfwd on a '!' grid against ez_lamb_llwfgdw / ez_lamb_gdwfllw restated on the reference's own projection routine; winds to 1e-5 |V|
    (the REAL sines and cosines of numpy, glibc and the device differ in the last place), directions to 1e-3 degrees"""
    t = TABLE[code]
    gd = ez.ezqkdef(t[7], t[8], "!", code, 0, 0, 0)
    assert gd >= 0
    rng = np.random.default_rng(code + 1)
    n = 20000
    lat = rng.uniform(-80, 80, n).astype(np.float32); lon = rng.uniform(0, 360, n).astype(np.float32)
    u = rng.normal(0, 12, n).astype(np.float32); v = rng.normal(0, 12, n).astype(np.float32)
    u[:50] = 0.0; v[25:75] = 0.0                                   # the uu == 0 and calm branches
    rc, spd, wd = ez.gdwdfuv(gd, u, v, lat, lon)
    assert rc >= 0
    wspd, wwd = _np_llwfgdw(code, u, v, lon)
    vmax = float(np.abs(wspd).max())
    assert np.abs(spd - wspd).max() <= 1e-5 * vmax, np.abs(spd - wspd).max()
    moving = wspd > 1e-3 * vmax
    assert _ang_diff(wd[moving], wwd[moving]).max() < 1e-3, _ang_diff(wd[moving], wwd[moving]).max()
    assert (spd[25:50] == 0).all() and (wd[25:50] == 0).all()     # calm: speed 0, direction 0 (:62-63)
    # and back (the angles of the two directions are opposite: the round trip closes)
    rc, u2, v2 = ez.gduvfwd(gd, spd, wd, lat, lon)
    assert rc >= 0
    wu, wv = _np_gdwfllw(code, spd, wd, lon)
    assert np.abs(u2 - wu).max() <= 1e-5 * vmax and np.abs(v2 - wv).max() <= 1e-5 * vmax, (np.abs(u2 - wu).max(), np.abs(v2 - wv).max())
    assert np.abs(u2 - u).max() <= 2e-4 * vmax and np.abs(v2 - v).max() <= 2e-4 * vmax
    # a wind along the grid's y axis at the cone's central meridian blows from the south: direction 180
    la0 = np.array([40.0], np.float32); lo0 = np.array([np.float32(t[4]) % np.float32(360.0)], np.float32)
    rc, s0, d0 = ez.gdwdfuv(gd, np.zeros(1, np.float32), np.full(1, 5.0, np.float32), la0, lo0)
    assert abs(float(s0[0]) - 5.0) < 1e-5 and _ang_diff(d0, np.array([180.0]))[0] < 1e-2, (s0, d0)


@pytest.mark.gpu
@pytest.mark.parametrize("degree", ["cubic", "linear"])
def test_lambert_uvint_is_interpolation_then_the_two_wind_legs(degree):
    """c_ezuvint from / towards a '!' grid (ezuvint.c:51-94): both components interpolated as scalars, c_gdwdfuv on the source grid and c_gduvfwd on the target
    grid at the target's points.  The legs are pinned by the test above, the scalars by test_lambert_scalar_interpolation_is_the_pointwise_one"""
    import ezcases as ec
    assert ez.ezsetopt("interp_degree", degree) == 0
    try:
        gl = ez.ezqkdef(120, 90, "!", 901, 0, 0, 0)
        gt = ez.ezqkdef(100, 60, "L", 20, 20, 12400, 25600)           # 0.2 degrees from 34 N, 256 E: inside the cone grid (outside it a VECTOR component is filled with 0, ez_corrval.c)
        gg = ez.ezqkdef(360, 181, "G", 0, 0, 0, 0)
        for gsrc, gdst, ns, nd in ((gl, gt, 120 * 90, 100 * 60), (gg, gl, 360 * 181, 120 * 90)):
            nis, njs = (120, 90) if gsrc == gl else (360, 181)
            u = ec.synth_field(nis, njs, seed=21) - np.float32(250.0); v = ec.synth_field(nis, njs, seed=22) - np.float32(250.0)
            assert ez.ezdefset(gdst, gsrc) == 1
            rc, uo, vo = ez.ezuvint(u, v, nd)
            assert rc >= 0, rc
            rc, us = ez.ezsint(u, nd); rc, vs = ez.ezsint(v, nd)
            rc, tlat, tlon = ez.gdll(gdst, nd)
            if gsrc == gl:
                rc, x, y = ez.gdxyfll(gl, tlat, tlon)
                assert x.min() >= 2 and x.max() <= 119 and y.min() >= 2 and y.max() <= 89, (x.min(), x.max(), y.min(), y.max())
            rc, spd, wd = ez.gdwdfuv(gsrc, us, vs, tlat, tlon)
            assert rc >= 0
            rc, wu, wv = ez.gduvfwd(gdst, spd, wd, tlat, tlon)
            assert rc >= 0
            vmax = float(np.abs(spd).max())
            assert np.abs(uo - wu).max() <= 1e-5 * vmax and np.abs(vo - wv).max() <= 1e-5 * vmax, (np.abs(uo - wu).max(), np.abs(vo - wv).max(), vmax)
            # the rotation is not the identity on this pair (the test would pass trivially otherwise)
            assert np.abs(uo - us).max() > 1e-2 * vmax
            rc, so, do = ez.ezwdint(u, v, nd)
            assert rc >= 0 and np.abs(so - spd).max() <= 1e-5 * vmax
    finally:
        ez.ezsetopt("interp_degree", "cubic")
