"""GPU parity tests: the HIP path (through the C ABI of librmn_ez_hip.so) against the CPU oracle on
the same seeded inputs and against the committed golden vectors (tests/golden/ez_golden.npz).

Tolerances (BASELINE.json north_star): float interpolation within 1e-5 relative.  Where the HIP
kernel restates the reference arithmetic operation by operation (nearest, linear, and every degree
on the per-point kernel) the comparison is bit-exact."""
import ctypes, os
from conftest import run_child
import numpy as np
import pytest

import oraclelib as ol
import ezcases as ec

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
from librmn_amd import ezscint as ez   # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ez_golden.npz"))
CASES = ec.scalar_cases()
DEGN = {0: "nearest", 1: "linear", 3: "cubic"}
RTOL = 1e-5


def hip_define(spec):
    ni, nj, grtyp, ig, grref, axes = spec
    if grtyp in ("Z", "Y"):
        ax, ay = axes(ni, nj)
        return ez.ezgdef_fmem(ni, nj, grtyp, grref, ig[0], ig[1], ig[2], ig[3], ax, ay)
    return ez.ezqkdef(ni, nj, grtyp, ig[0], ig[1], ig[2], ig[3])


def case_inputs(name, case):
    ni, nj = case["src"][:2]
    zin = ec.synth_field(ni, nj, seed=11)
    uu, vv = ec.synth_wind(ni, nj, seed=21)
    if case["src"][2] in ("Z", "B") or name == "Lrepeat_to_L":
        for a in (zin, uu, vv):
            a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
    return zin, uu, vv


RELERR_FLOOR = float(os.environ.get("EZ_RELERR_FLOOR", "0"))


def relerr(got, want):
    """the error measure of the 1e-5 bar: the PURE relative error |got - want| / |want| at every point (round 6; until then reference values below 1e-3 max|want| were
    held to an absolute 1e-8 max|want| instead -- a floor no test of this file needed: the scalars are evaluated in REAL*8 and rounded once, their relative error does
    not grow towards a zero crossing, and where the reference returns an exact 0 so does the product).  EZ_RELERR_FLOOR=f brings a floor of f x max|want| back."""
    scale = np.maximum(np.abs(want), np.abs(want).max() * RELERR_FLOOR + 1e-30)
    return np.abs(got.astype(np.float64) - want.astype(np.float64)) / scale


def err_report(got, want):
    """(max pure relative error over the points with |want| >= 1e-3 max|want|, max absolute error over the others, number of the others)"""
    w = want.astype(np.float64); d = np.abs(got.astype(np.float64) - w)
    big = np.abs(w) >= np.abs(w).max() * 1e-3
    rel = float((d[big] / np.abs(w[big])).max()) if big.any() else 0.0
    ab = float(d[~big].max()) if (~big).any() else 0.0
    return rel, ab, int((~big).sum())


def setopts(degree, polar, extrap="maximum"):
    assert ez.ezsetopt("interp_degree", DEGN[degree]) == 0
    assert ez.ezsetopt("polar_correction", "yes" if polar else "no") == 0
    assert ez.ezsetopt("extrap_degree", extrap) == 0


@pytest.fixture(autouse=True)
def _reset_opts():
    yield
    setopts(3, 1)
    os.environ.pop("EZHIP_FORCE_PTS", None)
    os.environ.pop("EZHIP_NO_SEPX", None)


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("force_pts", [0, 1, 2])      # 0: default (k_sepx where separable), 1: per-point k_pts, 2: fallback tile kernel k_sep
def test_ezsint_vs_golden(name, force_pts):
    """host-pointer c_ezsint, both kernel families, all degrees, polar correction on/off"""
    case = CASES[name]
    if force_pts == 1:
        os.environ["EZHIP_FORCE_PTS"] = "1"
    if force_pts == 2:
        os.environ["EZHIP_NO_SEPX"] = "1"
    gdin = hip_define(case["src"]); gdout = hip_define(ec.dst_spec(case))
    assert gdin >= 0 and gdout >= 0
    assert ez.ezdefset(gdout, gdin) == 1
    no, mo = case["dst"][:2]
    zin, _, _ = case_inputs(name, case)
    rotated = case["src"][2] == "Z" and case["src"][4] == "E"
    for degree in (0, 1, 3):
        for polar in (1, 0):
            setopts(degree, polar)
            mode = ez.set_mode()
            rc, z = ez.ezsint(zin, no * mo)
            want = GOLD[f"{name}/z_d{degree}_p{polar}"]
            assert rc == int(GOLD[f"{name}/rc_d{degree}_p{polar}"]), (name, degree, polar, mode)
            err = relerr(z, want)
            assert err.max() <= RTOL, (name, degree, polar, mode, float(err.max()), int(np.argmax(err)))
            # rotated sources: x,y come from the exact host locate, the per-point kernel restates the leaf kernels -> bit-exact too
            exact = (degree in (0, 1) or mode == 2) and not (polar and name not in ("Lregional_to_L",))
            if exact:
                assert np.array_equal(z.view(np.uint32), want.view(np.uint32)), (name, degree, polar, mode)


@pytest.mark.parametrize("name", [n for n in sorted(CASES) if n not in ("G_to_G", "L_to_G")])
def test_ezuvint_vs_golden(name):
    case = CASES[name]
    gdin = hip_define(case["src"]); gdout = hip_define(ec.dst_spec(case))
    ez.ezdefset(gdout, gdin)
    no, mo = case["dst"][:2]
    _, uu, vv = case_inputs(name, case)
    for degree in (0, 1, 3):
        for polar in (1, 0):
            setopts(degree, polar)
            rc, u, v = ez.ezuvint(uu, vv, no * mo)
            assert rc >= 0
            wu = GOLD[f"{name}/u_d{degree}_p{polar}"]; wv = GOLD[f"{name}/v_d{degree}_p{polar}"]
            # wind components: compare against the vector magnitude scale (a component may cross zero)
            scale = np.maximum(np.sqrt(wu.astype(np.float64) ** 2 + wv.astype(np.float64) ** 2), 1e-3)
            eu = np.abs(u - wu) / scale; ev = np.abs(v - wv) / scale
            tol = RTOL          # north_star: 1e-5 relative, no outlier allowances
            assert eu.max() <= tol and ev.max() <= tol, (name, degree, polar, float(eu.max()), float(ev.max()))


@pytest.mark.parametrize("name", [n for n in sorted(CASES) if n not in ("G_to_G", "L_to_G")])
def test_ezwdint_vs_golden(name):
    """c_ezwdint (speed / direction on the target grid), bicubic, polar correction on / off, against the reference's
    own outputs (golden fixture).  Direction is compared modulo 360 and only where the wind is not calm."""
    case = CASES[name]
    gdin = hip_define(case["src"]); gdout = hip_define(ec.dst_spec(case))
    assert ez.ezdefset(gdout, gdin) == 1
    no, mo = case["dst"][:2]
    _, uu, vv = case_inputs(name, case)
    for polar in (1, 0):
        setopts(3, polar)
        rc, spd, wd = ez.ezwdint(uu, vv, no * mo)
        assert rc in (0, 2)
        ws, wdir = GOLD[f"{name}/spd_d3_p{polar}"], GOLD[f"{name}/dir_d3_p{polar}"]
        ws_scale = np.abs(ws).max()
        serr = np.abs(spd.astype(np.float64) - ws) / np.maximum(np.abs(ws), ws_scale * 1e-3 + 1e-30)
        assert serr.max() <= RTOL, (name, polar, float(serr.max()))
        # direction: the error of an angle is measured across the wind, |V| * d(dir) against the field's speed scale
        dd = np.abs(((wd.astype(np.float64) - wdir.astype(np.float64)) + 180.0) % 360.0 - 180.0)
        cross = np.deg2rad(dd) * ws / ws_scale
        assert cross.max() <= RTOL, (name, polar, float(cross.max()), float(dd.max()))


def test_gdxysint_matches_oracle_bit_exact():
    """c_gdxysint at arbitrary x,y: every leaf kernel restated operation by operation -> bit-exact"""
    O = ol.oracle()
    rng = np.random.default_rng(7)
    for name in ("G_to_L", "Lglobal_to_L", "Lregional_to_L", "ZE_to_L", "B_to_L"):
        case = CASES[name]
        ni, nj = case["src"][:2]
        gdin = hip_define(case["src"])
        spec = case["src"]
        gi = ol.grid_define(ni, nj, spec[2], spec[3], spec[4], *(spec[5](ni, nj) if spec[5] else (None, None)))
        zin, _, _ = case_inputs(name, case)
        n = 5000
        x = rng.uniform(-1.0, ni + 2.0, n).astype(np.float32); y = rng.uniform(-1.0, nj + 2.0, n).astype(np.float32)
        for degree in (0, 1, 3):
            setopts(degree, 1)
            rc, z = ez.gdxysint(zin, gdin, x, y)
            assert rc == 0
            want = np.zeros(n, np.float32)
            O.orc_gdinterp(gi, degree, ol.fptr(want), ol.fptr(zin), ol.fptr(x), ol.fptr(y), n)
            assert np.array_equal(z.view(np.uint32), want.view(np.uint32)), (name, degree, int(np.count_nonzero(z != want)))


def test_gdllsval_gdllvval_match_oracle_bit_exact():
    """c_gdllsval / c_gdllvval = locate (c_gdxyfll) + c_gdxysval / 2 x c_gdxysint at caller-supplied lat/lon points"""
    O = ol.oracle()
    rng = np.random.default_rng(11)
    for name in ("G_to_L", "Lglobal_to_L", "Lregional_to_L"):
        case = CASES[name]
        ni, nj = case["src"][:2]
        gdin = hip_define(case["src"])
        spec = case["src"]
        gi = ol.grid_define(ni, nj, spec[2], spec[3], spec[4], *(spec[5](ni, nj) if spec[5] else (None, None)))
        zin, uu, vv = case_inputs(name, case)
        n = 3000
        lat = rng.uniform(-89.0, 89.0, n).astype(np.float32); lon = rng.uniform(0.0, 359.9, n).astype(np.float32)
        x = np.zeros(n, np.float32); y = np.zeros(n, np.float32)
        O.orc_gdxyfll(gi, ol.fptr(x), ol.fptr(y), ol.fptr(lat.copy()), ol.fptr(lon.copy()), n)
        for degree in (1, 3):
            setopts(degree, 1)
            rc, z = ez.gdllsval(gdin, zin, lat.copy(), lon.copy())
            assert rc == 0
            want = np.zeros(n, np.float32)
            O.orc_gdinterp(gi, degree, ol.fptr(want), ol.fptr(zin), ol.fptr(x), ol.fptr(y), n)
            assert np.array_equal(z.view(np.uint32), want.view(np.uint32)), (name, degree)
            rc, u, v = ez.gdllvval(gdin, uu, vv, lat.copy(), lon.copy())
            assert rc == 0
            wu = np.zeros(n, np.float32); wv = np.zeros(n, np.float32)
            O.orc_gdinterp(gi, degree, ol.fptr(wu), ol.fptr(uu), ol.fptr(x), ol.fptr(y), n)
            O.orc_gdinterp(gi, degree, ol.fptr(wv), ol.fptr(vv), ol.fptr(x), ol.fptr(y), n)
            assert np.array_equal(u.view(np.uint32), wu.view(np.uint32)) and np.array_equal(v.view(np.uint32), wv.view(np.uint32)), (name, degree)


def test_device_locate_matches_host_locate():
    """k_locate vs the exact host locate for non-rotated sources: bit-exact x,y"""
    for name in ("G_to_L", "Lglobal_to_L", "Lregional_to_L", "A_to_L", "B_to_L"):
        case = CASES[name]
        gdin = hip_define(case["src"])
        lat = GOLD[f"{name}/lat"].copy(); lon = GOLD[f"{name}/lon"].copy()
        n = lat.size
        d_lat = torch.from_numpy(lat).cuda(); d_lon = torch.from_numpy(lon).cuda()
        d_x = torch.empty(n, dtype=torch.float32, device="cuda"); d_y = torch.empty_like(d_x)
        ez.use_stream(0)
        assert ez.gdxyfll_dev(gdin, d_x, d_y, d_lat, d_lon, n) == 0
        torch.cuda.synchronize()
        assert np.array_equal(d_x.cpu().numpy().view(np.uint32), GOLD[f"{name}/x"].view(np.uint32)), name
        assert np.array_equal(d_y.cpu().numpy().view(np.uint32), GOLD[f"{name}/y"].view(np.uint32)), name


def test_full_size_cfg2_properties():
    """BASELINE cfg2 (G 4400x2200 -> L 7200x3601 bicubic) at full size, device-resident:
    size-independent properties + sampled points against the oracle's leaf kernel."""
    ni, nj, no, mo = 4400, 2200, 7200, 3601
    gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
    ez.ezdefset(gdout, gdin)
    setopts(3, 1)
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    d_out = torch.empty(no * mo, dtype=torch.float32, device="cuda")
    # 1. constants are reproduced (weights sum to 1 up to rounding).  The polar rows are excluded:
    #    the reference's pole value is a SEQUENTIAL float sum / ni, which is itself inexact for a
    #    constant row (273.1378 instead of 273.15 at ni = 4400) -- reproduced, and checked in step 4.
    d_in = torch.full((ni * nj,), 273.15, dtype=torch.float32, device="cuda")
    assert ez.ezsint_dev(d_out, d_in) == 0
    torch.cuda.synchronize()
    assert float((d_out.view(mo, no)[4:-4] - 273.15).abs().max()) <= 273.15 * 2e-7
    # 2. linearity: interp(a*f + b*g) == a*interp(f) + b*interp(g) within rounding
    f = torch.from_numpy(ec.synth_field(ni, nj, seed=2)).cuda(); g = torch.from_numpy(ec.synth_field(ni, nj, seed=9)).cuda()
    of = torch.empty_like(d_out); og = torch.empty_like(d_out); oc = torch.empty_like(d_out)
    ez.ezsint_dev(of, f); ez.ezsint_dev(og, g); ez.ezsint_dev(oc, 0.25 * f + 0.75 * g)
    torch.cuda.synchronize()
    lin = (oc - (0.25 * of + 0.75 * og)).abs().max() / oc.abs().max()
    assert float(lin) <= 2e-6
    # 3. all pole-row points carry one value; output within the source range (cubic overshoot bounded)
    o2 = of.view(mo, no)
    assert float(o2[0].max() - o2[0].min()) == 0.0 and float(o2[-1].max() - o2[-1].min()) == 0.0
    span = float(f.max() - f.min())
    assert float(of.max()) <= float(f.max()) + 0.2 * span and float(of.min()) >= float(f.min()) - 0.2 * span
    # 4. sampled points (incl. seam columns and the rows next to both poles) vs the oracle leaf kernel
    O = ol.oracle()
    gi = ol.grid_define(ni, nj, "G"); go = ol.grid_define(no, mo, "L", (5, 5, 0, 0))
    gs = O.orc_defset(go, gi)
    rows = np.array([0, 1, 2, 3, 4, 17, 1800, 1801, 3596, 3597, 3598, 3599, 3600])
    cols = np.array([0, 1, 2, 3, 100, 3599, 3600, 7190, 7196, 7197, 7198, 7199])
    # the oracle needs the whole located set once (25.9 M points, a few seconds)
    want = np.zeros(no * mo, np.float32)
    opts = ol.default_opts()
    zin = f.cpu().numpy()
    O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(want), ol.fptr(zin))
    got = of.cpu().numpy()
    err = relerr(got, want)
    assert err.max() <= RTOL, float(err.max())
    sub = np.ix_(rows, cols)
    assert relerr(got.reshape(mo, no)[sub], want.reshape(mo, no)[sub]).max() <= 2e-7   # <= 1 ulp incl. pole rows and seams
    frac_exact = np.count_nonzero(got == want) / got.size
    assert frac_exact > 0.999, frac_exact


@pytest.mark.parametrize("degree", [0, 1, 3])
@pytest.mark.parametrize("polar", [0, 1])
def test_batch_launch_equals_field_by_field(degree, polar):
    """c_ezsint_batch_dev (one k_sepx launch for all fields, blockIdx.z = field, pole values of the
    whole batch from one k_polevals launch) == c_ezsint_dev field by field, bit for bit."""
    ni, nj, no, mo = 360, 181, 500, 251
    gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 72, 72, 0, 0)   # 0.72 deg global target
    ez.ezdefset(gdout, gdin)
    setopts(degree, polar)
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    F = 5
    d_in = torch.stack([torch.from_numpy(ec.synth_field(ni, nj, seed=40 + f)) for f in range(F)]).cuda().contiguous()
    d_b = torch.full((F, no * mo), -1.0, dtype=torch.float32, device="cuda")
    d_s = torch.full((F, no * mo), -2.0, dtype=torch.float32, device="cuda")
    assert ez.ezsint_batch_dev(d_b, d_in, F) == 0
    for f in range(F):
        assert ez.ezsint_dev(d_s[f], d_in[f]) == 0
    torch.cuda.synchronize()
    assert torch.equal(d_b, d_s)
    # and against the oracle for one field of the batch
    O = ol.oracle()
    gi = ol.grid_define(ni, nj, "G"); go = ol.grid_define(no, mo, "L", (72, 72, 0, 0))
    gs = O.orc_defset(go, gi)
    opts = ol.default_opts(); opts.degre_interp = degree; opts.polar_correction = polar
    want = np.zeros(no * mo, np.float32)
    zin = d_in[3].cpu().numpy()
    O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(want), ol.fptr(zin))
    assert relerr(d_b[3].cpu().numpy(), want).max() <= RTOL


def _zl_axes(ni, nj, lon0, lon1, lat0, lat1, seed):
    """irregular (stretched) lat-lon axes for a 'Z' grid on an 'L' reference"""
    u = ec.hash_uniform(seed, ni + nj).astype(np.float64)
    fx = np.cumsum(1.0 + 0.6 * (u[:ni] - 0.5)); fx = (fx - fx[0]) / (fx[-1] - fx[0])
    fy = np.cumsum(1.0 + 0.6 * (u[ni:] - 0.5)); fy = (fy - fy[0]) / (fy[-1] - fy[0])
    return (lon0 + (lon1 - lon0) * fx).astype(np.float32), (lat0 + (lat1 - lat0) * fy).astype(np.float32)


# (name, source (ni, nj, grtyp, ig, grref, axes), target (...)) -- shapes the small golden cases do not reach:
# several 256-column strips with a partial last one, many 16-row blocks, ring wrap over many steps, strips that
# cross the longitude seam, up- and down-sampling, irregular source and target axes, regional sources (DEHORS fill)
SEPX_SHAPES = [
    ("G_up",        (360, 180, "G", (0, 0, 0, 0)),                 (777, 391, "L", (46, 46, 0, 0))),          # 1.64 / 2.2 x up-sampling, global
    ("G_down",      (720, 360, "G", (0, 0, 0, 0)),                 (300, 151, "L", (120, 120, 0, 0))),        # 2.4 x down-sampling
    ("L_up_offset", (400, 201, "L", (90, 90, 0, 0)),               (1030, 520, "L", (30, 30, 1200, 5000))),   # target window 12S..168N?? clipped by the L definition
    ("Zregional",   (300, 200, "Z", (100, 100, 0, 0), "L", lambda ni, nj: _zl_axes(ni, nj, 200.0, 300.0, 10.0, 70.0, 5)),
                    (600, 330, "L", (20, 20, 9500, 19000))),                                                   # regional Z source: DEHORS columns and rows
    ("L_same_res",  (500, 300, "L", (10, 10, 6000, 20000)),        (700, 420, "L", (10, 10, 5500, 19500))),   # equal resolution: 16 target rows span a 20-row source window -> ring + patch > 64 KB of LDS; regional (DEHORS fill)
    ("Zglobal_tgt", (360, 180, "G", (0, 0, 0, 0)),
                    (500, 300, "Z", (100, 100, 0, 0), "L", lambda ni, nj: _zl_axes(ni, nj, 0.0, 359.0, -88.0, 88.0, 6))),   # irregular target axes
]


@pytest.mark.parametrize("shape", SEPX_SHAPES, ids=[s[0] for s in SEPX_SHAPES])
@pytest.mark.parametrize("batch", [1, 3])
def test_sepx_shapes_vs_oracle(shape, batch):
    """k_sepx (default separable kernel) on mid-size shapes against the CPU oracle, all degrees, polar on/off,
    single-field and batch launches; nearest and bilinear bit-exact, bicubic <= 1e-5 relative."""
    name, src, dst = shape

    def define(spec, hip):
        ni, nj, grtyp, ig = spec[:4]
        grref = spec[4] if len(spec) > 4 else " "
        axes = spec[5](ni, nj) if len(spec) > 5 else (None, None)
        if hip:
            if grtyp in ("Z", "Y"):
                return ez.ezgdef_fmem(ni, nj, grtyp, grref, ig[0], ig[1], ig[2], ig[3], axes[0], axes[1])
            return ez.ezqkdef(ni, nj, grtyp, ig[0], ig[1], ig[2], ig[3])
        return ol.grid_define(ni, nj, grtyp, ig, grref, axes[0], axes[1])

    gdin, gdout = define(src, True), define(dst, True)
    assert gdin >= 0 and gdout >= 0
    assert ez.ezdefset(gdout, gdin) == 1
    O = ol.oracle()
    gs = O.orc_defset(define(dst, False), define(src, False))
    ni, nj = src[:2]; no, mo = dst[:2]
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    fields = [ec.synth_field(ni, nj, seed=70 + f) for f in range(batch)]
    d_in = torch.stack([torch.from_numpy(f) for f in fields]).cuda().contiguous()
    for degree in (0, 1, 3):
        for polar in (1, 0):
            setopts(degree, polar)
            mode = ez.set_mode()         # 1 separable (k_sepx); 2 per point (e.g. ez_irgdint_3_nw: regional irregular cubic)
            d_out = torch.full((batch, no * mo), -7.0, dtype=torch.float32, device="cuda")
            rc = ez.ezsint_batch_dev(d_out, d_in, batch) if batch > 1 else ez.ezsint_dev(d_out[0], d_in[0])
            assert rc in (0, 2)
            torch.cuda.synchronize()
            got = d_out.cpu().numpy()
            opts = ol.default_opts(); opts.degre_interp = degree; opts.polar_correction = polar
            for f in range(batch):
                want = np.zeros(no * mo, np.float32)
                O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(want), ol.fptr(fields[f]))
                err = relerr(got[f], want)
                assert err.max() <= RTOL, (name, degree, polar, mode, f, float(err.max()), int(np.argmax(err)))
                if degree in (0, 1) and not polar:
                    assert np.array_equal(got[f].view(np.uint32), want.view(np.uint32)), (name, degree, polar, f)


def test_two_host_threads_share_a_grid_set():
    """the reference's threading contract: grids defined once, then every thread calls c_ezdefset + c_ezsint
    on its own.  Two host threads, each with its own HIP stream, interpolate different fields on the SAME set
    concurrently, first use (plan build) included; every result is checked against the oracle."""
    import threading
    ni, nj, no, mo = 360, 181, 777, 391
    gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 47, 47, 0, 0)
    O = ol.oracle()
    gs = O.orc_defset(ol.grid_define(no, mo, "L", (47, 47, 0, 0)), ol.grid_define(ni, nj, "G"))
    setopts(3, 1)
    res, errs = {}, []

    def work(tid):
        try:
            st = torch.cuda.Stream()
            assert ez.ezdefset(gdout, gdin) == 1            # thread-local current set and options
            setopts(3, 1)
            ez.use_stream(st.cuda_stream)
            outs = []
            with torch.cuda.stream(st):
                for k in range(5):
                    f = ec.synth_field(ni, nj, seed=500 + 10 * tid + k)
                    d_in = torch.from_numpy(f).cuda()
                    d_out = torch.empty(no * mo, dtype=torch.float32, device="cuda")
                    assert ez.ezsint_dev(d_out, d_in) == 0
                    outs.append((f, d_out))
                st.synchronize()
            res[tid] = outs
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
    opts = ol.default_opts()
    for tid in res:
        for f, d_out in res[tid]:
            want = np.zeros(no * mo, np.float32)
            O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(want), ol.fptr(f))
            assert relerr(d_out.cpu().numpy(), want).max() <= RTOL, tid
    ez.use_stream(torch.cuda.current_stream().cuda_stream)


FULL = os.path.join(os.path.dirname(__file__), "golden", "cfg2_full_golden.npz")


def _bit_hash(t):
    u = t.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    x = 0
    for part in torch.chunk(t.view(torch.int32), 64):
        x ^= int(np.bitwise_xor.reduce(part.cpu().numpy().view(np.uint32)))
    return int(u.sum().item()) & 0xFFFFFFFF, x


@pytest.mark.parametrize("fname", ["probe", "synth"])
def test_full_size_cfg2_against_reference_run(fname):
    """BASELINE cfg2 at FULL size against the reference's own run of it (tests/golden/make_cfg2_full.py):
    sampled rows/columns (both pole rows, the rows next to them, the seam columns), the float64 sum of
    all 25.9 M points and -- where the HIP path is bit-exact (nearest, linear; polar correction off) --
    a bit hash of the whole output.  'probe' is the survey's drv2 field."""
    G = np.load(FULL)
    ni, nj, no, mo = 4400, 2200, 7200, 3601
    if fname == "probe":
        zin = (np.float32(280) + (np.float32(20) * G["probe_sin"])[None, :] * G["probe_cos"][:, None]).astype(np.float32)
    else:
        zin = ec.synth_field(ni, nj, seed=2)
    gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
    assert ez.ezdefset(gdout, gdin) == 1
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    d_in = torch.from_numpy(np.ascontiguousarray(zin).ravel()).cuda()
    d_out = torch.empty(no * mo, dtype=torch.float32, device="cuda")
    rows = torch.from_numpy(G["rows"]).cuda(); cols = torch.from_numpy(G["cols"]).cuda()
    for degree in (3, 1, 0):
        for polar in (1, 0):
            setopts(degree, polar)
            assert ez.ezsint_dev(d_out, d_in) == 0
            torch.cuda.synchronize()
            key = f"{fname}/d{degree}_p{polar}"
            o2 = d_out.view(mo, no)
            for got, want in ((o2[rows].cpu().numpy(), G[key + "/rows"]), (o2[:, cols].cpu().numpy(), G[key + "/cols"])):
                assert relerr(got, want).max() <= RTOL, (key, float(relerr(got, want).max()))
                rel, ab, nsmall = err_report(got, want)          # the two parts of that measure, printed (pytest -s): pure relative / absolute at small values
                print(f"{key}: max pure relative error {rel:.3e} (|want| >= 1e-3 max), max absolute error {ab:.3e} at the {nsmall} smaller values")
                assert rel <= RTOL and ab <= RTOL * 1e-3 * float(np.abs(want).max())
                if degree in (0, 1) and not polar:
                    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), key
            s = float(d_out.double().sum().item())
            assert abs(s - float(G[key + "/sum"])) <= 1e-8 * abs(s), (key, s, float(G[key + "/sum"]))
            if degree in (0, 1) and not polar:
                assert _bit_hash(d_out) == tuple(int(v) for v in G[key + "/hash"]), key
    if fname == "probe":        # the survey's printed anchors of this exact run
        setopts(3, 1)
        ez.ezsint_dev(d_out, d_in); torch.cuda.synchronize()
        f = d_out.cpu().numpy()
        assert "%.8e %.3f %.3f %.3f" % (f.astype(np.float64).sum(), f[0], f[f.size // 2], f[-1]) == "7.26056487e+09 281.818 276.863 278.271"


PS_CASES = ["N_to_L", "L_to_N", "S_to_L", "G_to_S", "N_to_N"]


@pytest.mark.parametrize("name", PS_CASES)
def test_polar_stereographic_coordinates_and_locate(name):
    """N / S grids: c_gdll of the target (GRPS, host) and c_gdxyfll on the source (ez_vxyfll, host)
    bit-exact against the reference's values; the device locate (double sin/cos/sqrt of the device library) within 1 ulp"""
    case = CASES[name]
    gdin = hip_define(case["src"]); gdout = hip_define(ec.dst_spec(case))
    no, mo = case["dst"][:2]
    rc, lat, lon = ez.gdll(gdout, no * mo)
    assert rc == 0
    assert np.array_equal(lat, GOLD[f"{name}/lat"])           # value equality: an 'L' target's first longitude is -0.0 in the reference
    assert np.array_equal(lon, GOLD[f"{name}/lon"])
    if case["dst"][2] in ("N", "S"):
        assert np.array_equal(lat.view(np.uint32), GOLD[f"{name}/lat"].view(np.uint32))
        assert np.array_equal(lon.view(np.uint32), GOLD[f"{name}/lon"].view(np.uint32))
    rc, x, y = ez.gdxyfll(gdin, lat.copy(), lon.copy())
    assert rc == 0
    if case["src"][2] in ("N", "S"):
        assert np.array_equal(x.view(np.uint32), GOLD[f"{name}/x"].view(np.uint32))
        assert np.array_equal(y.view(np.uint32), GOLD[f"{name}/y"].view(np.uint32))
    d_lat = torch.from_numpy(lat).cuda(); d_lon = torch.from_numpy(lon).cuda()
    d_x = torch.empty(no * mo, dtype=torch.float32, device="cuda"); d_y = torch.empty_like(d_x)
    ez.use_stream(0)
    assert ez.gdxyfll_dev(gdin, d_x, d_y, d_lat, d_lon, no * mo) == 0
    torch.cuda.synchronize()
    for got, want in ((d_x.cpu().numpy(), GOLD[f"{name}/x"]), (d_y.cpu().numpy(), GOLD[f"{name}/y"])):
        assert np.abs(got.astype(np.float64) - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), name


@pytest.mark.parametrize("tname", sorted(ec.yy_targets()))
def test_yinyang_source_vs_golden(tname):
    """Yin-Yang 'U' source (two Z-on-E subgrids, c_ezgdef_supergrid) -> L / G / N target through c_ezsint / c_ezuvint:
    mask and point lists on the host (bit-exact locate), per-point kernel per subgrid, merge on the device.
    Scalars are bit-exact (the per-point kernel restates the leaf kernels); winds within 1e-5 (device trig)."""
    ni, nj = ec.YY_NI, ec.YY_NJ
    ax, ay = ec.yy_axes(ni, nj)
    gy = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.YIN_IG, ax, ay); ga = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.YAN_IG, ax, ay)
    gu = ez.ezgdef_supergrid(ni, 2 * nj, "U", "F", 1, [gy, ga])
    no, mo, gt, ig = ec.yy_targets()[tname]
    go = ez.ezqkdef(no, mo, gt, *ig)
    assert gu >= 0 and go >= 0 and ez.ezdefset(go, gu) == 1
    assert ez.ezgdef_supergrid(ni, 2 * nj, "U", "F", 1, [gy, ga]) == gu          # identical definitions dedupe
    z, uu, vv = ec.yy_fields()
    for degree in (0, 1, 3):
        setopts(degree, 1)
        rc, got = ez.ezsint(z, no * mo)
        assert rc == 0
        want = GOLD[f"YY_to_{tname}/z_d{degree}"]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (tname, degree, float(relerr(got, want).max()))
        rc, gu_, gv = ez.ezuvint(uu, vv, no * mo)
        assert rc == 0
        wu = GOLD[f"YY_to_{tname}/u_d{degree}"]; wv = GOLD[f"YY_to_{tname}/v_d{degree}"]
        spd = np.sqrt(wu.astype(np.float64) ** 2 + wv.astype(np.float64) ** 2)
        tol = RTOL * np.maximum(spd, spd.max() * 1e-3)
        assert np.all(np.abs(gu_ - wu) <= tol) and np.all(np.abs(gv - wv) <= tol), (tname, degree)
    # c_ezwdint from the 'U' source (c_ezyywdint): the merged speed / direction
    setopts(3, 1)
    rc, gs, gd = ez.ezwdint(uu, vv, no * mo)
    ws = GOLD[f"YY_to_{tname}/spd_d3"]; wd = GOLD[f"YY_to_{tname}/dir_d3"]
    assert rc == 0 and np.all(np.abs(gs - ws) <= RTOL * np.maximum(ws, ws.max() * 1e-3))
    ddir = np.abs(((gd - wd + 180.0) % 360.0) - 180.0)
    assert np.all(ddir[ws > 1e-2] <= 5e-3), float(ddir.max())
    # use_1subgrid: the caller names ONE subgrid as the source (ezyysint.c:99-123) == a plain interpolation from that subgrid
    if tname == "L":
        assert ez.ezsetopt("use_1subgrid", "yes") == 0 and ez.ezsetival("subgridid", ga) == 0
        try:
            rc1, got1 = ez.ezsint(z, no * mo)
        finally:
            ez.ezsetopt("use_1subgrid", "no")
        assert ez.ezdefset(go, ga) == 1
        rc2, want1 = ez.ezsint(z[ni * nj:], no * mo)
        assert rc1 == rc2 and np.array_equal(got1.view(np.uint32), want1.view(np.uint32))
        assert ez.ezdefset(go, gu) == 1
    # device-resident call and a batch of 2 fields
    d_in = torch.from_numpy(np.stack([z, z[::-1].copy()])).cuda(); d_out = torch.empty((2, no * mo), dtype=torch.float32, device="cuda")
    setopts(3, 1)
    ez.use_stream(0)
    assert ez.ezsint_batch_dev(d_out, d_in, 2) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_out[0].cpu().numpy().view(np.uint32), GOLD[f"YY_to_{tname}/z_d3"].view(np.uint32))


def test_yinyang_target_vs_golden():
    """a Yin-Yang 'U' TARGET (ezyysint.c:79-86, :162-230): from an ordinary grid = two plain interpolations to its Z-on-E
    subgrids; from another 'U' grid = per target subgrid its own mask and point lists.  Scalars against the reference's
    outputs; winds towards the rotated subgrids are refused."""
    ni, nj = ec.YY_NI, ec.YY_NJ
    ax, ay = ec.yy_axes(ni, nj)
    gy = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.YIN_IG, ax, ay); ga = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.YAN_IG, ax, ay)
    gu = ez.ezgdef_supergrid(ni, 2 * nj, "U", "F", 1, [gy, ga])
    tni, tnj = ec.YYT_NI, ec.YYT_NJ
    tax, tay = ec.yyt_axes(tni, tnj)
    ty = ez.ezgdef_fmem(tni, tnj, "Z", "E", *ec.YIN_IG, tax, tay); ta = ez.ezgdef_fmem(tni, tnj, "Z", "E", *ec.YAN_IG, tax, tay)
    tu = ez.ezgdef_supergrid(tni, 2 * tnj, "U", "F", 1, [ty, ta])
    gsrc = ez.ezqkdef(64, 32, "G", 0, 0, 0, 0)
    zg = ec.synth_field(64, 32, seed=11)
    z, uu, vv = ec.yy_fields()
    nout = 2 * tni * tnj
    for degree in (0, 1, 3):
        setopts(degree, 1)
        assert ez.ezdefset(tu, gsrc) == 1
        rc, got = ez.ezsint(zg, nout)
        want = GOLD[f"G_to_YY/z_d{degree}"]
        assert rc >= 0 and relerr(got, want).max() <= RTOL, (degree, float(relerr(got, want).max()))
        assert ez.ezdefset(tu, gu) == 1
        rc, got = ez.ezsint(z, nout)
        want = GOLD[f"YY_to_YY/z_d{degree}"]
        assert rc >= 0
        if degree == 0:      # a nearest-neighbour pick may flip at a cell edge (device trig in the rotated locate of the G source case only)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert relerr(got, want).max() <= RTOL, (degree, float(relerr(got, want).max()))
    rc, lat, lon = ez.gdll(tu, nout)
    assert rc == 0 and lat.size == nout
    # winds towards the 'U' grid: c_ezgfwfllw on each rotated subgrid (ezyyuvint.c), from the G grid and from the other 'U' grid
    ug, vg = ec.synth_wind(64, 32, seed=21)
    for degree in (1, 3):
        setopts(degree, 1)
        for src, a, b, key in ((gsrc, ug, vg, "G_to_YY"), (gu, uu, vv, "YY_to_YY")):
            assert ez.ezdefset(tu, src) == 1
            rc, gu_, gv_ = ez.ezuvint(a, b, nout)
            wu = GOLD[f"{key}/u_d{degree}"]; wv = GOLD[f"{key}/v_d{degree}"]
            spd = np.sqrt(wu.astype(np.float64) ** 2 + wv.astype(np.float64) ** 2)
            tol = RTOL * np.maximum(spd, spd.max() * 1e-3)
            assert rc >= 0 and np.all(np.abs(gu_ - wu) <= tol) and np.all(np.abs(gv_ - wv) <= tol), (key, degree)


def _mask_field(ni, nj, seed):
    h = ec.hash_uniform(seed, ((ni + 1) // 2) * ((nj + 1) // 2)).reshape((nj + 1) // 2, (ni + 1) // 2)
    m = (np.repeat(np.repeat(h, 2, axis=0), 2, axis=1)[:nj, :ni] > 0.25).astype(np.int32)
    return np.ascontiguousarray(m.reshape(-1))


@pytest.mark.parametrize("name", ["Lregional_to_L", "G_to_L", "N_to_L", "L_to_N", "G_to_Y"])
@pytest.mark.parametrize("alg", ["linear", "distance"])
def test_masks_match_oracle(name, alg):
    """c_ezsint_mask / c_ezget_mask_zones / c_ezsint_mdm / c_gdsetmask (ez_mask.c) on the device against the oracle
    (itself pinned against the reference build): integer masks identical, the filled field bit-exact"""
    case = CASES[name]
    ni, nj = case["src"][:2]; no, mo = case["dst"][:2]
    gdin = hip_define(case["src"]); gdout = hip_define(ec.dst_spec(case))
    assert ez.ezdefset(gdout, gdin) == 1
    setopts(1, 1)
    assert ez.ezsetopt("cloud_interp_alg", alg) == 0
    zin, _, _ = case_inputs(name, case)
    mask_in = _mask_field(ni, nj, seed=ni + nj)
    try:
        rc, zo, mo_ = ez.ezsint_mdm(zin, mask_in, no * mo)
        assert rc == 0
        rc, mz = ez.ezget_mask_zones(mask_in, no * mo)
        assert rc == 0
        rc, m2 = ez.ezsint_mask(mask_in, no * mo)
        assert rc == 0 and np.array_equal(m2, mo_)
    finally:
        ez.ezsetopt("cloud_interp_alg", "distance")
    O = ol.oracle()
    O.orc_ezsint_mask.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    O.orc_ezget_mask_zones.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    O.orc_mask_fill2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    import test_oracle_golden as tog
    gi = tog.orc_define(case["src"]); go = tog.orc_define(ec.dst_spec(case))
    gs = O.orc_defset(go, gi)
    opts = ol.default_opts(degre_interp=1)
    want = np.zeros(no * mo, np.float32); wm = np.zeros(no * mo, np.int32); wz = np.zeros(no * mo, np.int32)
    O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(want), ol.fptr(zin))
    O.orc_ezsint_mask(ctypes.cast(gs, ctypes.c_void_p), int(alg == "linear"), wm.ctypes.data, mask_in.ctypes.data)
    O.orc_mask_fill2(want.ctypes.data, wm.ctypes.data, no * mo)
    O.orc_ezget_mask_zones(ctypes.cast(gs, ctypes.c_void_p), wz.ctypes.data, mask_in.ctypes.data)
    assert np.array_equal(mo_, wm), (name, alg, int((mo_ != wm).sum()))
    assert np.array_equal(mz, wz), (name, alg)
    assert np.array_equal(zo.view(np.uint32), want.view(np.uint32)), (name, alg)
    # the grid-attached mask is just stored and returned
    assert ez.gdsetmask(gdin, mask_in) == 0
    rc, back = ez.gdgetmask(gdin, ni * nj)
    assert rc == 0 and np.array_equal(back, mask_in)


@pytest.mark.parametrize("kind", ["L", "N", "S", "ZE", "G"])
def test_gdwdfuv_gduvfwd_match_oracle(kind):
    """c_gdwdfuv / c_gduvfwd on their own (scattered points) against the oracle (pinned against the reference build):
    non-rotated grids are the reference's REAL arithmetic with the device's float trig (<= 1e-5), rotated sources idem"""
    import test_oracle_golden as tog
    spec = {"L": (40, 20, "L", (900, 900, 450, 0), " ", None), "N": (101, 91, "N", ec.N_IG, " ", None), "S": (81, 121, "S", ec.S_IG, " ", None),
            "ZE": (65, 32, "Z", ec.E_IG, "E", ec.ze_axes), "G": (64, 32, "G", (0, 0, 0, 0), " ", None)}[kind]
    g = hip_define(spec); og = tog.orc_define(spec)
    O = ol.oracle()
    n = 777
    lat = (ec.hash_uniform(15, n).astype(np.float64) * 170.0 - 85.0).astype(np.float32)
    lon = (ec.hash_uniform(16, n).astype(np.float64) * 360.0).astype(np.float32)
    uu = ((ec.hash_uniform(17, n) - 0.5) * 60).astype(np.float32); vv = ((ec.hash_uniform(18, n) - 0.5) * 60).astype(np.float32)
    uu[:3] = 0.0; vv[1] = 0.0
    ws = np.zeros(n, np.float32); wd = np.zeros(n, np.float32)
    O.orc_gdwdfuv(og, ol.fptr(ws), ol.fptr(wd), ol.fptr(uu), ol.fptr(vv), ol.fptr(lat), ol.fptr(lon), n)
    rc, gs, gd = ez.gdwdfuv(g, uu, vv, lat, lon)
    assert rc == 0
    assert np.all(np.abs(gs - ws) <= 1e-5 * np.maximum(ws, 1e-3))
    ddir = np.abs(((gd - wd + 180.0) % 360.0) - 180.0)
    assert np.all(ddir[ws > 1e-3] <= 2e-3), float(ddir.max())           # degrees: 1e-5 of a full turn
    wu = np.zeros(n, np.float32); wv = np.zeros(n, np.float32)
    O.orc_gduvfwd(og, ol.fptr(wu), ol.fptr(wv), ol.fptr(ws), ol.fptr(wd), ol.fptr(lat), ol.fptr(lon), n)
    rc, gu, gv = ez.gduvfwd(g, ws, wd, lat, lon)                      # 'ZE': c_ezgfwfllw towards the rotated frame
    assert rc == 0
    tol = 1e-5 * np.maximum(ws, 1e-3) + 1e-6
    assert np.all(np.abs(gu - wu) <= tol) and np.all(np.abs(gv - wv) <= tol)


@pytest.mark.parametrize("kind", ["L", "N", "ZE", "G"])
def test_gdxywdval_gdllwdval_match_oracle(kind):
    """c_gdxywdval = c_gdxyvval + c_gdllfxy + c_gdwdfuv, c_gdllwdval = c_gdllvval + c_gdwdfuv (gdxywdval.c:38, gdllwdval.c:36)"""
    import test_oracle_golden as tog
    spec = {"L": (40, 20, "L", (900, 900, 450, 0), " ", None), "N": (101, 91, "N", ec.N_IG, " ", None),
            "ZE": (65, 32, "Z", ec.E_IG, "E", ec.ze_axes), "G": (64, 32, "G", (0, 0, 0, 0), " ", None)}[kind]
    ni, nj = spec[:2]
    g = hip_define(spec); og = tog.orc_define(spec)
    O = ol.oracle()
    n = 500
    x = (ec.hash_uniform(35, n).astype(np.float64) * (ni - 3.0) + 2.0).astype(np.float32)
    y = (ec.hash_uniform(36, n).astype(np.float64) * (nj - 3.0) + 2.0).astype(np.float32)
    uu, vv = ec.synth_wind(ni, nj, seed=9)
    if spec[2] == "Z":
        for a in (uu, vv):
            a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
    setopts(3, 1)
    rc, lat, lon = ez.gdllfxy(g, x, y)
    assert rc == 0
    iu = np.zeros(n, np.float32); iv = np.zeros(n, np.float32); ws = np.zeros(n, np.float32); wd = np.zeros(n, np.float32)
    O.orc_gdinterp(og, 3, ol.fptr(iu), ol.fptr(uu), ol.fptr(x), ol.fptr(y), n)
    O.orc_gdinterp(og, 3, ol.fptr(iv), ol.fptr(vv), ol.fptr(x), ol.fptr(y), n)
    O.orc_gdwdfuv(og, ol.fptr(ws), ol.fptr(wd), ol.fptr(iu), ol.fptr(iv), ol.fptr(lat), ol.fptr(lon), n)
    for rc, gs, gd in (ez.gdxywdval(g, uu, vv, x, y), ez.gdllwdval(g, uu, vv, lat, lon)):
        assert rc == 0
        assert np.all(np.abs(gs - ws) <= 2e-5 * np.maximum(ws, 1e-2)), float(np.abs(gs - ws).max())
        ddir = np.abs(((gd - wd + 180.0) % 360.0) - 180.0)
        assert np.all(ddir[ws > 1e-2] <= 5e-3), float(ddir.max())


HEMI = ec.hemi_cases()


@pytest.mark.parametrize("name", sorted(HEMI))
def test_hemispheric_scalar_vs_golden(name):
    """hemispheric A / B grids (ig1 = 1, 2): the source field is mirrored into the other hemisphere on the device (k_hemi_expand,
    ez_xpnsrcgd) and the per-point kernel indexes rows j1..j2 like the reference's leaf kernels -> bit-exact; targets: coordinates"""
    case = HEMI[name]
    ni, nj = case["src"][:2]; no, mo = case["dst"][:2]
    gdin = hip_define(case["src"]); gdout = hip_define(ec.dst_spec(case))
    assert gdin >= 0 and gdout >= 0 and ez.ezdefset(gdout, gdin) == 1
    zin = ec.synth_field(ni, nj, seed=11)
    if case["src"][2] == "B":
        z2 = zin.reshape(nj, ni); z2[:, -1] = z2[:, 0]; zin = np.ascontiguousarray(z2.reshape(-1))
    hemi_src = case["src"][2] in ("A", "B") and case["src"][3][0] != 0
    for degree in (0, 1, 3):
        for polar in (1, 0):
            setopts(degree, polar)
            rc, z = ez.ezsint(zin, no * mo)
            want = GOLD[f"{name}/z_d{degree}_p{polar}"]
            assert rc == int(GOLD[f"{name}/rc_d{degree}_p{polar}"]), (name, degree, polar)
            assert relerr(z, want).max() <= RTOL, (name, degree, polar, float(relerr(z, want).max()))
            if hemi_src and not polar:
                assert np.array_equal(z.view(np.uint32), want.view(np.uint32)), (name, degree, polar)
    rc, lat, lon = ez.gdll(gdout, no * mo)
    assert rc == 0 and np.array_equal(lat, GOLD[f"{name}/lat"]) and np.array_equal(lon, GOLD[f"{name}/lon"])
    if hemi_src:
        uu, vv = ec.synth_wind(ni, nj, seed=3)
        assert ez.ezuvint(uu, vv, no * mo)[0] == -1          # winds from a hemispheric source: refused


@pytest.mark.parametrize("name", ["Anord_to_L", "Bsud_to_L", "Ainv_to_L", "AnordInv_to_L"])
def test_gdxysint_hemispheric_and_inverted_sources(name):
    """c_gdxysint on a hemispheric / y-inverted source (gdxysint.c:35-47: PERMUT, ez_xpnsrcgd, then the leaf kernel on rows j1..j2)
    against the oracle's c_gdinterp on the field it permutes / expands itself: bit-exact; and a batch call of the same set"""
    case = HEMI[name]
    ni, nj = case["src"][:2]
    gdin = hip_define(case["src"])
    import test_oracle_golden as tog
    gi = tog.orc_define(case["src"])
    O = ol.oracle()
    O.orc_xpnsrcgd.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    O.orc_permut.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    zin = ec.synth_field(ni, nj, seed=13)
    if case["src"][2] == "B":
        z2 = zin.reshape(nj, ni); z2[:, -1] = z2[:, 0]; zin = np.ascontiguousarray(z2.reshape(-1))
    hem, yinv = case["src"][3][0], case["src"][3][1]
    src = zin.copy()
    if yinv:
        O.orc_permut(src.ctypes.data, ni, nj)
    j1, j2 = gi.contents.j1, gi.contents.j2
    if hem:
        ex = np.zeros(ni * (j2 - j1 + 1), np.float32)
        O.orc_xpnsrcgd(ctypes.cast(gi, ctypes.c_void_p), ex.ctypes.data, src.ctypes.data, 1)
        src = ex
    n = 400
    x = (ec.hash_uniform(45, n).astype(np.float64) * (ni - 1.0) + 1.0).astype(np.float32)
    y = (ec.hash_uniform(46, n).astype(np.float64) * (j2 - j1 - 3.0) + (j1 + 1.5)).astype(np.float32)
    for degree in (0, 1, 3):
        setopts(degree, 1)
        rc, z = ez.gdxysint(zin, gdin, x, y)
        want = np.zeros(n, np.float32)
        O.orc_gdinterp(gi, degree, ol.fptr(want), ol.fptr(src), ol.fptr(x), ol.fptr(y), n)
        assert rc == 0 and np.array_equal(z.view(np.uint32), want.view(np.uint32)), (name, degree, float(relerr(z, want).max()))
    # the batch entry point on such a set loops over the fields
    no, mo = case["dst"][:2]
    gdout = hip_define(ec.dst_spec(case))
    assert ez.ezdefset(gdout, gdin) == 1
    setopts(3, 1)
    d_in = torch.from_numpy(np.stack([zin, zin[::-1].copy()])).cuda(); d_out = torch.empty((2, no * mo), dtype=torch.float32, device="cuda")
    ez.use_stream(0)
    assert ez.ezsint_batch_dev(d_out, d_in, 2) == 0
    torch.cuda.synchronize()
    rc, single = ez.ezsint(zin, no * mo)
    assert np.array_equal(d_out[0].cpu().numpy().view(np.uint32), single.view(np.uint32))


EXTRAP_NAMES = {0: "nearest", 1: "linear", 3: "cubic", 4: "maximum", 5: "minimum", 6: "value"}


@pytest.mark.parametrize("name", ["Lregional_to_L", "N_to_L", "ZEreg_to_L"])
@pytest.mark.parametrize("extrap", [0, 1, 3, 4, 5, 6])
def test_extrapolation_degrees_match_oracle(name, extrap):
    """extrap_degree (ez_corrval.c:60-110) for points outside a regional source: fill with max / min / value, or
    re-interpolation with another degree -- HIP path against the oracle (pinned against the reference for these modes)"""
    import test_oracle_golden as tog
    case = CASES[name]
    ni, nj = case["src"][:2]; no, mo = case["dst"][:2]
    gdin = hip_define(case["src"]); gdout = hip_define(ec.dst_spec(case))
    assert ez.ezdefset(gdout, gdin) == 1
    zin, _, _ = case_inputs(name, case)
    O = ol.oracle()
    gi = tog.orc_define(case["src"]); go = tog.orc_define(ec.dst_spec(case))
    gs = O.orc_defset(go, gi)
    for degree in (1, 3):
        setopts(degree, 1, EXTRAP_NAMES[extrap])
        assert ez.ezsetval("extrap_value", 123.5) == 0
        try:
            rc, z = ez.ezsint(zin, no * mo)
        finally:
            ez.ezsetopt("extrap_degree", "maximum")
        opts = ol.default_opts(degre_interp=degree, degre_extrap=extrap, valeur_extrap=123.5)
        want = np.zeros(no * mo, np.float32)
        rc_o = O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(want), ol.fptr(zin))
        assert rc == rc_o == 2, (name, extrap, degree, rc, rc_o)
        err = relerr(z, want)
        assert err.max() <= RTOL, (name, extrap, degree, float(err.max()))


ECASES = ec.e_cases()


@pytest.mark.parametrize("name", sorted(ECASES))
def test_regular_E_grids_vs_golden(name):
    """regular rotated 'E' grids: as a source (k_locate kind 3; polar correction off, the only setting the reference survives,
    SURVEY D.1) and as a scalar target (2-D coordinates through ez_gfllfxy on the host)"""
    case = ECASES[name]
    ni, nj = case["src"][:2]; no, mo = case["dst"][:2]
    gdin = hip_define(case["src"]); gdout = hip_define(ec.dst_spec(case))
    assert gdin >= 0 and gdout >= 0 and ez.ezdefset(gdout, gdin) == 1
    zin = ec.synth_field(ni, nj, seed=11)
    if case["src"][2] == "E":
        z2 = zin.reshape(nj, ni); z2[:, -1] = z2[:, 0]; zin = np.ascontiguousarray(z2.reshape(-1))
    for degree in (0, 1, 3):
        for polar in case["polar"]:
            setopts(degree, polar)
            rc, z = ez.ezsint(zin, no * mo)
            want = GOLD[f"{name}/z_d{degree}_p{polar}"]
            assert rc == int(GOLD[f"{name}/rc_d{degree}_p{polar}"])
            err = relerr(z, want)
            assert err.max() <= RTOL, (name, degree, polar, float(err.max()))
            if case["src"][2] == "E":      # exact host locate + per-point kernel: bit-exact
                assert np.array_equal(z.view(np.uint32), want.view(np.uint32)), (name, degree, polar)


def test_yinyang_source_medium_size_vs_oracle():
    """a 0.5-degree Yin-Yang pair -> 0.25-degree global lat-lon (1 M points: the threaded first-call locate, both point lists,
    the direct-to-target writes): scalars bit-exact against the oracle, which is pinned against the reference at small size"""
    ni, nj = 577, 205
    dx = 270.0 / (ni - 37)
    ax = (45.0 - 18 * dx + dx * np.arange(ni, dtype=np.float64)).astype(np.float32)
    dy = 90.0 / (nj - 25)
    ay = (-45.0 - 12 * dy + dy * np.arange(nj, dtype=np.float64)).astype(np.float32)
    gy = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.YIN_IG, ax, ay); ga = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.YAN_IG, ax, ay)
    gu = ez.ezgdef_supergrid(ni, 2 * nj, "U", "F", 1, [gy, ga])
    no, mo = 1440, 721
    go = ez.ezqkdef(no, mo, "L", 25, 25, 0, 0)
    assert ez.ezdefset(go, gu) == 1
    z = np.concatenate([ec.synth_field(ni, nj, seed=3), ec.synth_field(ni, nj, seed=4)])
    O = ol.oracle()
    sg = O.orc_supergrid_define(ol.grid_define(ni, nj, "Z", ec.YIN_IG, "E", ax, ay), ol.grid_define(ni, nj, "Z", ec.YAN_IG, "E", ax, ay))
    ogo = ol.grid_define(no, mo, "L", (25, 25, 0, 0))
    for degree in (3, 1):
        setopts(degree, 1)
        rc, got = ez.ezsint(z, no * mo)
        opts = ol.default_opts(degre_interp=degree)
        want = np.zeros(no * mo, np.float32)
        assert O.orc_ezyysint(sg, ogo, ctypes.byref(opts), ol.fptr(want), ol.fptr(z)) == 0
        assert rc == 0 and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (degree, int((got != want).sum()))


# ---------------------------------------------------------------------------------------------------------------
# BASELINE configs at their stated sizes (round 2): cfg1 exact, cfg3 full size, the 32-field batch of cfg4's share
# ---------------------------------------------------------------------------------------------------------------
def _load(name):
    return np.load(os.path.join(os.path.dirname(__file__), "golden", name))


def test_cfg1_exact_size_against_reference_run():
    """BASELINE cfg1 at its exact size: 'L' 400x200 (ig 90,90,45,0) -> 'L' 800x400 (ig 45,45,0,0), host-pointer c_ezsint.
    Bilinear (the configured degree) and nearest: every one of the 320 000 outputs bit-identical to the reference's own run
    (tests/golden/make_cfg1_full.py); bicubic within 1e-5 on the sampled rows and on the float64 sum."""
    G1 = _load("cfg1_full_golden.npz")
    ni, nj, no, mo = 400, 200, 800, 400
    gdin = ez.ezqkdef(ni, nj, "L", 90, 90, 45, 0); gdout = ez.ezqkdef(no, mo, "L", 45, 45, 0, 0)
    assert ez.ezdefset(gdout, gdin) == 1
    zin = ec.synth_field(ni, nj, seed=1)
    rows = G1["rows"]
    for degree in (1, 0, 3):
        for polar in (1, 0):
            setopts(degree, polar)
            rc, z = ez.ezsint(zin, no * mo)
            key = f"d{degree}_p{polar}"
            assert rc == int(G1[key + "/rc"]), key
            got = z.reshape(mo, no)
            assert relerr(got[rows], G1[key + "/rows"]).max() <= RTOL, key
            s = float(z.astype(np.float64).sum())
            assert abs(s - float(G1[key + "/sum"])) <= 1e-7 * abs(s), key
            if degree == 1:
                assert relerr(z, G1[key + "/z"].reshape(-1)).max() <= RTOL, key
            if degree in (0, 1) and not polar:
                if degree == 1:
                    assert np.array_equal(z.view(np.uint32), G1[key + "/z"].reshape(-1).view(np.uint32)), key
                u = z.view(np.uint32)
                h = (int(u.astype(np.uint64).sum()) & 0xFFFFFFFF, int(np.bitwise_xor.reduce(u)))
                assert h == tuple(int(v) for v in G1[key + "/hash"]), key


def _libm_arguments(seed, n):
    """REAL arguments for the C library comparison: every exponent with random mantissas, the rotated frame's ranges densely, and the special values"""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    dar = np.float32(np.pi / 180.0)
    b = (rng.uniform(-400.0, 760.0, n).astype(np.float32) * dar).astype(np.float32)          # dar * lon, dar * lat as ez_lac forms them
    c = rng.uniform(-1.0, 1.0, n).astype(np.float32)                                         # components of unit vectors (ez_cal)
    d = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 0.975, 0.4375, 0.6875, 1.1875, 2.4375, 120.0, -120.0, 0.785398, 0.7853982, 1e-30, -1e-30, 1e-40, 3e38,
                  np.inf, -np.inf, np.nan, 2.0 ** -12, 2.0 ** -27, 2.0 ** -29, 2.0 ** 25, 1.5707964, 3.1415927, 6.2831855], dtype=np.float32)
    near = np.concatenate([(d.view(np.int32) + np.int32(k)).view(np.float32) for k in (-2, -1, 1, 2)])
    return np.ascontiguousarray(np.concatenate([a, b, c, d, near]))


def test_libm_exact_on_the_device_equals_the_c_library():
    """librmn_amd/csrc/libm_exact.h (GNU libc 2.35's REAL sinf / cosf / asinf / atanf / atan2f, operation by operation) evaluated ON THE DEVICE against the
    C library of this machine, bit for bit, over ~3 M arguments per function (tools/check_libm_exact.c covers all 2^32 on the host, tools/check_libm_exact_gpu.py
    all 2^32 on the device): what makes the device locate of a rotated source bit-exact (ez_lac.inc:31-47, ez_cal.inc:22-47)"""
    O = ol.oracle()
    O.orc_libm_apply.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    a = _libm_arguments(5, 1 << 20); b = np.ascontiguousarray(_libm_arguments(6, 1 << 20)[::-1])
    d_a = torch.from_numpy(a).cuda(); d_b = torch.from_numpy(b).cuda(); d_o = torch.empty_like(d_a)
    want = np.empty_like(a)
    for fn, name in enumerate(("sinf", "cosf", "asinf", "atanf", "atan2f")):
        with np.errstate(all="ignore"):
            O.orc_libm_apply(fn, a.ctypes.data, b.ctypes.data, want.ctypes.data, a.size)
        assert ez.libm_exact_probe(fn, d_a, d_b, d_o) == 0
        torch.cuda.synchronize()
        got = d_o.cpu().numpy()
        differ = (got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))
        assert not differ.any(), (name, int(differ.sum()), a[differ][:4], b[differ][:4], got[differ][:4], want[differ][:4])


def test_rotated_source_locate_on_the_device_equals_the_host_locate(monkeypatch):
    """a set's x, y located by k_locate (rotated kinds 2 / 3 with libm_exact.h) against the same set located by the host code that calls the C library
    (EZHIP_HOST_LOCATE=1): every point, every bit; separable, cloud and rotated (2-D coordinates) targets, 1.1 M points in the largest"""
    big = dict(CASES["ZE_to_L"]); big["dst"] = (1500, 750, "L", (24, 24, 0, 0))
    onto_ze = dict(src=ECASES["E_to_L"]["src"], dst=(120, 60, "Z", ec.E_IG), dst_ref="E", dst_axes=ec.ze_axes)
    ze_onto_ze = dict(src=CASES["ZE_to_L"]["src"], dst=(120, 60, "Z", ec.YAN_IG), dst_ref="E", dst_axes=ec.ze_axes)          # (another rotation than the source's)
    for name, case in (("ZE_to_L", CASES["ZE_to_L"]), ("ZEreg_to_L", CASES["ZEreg_to_L"]), ("ZE_to_Y", CASES["ZE_to_Y"]), ("E_to_L", ECASES["E_to_L"]),
                       ("ZE_to_L_big", big), ("E_to_ZE", onto_ze), ("ZE_to_ZE", ze_onto_ze)):
        n = case["dst"][0] * case["dst"][1]
        got = []
        for host in (False, True):
            if host:
                monkeypatch.setenv("EZHIP_HOST_LOCATE", "1")
            else:
                monkeypatch.delenv("EZHIP_HOST_LOCATE", raising=False)
            gdin = hip_define(case["src"]); gdout = hip_define(ec.dst_spec(case))          # (fresh grids: a set keeps the x, y it located first)
            assert ez.ezdefset(gdout, gdin) == 1
            d_x = torch.empty(n, dtype=torch.float32, device="cuda"); d_y = torch.empty_like(d_x)
            assert ez.set_xy_dev(d_x, d_y) == 0
            torch.cuda.synchronize()
            got.append((d_x.cpu().numpy(), d_y.cpu().numpy()))
            ez.gdrls(gdout); ez.gdrls(gdin)
        monkeypatch.delenv("EZHIP_HOST_LOCATE", raising=False)
        assert np.array_equal(got[0][0].view(np.uint32), got[1][0].view(np.uint32)), name
        assert np.array_equal(got[0][1].view(np.uint32), got[1][1].view(np.uint32)), name


def test_rotated_source_locate_is_bit_exact():
    """the x,y a set with a rotated source ('E', Z-on-'E': ez_gfxyfll.c:38-57) interpolates with -- read back through
    ezhip_set_xy_dev -- equal the reference's c_gdxyfll bit for bit: located ON THE DEVICE (k_locate with the C library's REAL functions restated in
    libm_exact.h) since round 5, by host threads before"""
    for name in ("ZE_to_L", "ZEreg_to_L", "ZE_to_Y"):
        case = CASES[name]
        gdin = hip_define(case["src"]); gdout = hip_define(ec.dst_spec(case))
        assert ez.ezdefset(gdout, gdin) == 1
        n = case["dst"][0] * case["dst"][1]
        d_x = torch.empty(n, dtype=torch.float32, device="cuda"); d_y = torch.empty_like(d_x)
        assert ez.set_xy_dev(d_x, d_y) == 0
        torch.cuda.synchronize()
        assert np.array_equal(d_x.cpu().numpy().view(np.uint32), GOLD[f"{name}/x"].view(np.uint32)), name
        assert np.array_equal(d_y.cpu().numpy().view(np.uint32), GOLD[f"{name}/y"].view(np.uint32)), name
    for name in ("E_to_L",):
        case = ECASES[name]
        gdin = hip_define(case["src"]); gdout = hip_define(ec.dst_spec(case))
        assert ez.ezdefset(gdout, gdin) == 1
        n = case["dst"][0] * case["dst"][1]
        d_x = torch.empty(n, dtype=torch.float32, device="cuda"); d_y = torch.empty_like(d_x)
        assert ez.set_xy_dev(d_x, d_y) == 0
        torch.cuda.synchronize()
        if f"{name}/x" in GOLD:
            assert np.array_equal(d_x.cpu().numpy().view(np.uint32), GOLD[f"{name}/x"].view(np.uint32)), name
            assert np.array_equal(d_y.cpu().numpy().view(np.uint32), GOLD[f"{name}/y"].view(np.uint32)), name


def _hash_t(t):
    return _bit_hash(t)


def test_cfg3_full_size_pair_batch_equals_single_calls():
    """BASELINE cfg3's grid pair at full size (Z-on-E 2560x1280 -> L 4000x2000): three wind pairs through c_ezuvint_batch_dev in one launch against three c_ezuvint_dev
    calls, every bit of 8 M points per component (the single calls themselves are held to the reference run by test_cfg3_full_size_against_reference_run)"""
    ni, nj, no, mo, K = 2560, 1280, 4000, 2000, 3
    ax, ay = ec.ze_axes(ni, nj)
    gdin = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); gdout = ez.ezqkdef(no, mo, "L", 9, 9, 0, 0)
    assert ez.ezdefset(gdout, gdin) == 1
    setopts(3, 1)
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    uu, vv = ec.synth_wind(ni, nj, seed=3)
    for a in (uu, vv):
        a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
    bu = torch.from_numpy(uu).cuda(); bv = torch.from_numpy(vv).cuda()
    d_u = torch.stack([bu * (1.0 + 0.125 * f) for f in range(K)]).contiguous(); d_v = torch.stack([bv * (1.0 - 0.25 * f) for f in range(K)]).contiguous()
    r_u = torch.empty((K, no * mo), dtype=torch.float32, device="cuda"); r_v = torch.empty_like(r_u)
    for rep in range(2):                                          # (the second round: k_uvt)
        for f in range(K):
            assert ez.ezuvint_dev(r_u[f], r_v[f], d_u[f], d_v[f]) == 0
    o_u = torch.full_like(r_u, float("nan")); o_v = torch.full_like(r_v, float("nan"))
    assert ez.ezuvint_batch_dev(o_u, o_v, d_u, d_v, K) == 0
    torch.cuda.synchronize()
    assert torch.equal(o_u.view(torch.int32), r_u.view(torch.int32)) and torch.equal(o_v.view(torch.int32), r_v.view(torch.int32))
    ez.gdrls(gdout); ez.gdrls(gdin)


@pytest.mark.parametrize("polar", [1, 0])
@pytest.mark.parametrize("npairs", [2, 5])
def test_ezuvint_batch_equals_single_calls(npairs, polar, monkeypatch):
    """c_ezuvint_batch_dev (additive): npairs wind pairs of one grid set in ONE staged-tile launch (k_uvt's batch form: x, y and the rotation of a point read once for all
    pairs, two polar-wind producer blocks per pair, the special points' kernel once with a pair index) against npairs c_ezuvint_dev calls: every bit of every pair,
    with and without polar correction; also when the batch is the set's FIRST call (the first pair then goes alone and builds the caches) and with the batch form
    switched off (EZHIP_NO_PAIR_BATCH=1: pair by pair).  ezuvint.c:51-94"""
    ni, nj, no, mo = 320, 160, 1000, 500
    ax, ay = ec.ze_axes(ni, nj)
    winds = [ec.synth_wind(ni, nj, seed=70 + f) for f in range(npairs)]
    for uu, vv in winds:
        for a in (uu, vv):
            a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
    d_u = torch.from_numpy(np.stack([w[0] for w in winds])).cuda().contiguous(); d_v = torch.from_numpy(np.stack([w[1] for w in winds])).cuda().contiguous()
    setopts(3, polar)
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    outs = {}
    for mode in ("single", "batch_first", "batch", "batch_off"):
        gdin = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); gdout = ez.ezqkdef(no, mo, "L", 36, 36, 0, 0)
        assert ez.ezdefset(gdout, gdin) == 1
        o_u = torch.full((npairs, no * mo), -7.0, dtype=torch.float32, device="cuda"); o_v = torch.full_like(o_u, -9.0)
        if mode == "single":
            for rep in range(2):                    # (the second round runs from the staged windows)
                for f in range(npairs):
                    assert ez.ezuvint_dev(o_u[f], o_v[f], d_u[f], d_v[f]) == 0
        else:
            if mode == "batch_off":
                monkeypatch.setenv("EZHIP_NO_PAIR_BATCH", "1")
            if mode != "batch_first":
                assert ez.ezuvint_dev(o_u[0], o_v[0], d_u[0], d_v[0]) == 0      # the set's caches exist: the batch form takes all pairs
                o_u.fill_(-7.0); o_v.fill_(-9.0)
            assert ez.ezuvint_batch_dev(o_u, o_v, d_u, d_v, npairs) == 0
            monkeypatch.delenv("EZHIP_NO_PAIR_BATCH", raising=False)
        torch.cuda.synchronize()
        outs[mode] = (o_u.clone(), o_v.clone())
        ez.gdrls(gdout); ez.gdrls(gdin)
    for mode in ("batch_first", "batch", "batch_off"):
        assert torch.equal(outs[mode][0], outs["single"][0]) and torch.equal(outs[mode][1], outs["single"][1]), mode
    assert float(outs["single"][0].abs().max()) > 1.0 and not bool((outs["single"][0] == -7.0).any())


def test_cfg3_full_size_against_reference_run():
    """BASELINE cfg3 at full size (Z-on-E 2560x1280 rotated global grid -> L 4000x2000), device resident, against the
    reference's own c_gdxyfll / c_ezsint / c_ezuvint run (tests/golden/make_cfg3_full.py): located x,y bit-exact over all
    8 M points (hash); scalars bit-exact over all points (hash) for every degree without polar correction and within
    1e-5 with it; wind components within 1e-5 of |V| on the sampled rows / columns and on the float64 sums."""
    G3 = _load("cfg3_full_golden.npz")
    ni, nj, no, mo = 2560, 1280, 4000, 2000
    ax, ay = ec.ze_axes(ni, nj)
    gdin = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); gdout = ez.ezqkdef(no, mo, "L", 9, 9, 0, 0)
    assert ez.ezdefset(gdout, gdin) == 1
    uu, vv = ec.synth_wind(ni, nj, seed=3)
    for a in (uu, vv):
        a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
    rows = torch.from_numpy(G3["rows"]).cuda(); cols = torch.from_numpy(G3["cols"]).cuda()
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    d_u = torch.from_numpy(uu).cuda(); d_v = torch.from_numpy(vv).cuda()
    o_u = torch.empty(no * mo, dtype=torch.float32, device="cuda"); o_v = torch.empty_like(o_u); o_z = torch.empty_like(o_u)
    d_x = torch.empty_like(o_u); d_y = torch.empty_like(o_u)
    assert ez.set_xy_dev(d_x, d_y) == 0
    torch.cuda.synchronize()
    for nm, t in (("x", d_x), ("y", d_y)):
        t2 = t.view(mo, no)
        assert np.array_equal(t2[rows].cpu().numpy().view(np.uint32), G3[nm + "/rows"].view(np.uint32)), nm
        assert np.array_equal(t2[:, cols].cpu().numpy().view(np.uint32), G3[nm + "/cols"].view(np.uint32)), nm
        assert _hash_t(t) == tuple(int(v) for v in G3[nm + "/hash"]), nm
    for degree in (3, 1, 0):
        for polar in (1, 0):
            setopts(degree, polar)
            key = f"d{degree}_p{polar}"
            assert ez.ezsint_dev(o_z, d_u) == 0
            assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) == 0
            torch.cuda.synchronize()
            if (degree, polar) == (3, 1):
                # the first call of a grid set lists its special points with the gathering kernel; from the second call on the pair runs from LDS-staged
                # stencil windows (k_uvt): the same numbers bit for bit, and it is THAT call's output the checks below look at
                first_u, first_v = o_u.clone(), o_v.clone()
                o_u.zero_(); o_v.zero_()
                assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) == 0
                torch.cuda.synchronize()
                assert torch.equal(o_u, first_u) and torch.equal(o_v, first_v)
            z2 = o_z.view(mo, no)
            for got, want in ((z2[rows].cpu().numpy(), G3[key + "/z/rows"]), (z2[:, cols].cpu().numpy(), G3[key + "/z/cols"])):
                assert relerr(got, want).max() <= RTOL, (key, float(relerr(got, want).max()))
            if not polar:
                assert _hash_t(o_z) == tuple(int(v) for v in G3[key + "/z/hash"]), key
            u2 = o_u.view(mo, no); v2 = o_v.view(mo, no)
            for sel, tag in ((lambda t: t[rows], "rows"), (lambda t: t[:, cols], "cols")):
                gu, gv = sel(u2).cpu().numpy(), sel(v2).cpu().numpy()
                wu, wv = G3[f"{key}/u/{tag}"], G3[f"{key}/v/{tag}"]
                scale = np.maximum(np.sqrt(wu.astype(np.float64) ** 2 + wv.astype(np.float64) ** 2), 1e-3)
                eu = np.abs(gu - wu) / scale; ev = np.abs(gv - wv) / scale
                assert eu.max() <= RTOL and ev.max() <= RTOL, (key, tag, float(eu.max()), float(ev.max()))
            for nm, t in (("u", o_u), ("v", o_v), ("z", o_z)):
                s = float(t.double().sum().item()); w = float(G3[f"{key}/{nm}/sum"])
                assert abs(s - w) <= 2e-7 * float(t.double().abs().sum().item()), (key, nm, s, w)


def test_batch_of_32_full_size_fields_equals_single_calls_and_reference():
    """the call bench.py times -- c_ezsint_batch_dev on 32 device-resident full-size cfg2 fields (cfg4's per-GPU share) --
    compared (a) bit for bit with c_ezsint_dev on fields 0, 15 and 31 and (b) for the field whose input is the fixture's
    'synth' field, with the reference's own full-size run (sampled rows / columns + float64 sum)."""
    G = _load("cfg2_full_golden.npz")
    ni, nj, no, mo = 4400, 2200, 7200, 3601
    gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
    assert ez.ezdefset(gdout, gdin) == 1
    setopts(3, 1)
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    F = 32
    d_in = torch.empty((F, ni * nj), dtype=torch.float32, device="cuda")
    for f in range(F):
        d_in[f] = torch.from_numpy(ec.synth_field(ni, nj, seed=2 if f == 15 else 1000 + f)).cuda()
    d_out = torch.full((F, no * mo), -1.0, dtype=torch.float32, device="cuda")
    assert ez.ezsint_batch_dev(d_out, d_in, F) == 0
    torch.cuda.synchronize()
    one = torch.empty(no * mo, dtype=torch.float32, device="cuda")
    for f in (0, 15, 31):
        assert ez.ezsint_dev(one, d_in[f]) == 0
        torch.cuda.synchronize()
        assert torch.equal(one, d_out[f]), f
    rows = torch.from_numpy(G["rows"]).cuda(); cols = torch.from_numpy(G["cols"]).cuda()
    o2 = d_out[15].view(mo, no)
    key = "synth/d3_p1"
    for got, want in ((o2[rows].cpu().numpy(), G[key + "/rows"]), (o2[:, cols].cpu().numpy(), G[key + "/cols"])):
        assert relerr(got, want).max() <= RTOL, float(relerr(got, want).max())
    s = float(d_out[15].double().sum().item())
    assert abs(s - float(G[key + "/sum"])) <= 1e-8 * abs(s)
    # no field of the batch was skipped or written twice into another's slot: all differ from the fill and from each other's sums
    sums = d_out.double().sum(dim=1).cpu().numpy()
    assert np.all(np.isfinite(sums)) and len(set(sums.tolist())) == F


def test_abort_and_rc_on_the_per_point_path_through_dev_entry_points():
    """a regional rotated (Z-on-E) source: the per-point path.  rc = 2 whenever target points lie outside the source
    (ez_corrval.c:54-55) and extrap_degree=abort returns -1 from the scalar calls (ez_corrval.c:56-60) -- also through the *_dev entry points,
    which do not go through ezhip_prepare_set."""
    case = CASES["ZEreg_to_L"]
    gdin = hip_define(case["src"]); gdout = hip_define(ec.dst_spec(case))
    assert ez.ezdefset(gdout, gdin) == 1
    ni, nj = case["src"][:2]; no, mo = case["dst"][:2]
    zin, uu, vv = case_inputs("ZEreg_to_L", case)
    d_in = torch.from_numpy(zin).cuda(); d_u = torch.from_numpy(uu).cuda(); d_v = torch.from_numpy(vv).cuda()
    d_out = torch.empty(no * mo, dtype=torch.float32, device="cuda"); d_o2 = torch.empty_like(d_out)
    ez.use_stream(0)
    setopts(3, 1)
    assert ez.ezsint_dev(d_out, d_in) == 2
    assert ez.ezsint_batch_dev(d_out, d_in, 1) == 2
    assert ez.ezuvint_dev(d_out, d_o2, d_u, d_v) == 2
    setopts(3, 1, "abort")
    try:
        assert ez.ezsint_dev(d_out, d_in) == -1
        assert ez.ezsint_batch_dev(d_out, d_in, 1) == -1
        # the winds: the reference's c_ezuvint ignores the -1 of its two c_ezsint calls (ezuvint.c:68-74 looks for 2 only) and returns 0 with the
        # winds of the same call without the polar correction -- reproduced (tools/fuzz_vs_ref2.py compares it with the reference build)
        assert ez.ezuvint_dev(d_out, d_o2, d_u, d_v) == 0
        setopts(3, 0, "abort")
        d_p = torch.empty_like(d_out); d_q = torch.empty_like(d_out)
        assert ez.ezuvint_dev(d_p, d_q, d_u, d_v) == 0
        torch.cuda.synchronize()
        assert torch.equal(d_p, d_out) and torch.equal(d_q, d_o2)
        setopts(3, 1, "abort")
        rc, _ = ez.ezsint(zin, no * mo)
        assert rc == -1
    finally:
        ez.ezsetopt("extrap_degree", "maximum")
    # polar correction off: no correction pass, rc 0 like the reference (ezsint.c:120-123)
    setopts(3, 0)
    assert ez.ezsint_dev(d_out, d_in) == 0
    torch.cuda.synchronize()


def test_pole_wait_timeout_is_reported_not_silent():
    """k_sepx's bounded wait for the in-launch pole values (EZHIP_TEST_POLE_TIMEOUT makes the launch behave as if it had
    given up): the polar rows become NaN -- never stale values -- and the call / the next entry point returns -1"""
    ni, nj, no, mo = 360, 181, 500, 251
    gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 72, 72, 0, 0)
    assert ez.ezdefset(gdout, gdin) == 1
    setopts(3, 1)
    zin = ec.synth_field(ni, nj, seed=5)
    rc, good = ez.ezsint(zin, no * mo)
    assert rc == 0 and np.all(np.isfinite(good))
    os.environ["EZHIP_TEST_POLE_TIMEOUT"] = "1"
    try:
        rc, z = ez.ezsint(zin, no * mo)
        assert rc == -1                                        # the host-pointer call synchronises: it reports its own failure
        z2 = z.reshape(mo, no)
        assert np.all(np.isnan(z2[0])) and np.all(np.isnan(z2[-1]))     # pole rows: NaN, not whatever the slot held
        assert np.array_equal(z2[5:-5], good.reshape(mo, no)[5:-5])      # the main rows do not depend on the pole values
        ez.use_stream(0)
        d_in = torch.from_numpy(zin).cuda(); d_out = torch.empty(no * mo, dtype=torch.float32, device="cuda")
        assert ez.ezsint_dev(d_out, d_in) == 0                 # asynchronous: the launch itself is fine ...
        torch.cuda.synchronize()
    finally:
        os.environ.pop("EZHIP_TEST_POLE_TIMEOUT", None)
    assert ez.ezsint_dev(d_out, d_in) == -1                    # ... the next entry point reports it (sticky), once
    assert ez.ezsint_dev(d_out, d_in) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint32), good.view(np.uint32))


def test_two_threads_host_pointer_ezsint_on_one_grid_pair():
    """two host threads call the host-pointer c_ezsint on the SAME grid pair at the same time, each on its own stream and field
    (staging buffers are per thread), while a third thread defines 200 more grids (the grid table never moves)"""
    import threading
    ni, nj, no, mo = 360, 181, 500, 251
    gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 72, 72, 0, 0)
    assert ez.ezdefset(gdout, gdin) == 1
    setopts(3, 1)
    fields = [ec.synth_field(ni, nj, seed=70 + k) for k in range(2)]
    want = []
    for k in range(2):
        rc, z = ez.ezsint(fields[k], no * mo)
        assert rc == 0
        want.append(z)
    errs = []

    def worker(k):
        try:
            st = torch.cuda.Stream()
            ez.use_stream(st.cuda_stream)
            assert ez.ezdefset(gdout, gdin) == 1               # the current pair is per thread
            for it in range(25):
                rc, z = ez.ezsint(fields[k], no * mo)
                if rc != 0 or not np.array_equal(z.view(np.uint32), want[k].view(np.uint32)):
                    errs.append((k, it, rc)); return
        except Exception as e:   # noqa: BLE001
            errs.append((k, repr(e)))

    def definer():
        try:
            for g in range(200):
                if ez.ezqkdef(10 + g, 8, "L", 100, 100, 0, 0) < 0:
                    errs.append(("define", g)); return
        except Exception as e:   # noqa: BLE001
            errs.append(("define", repr(e)))

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(2)] + [threading.Thread(target=definer)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    ez.use_stream(0)
    assert not errs, errs


@pytest.mark.parametrize("case", ["global_polar", "global_nopolar", "regional_fill", "linear", "nearest"])
def test_host_pointer_row_ranges_equal_whole_copies(case, monkeypatch):
    """c_ezsint between arrays registered with ezhip_register_host_buffer runs k_sepx in row ranges (source rows up, finished rows down,
    special rows and pole sums in the last range), EZHIP_HOST_CHUNKS sets their number.  Same bits as one launch."""
    import ctypes
    L = ez._lib()
    L.ezhip_register_host_buffer.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    L.ezhip_unregister_host_buffer.argtypes = [ctypes.c_void_p]
    L.c_ezsint.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    if case == "regional_fill":
        ni, nj, no, mo = 500, 300, 700, 420
        gdin = ez.ezqkdef(ni, nj, "L", 10, 10, 6000, 20000)          # 0.1 degree box; the target reaches beyond it
        gdout = ez.ezqkdef(no, mo, "L", 10, 10, 5500, 19500)
    else:
        ni, nj, no, mo = 720, 360, 1100, 551
        gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 33, 33, 0, 0)
    assert ez.ezdefset(gdout, gdin) == 1
    setopts({"linear": 1, "nearest": 0}.get(case, 3), 0 if case == "global_nopolar" else 1)
    zin = ec.synth_field(ni, nj, seed=11)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    monkeypatch.setenv("EZHIP_HOST_NO_CHUNKS", "1")
    want = np.zeros(no * mo, np.float32)
    rc0 = L.c_ezsint(p(want), p(zin))
    assert rc0 in (0, 2)
    monkeypatch.delenv("EZHIP_HOST_NO_CHUNKS")
    # row ranges run between page-locked arrays only (the product never cuts a copy to or from ordinary memory into ranges: such copies block the host
    # anyway, and many short device writes into one pageable array are what the runtime handles worst); EZHIP_HOST_CHUNKS sets their number
    got = np.full(no * mo, np.nan, np.float32)
    assert L.ezhip_register_host_buffer(p(zin), zin.nbytes) == 0 and L.ezhip_register_host_buffer(p(got), got.nbytes) == 0
    try:
        for chunks in ("2", "3", "7", "100"):
            monkeypatch.setenv("EZHIP_HOST_CHUNKS", chunks)
            got[:] = np.nan
            assert L.c_ezsint(p(got), p(zin)) == rc0
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), chunks
        monkeypatch.delenv("EZHIP_HOST_CHUNKS")
        for _ in range(3):
            got[:] = np.nan
            assert L.c_ezsint(p(got), p(zin)) == rc0
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    finally:
        assert L.ezhip_unregister_host_buffer(p(zin)) == 0 and L.ezhip_unregister_host_buffer(p(got)) == 0
    assert L.ezhip_unregister_host_buffer(p(got)) == -1                # not registered any more


@pytest.mark.parametrize("kind", ["rotated_source", "rotated_target"])
def test_wind_matrix_equals_the_chain(kind, monkeypatch):
    """c_ezuvint through a rotated frame applies the wind chain of the grid pair as a per-point matrix built from the chain itself
    (inside k_pts2 for per-point sets, k_wind_apply after separable launches): same winds as the chain run on every call
    (EZHIP_WIND_NO_MATRIX), to a few 1e-7 of |V|; the fused and the separate application agree bit for bit."""
    ni, nj, no, mo = 360, 180, 500, 250
    ax, ay = ec.ze_axes(ni, nj)
    if kind == "rotated_source":
        gdin = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); gdout = ez.ezqkdef(no, mo, "L", 72, 72, 0, 0)
        uu, vv = ec.synth_wind(ni, nj, seed=5); nout = no * mo
    else:
        gdin = ez.ezqkdef(no, mo, "G", 0, 0, 0, 0); gdout = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay)
        uu, vv = ec.synth_wind(no, mo, seed=5); nout = ni * nj
    assert ez.ezdefset(gdout, gdin) == 1
    for degree in (3, 1):
        setopts(degree, 1)
        res = {}
        for mode in ("matrix", "nofuse", "chain"):
            monkeypatch.delenv("EZHIP_WIND_NO_MATRIX", raising=False); monkeypatch.delenv("EZHIP_WIND_NO_FUSE", raising=False)
            if mode == "chain":
                monkeypatch.setenv("EZHIP_WIND_NO_MATRIX", "1")
            if mode == "nofuse":
                monkeypatch.setenv("EZHIP_WIND_NO_FUSE", "1")
            rc, u, v = ez.ezuvint(uu, vv, nout)
            assert rc >= 0
            res[mode] = (u.copy(), v.copy())
        scale = np.maximum(np.hypot(res["chain"][0].astype(np.float64), res["chain"][1].astype(np.float64)), 1e-3)
        for k in (0, 1):
            assert np.max(np.abs(res["matrix"][k].astype(np.float64) - res["chain"][k]) / scale) <= 2e-6, (kind, degree, k)
            assert np.array_equal(res["matrix"][k].view(np.uint32), res["nofuse"][k].view(np.uint32)), (kind, degree, k)


def test_regional_hash_tile_equals_the_z_grid():
    """c_ezgdef_fmem('#', ...) on a regional tile: the same bits as the 'Z' grid with the same axes, scalars and winds (the reference
    agrees with itself on this: tests/test_oracle_vs_ref.py)"""
    ni, nj, no, mo = 51, 41, 50, 40
    ax, ay = ec.zereg_axes(ni, nj)
    gz = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); gh = ez.ezgdef_fmem(ni, nj, "#", "E", *ec.E_IG, ax, ay)
    gdout = ez.ezqkdef(no, mo, "L", 100, 100, 9000, 24000)
    zin = ec.synth_field(ni, nj, seed=4); uu, vv = ec.synth_wind(ni, nj, seed=4)
    for degree in (0, 1, 3):
        setopts(degree, 1)
        out = []
        for g in (gz, gh):
            assert ez.ezdefset(gdout, g) == 1
            rc, z = ez.ezsint(zin, no * mo); rcv, u, v = ez.ezuvint(uu, vv, no * mo)
            out.append((rc, rcv, z, u, v))
        assert out[0][:2] == out[1][:2]
        for k in (2, 3, 4):
            assert np.array_equal(out[0][k].view(np.uint32), out[1][k].view(np.uint32)), (degree, k)


import test_oracle_vs_ref as tovr     # noqa: E402


@pytest.mark.parametrize("name", sorted(tovr.HEMI_G))
def test_hemispheric_gaussian_sources_vs_oracle(name):
    """hemispheric 'G' sources (the table of 2 nj latitudes, the field mirrored into rows j1 .. j2, the reference's two search lengths and
    its northern shift): the per-point kernel against the oracle, which equals the reference build on these cases bit for bit"""
    case = tovr.HEMI_G[name]
    ni, nj = case["src"][:2]; no, mo = case["dst"][:2]
    gdin = ez.ezqkdef(ni, nj, "G", *case["src"][3]); gdout = ez.ezqkdef(no, mo, "L", *case["dst"][3])
    assert gdin >= 0 and ez.ezdefset(gdout, gdin) == 1
    O = ol.oracle()
    gi = ol.grid_define(ni, nj, "G", case["src"][3]); go = ol.grid_define(no, mo, "L", case["dst"][3])
    gs = O.orc_defset(go, gi)
    zin = ec.synth_field(ni, nj, seed=21)
    for degree in (0, 1, 3):
        for polar in case["polar"]:
            setopts(degree, polar)
            rc, z = ez.ezsint(zin, no * mo)
            want = np.zeros(no * mo, np.float32)
            opts = ol.default_opts(degre_interp=degree, polar_correction=polar)
            rc_o = O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(want), ol.fptr(zin))
            assert rc == rc_o
            assert np.array_equal(z.view(np.uint32), want.view(np.uint32)), (name, degree, polar, int(np.count_nonzero(z != want)))
    uu, vv = ec.synth_wind(ni, nj, seed=2)
    assert ez.ezuvint(uu, vv, no * mo)[0] == -1            # winds from a hemisphere: undefined in the reference, refused


AVG_GPU_CASES = dict(tovr.AVG_CASES)
AVG_GPU_CASES["Lregional_outside_polar"] = ((120, 90, "L", (50, 50, 6000, 20000), " ", None), (40, 30, "L", (200, 250, 5500, 19500), " ", None), 1)   # the target leaves the source: fill / re-interpolated points


@pytest.mark.parametrize("name", sorted(AVG_GPU_CASES))
def test_average_degree_vs_oracle(name):
    """interp_degree = average (ez_avg.inc): k_average + the defined parts of the polar correction against the oracle (= the reference
    build on the inside cases, tests/test_oracle_vs_ref.py), bit for bit; extrapolation by value and by nearest for the outside points"""
    src, dst, polar = AVG_GPU_CASES[name]
    ni, nj = src[:2]; no, mo = dst[:2]
    gdin = ez.ezqkdef(ni, nj, src[2], *src[3]); gdout = ez.ezqkdef(no, mo, dst[2], *dst[3])
    assert ez.ezdefset(gdout, gdin) == 1
    O = ol.oracle()
    gi = ol.grid_define(ni, nj, src[2], src[3]); go = ol.grid_define(no, mo, dst[2], dst[3])
    gs = O.orc_defset(go, gi)
    zin = ec.synth_field(ni, nj, seed=13)
    if src[2] == "B":
        z2 = zin.reshape(nj, ni); z2[:, -1] = z2[:, 0]; zin = np.ascontiguousarray(z2.reshape(-1))
    for extrap, xcode in (("maximum", 4), ("nearest", 0), ("value", 6)):
        assert ez.ezsetopt("interp_degree", "average") == 0
        assert ez.ezsetopt("polar_correction", "yes" if polar else "no") == 0
        assert ez.ezsetopt("extrap_degree", extrap) == 0
        if extrap == "value":
            assert ez.ezsetval("extrap_value", -5.5) == 0
        rc, z = ez.ezsint(zin, no * mo)
        want = np.zeros(no * mo, np.float32)
        opts = ol.default_opts(degre_interp=4, polar_correction=polar, degre_extrap=xcode, valeur_extrap=-5.5)
        rc_o = O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(want), ol.fptr(zin))
        assert rc == rc_o, (rc, rc_o)
        assert np.array_equal(z.view(np.uint32), want.view(np.uint32)), (name, extrap, int(np.count_nonzero(z != want)))


def test_average_degree_refusals():
    gdin = ez.ezqkdef(128, 64, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(40, 61, "L", 300, 900, 0, 0)     # rows every 3 degrees: 87 N / S lie in the polar strips
    assert ez.ezdefset(gdout, gdin) == 1
    assert ez.ezsetopt("interp_degree", "average") == 0 and ez.ezsetopt("polar_correction", "yes") == 0
    zin = ec.synth_field(128, 64, seed=1)
    assert ez.ezsint(zin, 40 * 61)[0] == -1             # strip rows that are not pole rows: undefined in the reference, refused
    assert ez.ezsetopt("polar_correction", "no") == 0
    assert ez.ezsint(zin, 40 * 61)[0] == 0
    uu, vv = ec.synth_wind(128, 64, seed=1)
    assert ez.ezuvint(uu, vv, 40 * 61)[0] == -1         # winds: not with this degree
    assert ez.ezsetopt("interp_degree", "sph_average") == 0
    assert ez.ezsint(zin, 40 * 61)[0] == -1             # target rows at the poles: unbounded widening, refused
    # pole rows alone are fine: the pole values overwrite them (ez_corrval.c:125-140)
    gd2 = ez.ezqkdef(40, 21, "L", 900, 900, 0, 0)
    assert ez.ezdefset(gd2, gdin) == 1
    assert ez.ezsetopt("interp_degree", "average") == 0 and ez.ezsetopt("polar_correction", "yes") == 0
    rc, z = ez.ezsint(zin, 40 * 21)
    O = ol.oracle(); gs = O.orc_defset(ol.grid_define(40, 21, "L", (900, 900, 0, 0)), ol.grid_define(128, 64, "G"))
    want = np.zeros(40 * 21, np.float32)
    opts = ol.default_opts(degre_interp=4, polar_correction=1)
    assert O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(want), ol.fptr(zin)) == rc == 0
    assert np.array_equal(z.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("src", ["G", "A", "Lregional"])
def test_sph_average_degree_vs_oracle_leaf(src):
    """interp_degree = sph_average (ez_avg_sph.inc): the product through c_ezsint against the oracle's restatement of the routine (equal to
    the reference's own, tests/test_oracle_vs_ref.py) fed with the set's located x, y and the latitudes c_gdllfxy gives for the target rows"""
    if src == "G":
        ni, nj, gdin = 128, 64, ez.ezqkdef(128, 64, "G", 0, 0, 0, 0); ext = 2
        no, mo, gdout = 36, 15, ez.ezqkdef(36, 15, "L", 1000, 1000, 2000, 0)              # 70 S .. 70 N
    elif src == "A":
        ni, nj, gdin = 144, 72, ez.ezqkdef(144, 72, "A", 0, 0, 0, 0); ext = 2
        no, mo, gdout = 30, 14, ez.ezqkdef(30, 14, "L", 1000, 1200, 2500, 0)
    else:
        ni, nj, gdin = 120, 90, ez.ezqkdef(120, 90, "L", 50, 50, 6000, 20000); ext = 0
        no, mo, gdout = 14, 10, ez.ezqkdef(14, 10, "L", 150, 200, 7200, 21200)
    assert ez.ezdefset(gdout, gdin) == 1
    zin = ec.synth_field(ni, nj, seed=17)
    assert ez.ezsetopt("interp_degree", "sph_average") == 0 and ez.ezsetopt("polar_correction", "no") == 0
    rc, z = ez.ezsint(zin, no * mo)
    assert rc == 0
    d_x = torch.empty(no * mo, device="cuda"); d_y = torch.empty(no * mo, device="cuda")
    assert ez.set_xy_dev(d_x, d_y) == 0
    xx = d_x.cpu().numpy(); yy = d_y.cpu().numpy()
    L = ez._lib()
    L.c_gdllfxy.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lats = np.zeros(mo, np.float32); lons = np.zeros(mo, np.float32)
    xo = np.ones(mo, np.float32); yo = np.arange(1, mo + 1, dtype=np.float32)
    assert L.c_gdllfxy(gdout, lats.ctypes.data, lons.ctypes.data, xo.ctypes.data, yo.ctypes.data, mo) == 0
    want = np.zeros(no * mo, np.float32)
    O = ol.oracle(); O.orc_ez_avg_sph.restype = None
    O.orc_ez_avg_sph(ol.fptr(want), ol.fptr(xx), ol.fptr(yy), ol.fptr(lats), no, mo, ol.fptr(zin), ni, nj, ext)
    assert np.array_equal(z.view(np.uint32), want.view(np.uint32)), (src, int(np.count_nonzero(z != want)))


@pytest.mark.parametrize("degree", ["average", "sph_average"])
def test_averaging_degrees_against_the_reference_build(degree):
    """c_ezsint with the averaging degrees: the product on the GPU against the reference's own sources (oracle/_ref/libezref.so, which
    travels with the snapshot), bit for bit, global and regional sources, polar correction off and on where it is defined"""
    import reflib
    if not reflib.have_ref():
        pytest.skip("oracle/_ref/libezref.so not built")
    R = reflib.ref()
    fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    cases = [((128, 64, "G", (0, 0, 0, 0)), (36, 15, "L", (1000, 1000, 2000, 0)), 0),
             ((144, 72, "A", (0, 0, 0, 0)), (30, 14, "L", (1000, 1200, 2500, 0)), 0),
             ((120, 90, "L", (50, 50, 6000, 20000)), (14, 10, "L", (150, 200, 7200, 21200)), 1)]
    for src, dst, polar in cases:
        ni, nj = src[:2]; no, mo = dst[:2]
        zin = ec.synth_field(ni, nj, seed=23)
        gr_in = R.c_ezqkdef(ni, nj, src[2].encode(), *src[3], 0); gr_out = R.c_ezqkdef(no, mo, dst[2].encode(), *dst[3], 0)
        R.c_ezsetopt(b"interp_degree", degree.encode()); R.c_ezsetopt(b"polar_correction", b"yes" if polar else b"no")
        assert R.c_ezdefset(gr_out, gr_in) == 1
        zr = np.zeros(no * mo, np.float32)
        rc_r = R.c_ezsint(fp(zr), fp(zin))
        R.c_ezsetopt(b"interp_degree", b"cubic"); R.c_ezsetopt(b"polar_correction", b"yes")
        gdin = ez.ezqkdef(ni, nj, src[2], *src[3]); gdout = ez.ezqkdef(no, mo, dst[2], *dst[3])
        assert ez.ezdefset(gdout, gdin) == 1
        assert ez.ezsetopt("interp_degree", degree) == 0 and ez.ezsetopt("polar_correction", "yes" if polar else "no") == 0
        rc, z = ez.ezsint(zin, no * mo)
        assert rc == rc_r
        assert np.array_equal(z.view(np.uint32), zr.view(np.uint32)), (degree, src[2], int(np.count_nonzero(z != zr)))


def test_a_thread_that_ends_gives_its_workspaces_back():
    """per-thread device workspaces, page-locked bounce buffers and the side stream are released when the host thread ends:
    ten short-lived threads that each run c_ezsint on host arrays leave the device's free memory where it was"""
    import threading
    ni, nj, no, mo = 720, 360, 1440, 721
    gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 25, 25, 0, 0)
    zin = ec.synth_field(ni, nj, seed=5)
    errs = []

    def work():
        try:
            assert ez.ezdefset(gdout, gdin) == 1
            rc, z = ez.ezsint(zin, no * mo)
            assert rc == 0 and np.isfinite(z).all()
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    t = threading.Thread(target=work); t.start(); t.join()          # plans and tables (process-wide) exist after this one
    assert not errs, errs
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(10):
        t = threading.Thread(target=work); t.start(); t.join()
    assert not errs, errs
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (8 << 20), (free0, free1)               # without the release: >= 10 x (4.2 MB staging + workspaces)


def test_a_thread_on_another_device_is_refused_loudly():
    """the library binds to the device that was current at its first compute call; a call with another current device returns -1 and says which two devices
    (one process per GPU: plans, tables and workspaces are plain pointers of one device).  Simulated on the one-GPU box by binding a fresh process to device 5."""
    import subprocess, sys, textwrap
    ROOT_DIR = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    code = textwrap.dedent("""
        import sys, os
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import numpy as np
        from librmn_amd import ezscint as ez
        import ezcases as ec
        gi = ez.ezqkdef(64, 32, "G", 0, 0, 0, 0); go = ez.ezqkdef(90, 45, "L", 400, 400, 0, 0)
        assert ez.ezdefset(go, gi) == 1
        rc, z = ez.ezsint(ec.synth_field(64, 32, seed=1), 90 * 45)
        print("rc", rc)
    """ % (ROOT_DIR, ROOT_DIR))
    env = dict(os.environ); env["EZHIP_TEST_BOUND_DEVICE"] = "5"
    r = run_child([sys.executable, "-c", code], env=env, timeout=300)
    assert "rc -1" in r.stdout, (r.stdout, r.stderr[-1500:])
    assert "live on HIP device 5" in r.stderr and "current device is 0" in r.stderr, r.stderr[-1500:]


@pytest.mark.parametrize("th", [3232, 3216, 6416, 6408, -3232])
@pytest.mark.parametrize("shape", [(320, 160, 500, 250), (257, 130, 333, 167), (640, 320, 1000, 500), (1280, 640, 2000, 1000)])
@pytest.mark.parametrize("eig", ["cfg3", "tilted"])
def test_uvt_staged_tiles_equal_gathering_kernel(shape, eig, th):
    """k_uvt (wind pair from LDS-staged stencil windows, second call of a grid set on) against k_pts2_irgd3w (first call; EZHIP_NO_UVT=1): bit-identical
    outputs on rotated global sources whose target tiles include the longitude seam, both rotated poles (windows too large to stage) and ragged edges"""
    ni, nj, no, mo = shape
    ax, ay = ec.ze_axes(ni, nj)
    ig = ec.E_IG if eig == "cfg3" else ol.cxgaig("E", 62.0, 20.0, -15.0, 110.0)
    os.environ["EZHIP_UVT_SHAPE"] = str(abs(th))
    if th < 0:
        os.environ["EZHIP_UVT_NO_STREAMS"] = "1"      # the row-major x, y and matrix arrays instead of the set's tile-ordered copy
    try:
        gdin = ez.ezgdef_fmem(ni, nj, "Z", "E", *ig, ax, ay); gdout = ez.ezqkdef(no, mo, "L", *ol.cxgaig("L", -90.0, 0.0, 180.0 / (mo - 1), 360.0 / no))
        assert ez.ezdefset(gdout, gdin) == 1
        setopts(3, 1)
        uu, vv = ec.synth_wind(ni, nj, seed=11)
        for a in (uu, vv):
            a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
        ez.use_stream(torch.cuda.current_stream().cuda_stream)
        d_u = torch.from_numpy(uu).cuda(); d_v = torch.from_numpy(vv).cuda()
        outs = []
        for call in range(4):                                     # 0: k_pts2_irgd3w; 1: k_uvt (the special points inside its producer blocks, round 5); 2: their launch of their own; 3: the gathering kernel
            o_u = torch.full((no * mo,), float("nan"), dtype=torch.float32, device="cuda"); o_v = torch.full_like(o_u, float("nan"))
            if call == 2:
                os.environ["EZHIP_UVT_SPECIAL_LAUNCH"] = "1"
            if call == 3:
                os.environ["EZHIP_NO_UVT"] = "1"
            assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) == 0
            torch.cuda.synchronize()
            outs.append((o_u, o_v))
        for k in (1, 2, 3):
            assert torch.equal(outs[0][0].view(torch.int32), outs[k][0].view(torch.int32)), (k, int((outs[0][0] != outs[k][0]).sum()))
            assert torch.equal(outs[0][1].view(torch.int32), outs[k][1].view(torch.int32)), k
        assert not torch.isnan(outs[1][0]).any() and not torch.isnan(outs[1][1]).any()
    finally:
        os.environ.pop("EZHIP_UVT_SHAPE", None); os.environ.pop("EZHIP_NO_UVT", None); os.environ.pop("EZHIP_UVT_NO_STREAMS", None); os.environ.pop("EZHIP_UVT_SPECIAL_LAUNCH", None)
        ez.gdrls(gdin); ez.gdrls(gdout)


@pytest.mark.parametrize("degree", [3, 1])
@pytest.mark.parametrize("polar", [1, 0])
@pytest.mark.parametrize("shape", [(320, 160, 500, 250), (257, 130, 333, 167), (640, 320, 1000, 500), (1280, 640, 2000, 1000)])
@pytest.mark.parametrize("eig", ["cfg3", "tilted"])
def test_st_staged_tiles_equal_gathering_kernel(shape, eig, polar, degree):
    """k_st (c_ezsint from a rotated global source out of LDS-staged stencil windows, second call of a grid set on) against k_pts (first call; EZHIP_NO_ST=1):
    bit-identical fields, with the seam, both rotated poles, ragged edges, pole points and polar strips in the target; then against the oracle's gdxysint at
    the located points (the literal form of ez_irgdint_3_w)"""
    ni, nj, no, mo = shape
    ax, ay = ec.ze_axes(ni, nj)
    ig = ec.E_IG if eig == "cfg3" else ol.cxgaig("E", 62.0, 20.0, -15.0, 110.0)
    os.environ["EZHIP_ST_MIN_POINTS"] = "1"
    try:
        gdin = ez.ezgdef_fmem(ni, nj, "Z", "E", *ig, ax, ay); gdout = ez.ezqkdef(no, mo, "L", *ol.cxgaig("L", -90.0, 0.0, 180.0 / (mo - 1), 360.0 / no))
        assert ez.ezdefset(gdout, gdin) == 1
        setopts(degree, polar)                                    # 3: k_st, 1: k_st1 (the bilinear member)
        f = ec.synth_field(ni, nj, seed=17)
        f2 = f.reshape(nj, ni); f2[:, -1] = f2[:, 0]
        ez.use_stream(torch.cuda.current_stream().cuda_stream)
        d_f = torch.from_numpy(f).cuda()
        outs = []
        for call in range(5):                                     # 0: k_pts; 1, 2: k_st (the special points inside its producer blocks, round 5); 3: their launch of their own; 4: k_pts again
            o = torch.full((no * mo,), float("nan"), dtype=torch.float32, device="cuda")
            if call == 3:
                os.environ["EZHIP_ST_SPECIAL_LAUNCH"] = "1"
            if call == 4:
                os.environ["EZHIP_NO_ST"] = "1"
            assert ez.ezsint_dev(o, d_f) >= 0
            torch.cuda.synchronize()
            outs.append(o)
        for k in (1, 2, 3, 4):
            assert torch.equal(outs[0].view(torch.int32), outs[k].view(torch.int32)), (k, int((outs[0] != outs[k]).sum()))
        assert not torch.isnan(outs[1]).any()
    finally:
        os.environ.pop("EZHIP_ST_MIN_POINTS", None); os.environ.pop("EZHIP_NO_ST", None); os.environ.pop("EZHIP_ST_SPECIAL_LAUNCH", None)
        ez.gdrls(gdin); ez.gdrls(gdout)


@pytest.mark.parametrize("degree", [3, 1])
@pytest.mark.parametrize("target", ["inside", "beyond"])
@pytest.mark.parametrize("extrap", ["maximum", "value", "linear"])
@pytest.mark.parametrize("shape", [(400, 300, 700, 500), (801, 603, 1500, 1100)])
def test_st_staged_tiles_regional_source(shape, extrap, target, degree):
    """k_st on a source WITHOUT wrap (a regional Z-on-E grid: ez_irgdint_3_nw.inc, whose statement functions are REAL): second and third call of a set against the
    first (k_pts) and against EZHIP_NO_ST=1, bit for bit; a target inside the source's region and one that reaches beyond it (extrapolation zones: filled with a
    value, the field's maximum, or re-interpolated at a lower degree by the next kernel)"""
    ni, nj, no, mo = shape
    ax, ay = ec.zereg_axes(ni, nj)
    os.environ["EZHIP_ST_MIN_POINTS"] = "1"
    lat0, lon0, dlat, dlon = (-15.0, 155.0, 30.0 / (mo - 1), 40.0 / (no - 1)) if target == "inside" else (-28.0, 140.0, 56.0 / (mo - 1), 70.0 / (no - 1))
    try:
        gdin = ez.ezgdef_fmem(ni, nj, "Z", "E", *ol.cxgaig("E", 0.0, 180.0, 0.0, 0.0), ax, ay)      # the identity rotation: the rotated frame is the geographic one
        gdout = ez.ezqkdef(no, mo, "L", *ol.cxgaig("L", lat0, lon0, dlat, dlon))
        assert ez.ezdefset(gdout, gdin) == 1
        setopts(degree, 1, extrap)
        if extrap == "value":
            assert ez.ezsetval("extrap_value", -777.0) == 0
        f = ec.synth_field(ni, nj, seed=23)
        ez.use_stream(torch.cuda.current_stream().cuda_stream)
        d_f = torch.from_numpy(f).cuda()
        outs = []
        for call in range(4):
            o = torch.full((no * mo,), float("nan"), dtype=torch.float32, device="cuda")
            if call == 3:
                os.environ["EZHIP_NO_ST"] = "1"
            assert ez.ezsint_dev(o, d_f) >= 0
            torch.cuda.synchronize()
            outs.append(o)
        for k in (1, 2, 3):
            assert torch.equal(outs[0].view(torch.int32), outs[k].view(torch.int32)), (k, int((outs[0] != outs[k]).sum()))
        assert not torch.isnan(outs[1]).any()
    finally:
        os.environ.pop("EZHIP_ST_MIN_POINTS", None); os.environ.pop("EZHIP_NO_ST", None)
        ez.gdrls(gdin); ez.gdrls(gdout)


@pytest.mark.parametrize("degree", [3, 1])
@pytest.mark.parametrize("polar", [1, 0])
@pytest.mark.parametrize("kind", ["global_rotated", "regional"])
def test_st_batch_equals_single_calls(kind, polar, degree):
    """c_ezsint_batch_dev on a set with its staged-tile table: ONE k_st launch for the batch (x, y, zones, special points once) against the fields one call at a
    time, bit for bit; a rotated global source (pole points, polar strips, the seam) and a regional one"""
    os.environ["EZHIP_ST_MIN_POINTS"] = "1"
    try:
        if kind == "global_rotated":
            ni, nj, no, mo = 640, 320, 1000, 500
            ax, ay = ec.ze_axes(ni, nj)
            gdin = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); gdout = ez.ezqkdef(no, mo, "L", *ol.cxgaig("L", -90.0, 0.0, 180.0 / (mo - 1), 360.0 / no))
        else:
            ni, nj, no, mo = 400, 300, 700, 500
            ax, ay = ec.zereg_axes(ni, nj)
            gdin = ez.ezgdef_fmem(ni, nj, "Z", "E", *ol.cxgaig("E", 0.0, 180.0, 0.0, 0.0), ax, ay)
            gdout = ez.ezqkdef(no, mo, "L", *ol.cxgaig("L", -15.0, 155.0, 30.0 / (mo - 1), 40.0 / (no - 1)))
        assert ez.ezdefset(gdout, gdin) == 1
        setopts(degree, polar)
        F = 5
        fields = np.stack([ec.synth_field(ni, nj, seed=30 + f) for f in range(F)])
        if kind == "global_rotated":
            fields.reshape(F, nj, ni)[:, :, -1] = fields.reshape(F, nj, ni)[:, :, 0]
        ez.use_stream(torch.cuda.current_stream().cuda_stream)
        d_in = torch.from_numpy(fields).cuda().contiguous()
        singles = []
        for f in range(F):
            o = torch.full((no * mo,), float("nan"), dtype=torch.float32, device="cuda")
            assert ez.ezsint_dev(o, d_in[f]) >= 0
            singles.append(o)
        for mode in ("batch", "no_st_batch"):
            if mode == "no_st_batch":
                os.environ["EZHIP_NO_ST_BATCH"] = "1"
            d_out = torch.full((F, no * mo), float("nan"), dtype=torch.float32, device="cuda")
            assert ez.ezsint_batch_dev(d_out, d_in, F) >= 0
            torch.cuda.synchronize()
            for f in range(F):
                assert torch.equal(d_out[f].view(torch.int32), singles[f].view(torch.int32)), (mode, f, int((d_out[f] != singles[f]).sum()))
    finally:
        os.environ.pop("EZHIP_ST_MIN_POINTS", None); os.environ.pop("EZHIP_NO_ST_BATCH", None)
        ez.gdrls(gdin); ez.gdrls(gdout)


@pytest.mark.parametrize("target", ["inside", "beyond"])
@pytest.mark.parametrize("extrap", ["maximum", "value"])
@pytest.mark.parametrize("shape", [(400, 300, 700, 500), (801, 603, 1500, 1100)])
def test_uvt_staged_tiles_regional_source(shape, extrap, target):
    """k_uvt's literal twin on a source WITHOUT wrap (a regional rotated grid: both components in the REAL statement functions of ez_irgdint_3_nw.inc): second and
    third call of a set against the first (k_pts2) and against EZHIP_NO_UVT=1, bit for bit; a rotated frame, so the wind rotation is applied per point"""
    ni, nj, no, mo = shape
    ax, ay = ec.zereg_axes(ni, nj)
    lat0, lon0, dlat, dlon = (28.0, 255.0, 24.0 / (mo - 1), 36.0 / (no - 1)) if target == "inside" else (15.0, 235.0, 50.0 / (mo - 1), 75.0 / (no - 1))
    try:
        gdin = ez.ezgdef_fmem(ni, nj, "Z", "E", *ol.cxgaig("E", 40.0, 270.0, 50.0, 95.0), ax, ay)      # a tilted frame whose equator runs through the middle of North America
        gdout = ez.ezqkdef(no, mo, "L", *ol.cxgaig("L", lat0, lon0, dlat, dlon))
        assert ez.ezdefset(gdout, gdin) == 1
        setopts(3, 1, extrap)
        if extrap == "value":
            assert ez.ezsetval("extrap_value", -55.0) == 0
        uu, vv = ec.synth_wind(ni, nj, seed=19)
        ez.use_stream(torch.cuda.current_stream().cuda_stream)
        d_u = torch.from_numpy(uu).cuda(); d_v = torch.from_numpy(vv).cuda()
        outs = []
        for call in range(4):
            o_u = torch.full((no * mo,), float("nan"), dtype=torch.float32, device="cuda"); o_v = torch.full_like(o_u, float("nan"))
            if call == 3:
                os.environ["EZHIP_NO_UVT"] = "1"
            assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) >= 0
            torch.cuda.synchronize()
            outs.append((o_u, o_v))
        for k in (1, 2, 3):
            assert torch.equal(outs[0][0].view(torch.int32), outs[k][0].view(torch.int32)), (k, int((outs[0][0] != outs[k][0]).sum()))
            assert torch.equal(outs[0][1].view(torch.int32), outs[k][1].view(torch.int32)), k
        assert not torch.isnan(outs[1][0]).any()
    finally:
        os.environ.pop("EZHIP_NO_UVT", None)
        ez.gdrls(gdin); ez.gdrls(gdout)


def test_st_one_set_keeps_a_table_per_degree():
    """bicubic and bilinear calls alternate on one grid set: each degree builds its own staged-tile table behind its first call and uses it from its second call on;
    results equal the gathering kernels' (EZHIP_NO_ST=1) bit for bit whatever the order"""
    ni, nj, no, mo = 640, 320, 1000, 500
    ax, ay = ec.ze_axes(ni, nj)
    os.environ["EZHIP_ST_MIN_POINTS"] = "1"
    try:
        gdin = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); gdout = ez.ezqkdef(no, mo, "L", *ol.cxgaig("L", -90.0, 0.0, 180.0 / (mo - 1), 360.0 / no))
        assert ez.ezdefset(gdout, gdin) == 1
        f = ec.synth_field(ni, nj, seed=41); f.reshape(nj, ni)[:, -1] = f.reshape(nj, ni)[:, 0]
        ez.use_stream(torch.cuda.current_stream().cuda_stream)
        d_f = torch.from_numpy(f).cuda()
        want = {}
        os.environ["EZHIP_NO_ST"] = "1"
        for deg in (3, 1, 0):
            setopts(deg, 1)
            o = torch.empty(no * mo, dtype=torch.float32, device="cuda"); assert ez.ezsint_dev(o, d_f) >= 0; want[deg] = o
        os.environ.pop("EZHIP_NO_ST")
        for deg in (3, 1, 3, 0, 1, 1, 3, 3, 0, 1):
            setopts(deg, 1)
            o = torch.full((no * mo,), float("nan"), dtype=torch.float32, device="cuda")
            assert ez.ezsint_dev(o, d_f) >= 0
            assert torch.equal(o.view(torch.int32), want[deg].view(torch.int32)), deg
    finally:
        os.environ.pop("EZHIP_ST_MIN_POINTS", None); os.environ.pop("EZHIP_NO_ST", None)
        ez.gdrls(gdin); ez.gdrls(gdout)
