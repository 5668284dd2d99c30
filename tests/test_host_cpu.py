"""CPU-only checks of the product library: it loads, exports every symbol declared in include/*.h,
its host front-end reproduces the reference's setup math (golden vectors), and compute entry points
fail loudly without a GPU (no silent CPU fallback)."""
import os, re
import numpy as np
import pytest

import librmn_amd
from librmn_amd import ezscint as ez
import ezcases as ec
from conftest import run_child, both_legs

ROOT = os.path.join(os.path.dirname(__file__), "..")
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "ez_golden.npz"))


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(librmn_amd.library_path()):
        librmn_amd.build_library()


def declared_symbols():
    syms = []
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"^INTERPV_F_DECL\((\w+)\)", text, flags=re.M):        # the four forms of every src/interpv routine
            syms += [m.group(1) + sfx for sfx in ("_", "8_", "_x_", "_x8_")]
        text = re.sub(r"^[ \t]*#[ \t]*define(?:.*\\\n)*.*$", "", text, flags=re.M)    # macro bodies declare nothing
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text):
            name = m.group(1)
            if name in ("defined", "sizeof") or name.isupper():
                continue
            syms.append(name)
    return sorted(set(syms))


@both_legs
def test_library_exports_every_declared_symbol(leg):
    L = librmn_amd.load_library()
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, missing
    assert len(declared_symbols()) >= 30


@both_legs
def test_gaussian_latitudes_match_reference_golden(leg):
    for nj in (8, 32, 200, 2200):
        gd = ez.ezqkdef(2 * nj, nj, "G", 0, 0, 0, 0)
        rc, ax, ay = ez.gdgaxes(gd, 2 * nj, nj)
        assert rc == 0
        assert np.array_equal(ay.view(np.uint32), GOLD[f"gausslat_{nj}"].view(np.uint32))


def _define(spec):
    ni, nj, grtyp, ig, grref, axes = spec
    if grtyp in ("Z", "Y"):
        ax, ay = axes(ni, nj)
        return ez.ezgdef_fmem(ni, nj, grtyp, grref, ig[0], ig[1], ig[2], ig[3], ax, ay)
    return ez.ezqkdef(ni, nj, grtyp, ig[0], ig[1], ig[2], ig[3])


@both_legs
@pytest.mark.parametrize("name", sorted(ec.scalar_cases()))
def test_gdll_and_host_locate_match_reference_golden(name, leg):
    case = ec.scalar_cases()[name]
    gdin = _define(case["src"]); gdout = _define(ec.dst_spec(case))
    no, mo = case["dst"][:2]
    rc, lat, lon = ez.gdll(gdout, no * mo)
    assert rc == 0
    # the golden lon was read AFTER the reference's locate had edited it in place;
    # reproduce the same call order: define the set, force the 1-D locate, then read
    ez.ezdefset(gdout, gdin)
    ez.set_mode()
    rc, lat, lon = ez.gdll(gdout, no * mo)
    assert np.array_equal(lat, GOLD[f"{name}/lat"])
    assert np.array_equal(lon, GOLD[f"{name}/lon"])
    rc, x, y = ez.gdxyfll(gdin, GOLD[f"{name}/lat"], GOLD[f"{name}/lon"])
    assert rc == 0
    assert np.array_equal(x.view(np.uint32), GOLD[f"{name}/x"].view(np.uint32))
    assert np.array_equal(y.view(np.uint32), GOLD[f"{name}/y"].view(np.uint32))


@both_legs
def test_grid_table_semantics(leg):
    a = ez.ezqkdef(30, 15, "L", 100, 100, 0, 0)
    b = ez.ezqkdef(30, 15, "L", 100, 100, 0, 0)
    assert a == b and a >= 0                       # identical definitions dedupe (ez_identifygrid.c)
    assert ez.ezqkdef(30, 15, "N", 455, 505, 2100, 1000) >= 0  # polar-stereographic: supported
    assert ez.ezqkdef(30, 15, "!", 1, 1, 1, 1) == -1       # Lambert: out of scope, rejected loudly
    assert ez.ezqkdef(30, 15, "G", 1, 0, 0, 0) >= 0        # hemispheric Gaussian: supported (scalars)
    assert ez.ezqkdef(30, 15, "G", 3, 0, 0, 0) == -1       # ig1 is 0, 1 or 2
    assert ez.ezdefset(a, 9999) == -1


@both_legs
def test_options_round_trip(leg):
    assert ez.ezsetopt("INTERP_DEGREE", "LINEAR") == 0 and ez.ezgetopt("interp_degree") == "linear"
    assert ez.ezsetopt("degre_interp", "cubique") == 0 and ez.ezgetopt("interp_degree") == "cubic"
    assert ez.ezgetopt("degre_interp") == "cubique"
    assert ez.ezsetopt("extrap_degree", "value") == 0 and ez.ezgetopt("extrap_degree") == "value"
    assert ez.ezsetopt("extrap_degree", "maximum") == 0
    assert ez.ezsetopt("polar_correction", "non") == 0 and ez.ezgetopt("polar_correction") == "no"
    assert ez.ezsetopt("polar_correction", "oui") == 0
    assert ez.ezsetopt("interp_degree", "quintic") == -1
    assert ez.ezsetopt("no_such_option", "yes") == -1


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    gdin = ez.ezqkdef(64, 32, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(90, 46, "L", 400, 400, 0, 0)
    ez.ezdefset(gdout, gdin)
    rc, z = ez.ezsint(np.zeros(64 * 32, np.float32), 90 * 46)
    assert rc == -1                                 # no CPU fallback


@both_legs
def test_subgrid_queries_and_fll(leg):
    """c_ezget_nsubgrids / c_ezget_subgridids (ezget_nsubgrids.c, ezget_subgridids.c), c_ezgdef_fll == 'Y' on 'L' (ezgdef_fll.c)"""
    ax, ay = ec.yy_axes(ec.YY_NI, ec.YY_NJ)
    gy = ez.ezgdef_fmem(ec.YY_NI, ec.YY_NJ, "Z", "E", *ec.YIN_IG, ax, ay); ga = ez.ezgdef_fmem(ec.YY_NI, ec.YY_NJ, "Z", "E", *ec.YAN_IG, ax, ay)
    gu = ez.ezgdef_supergrid(ec.YY_NI, 2 * ec.YY_NJ, "U", "F", 1, [gy, ga])
    assert ez.ezget_nsubgrids(gu) == 2 and ez.ezget_nsubgrids(gy) == 1
    n, ids = ez.ezget_subgridids(gu)
    assert n == 2 and list(ids) == [gy, ga]
    n, ids = ez.ezget_subgridids(gy)
    assert n == 1 and list(ids) == [gy]
    lon, lat = ec.cloud_axes(37, 11)
    g1 = ez.ezgdef_fll(37, 11, lat, lon)
    assert g1 >= 0 and g1 == ez.ezgdef_fmem(37, 11, "Y", "L", 100, 100, 9000, 0, lon, lat)      # cxgaig('L', 0, 0, 1, 1)


@both_legs
def test_gdxyzfll_vs_reference(leg):
    """c_gdxyzfll (host only): regular types == c_gdxyfll, 'Z' grids in reference-grid coordinates -- against the reference build"""
    import reflib as rl
    if not rl.have_ref():
        pytest.skip("oracle/_ref/libezref.so not built")
    L = rl.ref()
    n = 500
    lat = (ec.hash_uniform(5, n).astype(np.float64) * 170.0 - 85.0).astype(np.float32)
    lon = (ec.hash_uniform(6, n).astype(np.float64) * 360.0).astype(np.float32)
    ax, ay = ec.ze_axes(65, 32)
    specs = [(lambda: ez.ezgdef_fmem(65, 32, "Z", "E", *ec.E_IG, ax, ay), lambda: L.c_ezgdef_fmem(65, 32, b"Z", b"E", *ec.E_IG, rl.fptr(ax), rl.fptr(ay))),
             (lambda: ez.ezqkdef(101, 91, "N", *ec.N_IG), lambda: L.c_ezqkdef(101, 91, b"N", *ec.N_IG, 0)),
             (lambda: ez.ezqkdef(40, 20, "L", 900, 900, 450, 0), lambda: L.c_ezqkdef(40, 20, b"L", 900, 900, 450, 0, 0))]
    for mine, theirs in specs:
        g = mine(); gr = theirs()
        rc, x, y = ez.gdxyzfll(g, lat, lon)
        xr = np.zeros(n, np.float32); yr = np.zeros(n, np.float32); lo2 = lon.copy()
        L.c_gdxyzfll(gr, rl.fptr(xr), rl.fptr(yr), rl.fptr(lat.copy()), rl.fptr(lo2), n)
        assert rc == 0 and np.array_equal(x.view(np.uint32), xr.view(np.uint32)) and np.array_equal(y.view(np.uint32), yr.view(np.uint32))


@both_legs
def test_gdllfxy_vs_reference(leg):
    """c_gdllfxy (host only), every supported grid type, against the reference build: bit-exact"""
    import reflib as rl
    if not rl.have_ref():
        pytest.skip("oracle/_ref/libezref.so not built")
    L = rl.ref()
    n = 600
    ax, ay = ec.ze_axes(65, 32)
    specs = {"ZE": (65, 32, lambda: ez.ezgdef_fmem(65, 32, "Z", "E", *ec.E_IG, ax, ay), lambda: L.c_ezgdef_fmem(65, 32, b"Z", b"E", *ec.E_IG, rl.fptr(ax), rl.fptr(ay))),
             "N": (101, 91, lambda: ez.ezqkdef(101, 91, "N", *ec.N_IG), lambda: L.c_ezqkdef(101, 91, b"N", *ec.N_IG, 0)),
             "S": (81, 121, lambda: ez.ezqkdef(81, 121, "S", *ec.S_IG), lambda: L.c_ezqkdef(81, 121, b"S", *ec.S_IG, 0)),
             "L": (40, 20, lambda: ez.ezqkdef(40, 20, "L", 900, 900, 450, 0), lambda: L.c_ezqkdef(40, 20, b"L", 900, 900, 450, 0, 0)),
             "A": (48, 24, lambda: ez.ezqkdef(48, 24, "A", 0, 0, 0, 0), lambda: L.c_ezqkdef(48, 24, b"A", 0, 0, 0, 0, 0)),
             "B": (49, 25, lambda: ez.ezqkdef(49, 25, "B", 0, 0, 0, 0), lambda: L.c_ezqkdef(49, 25, b"B", 0, 0, 0, 0, 0)),
             "G": (64, 32, lambda: ez.ezqkdef(64, 32, "G", 0, 0, 0, 0), lambda: L.c_ezqkdef(64, 32, b"G", 0, 0, 0, 0, 0)),
             "E": (41, 20, lambda: ez.ezqkdef(41, 20, "E", *ec.E_IG), lambda: L.c_ezqkdef(41, 20, b"E", *ec.E_IG, 0))}
    for name, (ni, nj, mine, theirs) in specs.items():
        x = (ec.hash_uniform(25, n).astype(np.float64) * (ni + 1.0) - 0.5).astype(np.float32)
        y = (ec.hash_uniform(26, n).astype(np.float64) * (nj + 1.0) - 0.5).astype(np.float32)
        x[0] = 1.0; y[0] = 1.0; x[1] = float(ni); y[1] = float(nj)
        if name == "N":
            x[2] = ec.N_IG[1] * 0.1; y[2] = ec.N_IG[0] * 0.1      # the pole itself (pi, pj)
        g = mine(); gr = theirs()
        rc, lat, lon = ez.gdllfxy(g, x, y)
        latr = np.zeros(n, np.float32); lonr = np.zeros(n, np.float32)
        L.c_gdllfxy(gr, rl.fptr(latr), rl.fptr(lonr), rl.fptr(x.copy()), rl.fptr(y.copy()), n)
        assert rc == 0 and np.array_equal(lat.view(np.uint32), latr.view(np.uint32)), (name, int((lat != latr).sum()))
        assert np.array_equal(lon.view(np.uint32), lonr.view(np.uint32)), (name, int((lon != lonr).sum()))


@both_legs
def test_fortran_twins_by_reference_and_hidden_lengths(leg):
    """the Fortran-ABI twins (f77name(x) = x_): scalars by reference, blank-padded strings with hidden trailing lengths"""
    import ctypes
    L = ez._lib()
    i = lambda v: ctypes.byref(ctypes.c_int32(v))
    g1 = L.ezqkdef_(i(40), i(20), b"L   ", i(900), i(900), i(450), i(0), i(0), 4)
    assert g1 == ez.ezqkdef(40, 20, "L", 900, 900, 450, 0)                      # same grid: deduplicated
    ax, ay = ec.ze_axes(65, 32)
    g2 = L.ezgdef_fmem_(i(65), i(32), b"Z", b"E", i(ec.E_IG[0]), i(ec.E_IG[1]), i(ec.E_IG[2]), i(ec.E_IG[3]), ctypes.c_void_p(ax.ctypes.data), ctypes.c_void_p(ay.ctypes.data), 1, 1)
    assert g2 == ez.ezgdef_fmem(65, 32, "Z", "E", *ec.E_IG, ax, ay)
    assert L.ezdefset_(i(g1), i(g2)) == 1
    assert L.ezsetopt_(b"INTERP_DEGREE   ", b"LINEAR  ", 16, 8) == 0            # upper case, blank padded
    assert ez.ezgetopt("interp_degree") == "linear"
    assert L.ezsetopt_(b"interp_degree", b"cubic", 13, 5) == 0
    lat = np.zeros(800, np.float32); lon = np.zeros(800, np.float32)
    assert L.gdll_(i(g1), ctypes.c_void_p(lat.ctypes.data), ctypes.c_void_p(lon.ctypes.data)) == 0 and abs(float(lat[0]) + 85.5) < 1e-4
    ids = np.array([g2, g2], np.int32)
    assert L.ezgdef_supergrid_(i(65), i(64), b"U", b"F", i(1), i(2), ctypes.c_void_p(ids.ctypes.data), 1, 1) >= 0


def test_threaded_host_locate_equals_serial():
    """c_gdxyfll slices large point sets over host threads (>= 400 k points): same bits as the serial path"""
    ax, ay = ec.ze_axes(65, 32)
    g = ez.ezgdef_fmem(65, 32, "Z", "E", *ec.E_IG, ax, ay)
    n = 900_001
    lat = (ec.hash_uniform(55, n).astype(np.float64) * 178.0 - 89.0).astype(np.float32)
    lon = (ec.hash_uniform(56, n).astype(np.float64) * 360.0).astype(np.float32)
    rc, x, y = ez.gdxyfll(g, lat, lon)
    assert rc == 0
    for a, b in ((0, 300_000), (300_000, 600_000), (600_000, n)):
        rc, xs, ys = ez.gdxyfll(g, lat[a:b], lon[a:b])
        assert rc == 0 and np.array_equal(xs.view(np.uint32), x[a:b].view(np.uint32)) and np.array_equal(ys.view(np.uint32), y[a:b].view(np.uint32))


def test_vertical_interpolation_fails_loudly_without_gpu():
    """device entry points return -1; the reference's Fortran subroutines (no status argument) abort the program"""
    import subprocess, sys, torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from librmn_amd import interpv as V
    z = np.zeros((4, 8), np.float32)
    assert V.findpos_dev(8, z, np.zeros((4, 8), np.int32), z) == -1
    assert V.column_dev(V.LINEAR, V.X_NONE, 8, z, z, z, None, z, z, z) == -1
    code = ("import numpy as np; from librmn_amd import interpv as V; z = np.ones((4, 8), np.float32);"
            "V.findpos(8, z, z); print('survived')")
    r = run_child([sys.executable, "-c", code], cwd=ROOT)
    assert r.returncode != 0 and "survived" not in r.stdout and "no CPU fallback" in r.stderr


@both_legs
def test_hash_tiles_through_fmem_are_tracked_as_their_own_grids(leg):
    """a regional '#' tile is computed as a 'Z' grid but keeps its own handle and reports '#' (c_ezgprm); global tiles and tiles
    without axes are refused"""
    import ctypes, ezcases as ec
    ni, nj = 51, 41
    ax, ay = ec.zereg_axes(ni, nj)
    gz = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay)
    gh = ez.ezgdef_fmem(ni, nj, "#", "E", *ec.E_IG, ax, ay)
    assert gz >= 0 and gh >= 0 and gz != gh
    assert ez.ezgdef_fmem(ni, nj, "#", "E", *ec.E_IG, ax, ay) == gh
    L = librmn_amd.load_library()
    t = ctypes.create_string_buffer(4); v = [ctypes.c_int32() for _ in range(6)]
    assert L.c_ezgprm(gh, t, *[ctypes.byref(x) for x in v]) == 0 and t.value[:1] == b"#"
    assert L.c_ezgprm(gz, t, *[ctypes.byref(x) for x in v]) == 0 and t.value[:1] == b"Z"
    gax = (np.arange(ni) * (360.0 / (ni - 1))).astype(np.float32)
    assert ez.ezgdef_fmem(ni, nj, "#", "E", *ec.E_IG, gax, ay) == -1            # a global tile: other polar kernels than 'Z' in the reference


@both_legs
@pytest.mark.parametrize("ig", [(2, 0, 0, 0), (1, 0, 0, 0), (2, 1, 0, 0), (0, 1, 0, 0)])
def test_hemispheric_and_inverted_gaussian_grids_locate_like_the_reference(ig, leg):
    """c_gdll, c_gdxyfll and c_gdxyfll_orig of hemispheric / y-inverted 'G' grids against the reference build: the table of 2 nj
    latitudes, the two search lengths (gr.nj for a set, gr.j2 for c_gdxyfll), the northern shift, the public routine's row inversion"""
    import reflib
    if not reflib.have_ref():
        pytest.skip("oracle/_ref/libezref.so not built")
    R = reflib.ref(); L = librmn_amd.load_library()
    import ctypes
    fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    ni, nj = 64, 16
    gr = R.c_ezqkdef(ni, nj, b"G", *ig, 0); gp = ez.ezqkdef(ni, nj, "G", *ig)
    assert gr >= 0 and gp >= 0
    la_r = np.zeros(ni * nj, np.float32); lo_r = np.zeros(ni * nj, np.float32); la_p = la_r.copy(); lo_p = la_r.copy()
    assert R.c_gdll(gr, fp(la_r), fp(lo_r)) >= 0 and L.c_gdll(gp, fp(la_p), fp(lo_p)) >= 0
    assert np.array_equal(la_r, la_p) and np.array_equal(lo_r, lo_p)
    rng = np.random.default_rng(3)
    lat = rng.uniform(-90, 90, 500).astype(np.float32); lon = rng.uniform(0, 360, 500).astype(np.float32)
    for name in ("c_gdxyfll", "c_gdxyfll_orig"):
        xr = np.zeros(500, np.float32); yr = xr.copy(); xp = xr.copy(); yp = xr.copy()
        getattr(R, name)(gr, fp(xr), fp(yr), fp(lat), fp(lon.copy()), 500)
        getattr(L, name)(gp, fp(xp), fp(yp), fp(lat), fp(lon.copy()), 500)
        assert np.array_equal(xr, xp) and np.array_equal(yr, yp), (ig, name)
    # the way back: the public c_gdllfxy is c_gdllfxy_new (gdllfxy.c:103-250), which counts the rows of a grid with ig2 == 1 from the north
    # (:190-195); found by tools/fuzz_vs_ref3.py, the product used the internal form (c_gdllfxy_orig) for both
    x = rng.uniform(0.6, ni + 0.4, 500).astype(np.float32); y = rng.uniform(0.6, nj + 0.4, 500).astype(np.float32)
    x[:50] = np.round(x[:50]).clip(1, ni); y[:50] = np.round(y[:50]).clip(1, nj)
    la_r = np.zeros(500, np.float32); lo_r = la_r.copy(); la_p = la_r.copy(); lo_p = la_r.copy()
    R.c_gdllfxy(gr, fp(la_r), fp(lo_r), fp(x), fp(y), 500); L.c_gdllfxy(gp, fp(la_p), fp(lo_p), fp(x), fp(y), 500)
    assert np.array_equal(la_r, la_p) and np.array_equal(lo_r, lo_p), ig


@both_legs
def test_options_set_and_read_back_like_the_reference(leg):
    """c_ezsetopt / c_ezgetopt / c_ezsetval / c_ezgetval / c_ezsetival / c_ezgetival against the reference build: every option name (English, French,
    upper case, unknown) with every value (synonyms, unknown, empty): the same return codes and the same strings read back (ezsetopt.c:59-215)"""
    import ctypes
    import reflib
    if not reflib.have_ref():
        pytest.skip("oracle/_ref/libezref.so not built")
    R = reflib.ref(); L = librmn_amd.load_library()
    opts = ["interp_degree", "degre_interp", "extrap_degree", "degre_extrap", "polar_correction", "correction_polaire", "verbose", "cloud_interp_alg",
            "use_1subgrid", "use_1sousgrille", "INTERP_DEGREE", "Extrap_Degree", "bogus_option", "missing_interp_alg", "extrap_value", "subgridid"]
    vals = ["nearest", "voisin", "linear", "lineair", "lineaire", "cubic", "cubique", "CUBIC", "average", "sph_average", "neutral", "neutre", "maximum",
            "minimum", "value", "valeur", "abort", "yes", "oui", "no", "non", "yesyesyes", "ouiouioui", "distance", "bogus", ""]
    getn = ["interp_degree", "extrap_degree", "polar_correction", "verbose", "cloud_interp_alg", "use_1subgrid", "degre_interp", "INTERP_DEGREE", "bogus_option"]

    def get(lib, name):
        b = ctypes.create_string_buffer(64)
        return lib.c_ezgetopt(name.encode(), b), b.value
    try:
        for o in opts:
            for v in vals:
                assert R.c_ezsetopt(o.encode(), v.encode()) == L.c_ezsetopt(o.encode(), v.encode()), (o, v)
                for n in getn:
                    assert get(R, n) == get(L, n), (o, v, n)
        for name, x in (("extrap_value", -3.5), ("extrap_value", 1e30), ("bogus", 1.0), ("EXTRAP_VALUE", 2.0)):
            a = ctypes.c_float(); b = ctypes.c_float()
            assert R.c_ezsetval(name.encode(), ctypes.c_float(x)) == L.c_ezsetval(name.encode(), ctypes.c_float(x))
            assert (R.c_ezgetval(name.encode(), ctypes.byref(a)), a.value) == (L.c_ezgetval(name.encode(), ctypes.byref(b)), b.value), name
        for name, x in (("subgridid", 3), ("bogus", 1), ("SUBGRIDID", 7)):
            a = ctypes.c_int(); b = ctypes.c_int()
            assert R.c_ezsetival(name.encode(), x) == L.c_ezsetival(name.encode(), x)
            assert (R.c_ezgetival(name.encode(), ctypes.byref(a)), a.value) == (L.c_ezgetival(name.encode(), ctypes.byref(b)), b.value), name
    finally:
        for lib in (R, L):
            for o, v in (("interp_degree", "cubic"), ("extrap_degree", "maximum"), ("polar_correction", "yes"), ("verbose", "no"), ("cloud_interp_alg", "distance"), ("use_1subgrid", "no")):
                lib.c_ezsetopt(o.encode(), v.encode())
            lib.c_ezsetval(b"extrap_value", ctypes.c_float(0.0))
