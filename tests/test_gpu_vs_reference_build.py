"""The product on the GPU against the reference's OWN sources (oracle/_ref/libezref.so, built where they lie and shipped with the snapshot)
on random grid pairs -- no oracle in between.  tools/fuzz_vs_ref.py: c_ezsint nearest / bilinear bit for bit, bicubic within 1e-5, c_ezuvint
within 1e-5 |V|, global / hemispheric / regional sources, every extrapolation mode; tools/fuzz_average.py: the averaging degrees bit for bit; tools/fuzz_vs_ref2.py: irregular (Z-on-L), rotated (Z-on-E, global and regional) and
polar-stereographic grids on either side, Gaussian and Z targets -- it found the two cases tested by name below; tools/fuzz_vs_ref3.py: c_gdll,
c_gdxyfll, c_gdllfxy, c_gdxysint, c_gdllsval, c_gdllvval, c_gdwdfuv, c_gduvfwd on random grids of every kind (it found the public c_gdllfxy's row
inversion, tests/test_host_cpu.py); tools/fuzz_vs_ref4.py: the mask entry points, c_ezwdint and Yin-Yang grids as source and as target (the
reference's Yin-Yang answers from a process per case: in a long session it corrupts its own heap there)."""
import os, subprocess, sys
from conftest import run_child
import pytest
import reflib

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


@pytest.mark.skipif(not reflib.have_ref(), reason="oracle/_ref/libezref.so not built")
@pytest.mark.parametrize("tool,ncases,seed", [("fuzz_vs_ref.py", 400, 1), ("fuzz_vs_ref.py", 400, 2), ("fuzz_average.py", 200, 1),
                                              ("fuzz_vs_ref2.py", 600, 1), ("fuzz_vs_ref2.py", 600, 7), ("fuzz_vs_ref3.py", 300, 1), ("fuzz_vs_ref3.py", 300, 5), ("fuzz_vs_ref4.py", 150, 1), ("fuzz_vs_ref2.py hemi", 600, 3)])
def test_random_grid_pairs_against_the_reference_build(tool, ncases, seed):
    env = dict(os.environ)
    if tool.endswith(" hemi"): tool = tool.split()[0]; env["FUZZ_HEMI"] = "1"          # + hemispheric / y-inverted A B G on either side
    # The reference build corrupts its own heap now and then in a long session ("double free or corruption", "malloc(): corrupted top size", with no
    # product call on the stack -- tools/fuzz_vs_ref4.py, DESIGN_LOG.md 2): a run that a SIGNAL ended before any mismatch was printed proves nothing about
    # parity.  Whose crash it was is settled by running the SAME seed again with the reference's entry points replaced by no-ops (FUZZ_PRODUCT_ONLY,
    # tests/reflib.py): the product then makes exactly the same calls alone.  If that run dies too, the product crashed: the test fails at once.  If it
    # survives, the death was the reference's: the comparison is repeated with another seed; three such deaths in a row fail the test.
    for attempt in range(3):
        cmd = [sys.executable, os.path.join(ROOT, "tools", tool), str(ncases), str(seed + 1000 * attempt)]
        r = run_child(cmd, cwd=ROOT, env=env)
        if r.returncode >= 0 or "MISMATCH" in r.stdout:
            break
        alone = run_child(cmd, cwd=ROOT, env=dict(env, FUZZ_PRODUCT_ONLY="1"))
        assert alone.returncode >= 0, "the PRODUCT alone dies on this sequence of calls (signal %d):\n" % -alone.returncode + alone.stderr[-2000:]
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


def _ref_and_product(src, dst, degree, polar, extrap="maximum"):
    """(reference u, v, rc), (product u, v, rc) of c_ezuvint for grids given as (ni, nj, grtyp, ig, grref, ax, ay)"""
    import ctypes
    import numpy as np
    import ezcases as ec
    from librmn_amd import ezscint as ez
    R = reflib.ref()
    fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    def define(ref_side, g):
        ni, nj, t, ig, rf, ax, ay = g
        if ax is None:
            return R.c_ezqkdef(ni, nj, t.encode(), *ig, 0) if ref_side else ez.ezqkdef(ni, nj, t, *ig)
        return R.c_ezgdef_fmem(ni, nj, t.encode(), rf.encode(), *ig, fp(ax), fp(ay)) if ref_side else ez.ezgdef_fmem(ni, nj, t, rf, *ig, ax, ay)
    ni, nj = src[:2]; n = dst[0] * dst[1]
    uu, vv = ec.synth_wind(ni, nj, seed=6)
    gi, go = define(True, src), define(True, dst)
    for name, val in (("interp_degree", degree), ("polar_correction", "yes" if polar else "no"), ("extrap_degree", extrap)):
        R.c_ezsetopt(name.encode(), val.encode()); ez.ezsetopt(name, val)
    assert R.c_ezdefset(go, gi) == 1
    ur = np.zeros(n, np.float32); vr = ur.copy()
    rcr = R.c_ezuvint(fp(ur), fp(vr), fp(uu), fp(vv))
    a, b = define(False, src), define(False, dst)
    assert ez.ezdefset(b, a) == 1
    rc, u, v = ez.ezuvint(uu, vv, n)
    for name, val in (("interp_degree", "cubic"), ("polar_correction", "yes"), ("extrap_degree", "maximum")):
        R.c_ezsetopt(name.encode(), val.encode()); ez.ezsetopt(name, val)
    return (ur, vr, rcr), (u, v, rc), (uu, vv)


@pytest.mark.skipif(not reflib.have_ref(), reason="oracle/_ref/libezref.so not built")
@pytest.mark.parametrize("degree", ["nearest", "linear", "cubic"])
@pytest.mark.parametrize("polar", [0, 1])
def test_winds_with_one_grid_on_both_sides(degree, polar):
    """gdin == gdout (ezuvint.c:68-86): the fields are copied, not interpolated, and only the wind chain touches them"""
    import numpy as np
    g = (244, 122, "G", (0, 0, 0, 0), " ", None, None)
    (ur, vr, rcr), (u, v, rc), (uu, vv) = _ref_and_product(g, g, degree, polar)
    assert rc == rcr == 0
    sc = np.maximum(np.hypot(ur, vr), 1e-3)
    assert np.all(np.abs(u - ur) <= 1e-5 * sc) and np.all(np.abs(v - vr) <= 1e-5 * sc)


@pytest.mark.skipif(not reflib.have_ref(), reason="oracle/_ref/libezref.so not built")
def test_wind_chain_overflow_of_the_speed():
    """raw cubic extrapolation far outside a polar-stereographic source reaches |V| > 1.8e19, where the REAL sqrt(u*u + v*v) of the reference's
    speed / direction chain is inf and what follows NaN (rotated target): the per-point matrix of the product must say the same, point for point"""
    import numpy as np
    no, mo = 121, 60
    ax = np.ascontiguousarray((np.arange(no) * 360.0 / (no - 1)).astype(np.float32)); ay = np.ascontiguousarray((-90.0 + (np.arange(mo) + 0.5) * 180.0 / mo).astype(np.float32))
    src = (49, 47, "N", (345, 392, 26357, 1632), " ", None, None); dst = (no, mo, "Z", (380, 332, 31125, 35041), "E", ax, ay)
    (ur, vr, rcr), (u, v, rc), _ = _ref_and_product(src, dst, "cubic", 0, "cubic")
    assert rc == rcr
    bad_r = ~(np.isfinite(ur) & np.isfinite(vr))
    assert bad_r.sum() > 0, "the case no longer reaches the overflow"
    assert np.array_equal(bad_r, ~(np.isfinite(u) & np.isfinite(v)))
    assert np.array_equal(np.isnan(ur), np.isnan(u)) and np.array_equal(np.isnan(vr), np.isnan(v))
    m = ~bad_r
    sc = np.maximum(np.hypot(ur[m].astype(np.float64), vr[m].astype(np.float64)), 1e-3)
    assert np.all(np.abs(u[m] - ur[m]) <= 2e-5 * sc) and np.all(np.abs(v[m] - vr[m]) <= 2e-5 * sc)      # extrapolated magnitudes up to 1e18: rounding of the chain, not of the interpolation


@pytest.mark.skipif(not reflib.have_ref(), reason="oracle/_ref/libezref.so not built")
@pytest.mark.parametrize("ig", [(0, 0, 0, 0), (0, 1, 0, 0), (1, 0, 0, 0), (2, 1, 0, 0)])
@pytest.mark.parametrize("degree", ["nearest", "linear", "cubic"])
def test_rotated_source_to_B_target_pole_rows(ig, degree):
    """ez_corrbgd (ez_corrval.c:146-148): the pole rows of a 'B' target behind a Z-on-E source are their sequential REAL mean -- k_corrbgd"""
    import ctypes
    import numpy as np
    import ezcases as ec
    from librmn_amd import ezscint as ez
    R = reflib.ref()
    fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    ni, nj = 65, 32; ax, ay = ec.ze_axes(ni, nj)
    no, mo = 113, (57 if ig[0] == 0 else 29)
    src = (ni, nj, "Z", ec.E_IG, "E", ax, ay); dst = (no, mo, "B", ig, " ", None, None)
    (ur, vr, rcr), (u, v, rc), _ = _ref_and_product(src, dst, degree, 1)
    assert rc == rcr
    sc = np.maximum(np.hypot(ur, vr), 1e-3)
    assert np.all(np.abs(u - ur) <= 1e-5 * sc) and np.all(np.abs(v - vr) <= 1e-5 * sc)
    zin = ec.synth_field(ni, nj, seed=5); z2 = zin.reshape(nj, ni); z2[:, -1] = z2[:, 0]
    gi = R.c_ezgdef_fmem(ni, nj, b"Z", b"E", *ec.E_IG, fp(ax), fp(ay)); go = R.c_ezqkdef(no, mo, b"B", *ig, 0)
    a = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); b = ez.ezqkdef(no, mo, "B", *ig)
    R.c_ezsetopt(b"interp_degree", degree.encode()); ez.ezsetopt("interp_degree", degree)
    assert R.c_ezdefset(go, gi) == 1 and ez.ezdefset(b, a) == 1
    zr = np.zeros(no * mo, np.float32); rcr = R.c_ezsint(fp(zr), fp(zin)); rc, z = ez.ezsint(zin, no * mo)
    R.c_ezsetopt(b"interp_degree", b"cubic"); ez.ezsetopt("interp_degree", "cubic")
    assert rc == rcr
    if degree == "cubic": assert np.all(np.abs(z - zr) <= 1e-5 * np.abs(zr))
    else: assert np.array_equal(z.view(np.uint32), zr.view(np.uint32))


@pytest.mark.skipif(not reflib.have_ref(), reason="oracle/_ref/libezref.so not built")
@pytest.mark.parametrize("grid", [(64, 16, "G", (1, 0, 0, 0)), (64, 16, "G", (2, 1, 0, 0)), (50, 12, "A", (2, 0, 0, 0)), (51, 13, "B", (1, 1, 0, 0))])
def test_winds_at_points_of_a_hemispheric_source_mirror_v_with_its_sign_changed(grid):
    """c_gdxyvval / c_gdllvval (gdxyvval.c:103-113): u is interpolated with symmetrie = SYM, v with ANTISYM -- in the mirrored hemisphere of a
    hemispheric source v changes sign (ez_xpngdag2.inc:37-41).  Found by tools/fuzz_vs_ref3.py: the product mirrored both alike."""
    import ctypes
    import numpy as np
    import ezcases as ec
    from librmn_amd import ezscint as ez
    R = reflib.ref(); L = ez._lib()
    fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    ni, nj, t, ig = grid
    uu, vv = ec.synth_wind(ni, nj, seed=9)
    if t == "B":
        for a in (uu, vv):
            a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
    gr = R.c_ezqkdef(ni, nj, t.encode(), *ig, 0); gp = ez.ezqkdef(ni, nj, t, *ig)
    rng = np.random.default_rng(4)
    lat = rng.uniform(-89, 89, 600).astype(np.float32); lon = rng.uniform(0, 360, 600).astype(np.float32)
    for degree in ("nearest", "linear", "cubic"):
        R.c_ezsetopt(b"interp_degree", degree.encode()); ez.ezsetopt("interp_degree", degree)
        ur = np.zeros(600, np.float32); vr = ur.copy(); R.c_gdllvval(gr, fp(ur), fp(vr), fp(uu), fp(vv), fp(lat), fp(lon.copy()), 600)
        rc, u, v = ez.gdllvval(gp, uu, vv, lat, lon)
        assert rc == 0
        if degree == "cubic":
            assert np.all(np.abs(u - ur) <= 1e-5 * np.maximum(np.abs(ur), 1e-3)) and np.all(np.abs(v - vr) <= 1e-5 * np.maximum(np.abs(vr), 1e-3))
        else:
            assert np.array_equal(u, ur) and np.array_equal(v, vr), degree
    R.c_ezsetopt(b"interp_degree", b"cubic"); ez.ezsetopt("interp_degree", "cubic")
