"""Pins the REAL (fp32) evaluation of wind pairs from rotated sources (k_uvt / k_pts2_irgd3w: pair_eval in ez_kernels.hip) where it can break
(the reference evaluates in REAL*8: src/interp/ez_irgdint_3_w.inc:78-235).

Rule under test (PAIR_RULE in ez_kernels.hip): the REAL sums err by c x M_eff, M_eff = max(largest |cell| of the two central stencil rows,
|w_y| x largest |cell| of an outer row); a point whose larger component is below M_eff / 3 is evaluated again in the reference's Newton form, REAL*8.  The tests hold
    (1)  |product - reference| <= 1e-5 |V|                        -- north_star's tolerance, at EVERY point, and
    (2)  |product - reference| <= 2.5e-6 M_eff + 2e-6 |V|         -- the bound that makes (1) follow from the rule for points that kept their REAL result
         (3 x 2.5e-6 + 2e-6 = 9.5e-6; the |V| term is the reference chain's own noise: its wind direction passes through REAL degrees)
against the reference's own c_ezuvint (oracle/_ref/libezref.so) run in a fresh child process (tests/ref_child.py through the suite's fork-server):
  * BASELINE cfg3 at full size, all 16 M values;
  * adversarial fields at three sizes: a vortex centre (u, v through zero under +-30 m/s neighbours), a coarse source of +-15 m/s cells of random sign
    (the calm-point-between-jets case tools/fuzz_vs_ref2.py 600 7 found, by construction: hundreds of target points with |V| < 0.1 under full-size stencils),
    and cells x 100 of alternating sign in OUTER stencil rows over calm central rows (invisible to the rule until round 6).
"""
import os, sys
import numpy as np
import pytest
import torch
from conftest import run_child
import reflib
import ezcases as ec
import oraclelib as ol
from librmn_amd import ezscint as ez

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not reflib.have_ref(), reason="oracle/_ref/libezref.so not built")]
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..")
TOL_V = 1e-5            # of |V|
C_M, C_V, RULE = 2.5e-6, 2.0e-6, 3.0        # e <= C_M M_eff + C_V |V|; a REAL result stands only where M_eff <= RULE x the larger component
assert RULE * C_M + C_V <= TOL_V


def _setopts():
    assert ez.ezsetopt("interp_degree", "cubic") == 0 and ez.ezsetopt("polar_correction", "yes") == 0


def _meff(x, y, uu, vv, ni, nj, ay):
    """M_eff of every target point whose stencil is an ordinary one (away from the longitude seam: columns i-1 .. i+2 consecutive), NaN elsewhere.
    x, y: the set's located coordinates (1-based, float32); stencil and weights as pair_eval / ez_irgdint_3_w.inc:158-235"""
    px = x.astype(np.float64); py = y.astype(np.float64)
    i = np.clip(px.astype(np.int64), 2, ni - 2); j = np.clip(py.astype(np.int64), 2, nj - 2)          # 1-based; j1 = 1, j2 = nj
    ok = (px.astype(np.int64) >= 2) & (px.astype(np.int64) <= ni - 2)
    U = np.abs(uu.reshape(nj, ni)); V = np.abs(vv.reshape(nj, ni))
    C = np.maximum(U, V)
    a = ay.astype(np.float64)
    y1, y2, y3, y4 = a[j - 2], a[j - 1], a[j], a[j + 1]
    yy = y2 + (y3 - y2) * (py - j)
    w0 = (yy - y2) * (yy - y3) * (yy - y4) / ((y1 - y2) * (y1 - y3) * (y1 - y4))
    w3 = (yy - y1) * (yy - y2) * (yy - y3) / ((y4 - y1) * (y4 - y2) * (y4 - y3))
    def rowmax(r):      # r = 0 .. 3 -> source row j - 2 + r (0-based)
        m = C[j - 2 + r, i - 2]
        for c in range(1, 4):
            m = np.maximum(m, C[j - 2 + r, i - 2 + c])
        return m
    m = np.maximum(np.maximum(rowmax(1), rowmax(2)), np.maximum(np.abs(w0) * rowmax(0), np.abs(w3) * rowmax(3)))
    m[~ok] = np.nan
    return m


def _compare(tag, u, v, ur, vr, meff):
    du = np.abs(u.astype(np.float64) - ur); dv = np.abs(v.astype(np.float64) - vr)
    e = np.maximum(du, dv)
    V = np.hypot(ur.astype(np.float64), vr.astype(np.float64))
    relv = e / np.maximum(V, 1e-3)
    k = int(np.argmax(relv))
    assert relv[k] <= TOL_V, "%s: |V| bound: %.3g at point %d: reference (%.9g, %.9g) product (%.9g, %.9g) M_eff %.6g; %d points above" % (
        tag, relv[k], k, ur[k], vr[k], u[k], v[k], meff[k], int(np.count_nonzero(relv > TOL_V)))
    okm = np.isfinite(meff) & (meff > 0)
    relm = np.zeros_like(e); relm[okm] = (e[okm] - C_V * V[okm]) / meff[okm]
    k = int(np.argmax(relm))
    assert relm[k] <= C_M, "%s: M_eff bound: (e - %.1e |V|) / M_eff = %.3g at point %d: M_eff %.6g |V| %.6g e %.3g" % (tag, C_V, relm[k], k, meff[k], V[k], e[k])
    calm = int(np.count_nonzero(okm & (np.maximum(np.abs(ur), np.abs(vr)) * RULE < meff)))
    return float(relv.max()), float(relm.max()), calm


def _product(ni, nj, no, mo, ig_src, ax, ay, dst_ig, uu, vv, degree="cubic", polar="yes"):
    """both calls of a set (first: gathering kernel + special-point listing; second: staged windows), x, y of the set"""
    gdin = ez.ezgdef_fmem(ni, nj, "Z", "E", *ig_src, ax, ay); gdout = ez.ezqkdef(no, mo, "L", *dst_ig)
    assert ez.ezdefset(gdout, gdin) == 1
    assert ez.ezsetopt("interp_degree", degree) == 0 and ez.ezsetopt("polar_correction", polar) == 0
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    d_u = torch.from_numpy(uu).cuda(); d_v = torch.from_numpy(vv).cuda()
    outs = []
    for _ in range(2):
        o_u = torch.zeros(no * mo, dtype=torch.float32, device="cuda"); o_v = torch.zeros_like(o_u)
        assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) == 0
        torch.cuda.synchronize()
        outs.append((o_u.cpu().numpy(), o_v.cpu().numpy()))
    d_x = torch.empty(no * mo, dtype=torch.float32, device="cuda"); d_y = torch.empty_like(d_x)
    assert ez.set_xy_dev(d_x, d_y) == 0
    torch.cuda.synchronize()
    x, y = d_x.cpu().numpy(), d_y.cpu().numpy()
    ez.gdrls(gdout); ez.gdrls(gdin)
    _setopts()
    return outs, x, y


@pytest.mark.parametrize("degree,polar", [(1, 1), (0, 1), (3, 0)])
def test_cfg3_every_wind_value_other_options(degree, polar, tmp_path):
    """the same whole-field comparison for bilinear and nearest-neighbour winds (other kernels in front of the same wind chain) and without the polar correction"""
    ni, nj, no, mo = 2560, 1280, 4000, 2000
    out = str(tmp_path / "cfg3_ref.npy")
    r = run_child([sys.executable, os.path.join(HERE, "ref_child.py"), "cfg3_uvint", "--reps", "0", "--out", out, "--degree", str(degree), "--polar", str(polar)], cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = np.load(out)
    ax, ay = ec.ze_axes(ni, nj)
    uu, vv = ec.synth_wind(ni, nj, seed=3)
    for a in (uu, vv):
        a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
    outs, x, y = _product(ni, nj, no, mo, ec.E_IG, ax, ay, (9, 9, 0, 0), uu, vv, {0: "nearest", 1: "linear", 3: "cubic"}[degree], "yes" if polar else "no")
    for call, (pu, pv) in enumerate(outs):
        e = np.maximum(np.abs(pu.astype(np.float64) - ref[0]), np.abs(pv.astype(np.float64) - ref[1]))
        V = np.maximum(np.hypot(ref[0].astype(np.float64), ref[1].astype(np.float64)), 1e-3)
        rel = e / V
        k = int(np.argmax(rel))
        print("cfg3 winds degree %d polar %d call %d: max err %.3g |V| over %d values (%d values differ at all)" % (degree, polar, call, rel[k], 2 * no * mo, int(np.count_nonzero(pu != ref[0]) + np.count_nonzero(pv != ref[1]))))
        assert rel[k] <= TOL_V, (degree, polar, call, float(rel[k]), k, float(ref[0][k]), float(ref[1][k]), float(pu[k]), float(pv[k]))


@pytest.mark.parametrize("degree", [0, 1, 3])
def test_cfg3_exact_winds_equal_the_reference_bit_for_bit(degree, tmp_path):
    """ezhip_set_wind_exact(1): the reference's wind chain as written on every call (k_wind_rotate with the C library's REAL trig) on components from the scalar kernels.
    Nearest, bilinear AND bicubic winds at cfg3: all 16 M values equal the reference build's bit for bit (with the pole rows from the device library's trig nine bicubic
    values differed in the last place: the exact mode's k_polar_wind uses libm_exact.h too)."""
    ni, nj, no, mo = 2560, 1280, 4000, 2000
    out = str(tmp_path / "cfg3_ref.npy")
    r = run_child([sys.executable, os.path.join(HERE, "ref_child.py"), "cfg3_uvint", "--reps", "0", "--out", out, "--degree", str(degree)], cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = np.load(out)
    ax, ay = ec.ze_axes(ni, nj)
    uu, vv = ec.synth_wind(ni, nj, seed=3)
    for a in (uu, vv):
        a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
    ez.set_wind_exact(1)
    try:
        outs, x, y = _product(ni, nj, no, mo, ec.E_IG, ax, ay, (9, 9, 0, 0), uu, vv, {0: "nearest", 1: "linear", 3: "cubic"}[degree], "yes")
    finally:
        ez.set_wind_exact(0)
    for call, (pu, pv) in enumerate(outs):
        ndiff = int(np.count_nonzero(pu.view(np.uint32) != ref[0].view(np.uint32)) + np.count_nonzero(pv.view(np.uint32) != ref[1].view(np.uint32)))
        e = np.maximum(np.abs(pu.astype(np.float64) - ref[0]), np.abs(pv.astype(np.float64) - ref[1])) / np.maximum(np.hypot(ref[0].astype(np.float64), ref[1].astype(np.float64)), 1e-3)
        print("cfg3 exact winds degree %d call %d: %d of %d values differ from the reference's bits (max %.3g |V|)" % (degree, call, ndiff, 2 * no * mo, e.max()))
        assert ndiff == 0, ndiff


def test_cfg3_every_wind_value_against_the_reference_run(tmp_path):
    ni, nj, no, mo = 2560, 1280, 4000, 2000
    out = str(tmp_path / "cfg3_ref.npy")
    r = run_child([sys.executable, os.path.join(HERE, "ref_child.py"), "cfg3_uvint", "--reps", "0", "--out", out], cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = np.load(out)
    ax, ay = ec.ze_axes(ni, nj)
    uu, vv = ec.synth_wind(ni, nj, seed=3)
    for a in (uu, vv):
        a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
    outs, x, y = _product(ni, nj, no, mo, ec.E_IG, ax, ay, (9, 9, 0, 0), uu, vv)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])          # gathering call == staged call, bit for bit
    meff = _meff(x, y, uu, vv, ni, nj, ay)
    relv, relm, calm = _compare("cfg3", outs[1][0], outs[1][1], ref[0], ref[1], meff)
    print("cfg3 full field: max err %.3g |V|, max (e - %.1e |V|) / M_eff %.3g over %d values; %d points under the M_eff / %g rule" % (relv, C_V, relm, 2 * no * mo, calm, RULE))


def _vortex(ni, nj, seed):
    """solid rotation in index space around a centre between grid nodes, +-30 m/s one cell away, saturating: u, v pass through zero inside one cell"""
    ii = np.arange(ni, dtype=np.float64)[None, :]; jj = np.arange(nj, dtype=np.float64)[:, None]
    u = np.zeros((nj, ni)); v = np.zeros((nj, ni))
    for c, (fi, fj) in enumerate(((0.31, 0.42), (0.62, 0.58), (0.13, 0.77), (0.83, 0.21))):
        i0 = int(fi * ni) + 0.37 + 0.1 * c; j0 = int(fj * nj) + 0.61 - 0.1 * c
        r2 = (ii - i0) ** 2 + (jj - j0) ** 2
        g = 30.0 / np.sqrt(np.maximum(r2, 0.5)) * np.exp(-r2 / (0.02 * ni * nj))
        u += -(jj - j0) * g; v += (ii - i0) * g
    u += 0.02 * (ec.hash_uniform(seed, ni * nj).reshape(nj, ni) - 0.5); v += 0.02 * (ec.hash_uniform(seed + 1, ni * nj).reshape(nj, ni) - 0.5)
    return u, v


def _jets(ni, nj, seed):
    """+-15 m/s cells of random sign (a coarse source between jets): the interpolants cross zero all over the target under full-size stencils"""
    u = 15.0 * np.sign(ec.hash_uniform(seed, ni * nj).reshape(nj, ni) - 0.5) * (0.6 + 0.4 * ec.hash_uniform(seed + 2, ni * nj).reshape(nj, ni))
    v = 15.0 * np.sign(ec.hash_uniform(seed + 1, ni * nj).reshape(nj, ni) - 0.5) * (0.6 + 0.4 * ec.hash_uniform(seed + 3, ni * nj).reshape(nj, ni))
    return u.astype(np.float64), v.astype(np.float64)


def _outer_rows(ni, nj, seed):
    """calm winds (|u|, |v| <= 1) everywhere but in two source rows out of every seven, which hold cells of +-100 .. 300 of alternating sign: for target points
    between the rows in the middle of such a pair the large cells sit in the stencil's OUTER rows only, and cancel in places"""
    u = 2.0 * (ec.hash_uniform(seed, ni * nj).reshape(nj, ni) - 0.5); v = 2.0 * (ec.hash_uniform(seed + 1, ni * nj).reshape(nj, ni) - 0.5)
    alt = np.where(np.arange(ni) % 2 == 0, 1.0, -1.0)[None, :]
    amp = 100.0 + 200.0 * ec.hash_uniform(seed + 2, ni * nj).reshape(nj, ni)
    for j0 in range(3, nj - 4, 7):
        for j in (j0, j0 + 3):          # rows j0 + 1, j0 + 2 stay calm: the band between them sees rows j0 and j0 + 3 as its outer rows
            u[j] = alt * amp[j]; v[j] = -alt * amp[j][::-1]
    return u, v


@pytest.mark.parametrize("shape", [(320, 160, 500, 250), (640, 320, 1000, 500), (1280, 640, 2000, 1000)])
@pytest.mark.parametrize("kind", ["vortex", "jets", "outer_rows"])
def test_adversarial_winds_against_the_reference_run(kind, shape, tmp_path):
    ni, nj, no, mo = shape
    ax, ay = ec.ze_axes(ni, nj)
    u, v = {"vortex": _vortex, "jets": _jets, "outer_rows": _outer_rows}[kind](ni, nj, 11 + ni)
    u[:, -1] = u[:, 0]; v[:, -1] = v[:, 0]                       # the global Z grid's duplicate column
    uu = np.ascontiguousarray(u.astype(np.float32).reshape(-1)); vv = np.ascontiguousarray(v.astype(np.float32).reshape(-1))
    dst_ig = ol.cxgaig("L", -90.0, 0.0, 180.0 / (mo - 1), 360.0 / no)
    case = str(tmp_path / "case.npz"); out = str(tmp_path / "out.npz")
    np.savez(case, src_ni=ni, src_nj=nj, src_grtyp="Z", src_grref="E", src_ig=np.array(ec.E_IG), src_ax=ax, src_ay=ay,
             dst_ni=no, dst_nj=mo, dst_grtyp="L", dst_grref=" ", dst_ig=np.array(dst_ig), dst_ax=np.zeros(0, np.float32), dst_ay=np.zeros(0, np.float32),
             degree=3, polar=1, uu=uu, vv=vv)
    r = run_child([sys.executable, os.path.join(HERE, "ref_child.py"), "uvint_case", case, out], cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    R = np.load(out)
    assert int(R["rc"]) == 0
    outs, x, y = _product(ni, nj, no, mo, ec.E_IG, ax, ay, dst_ig, uu, vv)
    meff = _meff(x, y, uu, vv, ni, nj, ay)
    for call, (pu, pv) in enumerate(outs):
        relv, relm, calm = _compare("%s %s call %d" % (kind, shape, call), pu, pv, R["ur"], R["vr"], meff)
    assert calm > 0, "the case does not exercise the second pass"
    print("%s %dx%d -> %dx%d: max err %.3g |V|, max (e - %.1e |V|) / M_eff %.3g; %d of %d points under the M_eff / %g rule" % (kind, ni, nj, no, mo, relv, C_V, relm, calm, no * mo, RULE))


@pytest.mark.parametrize("which", ["cfg2", "cfg3"])
def test_every_scalar_value_against_the_reference_run(which, tmp_path):
    """c_ezsint bicubic with polar correction at BASELINE configs[1] (k_sepx) and on cfg3's grid pair (k_st): ALL output values against the reference's own run in a
    fresh child -- the sampled rows / columns of tests/test_gpu_interp.py cover 0.5 % of them.  Pure relative error <= 1e-5 at every point (measured: 0 values differ at all)."""
    out = str(tmp_path / "ref.npy")
    r = run_child([sys.executable, os.path.join(HERE, "ref_child.py"), which + "_sint", "--out", out], cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    want = np.load(out).astype(np.float64)
    if which == "cfg2":
        ni, nj, no, mo = 4400, 2200, 7200, 3601
        gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
        zin = ec.synth_field(ni, nj, seed=2)
    else:
        ni, nj, no, mo = 2560, 1280, 4000, 2000
        ax, ay = ec.ze_axes(ni, nj)
        gdin = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); gdout = ez.ezqkdef(no, mo, "L", 9, 9, 0, 0)
        zin = ec.synth_wind(ni, nj, seed=3)[0]; z2 = zin.reshape(nj, ni); z2[:, -1] = z2[:, 0]
    assert ez.ezdefset(gdout, gdin) == 1
    _setopts()
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    d_in = torch.from_numpy(zin).cuda()
    for call in range(2):          # cfg3: the first call of a set gathers, the second runs from staged windows
        d_out = torch.zeros(no * mo, dtype=torch.float32, device="cuda")
        assert ez.ezsint_dev(d_out, d_in) == 0
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().astype(np.float64)
        d = np.abs(got - want); aw = np.abs(want)
        rel = float((d / np.maximum(aw, 1e-30)).max())
        nbits = int(np.count_nonzero(got != want))
        print("%s c_ezsint call %d: %d of %d values differ from the reference's; max relative error %.3g" % (which, call, nbits, want.size, rel))
        assert rel <= TOL_V, (which, call, rel)
    ez.gdrls(gdout); ez.gdrls(gdin)


def test_exact_winds_through_the_batch_entry_and_back_to_the_default():
    """c_ezuvint_batch_dev under ezhip_set_wind_exact(1) goes pair by pair through the exact route (same bits as single exact calls); switching the mode off brings the
    default route's bits back (the set's caches of either mode do not leak into the other)"""
    ni, nj, no, mo = 640, 320, 1000, 500
    ax, ay = ec.ze_axes(ni, nj)
    gdin = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); gdout = ez.ezqkdef(no, mo, "L", *ol.cxgaig("L", -90.0, 0.0, 180.0 / (mo - 1), 360.0 / no))
    assert ez.ezdefset(gdout, gdin) == 1
    _setopts()
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    K = 3
    winds = [ec.synth_wind(ni, nj, seed=50 + k) for k in range(K)]
    for u_, v_ in winds:
        for a in (u_, v_):
            a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
    d_u = torch.from_numpy(np.stack([w[0] for w in winds])).cuda().contiguous(); d_v = torch.from_numpy(np.stack([w[1] for w in winds])).cuda().contiguous()
    def single(k):
        o_u = torch.zeros(no * mo, device="cuda"); o_v = torch.zeros_like(o_u)
        assert ez.ezuvint_dev(o_u, o_v, d_u[k], d_v[k]) == 0
        torch.cuda.synchronize()
        return o_u, o_v
    def batch():
        o_u = torch.zeros((K, no * mo), device="cuda"); o_v = torch.zeros_like(o_u)
        assert ez.ezuvint_batch_dev(o_u, o_v, d_u, d_v, K) == 0
        torch.cuda.synchronize()
        return o_u, o_v
    fast0 = [single(k) for k in range(K)]; fast0 = [single(k) for k in range(K)]           # (second calls: the staged-tile route)
    fast_b = batch()
    ez.set_wind_exact(1)
    try:
        exact = [single(k) for k in range(K)]
        exact_b = batch()
    finally:
        ez.set_wind_exact(0)
    fast1 = [single(k) for k in range(K)]
    fast_b1 = batch()
    for k in range(K):
        assert torch.equal(exact_b[0][k], exact[k][0]) and torch.equal(exact_b[1][k], exact[k][1]), k
        assert torch.equal(fast_b[0][k], fast0[k][0]) and torch.equal(fast_b[1][k], fast0[k][1]), k
        assert torch.equal(fast1[k][0], fast0[k][0]) and torch.equal(fast1[k][1], fast0[k][1]), k
        assert torch.equal(fast_b1[0][k], fast0[k][0]) and torch.equal(fast_b1[1][k], fast0[k][1]), k
        assert not torch.equal(exact[k][0], fast0[k][0])                                     # the two modes do differ (in the last places)
        e = float(((exact[k][0] - fast0[k][0]).abs().max() / exact[k][0].abs().max()).item())
        assert e <= 1e-5, e
    ez.gdrls(gdout); ez.gdrls(gdin)
