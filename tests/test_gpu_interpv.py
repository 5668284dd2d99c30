"""GPU parity of the vertical interpolation: the HIP path through the C ABI -- the reference's own
Fortran-callable symbols on host arrays and the device-pointer entry points -- against the oracle, bit for bit
(REAL and REAL*8, levels ascending and descending, every LDS tiling of the kernel), plus the reference test program's
own data / criteria and the hand-derived known answers."""
import os
from conftest import run_child
import subprocess
import sys
import numpy as np
import pytest
import interpvcases as iv

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
from librmn_amd import interpv as V          # noqa: E402

ALGO = {"nearestneighbour": V.NEAREST, "linear": V.LINEAR, "cubiclagrange": V.CUBIC_LAGRANGE, "cubicwithderivs": V.CUBIC_DERIVS}
XKIND = {"fixed": V.X_FIXED, "lapserate": V.X_LAPSERATE}
#          n    ns   nd  sij  dij
SHAPES = [(7, 2, 5, 7, 7), (64, 5, 9, 64, 70), (33, 28, 17, 40, 33), (300, 80, 61, 300, 310), (1, 4, 3, 2, 3), (1000, 4, 130, 1000, 1000)]
# srcNumLevels that push the float / double kernel through 64, 32, 16 columns per block and the no-LDS form
TILINGS = [(500, 80), (500, 200), (300, 400), (200, 700), (130, 1400), (70, 2800)]


def _same(a, b):
    return a.tobytes() == b.tobytes()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("ascending", [True, False])
@pytest.mark.parametrize("shape", SHAPES)
def test_host_symbols_equal_oracle(shape, ascending, dtype):
    n, ns, nd, sij, dij = shape
    c = iv.make_case(n, ns, nd, sij, dij, ascending, dtype, seed=n + 7 * ns, outside=0.3)
    want = iv.orc_findpos(c)
    got = V.findpos(n, c["vls"], c["vld"])
    assert np.array_equal(got, want)                         # columns n.. keep the caller's values (-999 in both)
    for name, algo in ALGO.items():
        if name == "cubiclagrange" and ns < 4:
            continue
        for xd, xu in ((0, 0), (1, 1), (1, 0)):
            a, ad = iv.new_out(c); b, bd = iv.new_out(c)
            iv.orc_apply(name, c, want, a, ad, xd, xu)
            V.interp(algo, n, c["vls"], c["ss"], c["sds"], want, c["vld"], b, bd, xd, xu, extended=(xd != xu))
            assert _same(a, b) and _same(ad, bd), (name, xd, xu)
    for name, kind in XKIND.items():
        for xd, xu in ((1, 1), (0, 1), (1, 0), (0, 0)):
            a, ad = iv.new_out(c); b, bd = iv.new_out(c)
            iv.orc_apply(name, c, want, a, ad, xd, xu, -3.75, 0.4375)
            V.extrap(kind, n, c["vls"], c["ss"], c["sds"], want, c["vld"], b, bd, xd, xu, -3.75, 0.4375)
            assert _same(a, b) and _same(ad, bd), (name, xd, xu)


def _dev(c):
    return {k: torch.from_numpy(c[k]).cuda() for k in ("vls", "ss", "sds", "vld")}


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("ascending", [True, False])
@pytest.mark.parametrize("n,ns", TILINGS)
def test_device_entry_points_and_fused_pass_equal_oracle(n, ns, ascending, dtype):
    nd = 37
    c = iv.make_case(n, ns, nd, n + 3, n + 1, ascending, dtype, seed=ns, outside=0.3)
    d = _dev(c)
    want = iv.orc_findpos(c)
    posn = torch.full((nd, c["dij"]), -999, dtype=torch.int32, device="cuda")
    assert V.findpos_dev(n, d["vls"], posn, d["vld"]) == 0
    assert np.array_equal(posn.cpu().numpy(), want)
    for name, algo in ALGO.items():
        a, ad = iv.new_out(c)
        iv.orc_apply(name, c, want, a, ad, 1, 0)
        iv.orc_apply("lapserate", c, want, a, ad, 1, 0, 0.25, -0.5)
        # step by step on the device
        b = torch.full((nd, c["dij"]), 123.25, dtype=d["vls"].dtype, device="cuda"); bd = -b
        assert V.interp_dev(algo, n, d["vls"], d["ss"], d["sds"], posn, d["vld"], b, bd, 1, 0) == 0
        assert V.extrap_dev(V.X_LAPSERATE, n, d["vls"], d["ss"], d["sds"], posn, d["vld"], b, bd, 1, 0, 0.25, -0.5) == 0
        assert _same(a, b.cpu().numpy()) and _same(ad, bd.cpu().numpy()), name
        # search + interpolation + extrapolation in one pass, brackets never stored
        f = torch.full((nd, c["dij"]), 123.25, dtype=d["vls"].dtype, device="cuda"); fd = -f
        assert V.column_dev(algo, V.X_LAPSERATE, n, d["vls"], d["ss"], d["sds"], None, d["vld"], f, fd, 1, 0, 0.25, -0.5) == 0
        assert _same(a, f.cpu().numpy()) and _same(ad, fd.cpu().numpy()), name


@pytest.mark.parametrize("cols", ["64", "32", "16", "0"])
def test_every_tiling_gives_the_same_bits(cols, monkeypatch):
    monkeypatch.setenv("INTERPV_HIP_COLS", cols)
    c = iv.make_case(700, 40, 90, ascending=False, dtype=np.float32, seed=5, outside=0.3)
    d = _dev(c)
    want = iv.orc_findpos(c)
    a, ad = iv.new_out(c)
    iv.orc_apply("cubicwithderivs", c, want, a, ad, 0, 1)
    iv.orc_apply("fixed", c, want, a, ad, 0, 1, 1.0, -9.5)
    f = torch.full((90, 700), 123.25, device="cuda"); fd = -f
    posn = torch.zeros((90, 700), dtype=torch.int32, device="cuda")
    assert V.column_dev(V.CUBIC_DERIVS, V.X_FIXED, 700, d["vls"], d["ss"], d["sds"], posn, d["vld"], f, fd, 0, 1, 1.0, -9.5) == 0
    assert np.array_equal(posn.cpu().numpy(), want)
    assert _same(a, f.cpu().numpy()) and _same(ad, fd.cpu().numpy())


@pytest.mark.parametrize("ascending", [True, False])
def test_reference_test_program_on_the_gpu(ascending):
    """src/interpv/test/Test_Interp1D.F90: its data through the library's Fortran symbols, its pass criteria"""
    c, lsrc, ltgt, sa, da = iv.reference_test_case(ascending)
    posn = V.findpos(c["n"], c["vls"], c["vld"])
    want = [1, 3, 3, 1, 3] if ascending else [3, 1, 1, 3, 1]
    assert posn[:, :2].T.tolist() == [want, want]
    for name in ("cubicwithderivs", "linear", "cubiclagrange"):
        sd, sdd = iv.new_out(c, 0.0)
        V.interp(ALGO[name], c["n"], c["vls"], c["ss"], c["sds"], posn, c["vld"], sd, sdd, False, False, extended=(name == "cubiclagrange"))
        assert iv.reference_test_criteria(lsrc, ltgt, sa, da, sd, sdd), name
    sd, sdd = iv.new_out(c, 0.0)
    V.interp(V.NEAREST, c["n"], c["vls"], c["ss"], c["sds"], posn, c["vld"], sd, sdd)
    for i in range(2):
        assert [float(x) for x in sd[:, i]] == [float(sa[k, i]) for k in (1, 2, 3, 0, 3)]
    sd, sdd = iv.new_out(c, 0.0)
    V.extrap(V.X_LAPSERATE, c["n"], c["vls"], c["ss"], c["sds"], posn, c["vld"], sd, sdd, True, True, 0.4, -0.5)
    assert abs(float(sd[3, 0]) - 0.5411954) <= 1e-7 and abs(float(sd[4, 0]) - 0.1058001) <= 1e-7
    assert abs(float(sd[3, 1]) - 0.6885440) <= 1e-7 and abs(float(sd[4, 1]) - (-0.2382999)) <= 1e-7


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_hand_known_answers_on_the_gpu(dtype):
    lev = np.array([1, 2, 4, 8], dtype).reshape(4, 1)
    c = dict(n=1, ns=4, nd=6, sij=1, dij=1, vls=lev, ss=lev ** 2, sds=2 * lev, vld=np.array([3, 1.5, 6, 0, 10, 2], dtype).reshape(6, 1), dtype=np.dtype(dtype))
    posn = V.findpos(1, c["vls"], c["vld"])
    assert posn[:, 0].tolist() == [2, 1, 3, 1, 3, 2]
    sd, sdd = iv.new_out(c)
    V.interp(V.LINEAR, 1, c["vls"], c["ss"], c["sds"], posn, c["vld"], sd, sdd, True, True)
    assert sd[:, 0].tolist() == [10, 2.5, 40, -2, 88, 4]
    V.interp(V.CUBIC_LAGRANGE, 1, c["vls"], c["ss"], c["sds"], posn, c["vld"], sd, sdd, False, False)
    assert sd[:, 0].tolist() == [9, 2.25, 36, 1, 64, 4]
    V.interp(V.CUBIC_DERIVS, 1, c["vls"], c["ss"], c["sds"], posn, c["vld"], sd, sdd, True, True)
    assert sd[:, 0].tolist() == [9, 2.25, 36, 0, 100, 4] and sdd[:, 0].tolist() == [6, 3, 12, 0, 20, 4]
    V.interp(V.NEAREST, 1, c["vls"], c["ss"], c["sds"], posn, c["vld"], sd, sdd)
    assert sd[:, 0].tolist() == [16, 4, 64, 1, 64, 4]
    sd[:] = 7
    V.extrap(V.X_LAPSERATE, 1, c["vls"], c["ss"], c["sds"], posn, c["vld"], sd, sdd, True, True, 0.5, -0.25)
    assert sd[:, 0].tolist() == [7, 7, 7, 0.5, 63.5, 7]


def test_full_column_count_properties():
    """2 M columns x 60 -> 50 levels (the horizontal grid of a 2000 x 1000 field): linear interpolation of a function that is
    linear in the level is exact to rounding wherever it is not clamped; sampled columns equal the oracle bit for bit"""
    n, ns, nd = 2_000_000, 60, 50
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    inc = torch.rand((ns, n), device="cuda", generator=g) + 0.2
    vls = torch.cumsum(inc, 0)
    ss = 3.0 * vls - 7.0
    vld = vls[0] + torch.rand((nd, n), device="cuda", generator=g) * (vls[-1] - vls[0]) * 1.1 - 0.05 * (vls[-1] - vls[0])
    sd = torch.empty((nd, n), device="cuda"); sdd = torch.empty((nd, n), device="cuda")
    assert V.column_dev(V.LINEAR, V.X_NONE, n, vls, ss, ss, None, vld, sd, sdd, True, True) == 0
    torch.cuda.synchronize()
    err = (sd - (3.0 * vld - 7.0)).abs().max().item()
    assert err <= 2e-4, err
    cols = np.r_[0:64, 999_936:1_000_064, n - 64:n]
    c = dict(n=len(cols), ns=ns, nd=nd, sij=len(cols), dij=len(cols), vls=vls[:, cols].cpu().numpy().copy(), ss=ss[:, cols].cpu().numpy().copy(),
             sds=ss[:, cols].cpu().numpy().copy(), vld=vld[:, cols].cpu().numpy().copy(), dtype=np.dtype(np.float32))
    if not (c["vls"][1, 0] > c["vls"][0, 0]):
        pytest.skip("direction column differs")
    a, ad = iv.new_out(c)
    iv.orc_apply("linear", c, iv.orc_findpos(c), a, ad, 1, 1)
    assert _same(a, sd[:, cols].cpu().numpy().copy())


def test_abort_returns_2_and_the_fortran_symbol_exits_2():
    c, *_ = iv.reference_test_case(True)
    d = _dev(c)
    posn = torch.from_numpy(iv.orc_findpos(c)).cuda()
    sd = torch.zeros((5, 6), device="cuda")
    assert V.extrap_dev(V.X_ABORT, 2, d["vls"], d["ss"], d["sds"], posn, d["vld"], sd, sd, 1, 1) == 2      # 0.5 and 3.1 lie outside
    assert V.extrap_dev(V.X_ABORT, 2, d["vls"], d["ss"], d["sds"], posn, d["vld"], sd, sd, 0, 0) == 0
    c["vld"][3, :2] = 0.65; c["vld"][4, :2] = 2.9                                                             # Test_Interp1D.F90:394-409
    d = _dev(c); posn = torch.from_numpy(iv.orc_findpos(c)).cuda()
    assert V.extrap_dev(V.X_ABORT, 2, d["vls"], d["ss"], d["sds"], posn, d["vld"], sd, sd, 1, 1) == 0
    code = ("import sys; sys.path.insert(0, 'tests'); import numpy as np, interpvcases as iv; from librmn_amd import interpv as V;"
            "c, *_ = iv.reference_test_case(True); p = V.findpos(2, c['vls'], c['vld']); sd, sdd = iv.new_out(c);"
            "V.extrap(V.X_ABORT, 2, c['vls'], c['ss'], c['sds'], p, c['vld'], sd, sdd, True, True); print('survived')")
    r = run_child([sys.executable, "-c", code], cwd=os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    assert r.returncode == 2 and "survived" not in r.stdout and "Extrap1D_Abort: Attempting extrapolation to level" in r.stderr
    assert "below the lowest level" in r.stderr                                                              # vt = 4 (0.5) comes before vt = 5 (3.1)


def test_too_few_levels_is_refused_like_the_reference():
    c = iv.make_case(3, 3, 4, seed=1)
    posn = iv.orc_findpos(c)
    sd, sdd = iv.new_out(c)
    V.interp(V.CUBIC_LAGRANGE, 3, c["vls"], c["ss"], c["sds"], posn, c["vld"], sd, sdd)
    assert np.all(sd == 123.25)                         # an error line, nothing computed (Interp1D_CubicLagrange_Body.inc:88-91)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("ascending", [True, False])
def test_search_corner_cases(ascending, dtype):
    """destination levels exactly on source levels, beyond both ends, infinite and NaN; columns that are not strictly monotonic (equal
    neighbours, a reversed pair, a NaN level, the other direction than column 1): the literal search of the reference decides, so
    every bracket equals the oracle's"""
    n, ns, nd = 96, 37, 50
    c = iv.make_case(n, ns, nd, n, n, ascending, dtype, seed=17, outside=0.3, ties=0.4)
    c["vls"][5, 10] = c["vls"][4, 10]                       # equal neighbours
    c["vls"][[7, 8], 20] = c["vls"][[8, 7], 20]            # a reversed pair
    c["vls"][3, 30] = np.nan
    c["vls"][:, 40] = c["vls"][::-1, 40].copy()            # the other direction
    c["vld"][0, :] = np.inf; c["vld"][1, :] = -np.inf; c["vld"][2, ::3] = np.nan
    c["vld"][3, :] = c["vls"][0, :]; c["vld"][4, :] = c["vls"][-1, :]
    with np.errstate(invalid="ignore"):
        want = iv.orc_findpos(c)
    d = _dev(c)
    posn = torch.full((nd, n), -999, dtype=torch.int32, device="cuda")
    assert V.findpos_dev(n, d["vls"], posn, d["vld"]) == 0
    got = posn.cpu().numpy()
    bad = np.argwhere(got != want)
    assert bad.size == 0, (bad[:5].tolist(), got[tuple(bad[0])], want[tuple(bad[0])])
