"""CPU tests of the vertical-interpolation oracle (oracle/orc_interpv.c).

Pins, in decreasing strength:
  * FindPos, NearestNeighbour, Extrap1D_Fixed, Extrap1D_LapseRate: bit for bit against the reference's own files
    compiled where they lie (oracle/_ref/libinterpvref.so, oracle/build_ref.sh);
  * Linear, CubicLagrange, CubicWithDerivs (`use app`: unbuildable here): the reference's own test program,
    src/interpv/test/Test_Interp1D.F90 -- its data, its pass criteria and its literal lapse-rate answers --
    plus known answers worked out by hand from the reference text and exactness on polynomials.
"""
import numpy as np
import pytest
import interpvcases as iv

needs_refv = pytest.mark.skipif(not iv.have_refv(), reason="oracle/_ref/libinterpvref.so not built")
SHAPES = [(7, 2, 5, 7, 7), (64, 5, 9, 64, 70), (33, 28, 17, 40, 33), (129, 80, 60, 129, 129), (1, 4, 3, 2, 3)]


@needs_refv
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("ascending", [True, False])
@pytest.mark.parametrize("shape", SHAPES)
def test_findpos_equals_reference_build(shape, ascending, dtype):
    n, ns, nd, sij, dij = shape
    c = iv.make_case(n, ns, nd, sij, dij, ascending, dtype, seed=n + ns)
    po, pr = iv.orc_findpos(c), iv.ref_findpos(c)
    assert np.array_equal(po, pr)
    assert po[:, :n].min() >= 1 and po[:, :n].max() <= ns - 1
    assert np.all(po[:, n:] == -999)                      # dimensioned-only columns are never written


@needs_refv
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("ascending", [True, False])
@pytest.mark.parametrize("name,flags", [("nearestneighbour", (0, 0)), ("fixed", (1, 1)), ("fixed", (1, 0)), ("fixed", (0, 1)),
                                        ("lapserate", (1, 1)), ("lapserate", (0, 1)), ("lapserate", (1, 0)), ("lapserate", (0, 0))])
@pytest.mark.parametrize("shape", SHAPES)
def test_buildable_routines_equal_reference_build(shape, name, flags, ascending, dtype):
    n, ns, nd, sij, dij = shape
    c = iv.make_case(n, ns, nd, sij, dij, ascending, dtype, seed=3 * n + ns, outside=0.4)
    posn = iv.ref_findpos(c)
    a, ad = iv.new_out(c); b, bd = iv.new_out(c)
    iv.orc_apply(name, c, posn, a, ad, flags[0], flags[1], -3.75, 0.4375)
    iv.ref_apply(name, c, posn, b, bd, flags[0], flags[1], -3.75, 0.4375)
    assert a.tobytes() == b.tobytes() and ad.tobytes() == bd.tobytes()


@pytest.mark.parametrize("ascending", [True, False])
def test_reference_test_program_criteria(ascending):
    """Test_Interp1D.F90:141-166 (descending) and :209-322 (ascending): same calls, same pass criteria"""
    c, lsrc, ltgt, sa, da = iv.reference_test_case(ascending)
    posn = iv.orc_findpos(c)
    want = [1, 3, 3, 1, 3] if ascending else [3, 1, 1, 3, 1]          # brackets of 1.13 2.62 2.79 0.5 3.1 in 0.64 1.25 2.44 2.97
    assert posn[:, :2].T.tolist() == [want, want]
    for name in ("cubicwithderivs", "linear", "cubiclagrange"):
        sd, sdd = iv.new_out(c, 0.0)
        assert iv.orc_apply(name, c, posn, sd, sdd, False, False, 0.4, -0.5) == 0
        assert iv.reference_test_criteria(lsrc, ltgt, sa, da, sd, sdd), name
    sd, sdd = iv.new_out(c, 0.0)
    iv.orc_apply("nearestneighbour", c, posn, sd, sdd)
    idx = [1, 2, 3, 0, 3]                                                # Test_Interp1D.F90:262-270 (ascending indices 2 3 4 1 4)
    for i in range(2):
        assert [float(x) for x in sd[:, i]] == [float(sa[k, i]) for k in idx]
    # Extrap1D_LapseRate literal answers (:186-194, :329-337)
    sd, sdd = iv.new_out(c, 0.0)
    iv.orc_apply("lapserate", c, posn, sd, sdd, True, True, 0.4, -0.5)
    assert abs(float(sd[3, 0]) - 0.5411954) <= 1e-7 and abs(float(sd[4, 0]) - 0.1058001) <= 1e-7
    assert abs(float(sd[3, 1]) - 0.6885440) <= 1e-7 and abs(float(sd[4, 1]) - (-0.2382999)) <= 1e-7
    assert np.all(sd[:3] == 0)                                          # untouched inside the source range


def test_abort_reports_first_offender_in_reference_loop_order():
    """Extrap1D_Abort_Body.inc:70-92: vt outer, i inner; the reference test's non-aborting data (:394-409)"""
    c, *_ = iv.reference_test_case(True)
    c["vld"][3, :2] = 0.65; c["vld"][4, :2] = 2.9
    posn = iv.orc_findpos(c)
    sd, sdd = iv.new_out(c)
    assert iv.orc_apply("abort", c, posn, sd, sdd, True, True) == 0
    c["vld"][4, 1] = 3.5; c["vld"][2, 0] = 0.1
    posn = iv.orc_findpos(c)
    w = np.zeros(3, np.int32)
    assert iv.orc_apply("abort", c, posn, sd, sdd, True, True, where=w) == 2 and w.tolist() == [1, 3, 0]
    assert iv.orc_apply("abort", c, posn, sd, sdd, False, True, where=w) == 2 and w.tolist() == [2, 5, 1]
    assert iv.orc_apply("abort", c, posn, sd, sdd, False, False, where=w) == 0


# ------------------------------------------------------------------ known answers worked out by hand
def _one_column(levels, state, deriv, targets, dtype=np.float32):
    ns, nd = len(levels), len(targets)
    return dict(n=1, ns=ns, nd=nd, sij=1, dij=1, vls=np.array(levels, dtype).reshape(ns, 1), ss=np.array(state, dtype).reshape(ns, 1),
                sds=np.array(deriv, dtype).reshape(ns, 1), vld=np.array(targets, dtype).reshape(nd, 1), dtype=np.dtype(dtype))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_hand_known_answers(dtype):
    # levels 1 2 4 8, state = levels**2 (1 4 16 64), derivative 2*level
    c = _one_column([1, 2, 4, 8], [1, 4, 16, 64], [2, 4, 8, 16], [3, 1.5, 6, 0, 10, 2], dtype)
    posn = iv.orc_findpos(c)
    # FindPos by hand: ns = 4 -> index 2.5, uncertainty 1.5 -> 0.75 (one pass).  3: >= lev(2)=2 -> 3.25, <= lev(3)=4 -> 2.5 -> 2.
    # 1.5: < 2, <= 4 -> 1.75 -> 1.   6: >= 2 -> 3.25, not <= 4 -> 3.   0 -> 1.   10 -> 3.   2: >= 2 -> 3.25, <= 4 -> 2.5 -> 2
    assert posn[:, 0].tolist() == [2, 1, 3, 1, 3, 2]
    sd, sdd = iv.new_out(c)
    iv.orc_apply("linear", c, posn, sd, sdd, False, False)
    # slopes 3, 6, 12: 4 + 6*(3-2) = 10;  1 + 3*0.5 = 2.5;  16 + 12*2 = 40;  clamped 1 and 64;  4 + 6*0 = 4
    assert sd[:, 0].tolist() == [10, 2.5, 40, 1, 64, 4]
    iv.orc_apply("linear", c, posn, sd, sdd, True, True)
    assert sd[:, 0].tolist() == [10, 2.5, 40, -2, 88, 4]          # 1 + 3*(0-1);  16 + 12*(10-4)
    # cubic Lagrange through a quadratic is that quadratic (every quantity a small dyadic rational: exact)
    iv.orc_apply("cubiclagrange", c, posn, sd, sdd, True, True)
    assert sd[:, 0].tolist() == [9, 2.25, 36, 0, 100, 4]
    iv.orc_apply("cubiclagrange", c, posn, sd, sdd, False, False)
    assert sd[:, 0].tolist() == [9, 2.25, 36, 1, 64, 4]
    # cubic with derivatives between (2, 4, 4) and (4, 16, 8) at 3: centre 3, deltaLIn2 2, target offset 0:
    # dd = 0.125*4 = 0.5; sc0 10, sc1 6, sc2 2, sc3 ((8-6)-(6-4))/4 = 0; state = 10 - 0.5*2 = 9, derivative = 6
    iv.orc_apply("cubicwithderivs", c, posn, sd, sdd, True, True)
    assert sd[:, 0].tolist() == [9, 2.25, 36, 0, 100, 4] and sdd[:, 0].tolist() == [6, 3, 12, 0, 20, 4]
    iv.orc_apply("cubicwithderivs", c, posn, sd, sdd, False, False)
    assert sd[:, 0].tolist() == [9, 2.25, 36, 1, 64, 4] and sdd[:, 0].tolist() == [6, 3, 12, 2, 16, 4]
    # nearest neighbour: 3, 1.5 and 6 are equidistant from their brackets -> not strictly closer to the one below -> the one above
    iv.orc_apply("nearestneighbour", c, posn, sd, sdd)
    assert sd[:, 0].tolist() == [16, 4, 64, 1, 64, 4]
    # extrapolators only touch the end brackets and only strictly outside
    sd[:] = 7
    iv.orc_apply("fixed", c, posn, sd, sdd, True, True, -1.5, 2.5)
    assert sd[:, 0].tolist() == [7, 7, 7, -1.5, 2.5, 7]
    sd[:] = 7
    iv.orc_apply("lapserate", c, posn, sd, sdd, True, True, 0.5, -0.25)
    assert sd[:, 0].tolist() == [7, 7, 7, 0.5, 63.5, 7]          # 1 + 0.5*(0-1);  64 - 0.25*(10-8)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cubics_are_reproduced(dtype):
    """Lagrange on 4 points and the two-point cubic with derivatives are exact on cubics (to rounding)"""
    c = iv.make_case(40, 12, 30, ascending=True, dtype=dtype, seed=9, outside=0.0, ties=0.0)
    f = lambda x: 0.03 * x ** 3 - 0.4 * x ** 2 + 1.7 * x - 2
    g = lambda x: 0.09 * x ** 2 - 0.8 * x + 1.7
    lev = c["vls"].astype(np.float64)
    c["ss"] = f(lev).astype(dtype); c["sds"] = g(lev).astype(dtype)
    posn = iv.orc_findpos(c)
    tol = 2e-4 if dtype == np.float32 else 1e-11
    for name in ("cubiclagrange", "cubicwithderivs"):
        sd, sdd = iv.new_out(c)
        iv.orc_apply(name, c, posn, sd, sdd)
        x = c["vld"].astype(np.float64)
        assert np.max(np.abs(sd - f(x))) <= tol * np.max(np.abs(f(x))), name
        if name == "cubicwithderivs":
            assert np.max(np.abs(sdd - g(x))) <= 10 * tol * np.max(np.abs(g(x)))


def test_too_few_levels_is_an_error():
    c = iv.make_case(3, 3, 4, seed=1)
    posn = iv.orc_findpos(c)
    sd, sdd = iv.new_out(c)
    assert iv.orc_apply("cubiclagrange", c, posn, sd, sdd) == -1      # Interp1D_CubicLagrange_Body.inc:88-91
    assert np.all(sd == 123.25)
    assert iv.orc_apply("linear", c, posn, sd, sdd) == 0
