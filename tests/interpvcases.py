"""Shared inputs and loaders for the vertical-interpolation (src/interpv) tests.  TEST INFRASTRUCTURE.

Arrays follow the reference: Fortran (ijDim, levels), held here as C-ordered numpy (levels, ijDim).
posn holds 1-based level indices (Interp1D_FindPos_Body.inc:142).
"""
import ctypes, os
import numpy as np
import oraclelib as ol

_HERE = os.path.dirname(os.path.abspath(__file__))
REFV_SO = os.path.join(_HERE, "..", "oracle", "_ref", "libinterpvref.so")
ALGOS = ("nearestneighbour", "linear", "cubiclagrange", "cubicwithderivs")
EXTRAPS = ("fixed", "lapserate")
_refv = None


def have_refv():
    return os.path.exists(REFV_SO)


def refv():
    global _refv
    if _refv is None:
        _refv = ctypes.CDLL(os.path.abspath(REFV_SO))
    return _refv


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def ctype_of(dt):
    return ctypes.c_float if np.dtype(dt) == np.float32 else ctypes.c_double


def make_case(n, ns, nd, sij=None, dij=None, ascending=True, dtype=np.float32, seed=0, same_levels=False, outside=0.25, ties=0.05):
    """Monotone source levels per column (direction shared by all columns, as FindPos assumes), smooth state and
    derivative, destination levels in no particular order: a share `outside` beyond either end, a share `ties`
    exactly on a source level.  Dimensioned columns beyond n hold a poison value the routines must not touch."""
    sij = sij or n; dij = dij or n
    r = np.random.default_rng(seed)
    inc = r.uniform(0.2, 1.5, (ns, n))
    if same_levels:
        inc[:] = inc[:, :1]
    lev = np.cumsum(inc, axis=0) + r.uniform(-2, 2, (1, n))
    if not ascending:
        lev = lev[::-1].copy()
    vls = np.full((ns, sij), 1e30, dtype); vls[:, :n] = lev
    ss = np.full((ns, sij), -7e29, dtype); ss[:, :n] = np.sin(0.37 * lev) * 20 + 0.5 * lev + r.normal(0, 0.3, (ns, n))
    sds = np.full((ns, sij), 3e29, dtype); sds[:, :n] = 0.37 * 20 * np.cos(0.37 * lev) + 0.5
    lo = lev.min(axis=0); hi = lev.max(axis=0)
    u = r.uniform(0, 1, (nd, n))
    dest = lo + u * (hi - lo)
    k = r.uniform(0, 1, (nd, n))
    dest = np.where(k < outside / 2, lo - r.uniform(0.01, 3, (nd, n)), dest)
    dest = np.where((k >= outside / 2) & (k < outside), hi + r.uniform(0.01, 3, (nd, n)), dest)
    tie = r.uniform(0, 1, (nd, n)) < ties
    pick = r.integers(0, ns, (nd, n))
    dest = np.where(tie, np.take_along_axis(lev, pick, axis=0), dest)
    vld = np.full((nd, dij), -1e30, dtype); vld[:, :n] = dest
    if dtype == np.float32:        # ties must be exact in the working precision
        vld[:, :n] = np.where(tie, np.take_along_axis(vls[:, :n], pick, axis=0), vld[:, :n])
    return dict(n=n, ns=ns, nd=nd, sij=sij, dij=dij, vls=vls, ss=ss, sds=sds, vld=vld, dtype=np.dtype(dtype))


def new_out(c, fill=123.25):
    return np.full((c["nd"], c["dij"]), fill, c["dtype"]), np.full((c["nd"], c["dij"]), -fill, c["dtype"])


# ---------------------------------------------------------------- oracle
def orc_findpos(c):
    O = ol.oracle()
    posn = np.full((c["nd"], c["dij"]), -999, np.int32)
    f = getattr(O, "orc_interp1d_findpos" + ("8" if c["dtype"] == np.float64 else ""))
    f.restype = None
    f(c["n"], c["ns"], c["nd"], c["sij"], c["dij"], _p(c["vls"]), _p(posn), _p(c["vld"]))
    return posn


def orc_apply(name, c, posn, sd, sdd, xdown=False, xup=False, gdown=0.0, gup=0.0, where=None):
    """name: one of ALGOS, 'fixed', 'lapserate', 'abort'; works in place on sd / sdd; returns the oracle's code"""
    O = ol.oracle()
    pre = "orc_extrap1d_" if name in ("fixed", "lapserate", "abort") else "orc_interp1d_"
    f = getattr(O, pre + name + ("8" if c["dtype"] == np.float64 else ""))
    R = ctype_of(c["dtype"])
    f.restype = ctypes.c_int
    args = [ctypes.c_int(c["n"]), ctypes.c_int(c["ns"]), ctypes.c_int(c["nd"]), ctypes.c_int(c["sij"]), ctypes.c_int(c["dij"]),
            _p(c["vls"]), _p(c["ss"]), _p(c["sds"]), _p(posn), _p(c["vld"]), _p(sd), _p(sdd),
            ctypes.c_int(int(xdown)), ctypes.c_int(int(xup)), R(gdown), R(gup)]
    if name == "abort":
        w = where if where is not None else np.zeros(3, np.int32)
        args.append(_p(w))
    return f(*args)


# ---------------------------------------------------------------- reference build (Fortran ABI: everything by reference)
def _ibr(v):
    return ctypes.byref(ctypes.c_int32(v))


def ref_findpos(c):
    L = refv()
    posn = np.full((c["nd"], c["dij"]), -999, np.int32)
    f = getattr(L, "interp1d_findpos8_" if c["dtype"] == np.float64 else "interp1d_findpos_")
    f.restype = None
    f(_ibr(c["n"]), _ibr(c["ns"]), _ibr(c["nd"]), _ibr(c["sij"]), _ibr(c["dij"]), _p(c["vls"]), _p(posn), _p(c["vld"]))
    return posn


REF_NAMES = {"nearestneighbour": "interp1d_nearestneighbour", "fixed": "extrap1d_fixed", "lapserate": "extrap1d_lapserate"}


def fortran_call(L, sym, c, posn, sd, sdd, xdown, xup, gdown, gup):
    """<name>_ / <name>8_ of a library with the reference's Fortran interface (Interp1D_Arguments.inc)"""
    f = getattr(L, sym + ("8_" if c["dtype"] == np.float64 else "_"))
    f.restype = None
    R = ctype_of(c["dtype"])
    f(_ibr(c["n"]), _ibr(c["ns"]), _ibr(c["nd"]), _ibr(c["sij"]), _ibr(c["dij"]), _p(c["vls"]), _p(c["ss"]), _p(c["sds"]),
      _p(posn), _p(c["vld"]), _p(sd), _p(sdd), _ibr(1 if xdown else 0), _ibr(1 if xup else 0),
      ctypes.byref(R(gdown)), ctypes.byref(R(gup)))


def ref_apply(name, c, posn, sd, sdd, xdown=False, xup=False, gdown=0.0, gup=0.0):
    fortran_call(refv(), REF_NAMES[name], c, posn, sd, sdd, xdown, xup, gdown, gup)


# ---------------------------------------------------------------- the reference's own test data (Test_Interp1D.F90:61-139)
def reference_test_case(ascending, dtype=np.float32):
    hdim, nsrc, ntgt, hnum = 6, 4, 5, 2
    levels_src = np.array([0.64, 1.25, 2.44, 2.97], np.float32)
    levels_tgt = np.array([1.13, 2.62, 2.79, 0.5, 3.1], np.float32)
    sd_ = np.zeros((nsrc, hdim), np.float32); dd_ = np.zeros((nsrc, hdim), np.float32)       # "descending" tables (:92-113)
    sd_[:, 0] = [0.1708, 0.6454, 0.9489846, 0.5971954]
    dd_[:, 0] = [-0.9853, -0.7638, 0.3153224, 0.8020958]
    sd_[:, 1] = [-0.1733, -0.8450, 3.009570, 0.744544]
    dd_[:, 1] = [1.03003, 1.7140, 10.057510, 1.554346]
    sa_ = sd_[::-1].copy(); da_ = dd_[::-1].copy()                                            # (:117-122)
    vls = np.zeros((nsrc, hdim), np.float32)
    vls[:, :hnum] = (levels_src if ascending else levels_src[::-1])[:, None]
    vld = np.zeros((ntgt, hdim), np.float32); vld[:, :hnum] = levels_tgt[:, None]
    c = dict(n=hnum, ns=nsrc, nd=ntgt, sij=hdim, dij=hdim, vls=vls.astype(dtype), ss=(sa_ if ascending else sd_).astype(dtype),
             sds=(da_ if ascending else dd_).astype(dtype), vld=vld.astype(dtype), dtype=np.dtype(dtype))
    return c, levels_src, levels_tgt, sa_, da_


def reference_test_criteria(levels_src, levels_tgt, sa, da, state, deriv, hnum=2):
    """TestReportClear (Test_Interp1D.F90:652-707): state / deriv are (ntgt, hdim); sa / da the ascending tables"""
    ok = True
    for v, lv in enumerate(levels_tgt):
        if lv < levels_src[0]:
            for i in range(hnum):
                ok &= abs(float(state[v, i]) - float(sa[0, i])) <= 1e-15
                ok &= not (abs(float(deriv[v, i]) - float(da[0, i])) > 1e-15 and abs(float(deriv[v, i])) > 1e-15)
        elif lv > levels_src[-1]:
            for i in range(hnum):
                ok &= abs(float(state[v, i]) - float(sa[-1, i])) <= 1e-15
                ok &= not (float(deriv[v, i]) != float(da[-1, i]) and abs(float(deriv[v, i])) > 1e-15)
        else:
            ok &= abs(float(state[v, 0]) - np.sin(lv)) <= 0.06
            ok &= abs(float(state[v, 1]) - np.tan(lv)) <= 0.5 / np.cos(lv) ** 2
    if abs(float(deriv[0, 0])) > 1e-15:
        for v, lv in enumerate(levels_tgt):
            ok &= abs(float(deriv[v, 0]) - np.cos(lv)) <= 0.1
            ok &= abs(float(deriv[v, 1]) - 1.0 / np.cos(lv) ** 2) <= 0.7 * abs(np.tan(lv))
    return bool(ok)
