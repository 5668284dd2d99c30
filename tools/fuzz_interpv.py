"""Random shapes / pitches / options of the vertical interpolation through the device entry points against the oracle, bit for bit.
python tools/fuzz_interpv.py [ncases] [seed]"""
import os, sys
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
import interpvcases as iv
from librmn_amd import interpv as V

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ALGO = {"nearestneighbour": V.NEAREST, "linear": V.LINEAR, "cubiclagrange": V.CUBIC_LAGRANGE, "cubicwithderivs": V.CUBIC_DERIVS}
bad = 0
for k in range(ncases):
    n = int(rng.choice([1, 2, 31, 63, 64, 65, 200, 1000, 5000, 40000]))
    ns = int(rng.choice([2, 3, 4, 5, 17, 60, 80, 130, 161, 330, 700, 1500]))
    nd = int(rng.choice([1, 2, 3, 4, 5, 16, 33, 60, 257]))
    if n * (ns + nd) > 6_000_000:
        n = max(1, 6_000_000 // (ns + nd))
    sij = n + int(rng.integers(0, 4)); dij = n + int(rng.integers(0, 4))
    dtype = np.float32 if rng.random() < 0.5 else np.float64
    asc = bool(rng.random() < 0.5)
    c = iv.make_case(n, ns, nd, sij, dij, asc, dtype, seed=int(rng.integers(1 << 30)), outside=float(rng.choice([0.0, 0.2, 0.6])), ties=float(rng.choice([0.0, 0.05, 0.5])))
    names = [a for a in ALGO if not (a == "cubiclagrange" and ns < 4)]
    name = names[int(rng.integers(len(names)))]
    xd, xu = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
    kind = int(rng.choice([V.X_NONE, V.X_FIXED, V.X_LAPSERATE]))
    gd, gu = float(rng.normal()), float(rng.normal())
    want_p = iv.orc_findpos(c)
    a, ad = iv.new_out(c)
    iv.orc_apply(name, c, want_p, a, ad, xd, xu)
    if kind != V.X_NONE:
        iv.orc_apply("fixed" if kind == V.X_FIXED else "lapserate", c, want_p, a, ad, xd, xu, gd, gu)
    d = {q: torch.from_numpy(c[q]).cuda() for q in ("vls", "ss", "sds", "vld")}
    f = torch.full((nd, dij), 123.25, dtype=d["vls"].dtype, device="cuda"); fd = -f
    posn = torch.full((nd, dij), -999, dtype=torch.int32, device="cuda")
    keep_posn = rng.random() < 0.5
    rc = V.column_dev(ALGO[name], kind, n, d["vls"], d["ss"], d["sds"], posn if keep_posn else None, d["vld"], f, fd, xd, xu, gd, gu)
    ok = rc == 0 and a.tobytes() == f.cpu().numpy().tobytes() and ad.tobytes() == fd.cpu().numpy().tobytes()
    if keep_posn:
        ok = ok and np.array_equal(posn.cpu().numpy(), want_p)
    if not ok:
        bad += 1
        print("MISMATCH", dict(n=n, ns=ns, nd=nd, sij=sij, dij=dij, dtype=str(np.dtype(dtype)), asc=asc, algo=name, xd=xd, xu=xu, kind=kind, rc=rc), flush=True)
print(f"{ncases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
