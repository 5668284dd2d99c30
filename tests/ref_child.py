"""TEST / BASELINE INFRASTRUCTURE: the reference build (oracle/_ref/libezref.so) in a process of its own.

Why a process of its own: (1) the reference's Fortran leaf routines keep ~24 bytes per target point in automatic arrays --
cfg3's 8 M points need ~200 MB of stack, so the calls run on a thread whose stack this file sizes itself (no `ulimit -s` needed); (2) a caller that
already holds GPU mappings must not fork -- the GPU suite asks its fork-server (tests/conftest.py, created before the first GPU call) to start this file,
bench.py starts it before it imports torch.  It never touches a GPU and reads nothing but the repository copy (oracle/_ref travels with the snapshot).

    python tests/ref_child.py cfg3_uvint [--reps N] [--out FILE.npy] [--degree 0|1|3] [--polar 0|1]
        the reference's c_ezuvint (src/interp/ezuvint.c:51-94) on BASELINE configs[2]: Z-on-E 2560x1280 -> L 4000x2000, bicubic, polar correction,
        inputs = tests/golden/make_cfg3_full.py's.  Prints one JSON line {first_s, s_per_pair, reps, rc}; --out: the two result fields as a
        float32 [2, 8 M] array (u, v).
    python tests/ref_child.py cfg2_sint | cfg3_sint [--out FILE.npy]
        the reference's c_ezsint, bicubic with polar correction, on BASELINE configs[1] / on cfg3's grid pair: the whole result field
    python tests/ref_child.py uvint_case CASE.npz OUT.npz
        c_ezuvint on a case file: src_* / dst_* = (ni, nj, grtyp, grref, ig[4], ax, ay), degree, polar, uu, vv  ->  ur, vr, rc
"""
import ctypes, json, os, sys, threading, time

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)
import numpy as np          # noqa: E402
import reflib               # noqa: E402
import ezcases as ec        # noqa: E402

fp = reflib.fptr
DEG = {0: b"nearest", 1: b"linear", 3: b"cubic"}


def cfg3_inputs():
    ni, nj = 2560, 1280
    uu, vv = ec.synth_wind(ni, nj, seed=3)
    for a in (uu, vv):
        a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
    return uu, vv


def cfg3_uvint(argv):
    reps = int(argv[argv.index("--reps") + 1]) if "--reps" in argv else 3
    out = argv[argv.index("--out") + 1] if "--out" in argv else None
    L = reflib.ref()
    ni, nj, no, mo = 2560, 1280, 4000, 2000
    ax, ay = ec.ze_axes(ni, nj)
    gdin = L.c_ezgdef_fmem(ni, nj, b"Z", b"E", *ec.E_IG, fp(ax), fp(ay))
    gdout = L.c_ezqkdef(no, mo, b"L", 9, 9, 0, 0, 0)
    assert gdin >= 0 and gdout >= 0 and L.c_ezdefset(gdout, gdin) == 1
    degree = int(argv[argv.index("--degree") + 1]) if "--degree" in argv else 3
    polar = int(argv[argv.index("--polar") + 1]) if "--polar" in argv else 1
    L.c_ezsetopt(b"interp_degree", DEG[degree]); L.c_ezsetopt(b"polar_correction", b"yes" if polar else b"no")
    uu, vv = cfg3_inputs()
    uv = np.zeros((2, no * mo), np.float32)
    t0 = time.perf_counter()
    rc = L.c_ezuvint(fp(uv[0]), fp(uv[1]), fp(uu), fp(vv))                # first call of the set: lat / lon, locate through the rotation, zones
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(reps):
        rc = min(rc, L.c_ezuvint(fp(uv[0]), fp(uv[1]), fp(uu), fp(vv)))
    dt = (time.perf_counter() - t0) / max(reps, 1)
    if out:
        np.save(out, uv)
    print(json.dumps({"first_s": first, "s_per_pair": dt, "reps": reps, "rc": int(rc), "points": no * mo}), flush=True)


def _define(L, ni, nj, grtyp, grref, ig, ax, ay):
    ig = [int(v) for v in ig]
    if ax is None or np.size(ax) == 0:
        return L.c_ezqkdef(int(ni), int(nj), str(grtyp).encode(), *ig, 0)
    ax = np.ascontiguousarray(ax, np.float32); ay = np.ascontiguousarray(ay, np.float32)
    _define.keep.append((ax, ay))
    return L.c_ezgdef_fmem(int(ni), int(nj), str(grtyp).encode(), str(grref).encode(), *ig, fp(ax), fp(ay))
_define.keep = []


def uvint_case(argv):
    d = np.load(argv[0], allow_pickle=False)
    L = reflib.ref()
    gi = _define(L, d["src_ni"], d["src_nj"], d["src_grtyp"], d["src_grref"], d["src_ig"], d["src_ax"], d["src_ay"])
    go = _define(L, d["dst_ni"], d["dst_nj"], d["dst_grtyp"], d["dst_grref"], d["dst_ig"], d["dst_ax"], d["dst_ay"])
    assert gi >= 0 and go >= 0 and L.c_ezdefset(go, gi) == 1
    L.c_ezsetopt(b"interp_degree", DEG[int(d["degree"])]); L.c_ezsetopt(b"polar_correction", b"yes" if int(d["polar"]) else b"no")
    n = int(d["dst_ni"]) * int(d["dst_nj"])
    uu = np.ascontiguousarray(d["uu"], np.float32); vv = np.ascontiguousarray(d["vv"], np.float32)
    ur = np.zeros(n, np.float32); vr = np.zeros(n, np.float32)
    rc = L.c_ezuvint(fp(ur), fp(vr), fp(uu), fp(vv))
    np.savez(argv[1], ur=ur, vr=vr, rc=np.int32(rc))
    print(json.dumps({"rc": int(rc), "points": n}), flush=True)


def cfg_sint(argv, which):
    """c_ezsint (src/interp/ezsint.c) bicubic with polar correction on BASELINE configs[1] (G 4400x2200 -> L 7200x3601, tests/ezcases.synth_field seed 2) or on cfg3's grid
    pair (the u field of cfg3_inputs): --out FILE.npy = the whole result field"""
    out = argv[argv.index("--out") + 1] if "--out" in argv else None
    L = reflib.ref()
    if which == 2:
        ni, nj, no, mo = 4400, 2200, 7200, 3601
        gdin = L.c_ezqkdef(ni, nj, b"G", 0, 0, 0, 0, 0); gdout = L.c_ezqkdef(no, mo, b"L", 5, 5, 0, 0, 0)
        zin = ec.synth_field(ni, nj, seed=2)
    else:
        ni, nj, no, mo = 2560, 1280, 4000, 2000
        ax, ay = ec.ze_axes(ni, nj)
        gdin = L.c_ezgdef_fmem(ni, nj, b"Z", b"E", *ec.E_IG, fp(ax), fp(ay)); gdout = L.c_ezqkdef(no, mo, b"L", 9, 9, 0, 0, 0)
        zin = cfg3_inputs()[0]
    assert gdin >= 0 and gdout >= 0 and L.c_ezdefset(gdout, gdin) == 1
    L.c_ezsetopt(b"interp_degree", b"cubic"); L.c_ezsetopt(b"polar_correction", b"yes")
    z = np.zeros(no * mo, np.float32)
    t0 = time.perf_counter(); rc = L.c_ezsint(fp(z), fp(zin)); first = time.perf_counter() - t0
    if out:
        np.save(out, z)
    print(json.dumps({"first_s": first, "rc": int(rc), "points": no * mo}), flush=True)


MODES = {"cfg3_uvint": cfg3_uvint, "uvint_case": uvint_case, "cfg2_sint": lambda a: cfg_sint(a, 2), "cfg3_sint": lambda a: cfg_sint(a, 3)}


def main():
    if len(sys.argv) < 2 or sys.argv[1] not in MODES:
        sys.stderr.write(__doc__); return 2
    if not reflib.have_ref():
        sys.stderr.write("ref_child: oracle/_ref/libezref.so is not built\n"); return 3
    res = {}

    def run():
        try:
            MODES[sys.argv[1]](sys.argv[2:]); res["rc"] = 0
        except BaseException as e:      # noqa: BLE001
            sys.stderr.write("ref_child: %r\n" % (e,)); res["rc"] = 1
    # the Fortran automatic arrays: 24 B per target point and more, on the calling thread's stack (cfg2's 26 M points: ~0.7 GB).  The reservation is virtual; where a
    # 2 GB thread stack is refused, smaller ones are tried
    for size in (2 << 30, 1 << 30, 768 << 20, 512 << 20):
        try:
            threading.stack_size(size)
            t = threading.Thread(target=run); t.start()
        except (ValueError, RuntimeError):
            continue
        t.join()
        break
    return res.get("rc", 1)


if __name__ == "__main__":
    sys.exit(main())
