"""Compiled callers of librmn_ez_hip.so -- what a maintainer's application looks like after the switch: tests/callers/ez_c.c (gcc, built against
include/*.h, linked with -lrmn_ez_hip) and tests/callers/ez_f.f90 (AMD flang: hidden string lengths, by-reference scalars, LOGICAL arguments:
rpnmacros.h:21,32-49, ezqkdef.c:39-55, Interp1D_Linear.F90:22-99).  Each runs as a child process with no Python in it; its outputs must equal
the ctypes results of this process bit for bit."""
import os, subprocess
from conftest import run_child
import numpy as np
import pytest

import ezcases as ec

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
CALLERS = os.path.join(ROOT, "tests", "callers")
NI, NJ, NO, MO = 128, 64, 180, 91


def _inputs(tmp_path):
    z = ec.synth_field(NI, NJ, seed=3)
    u, v = ec.synth_wind(NI, NJ, seed=4)
    p = tmp_path / "in.bin"
    with open(p, "wb") as f:
        np.array([NI, NJ, NO, MO], np.int32).tofile(f); z.tofile(f); u.tofile(f); v.tofile(f)
    return z, u, v, str(p)


def _expected(z, u, v):
    from librmn_amd import ezscint as ez
    gi = ez.ezqkdef(NI, NJ, "G", 0, 0, 0, 0); go = ez.ezqkdef(NO, MO, "L", 200, 200, 0, 0)
    assert ez.ezsetopt("interp_degree", "cubic") == 0 and ez.ezsetopt("polar_correction", "yes") == 0
    assert ez.ezdefset(go, gi) == 1
    rc, zo = ez.ezsint(z, NO * MO)
    assert rc >= 0
    rc, uo, vo = ez.ezuvint(u, v, NO * MO)
    assert rc >= 0
    return zo, uo, vo


def _run(exe, fin, fout):
    path = os.path.join(CALLERS, exe)
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (__graft_entry__.build() makes it)")
    r = run_child([path, fin, fout], cwd=ROOT, timeout=600)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2000:])
    return r.stdout


def test_c_program_against_the_headers(tmp_path):
    z, u, v, fin = _inputs(tmp_path)
    fout = str(tmp_path / "out_c.bin")
    out = _run("ez_c", fin, fout)
    assert "ez_c:" in out
    n = NO * MO
    raw = np.fromfile(fout, np.uint32)
    zo, uo, vo = raw[:n].view(np.float32), raw[n:2 * n].view(np.float32), raw[2 * n:3 * n].view(np.float32)
    zlng = int(raw[3 * n:3 * n + 1].view(np.int32)[0]); rec = raw[3 * n + 1:]
    ez_, eu, ev = _expected(z, u, v)
    assert np.array_equal(zo.view(np.uint32), ez_.view(np.uint32))
    assert np.array_equal(uo.view(np.uint32), eu.view(np.uint32)) and np.array_equal(vo.view(np.uint32), ev.view(np.uint32))
    # the record: compact_float(16-bit slots) + armn_compress of the same field through the ctypes mirror
    import ctypes
    from librmn_amd.lib import load_library
    L = load_library()
    r2 = np.zeros(4 + n // 2 + 16, np.uint32)
    tag = ctypes.c_double(99999.0)
    L.compact_float.restype = ctypes.c_void_p
    L.compact_float.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
    assert L.compact_float(ez_.ctypes.data, r2.ctypes.data, r2[4:].ctypes.data, n, 16 + 64 * 16, 0, 1, 1, 0, ctypes.addressof(tag))
    L.armn_compress.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5
    zl2 = L.armn_compress(r2[4:].ctypes.data, NO, MO, 1, 16, 1)
    assert zlng == zl2 and zlng > 0
    m = 4 + (zlng - 1) // 4
    assert np.array_equal(rec[:m], r2[:m])


def test_fortran_program_through_the_fortran_symbols(tmp_path):
    from librmn_amd import interpv as iv
    z, u, v, fin = _inputs(tmp_path)
    fout = str(tmp_path / "out_f.bin")
    out = _run("ez_f", fin, fout)
    assert "ez_f:" in out
    n = NO * MO
    raw = np.fromfile(fout, np.float32)
    zo, uo, vo = raw[:n], raw[n:2 * n], raw[2 * n:3 * n]
    ez_, eu, ev = _expected(z, u, v)
    assert np.array_equal(zo.view(np.uint32), ez_.view(np.uint32))
    assert np.array_equal(uo.view(np.uint32), eu.view(np.uint32)) and np.array_equal(vo.view(np.uint32), ev.view(np.uint32))
    # the vertical step of the program, repeated through the ctypes mirror on the same arrays (Fortran (ijDim, levels) = C [levels][ijDim])
    ncol, ns, nd = min(NO, 64), 6, 4
    i = np.arange(1, ncol + 1, dtype=np.float32)
    lev_s = np.zeros((ns, ncol + 3), np.float32); st_s = np.zeros_like(lev_s); der_s = np.zeros_like(lev_s)
    lev_d = np.zeros((nd, ncol + 2), np.float32); st_d = np.full((nd, ncol + 2), -1.0, np.float32); der_d = np.zeros_like(lev_d)
    for k in range(ns):
        lev_s[k, :ncol] = np.float32(100.0) * np.float32(k + 1) + np.float32(0.25) * i
        st_s[k, :ncol] = ez_[k * NO:k * NO + ncol]
    for k in range(nd):
        lev_d[k, :ncol] = np.float32(130.0) + np.float32(110.0) * np.float32(k + 1) + np.float32(0.125) * i
    posn = iv.findpos(ncol, lev_s, lev_d)
    iv.interp(iv.LINEAR, ncol, lev_s, st_s, der_s, posn, lev_d, st_d, der_d, xdown=False, xup=True)
    prof = raw[3 * n:].reshape(nd, ncol)
    assert np.array_equal(prof.view(np.uint32), st_d[:, :ncol].view(np.uint32)), np.abs(prof - st_d[:, :ncol]).max()
