"""Python mirror of librmn's EZSCINT C interface (src/PUBLIC_INCLUDES/rmn/ezscint.h) bound to
librmn_ez_hip.so.  Host-pointer calls take numpy float32 arrays; ``*_dev`` calls take torch CUDA
(ROCm) tensors or raw device pointers and enqueue on the given / current stream."""
import ctypes
import numpy as np

from .lib import load_library

_F = ctypes.POINTER(ctypes.c_float)
_configured = False


def _lib():
    global _configured
    L = load_library()
    if not _configured:
        i32, vp, cp = ctypes.c_int32, ctypes.c_void_p, ctypes.c_char_p
        L.c_ezqkdef.argtypes = [i32, i32, cp, i32, i32, i32, i32, i32]
        L.c_ezgdef_fmem.argtypes = [i32, i32, cp, cp, i32, i32, i32, i32, vp, vp]
        L.c_ezdefset.argtypes = [i32, i32]
        L.c_ezgdef_supergrid.argtypes = [i32, i32, cp, cp, i32, i32, vp]
        L.c_ezsetopt.argtypes = [cp, cp]
        L.c_ezgetopt.argtypes = [cp, cp]
        L.c_ezsetval.argtypes = [cp, ctypes.c_float]
        L.c_ezsetival.argtypes = [cp, i32]
        L.c_ezsint.argtypes = [vp, vp]
        L.c_ezsetival.argtypes = [cp, i32]; L.c_ezgetival.argtypes = [cp, vp]
        L.c_gdwdfuv.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32]; L.c_gduvfwd.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32]
        L.c_gdllfxy.argtypes = [i32, vp, vp, vp, vp, i32]; L.c_gdxywdval.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32]; L.c_gdllwdval.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32]
        L.c_gdxyzfll.argtypes = [i32, vp, vp, vp, vp, i32]; L.c_ezgdef_fll.argtypes = [i32, i32, vp, vp]
        L.c_ezget_nsubgrids.argtypes = [i32]; L.c_ezget_subgridids.argtypes = [i32, vp]
        L.c_gdsetmask.argtypes = [i32, vp]; L.c_gdgetmask.argtypes = [i32, vp]
        L.c_ezsint_mask.argtypes = [vp, vp]; L.c_ezget_mask_zones.argtypes = [vp, vp]
        L.c_ezsint_mdm.argtypes = [vp, vp, vp, vp]; L.c_ezuvint_mdm.argtypes = [vp, vp, vp, vp, vp, vp]
        L.c_ezuvint.argtypes = [vp, vp, vp, vp]
        L.c_ezwdint.argtypes = [vp, vp, vp, vp]
        L.c_ezwdint_dev.argtypes = [vp, vp, vp, vp]
        L.c_gdxysint.argtypes = [vp, vp, i32, vp, vp, i32]
        L.c_gdxysval.argtypes = [i32, vp, vp, vp, vp, i32]
        L.c_gdxyfll.argtypes = [i32, vp, vp, vp, vp, i32]
        L.c_gdll.argtypes = [i32, vp, vp]
        L.c_gdgaxes.argtypes = [i32, vp, vp]
        L.c_gdrls.argtypes = [i32]
        L.ezhip_use_stream.argtypes = [vp]
        L.ezhip_use_stream.restype = None
        L.c_ezsint_dev.argtypes = [vp, vp]
        L.c_ezuvint_dev.argtypes = [vp, vp, vp, vp]
        L.c_gdxysint_dev.argtypes = [vp, vp, i32, vp, vp, i32]
        L.c_gdxyfll_dev.argtypes = [i32, vp, vp, vp, vp, i32]
        L.c_ezsint_batch_dev.argtypes = [vp, vp, i32]
        L.c_ezuvint_batch_dev.argtypes = [vp, vp, vp, vp, i32]
        _configured = True
    return L


def _np(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data


def _dptr(t):
    """device pointer of a torch tensor (or pass an int through)"""
    if isinstance(t, int):
        return t
    assert t.is_cuda and t.is_contiguous() and t.dtype.is_floating_point and t.element_size() == 4
    return t.data_ptr()


def available():
    return bool(_lib().ezhip_available())


def ezqkdef(ni, nj, grtyp, ig1, ig2, ig3, ig4, iunit=0):
    return _lib().c_ezqkdef(ni, nj, grtyp.encode(), ig1, ig2, ig3, ig4, iunit)


def ezgdef_fmem(ni, nj, grtyp, grref, ig1, ig2, ig3, ig4, ax, ay):
    ax, pax = _np(ax); ay, pay = _np(ay)
    return _lib().c_ezgdef_fmem(ni, nj, grtyp.encode(), grref.encode(), ig1, ig2, ig3, ig4, pax, pay)


def ezgdef_supergrid(ni, nj, grtyp, grref, vercode, subgrids):
    ids = np.asarray(subgrids, dtype=np.int32)
    return _lib().c_ezgdef_supergrid(ni, nj, grtyp.encode(), grref.encode(), vercode, ids.size, ids.ctypes.data)


def ezdefset(gdout, gdin):
    return _lib().c_ezdefset(gdout, gdin)


def ezsetopt(option, value):
    return _lib().c_ezsetopt(option.encode(), value.encode())


def ezgetopt(option):
    buf = ctypes.create_string_buffer(64)
    _lib().c_ezgetopt(option.encode(), buf)
    return buf.value.decode()


def ezsetval(option, value):
    return _lib().c_ezsetval(option.encode(), float(value))


def gdrls(gdid):
    return _lib().c_gdrls(gdid)


def gdgaxes(gdid, ni, nj):
    ax = np.zeros(ni, np.float32); ay = np.zeros(nj, np.float32)
    rc = _lib().c_gdgaxes(gdid, ax.ctypes.data, ay.ctypes.data)
    return rc, ax, ay


def gdll(gdid, npts):
    lat = np.zeros(npts, np.float32); lon = np.zeros(npts, np.float32)
    rc = _lib().c_gdll(gdid, lat.ctypes.data, lon.ctypes.data)
    return rc, lat, lon


def gdxyfll(gdid, lat, lon):
    lat, plat = _np(lat); lon = np.array(lon, dtype=np.float32, copy=True)
    x = np.zeros(lat.size, np.float32); y = np.zeros(lat.size, np.float32)
    rc = _lib().c_gdxyfll(gdid, x.ctypes.data, y.ctypes.data, plat, lon.ctypes.data, lat.size)
    return rc, x, y


def ezsint(zin, nout):
    """host-pointer c_ezsint: returns (rc, zout)"""
    zin, pin = _np(zin)
    zout = np.zeros(nout, np.float32)
    rc = _lib().c_ezsint(zout.ctypes.data, pin)
    return rc, zout


def ezuvint(uuin, vvin, nout):
    uuin, pu = _np(uuin); vvin, pv = _np(vvin)
    uo = np.zeros(nout, np.float32); vo = np.zeros(nout, np.float32)
    rc = _lib().c_ezuvint(uo.ctypes.data, vo.ctypes.data, pu, pv)
    return rc, uo, vo


def ezwdint(uuin, vvin, nout):
    """c_ezwdint: interpolated winds as (speed, meteorological direction) on the target grid"""
    uuin, pu = _np(uuin); vvin, pv = _np(vvin)
    so = np.zeros(nout, np.float32); do = np.zeros(nout, np.float32)
    rc = _lib().c_ezwdint(so.ctypes.data, do.ctypes.data, pu, pv)
    return rc, so, do


def gdxysint(zin, gdin, x, y):
    zin, pin = _np(zin); x, px = _np(x); y, py = _np(y)
    zout = np.zeros(x.size, np.float32)
    rc = _lib().c_gdxysint(zout.ctypes.data, pin, gdin, px, py, x.size)
    return rc, zout


def gdllsval(gdid, zin, lat, lon):
    """c_gdllsval: interpolation at lat/lon points (locate + c_gdxysval)"""
    zin, pin = _np(zin); lat, pla = _np(lat); lon, plo = _np(lon)
    zout = np.zeros(lat.size, np.float32)
    L = _lib()
    L.c_gdllsval.argtypes = [ctypes.c_int32] + [ctypes.c_void_p] * 4 + [ctypes.c_int32]
    rc = L.c_gdllsval(gdid, zout.ctypes.data, pin, pla, plo, lat.size)
    return rc, zout


def gdllvval(gdid, uuin, vvin, lat, lon):
    uuin, pu = _np(uuin); vvin, pv = _np(vvin); lat, pla = _np(lat); lon, plo = _np(lon)
    uo = np.zeros(lat.size, np.float32); vo = np.zeros(lat.size, np.float32)
    L = _lib()
    L.c_gdllvval.argtypes = [ctypes.c_int32] + [ctypes.c_void_p] * 6 + [ctypes.c_int32]
    rc = L.c_gdllvval(gdid, uo.ctypes.data, vo.ctypes.data, pu, pv, pla, plo, lat.size)
    return rc, uo, vo


def use_stream(stream_handle):
    """hipStream_t as an integer (torch: torch.cuda.current_stream().cuda_stream); 0/None = null stream"""
    _lib().ezhip_use_stream(ctypes.c_void_p(stream_handle or 0))


def develop_build():
    """True when the loaded library carries the development knock-outs (make develop); the shipped one never does"""
    return bool(_lib().ezhip_develop_build())


def ezsint_dev(zout, zin):
    return _lib().c_ezsint_dev(_dptr(zout), _dptr(zin))


def ezsint_batch_dev(zout, zin, nfields):
    return _lib().c_ezsint_batch_dev(_dptr(zout), _dptr(zin), nfields)


def ezuvint_dev(uuout, vvout, uuin, vvin):
    return _lib().c_ezuvint_dev(_dptr(uuout), _dptr(vvout), _dptr(uuin), _dptr(vvin))


def ezwdint_dev(spdout, dirout, uuin, vvin):
    return _lib().c_ezwdint_dev(_dptr(spdout), _dptr(dirout), _dptr(uuin), _dptr(vvin))


def gdxysint_dev(zout, zin, gdin, x, y, npts):
    return _lib().c_gdxysint_dev(_dptr(zout), _dptr(zin), gdin, _dptr(x), _dptr(y), npts)


def gdxyfll_dev(gdid, x, y, lat, lon, n):
    return _lib().c_gdxyfll_dev(gdid, _dptr(x), _dptr(y), _dptr(lat), _dptr(lon), n)


def prepare_set():
    return _lib().ezhip_prepare_set()


def set_mode():
    return _lib().ezhip_set_mode()


def set_wind_exact(on):
    """ezhip_set_wind_exact: 1 = c_ezuvint runs the reference's wind chain as written on every call (bit-exact nearest / bilinear winds), 0 = the per-point matrix (default)"""
    L = _lib()
    L.ezhip_set_wind_exact.argtypes = [ctypes.c_int32]; L.ezhip_set_wind_exact.restype = None
    L.ezhip_set_wind_exact(int(on))


def set_xy_dev(x, y):
    L = _lib()
    L.ezhip_set_xy_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    return L.ezhip_set_xy_dev(_dptr(x), _dptr(y))


def ezuvint_batch_dev(uuout, vvout, uuin, vvin, npairs):
    """c_ezuvint_batch_dev: npairs wind pairs, contiguous torch CUDA tensors [npairs, n]"""
    return _lib().c_ezuvint_batch_dev(_dptr(uuout), _dptr(vvout), _dptr(uuin), _dptr(vvin), npairs)


def libm_exact_probe(fn, a, b, out):
    """libm_exact.h on the device: fn 0 sinf, 1 cosf, 2 asinf, 3 atanf, 4 atan2f(a, b) over device arrays"""
    L = _lib()
    L.ezhip_libm_exact_probe.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    return L.ezhip_libm_exact_probe(fn, _dptr(a), _dptr(b) if b is not None else None, _dptr(out), a.numel())


def gdsetmask(gdid, mask):
    m = np.ascontiguousarray(mask, dtype=np.int32)
    return _lib().c_gdsetmask(gdid, m.ctypes.data)


def gdgetmask(gdid, n):
    m = np.zeros(n, np.int32)
    return _lib().c_gdgetmask(gdid, m.ctypes.data), m


def ezsint_mask(mask_in, nout):
    mi = np.ascontiguousarray(mask_in, dtype=np.int32); mo = np.zeros(nout, np.int32)
    return _lib().c_ezsint_mask(mo.ctypes.data, mi.ctypes.data), mo


def ezget_mask_zones(mask_in, nout):
    mi = np.ascontiguousarray(mask_in, dtype=np.int32); mo = np.zeros(nout, np.int32)
    return _lib().c_ezget_mask_zones(mo.ctypes.data, mi.ctypes.data), mo


def ezsint_mdm(zin, mask_in, nout):
    zin, pin = _np(zin); mi = np.ascontiguousarray(mask_in, dtype=np.int32)
    zo = np.zeros(nout, np.float32); mo = np.zeros(nout, np.int32)
    return _lib().c_ezsint_mdm(zo.ctypes.data, mo.ctypes.data, pin, mi.ctypes.data), zo, mo


def ezuvint_mdm(uuin, vvin, mask_in, nout):
    uuin, pu = _np(uuin); vvin, pv = _np(vvin); mi = np.ascontiguousarray(mask_in, dtype=np.int32)
    uo = np.zeros(nout, np.float32); vo = np.zeros(nout, np.float32); mo = np.zeros(nout, np.int32)
    return _lib().c_ezuvint_mdm(uo.ctypes.data, vo.ctypes.data, mo.ctypes.data, pu, pv, mi.ctypes.data), uo, vo, mo


def gdwdfuv(gdid, uu, vv, lat, lon):
    """grid components at (lat, lon) -> (speed, direction)"""
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (uu, vv, lat, lon)]
    spd = np.zeros_like(a[0]); wd = np.zeros_like(a[0])
    rc = _lib().c_gdwdfuv(gdid, spd.ctypes.data, wd.ctypes.data, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, a[0].size)
    return rc, spd, wd


def gduvfwd(gdid, spd, wd, lat, lon):
    """(speed, direction) at (lat, lon) -> grid components"""
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (spd, wd, lat, lon)]
    uu = np.zeros_like(a[0]); vv = np.zeros_like(a[0])
    rc = _lib().c_gduvfwd(gdid, uu.ctypes.data, vv.ctypes.data, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, a[0].size)
    return rc, uu, vv


def gdxyzfll(gdid, lat, lon):
    lat = np.ascontiguousarray(lat, dtype=np.float32); lon = np.array(lon, dtype=np.float32, copy=True)
    x = np.zeros_like(lat); y = np.zeros_like(lat)
    rc = _lib().c_gdxyzfll(gdid, x.ctypes.data, y.ctypes.data, lat.ctypes.data, lon.ctypes.data, lat.size)
    return rc, x, y


def ezgdef_fll(ni, nj, lat, lon):
    lat = np.ascontiguousarray(lat, dtype=np.float32); lon = np.ascontiguousarray(lon, dtype=np.float32)
    return _lib().c_ezgdef_fll(ni, nj, lat.ctypes.data, lon.ctypes.data)


def ezget_nsubgrids(gdid):
    return _lib().c_ezget_nsubgrids(gdid)


def ezget_subgridids(gdid):
    ids = np.zeros(4, np.int32)
    n = _lib().c_ezget_subgridids(gdid, ids.ctypes.data)
    return n, ids[:max(n, 0)]


def gdllfxy(gdid, x, y):
    x = np.ascontiguousarray(x, dtype=np.float32); y = np.ascontiguousarray(y, dtype=np.float32)
    lat = np.zeros_like(x); lon = np.zeros_like(x)
    rc = _lib().c_gdllfxy(gdid, lat.ctypes.data, lon.ctypes.data, x.ctypes.data, y.ctypes.data, x.size)
    return rc, lat, lon


def gdxywdval(gdid, uuin, vvin, x, y):
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (uuin, vvin, x, y)]
    s = np.zeros_like(a[2]); d = np.zeros_like(a[2])
    rc = _lib().c_gdxywdval(gdid, s.ctypes.data, d.ctypes.data, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, a[2].size)
    return rc, s, d


def gdllwdval(gdid, uuin, vvin, lat, lon):
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (uuin, vvin, lat, lon)]
    s = np.zeros_like(a[2]); d = np.zeros_like(a[2])
    rc = _lib().c_gdllwdval(gdid, s.ctypes.data, d.ctypes.data, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, a[2].size)
    return rc, s, d


def ezsetival(option, value):
    return _lib().c_ezsetival(option.encode(), int(value))


def ezgetival(option):
    v = ctypes.c_int32(0)
    rc = _lib().c_ezgetival(option.encode(), ctypes.byref(v))
    return rc, v.value


def _key2f(k):
    """the ordered-key encoding of the min/max reductions (ez_kernels.hip f2key) back to float32"""
    k = np.uint32(k)
    u = np.uint32(k & np.uint32(0x7FFFFFFF)) if (k & np.uint32(0x80000000)) else np.uint32(~k)
    return np.array([u], np.uint32).view(np.float32)[0]


def ezsint_batch_extrema_dev(zin, nfields, method="bounds"):
    """{min, max} of what c_ezsint_batch_dev would store for each of nfields device-resident source fields (current grid set), WITHOUT
    storing anything -- compact_float's first pass of the cfg5 pipeline.  method "interp": ezhip_ezsint_batch_minmax_only_dev (a k_sepx pass
    that stores nothing); "bounds": ezhip_ezsint_batch_minmax_bb_dev (bounds of the source windows + exact evaluation of the few that can
    hold an extremum).  Returns (rc, mins[nfields] float32, maxs[nfields] float32, flags[nfields] int32)."""
    import torch
    L = _lib()
    i32, vp, i64 = ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64
    L.ezhip_ezsint_batch_minmax_only_dev.argtypes = [vp, i32, vp, i64, vp]
    L.ezhip_ezsint_batch_minmax_bb_dev.argtypes = [vp, i32, vp, i64, vp, vp]
    stride = 3 * 8192
    part = torch.zeros((nfields, stride), dtype=torch.int32, device=zin.device)
    flags = torch.zeros(nfields, dtype=torch.int32, device=zin.device)
    npf = ctypes.c_int32(0)
    if method == "bounds":
        rc = L.ezhip_ezsint_batch_minmax_bb_dev(_dptr(zin), nfields, part.data_ptr(), stride, ctypes.byref(npf), flags.data_ptr())
    else:
        rc = L.ezhip_ezsint_batch_minmax_only_dev(_dptr(zin), nfields, part.data_ptr(), stride, ctypes.byref(npf))
    torch.cuda.synchronize()
    if rc < 0:
        return rc, None, None, None
    p = part.cpu().numpy().view(np.uint32)[:, :3 * npf.value].reshape(nfields, npf.value, 3)
    mins = np.array([_key2f(p[f, :, 0].min()) for f in range(nfields)], np.float32)
    maxs = np.array([_key2f(p[f, :, 1].max()) for f in range(nfields)], np.float32)
    return rc, mins, maxs, flags.cpu().numpy()
