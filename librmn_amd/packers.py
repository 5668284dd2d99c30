"""Python mirror of librmn's packer C interface (src/packers/packers.h, include/armn_compress.h) bound to
librmn_ez_hip.so.  Host calls take numpy arrays (uint32 word buffers for packed data)."""
import ctypes
import numpy as np

from .lib import load_library

FLOAT_PACK, FLOAT_UNPACK = 1, 2
_configured = False


def _lib():
    global _configured
    L = load_library()
    if not _configured:
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.compact_float.restype = vp
        L.compact_float.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
        L.compact_integer.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci]
        L.compact_double.restype = vp
        L.compact_double.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
        L.compact_short.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci]
        L.compact_char.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci]
        L.c_float_packer.argtypes = [vp, ci, vp, vp, ci]
        L.c_float_unpacker.argtypes = [vp, vp, vp, ci, vp]
        L.armn_compress.argtypes = [vp, ci, ci, ci, ci, ci]
        L.c_armn_compress_setlevel.argtypes = [ci]
        L.compact_float_dev.restype = vp
        L.compact_float_dev.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, ci]
        L.compact_integer_dev.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci]
        L.armn_compress_dev.argtypes = [vp, vp, ci, ci, ci]
        L.ezhip_pack16_compress_dev.argtypes = [vp, vp, ci, ci, ci]
        L.ezhip_ezsint_pack16_batch_dev.argtypes = [vp, ctypes.c_int64, vp, vp, ci, ci, ci]
        L.ezhip_pack_release.restype = None
        L.c_armn_compress_setswap.argtypes = [ci]
        L.armn_uncompress_dev.argtypes = [vp, vp, ctypes.c_int64, ci, ci, ci]
        L.ezhip_armn_uncompress_batch_dev.argtypes = [vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int64, ci, ci, ci, ci]
        L.ezhip_uncompress_unpack16_dev.argtypes = [vp, vp, ci, ci, ci, ci]
        L.ezhip_pack16_compress_batch_dev.argtypes = [vp, ctypes.c_int64, vp, ctypes.c_int64, ci, ci, ci, ci, ci, vp]
        _configured = True
    return L


def stream_words(n, bits, offset=0):
    return (offset + n * bits + 31) // 32


def compact_float_pack(a, nbits_arg, offset=0, stride=1, has_missing=0, tag=0.0, style1=False, prefill=0):
    """returns the uint32 image [4 header words | stream] (style 2) or the style-1 image where the stream
    starts at bit 24 of header word 3 (pass style1=True; offset is then forced to 24)"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    n = (a.size + stride - 1) // stride
    eff = (nbits_arg >> 6) if nbits_arg > 64 else nbits_arg
    tagv = np.array([tag], np.float32)
    if style1:
        buf = np.full(3 + stream_words(n, eff, 24) + 1, prefill, np.uint32)
        r = _lib().compact_float(a.ctypes.data, buf.ctypes.data, buf[3:].ctypes.data, n, nbits_arg, 24, stride, FLOAT_PACK, has_missing, tagv.ctypes.data)
    else:
        buf = np.full(4 + stream_words(n, eff, offset) + 1, prefill, np.uint32)
        r = _lib().compact_float(a.ctypes.data, buf.ctypes.data, buf[4:].ctypes.data, n, nbits_arg, offset, stride, FLOAT_PACK, has_missing, tagv.ctypes.data)
    return (None if not r else buf)


def compact_float_unpack(buf, n, nbits_arg, offset=0, stride=1, has_missing=0, tag=0.0):
    out = np.zeros(n * stride, np.float32)
    tagv = np.array([tag], np.float32)
    r = _lib().compact_float(out.ctypes.data, buf.ctypes.data, buf[4:].ctypes.data, n, nbits_arg, offset, stride, FLOAT_UNPACK, has_missing, tagv.ctypes.data)
    return None if not r else out


def compact_integer_pack(a, nbits, op, header=False, offset=0, stride=1, prefill=0):
    a = np.ascontiguousarray(a)
    n = (a.size + stride - 1) // stride
    nb = 32 if nbits == -1 else nbits
    out = np.full(stream_words(n, nb, offset) + 1, prefill, np.uint32)
    hdr = np.zeros(4, np.uint32)
    rc = _lib().compact_integer(a.ctypes.data, hdr.ctypes.data if header else None, out.ctypes.data, n, nbits, offset, stride, op)
    return rc, hdr, out


def compact_integer_unpack(out, n, nbits, op, hdr=None, offset=0, stride=1, dtype=np.uint32):
    a = np.zeros(n * stride, dtype)
    rc = _lib().compact_integer(a.ctypes.data, hdr.ctypes.data if hdr is not None else None, out.ctypes.data, n, nbits, offset, stride, op)
    return rc, a


def float_packer(a, nbits):
    a = np.ascontiguousarray(a, dtype=np.float32)
    hdr = np.zeros(3, np.int32); stream = np.zeros((a.size + 1) // 2, np.int32)
    rc = _lib().c_float_packer(a.ctypes.data, nbits, hdr.ctypes.data, stream.ctypes.data, a.size)
    return rc, hdr, stream


def float_unpacker(hdr, stream, n):
    out = np.zeros(n, np.float32); nb = ctypes.c_int(0)
    rc = _lib().c_float_unpacker(out.ctypes.data, hdr.ctypes.data, stream.ctypes.data, n, ctypes.byref(nb))
    return rc, out, nb.value


def armn_compress(words, ni, nj, nbits):
    """in place on a uint32 word buffer (two 16-bit tokens per word); returns zlng or -1"""
    return _lib().armn_compress(words.ctypes.data, ni, nj, 1, nbits, 1)


def armn_uncompress(buf, ni, nj, nbits):
    """in place on a uint32 buffer of at least 1 + ni*nj/2 words holding an armn stream; returns ni*nj*2 or -1"""
    return _lib().armn_compress(buf.ctypes.data, ni, nj, 1, nbits, 2)


def armn_setlevel(level):
    _lib().c_armn_compress_setlevel(level)


def armn_setswap(state):
    _lib().c_armn_compress_setswap(state)


def _dptr(t):
    return t if isinstance(t, int) else t.data_ptr()


def compact_float_pack_dev(d_field, d_header, d_stream, n, nbits_arg, offset=0, stride=1, style=2):
    tagv = np.array([0.0], np.float32)
    return _lib().compact_float_dev(_dptr(d_field), _dptr(d_header), _dptr(d_stream), n, nbits_arg, offset, stride, FLOAT_PACK, 0, tagv.ctypes.data, style)


def armn_compress_dev(d_z, d_words, ni, nj, nbits):
    return _lib().armn_compress_dev(_dptr(d_z), _dptr(d_words), ni, nj, nbits)


def armn_uncompress_dev(d_words, d_z, z_words, ni, nj, nbits):
    return _lib().armn_uncompress_dev(_dptr(d_words), _dptr(d_z), z_words, ni, nj, nbits)


def armn_uncompress_batch_dev(d_words, out_stride_words, d_z, z_stride_words, z_words, ni, nj, nbits, nfields):
    return _lib().ezhip_armn_uncompress_batch_dev(_dptr(d_words), out_stride_words, _dptr(d_z), z_stride_words, z_words, ni, nj, nbits, nfields)


def uncompress_unpack16_dev(d_field, d_record, ni, nj, nbits, compressed):
    return _lib().ezhip_uncompress_unpack16_dev(_dptr(d_field), _dptr(d_record), ni, nj, nbits, int(compressed))


def pack16_compress_dev(d_record, d_field, ni, nj, nbits):
    return _lib().ezhip_pack16_compress_dev(_dptr(d_record), _dptr(d_field), ni, nj, nbits)


def ezsint_pack16_batch_dev(d_records, record_stride_words, d_zout, d_zin, nfields, npts_out, nbits):
    """fused c_ezsint_batch_dev + compact_float 16-bit-slot pack of every output field (min/max from the interpolation kernel)"""
    return _lib().ezhip_ezsint_pack16_batch_dev(_dptr(d_records), record_stride_words, _dptr(d_zout), _dptr(d_zin), nfields, npts_out, nbits)


def pack_release():
    _lib().ezhip_pack_release()


def pack16_compress_batch_dev(d_records, record_stride_words, d_fields, field_stride, nfields, ni, nj, nbits, prepacked=0):
    """compact_float(16-bit slots, unless prepacked) + armn_compress of a batch; one synchronisation; returns (rc, zlng[nfields])"""
    zl = np.zeros(nfields, np.int32)
    rc = _lib().ezhip_pack16_compress_batch_dev(_dptr(d_records), record_stride_words, _dptr(d_fields) if d_fields is not None else None,
                                                field_stride, nfields, ni, nj, nbits, prepacked, zl.ctypes.data)
    return rc, zl


def ezsint_pack16_compress_batch_dev(d_records, record_stride_words, d_zin, nfields, ni_out, nj_out, nbits):
    """the fused cfg5 pipeline (interpolation twice: min/max only, then tokens; one-pass armn encoder writing in place);
    returns (rc, zlng[nfields])"""
    L = _lib()
    L.ezhip_ezsint_pack16_compress_batch_dev.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    zl = np.zeros(nfields, np.int32)
    rc = L.ezhip_ezsint_pack16_compress_batch_dev(_dptr(d_records), record_stride_words, _dptr(d_zin), nfields, ni_out, nj_out, nbits, zl.ctypes.data)
    return rc, zl


def compact_double_pack(a, nbits_arg, offset=0, stride=1, has_missing=0, tag=0.0, prefill=0):
    """compact_double FLOAT_PACK on a float64 array: returns the uint32 image [4 header words | stream] (header style 2)"""
    a = np.ascontiguousarray(a, dtype=np.float64)
    n = (a.size + stride - 1) // stride
    eff = (nbits_arg >> 6) if nbits_arg > 64 else nbits_arg
    tagv = np.array([tag], np.float64)
    buf = np.full(4 + stream_words(n, eff, offset) + 1, prefill, np.uint32)
    r = _lib().compact_double(a.ctypes.data, buf.ctypes.data, buf[4:].ctypes.data, n, nbits_arg, offset, stride, FLOAT_PACK, has_missing, tagv.ctypes.data)
    return None if not r else buf


def compact_double_unpack(buf, n, nbits_arg, offset=0, stride=1, has_missing=0, tag=0.0):
    out = np.zeros(n * stride, np.float64)
    tagv = np.array([tag], np.float64)
    r = _lib().compact_double(out.ctypes.data, buf.ctypes.data, buf[4:].ctypes.data, n, nbits_arg, offset, stride, FLOAT_UNPACK, has_missing, tagv.ctypes.data)
    return None if not r else out


def compact_narrow_pack(a, nbits, header=False, offset=0, stride=1, prefill=0):
    """compact_short (uint16 input, opCode 5) / compact_char (uint8 input, opCode 9)"""
    a = np.ascontiguousarray(a)
    assert a.dtype in (np.uint16, np.uint8)
    n = (a.size + stride - 1) // stride
    nb = 32 if nbits == -1 else nbits
    out = np.full(stream_words(n, nb, offset) + 1, prefill, np.uint32)
    hdr = np.zeros(4, np.uint32)
    fn = _lib().compact_short if a.dtype == np.uint16 else _lib().compact_char
    rc = fn(a.ctypes.data, hdr.ctypes.data if header else None, out.ctypes.data, n, nbits, offset, stride, 5 if a.dtype == np.uint16 else 9)
    return rc, hdr, out


def compact_narrow_unpack(out, n, nbits, dtype, hdr=None, offset=0, stride=1, fill=0):
    a = np.full(n * stride, fill, dtype)
    fn = _lib().compact_short if dtype == np.uint16 else _lib().compact_char
    rc = fn(a.ctypes.data, hdr.ctypes.data if hdr is not None else None, out.ctypes.data, n, nbits, offset, stride, 6 if dtype == np.uint16 else 10)
    return rc, a


def armn_compress32(fld, ni, nj, znbits):
    """c_armn_compress32 on a float32 field (host): returns (zlng, uint32 stream words) ; zlng = -1 when not compressible"""
    f = np.ascontiguousarray(fld, dtype=np.float32)
    L = _lib()
    L.c_armn_compress32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    z = np.zeros(ni * nj * max(znbits, 8) // 32 + 64, np.uint32)
    zlng = L.c_armn_compress32(z.ctypes.data, f.ctypes.data, ni, nj, 1, znbits)
    return zlng, z


def armn_uncompress32(z, ni, nj, znbits, out=None):
    """c_armn_uncompress32 (the reference's signature: no length); out: a float32 array of ni * nj to decode into (a caller that keeps its field buffer, as the
    reference's callers do, pays no page faults of a fresh one)"""
    L = _lib()
    L.c_armn_uncompress32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    if out is None:
        out = np.zeros(ni * nj, np.float32)
    assert out.dtype == np.float32 and out.size >= ni * nj and out.flags["C_CONTIGUOUS"]
    rc = L.c_armn_uncompress32(out.ctypes.data, z.ctypes.data, ni, nj, 1, znbits)
    return rc, out


def armn_uncompress32_lng(z, zbytes, ni, nj, znbits, out=None):
    """c_armn_uncompress32_lng: the stream's byte length is given (an upper bound inside the buffer), the tile chains are followed on the device"""
    L = _lib()
    L.c_armn_uncompress32_lng.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    if out is None:
        out = np.zeros(ni * nj, np.float32)
    assert out.dtype == np.float32 and out.size >= ni * nj and out.flags["C_CONTIGUOUS"]
    rc = L.c_armn_uncompress32_lng(out.ctypes.data, z.ctypes.data, int(zbytes), ni, nj, 1, znbits)
    return rc, out


def armn_compress32_dev(d_z, d_fld, ni, nj, znbits):
    """c_armn_compress32_dev: torch CUDA float32 field -> record in the torch CUDA int32 buffer d_z (ni*nj + 64 words); returns the byte count or -1"""
    L = _lib()
    L.c_armn_compress32_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return L.c_armn_compress32_dev(d_z.data_ptr(), d_fld.data_ptr(), ni, nj, 1, znbits)


def armn_uncompress32_zdev(d_fld, d_z, zbytes, ni, nj, znbits):
    """c_armn_uncompress32_zdev: record and field both torch CUDA tensors; returns ni*nj or -1"""
    L = _lib()
    L.c_armn_uncompress32_zdev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return L.c_armn_uncompress32_zdev(d_fld.data_ptr(), d_z.data_ptr(), int(zbytes), ni, nj, 1, znbits)


def fst_pack_data(field, ni, nj, nk, datyp, nbits):
    """the data part of an FST record (c_fstecr's packing switch): returns (words, datyp_out, nbits_out, uint32 buffer).  The element size is the
    array's (float32 / int32 / uint32: 4, float64: 8, int16 / uint16: 2, int8 / uint8: 1); datyp may carry the missing-value flag (+64)"""
    L = _lib()
    L.ezhip_fst_pack_data_ex.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p] + [ctypes.c_int32] * 6 + [ctypes.c_void_p] * 2
    f = np.ascontiguousarray(field)
    n = ni * nj * nk
    cap = n * max(nbits, 16) // 32 + n // 8 + 256 + (2 * n if f.itemsize == 8 else 0)
    buf = np.zeros(cap, np.uint32)
    d = ctypes.c_int32(-1); b = ctypes.c_int32(-1)
    w = L.ezhip_fst_pack_data_ex(buf.ctypes.data, cap, f.ctypes.data, ni, nj, nk, datyp, nbits, f.itemsize, ctypes.addressof(d), ctypes.addressof(b))
    return w, d.value, b.value, buf


def fst_unpack_data(buf, ni, nj, nk, datyp, nbits, dtype=np.float32):
    L = _lib()
    L.ezhip_fst_unpack_data_ex.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 6
    out = np.zeros(ni * nj * nk, dtype)
    rc = L.ezhip_fst_unpack_data_ex(out.ctypes.data, buf.ctypes.data, ni, nj, nk, datyp, nbits, out.itemsize)
    return rc, out


def fst_force_missing_value_usage(enable):
    return _lib().ezhip_fst_force_missing_value_usage(1 if enable else 0)


def fst_frame_record_dev(d_data, cap_words, d_record, zlng, ni, nj, nk, nbits):
    """a cfg5 record in HBM -> the data part of the datyp-129 (or, zlng < 0, datyp-1) FST record, on the device; returns (words, datyp_out).
    d_data / d_record: torch int32 CUDA tensors or raw device pointers"""
    L = _lib()
    L.ezhip_fst_frame_record_dev.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p] + [ctypes.c_int32] * 5 + [ctypes.c_void_p]
    d = ctypes.c_int32(-1)
    pd = d_data if isinstance(d_data, int) else d_data.data_ptr()
    pr = d_record if isinstance(d_record, int) else d_record.data_ptr()
    w = L.ezhip_fst_frame_record_dev(pd, cap_words, pr, int(zlng), ni, nj, nk, nbits, ctypes.addressof(d))
    return w, d.value
