"""Test twin of c_fstecr's packing switch (src/fstd98/fstd98.c:935-1000, :1140-1370) composed from the CPU ORACLE's packers: what the data part
of an FST record must contain for a (datyp, nbits) request.  TEST INFRASTRUCTURE: the product's composition is librmn_amd/csrc/fst_record_host.c."""
import ctypes
import numpy as np
import test_oracle_packers as top
import test_oracle_armn32 as ta32


def plain_words(datyp, n, nbits):
    hs = ss = 0
    if datyp in (6, 134):
        hs, ss = 3, (n + 1) // 2
        nw = ((hs + ss) * 32 + (32 if datyp == 134 else 0) + 63) // 64
    elif datyp == 129:
        nw = (n * max(nbits, 16) + 128 + 32 + 63) // 64
    elif datyp == 130:
        nw = (n * max(nbits, 16) + 32 + 63) // 64
    else:
        nw = (n * nbits + 120 + 63) // 64
    return 2 * nw, hs


def pack(field, ni, nj, nk, datyp, nbits):
    """returns (words, datyp_out, uint32 data, defined_mask) -- defined_mask marks the words the reference defines (garbage left behind by a
    refused compression is excluded)"""
    O = top.O(); O32 = ta32.O()
    n = ni * nj * nk
    if nbits > 16 and datyp != 133:          # fstd98.c:934
        datyp &= 0x7F
    if datyp == 6 and nbits > 24:
        datyp, nbits = 5, 32
    if datyp == 6 and nbits > 16:
        datyp = 1
    words, hs = plain_words(datyp, n, nbits)
    data = np.zeros(words + n + 64, np.uint32)
    tag = np.array([99999.0], np.float64)
    out = datyp
    f = np.ascontiguousarray(field)

    def cf(hdr_off, st_off, arg, offset):
        assert O.orc_compact_float(f.ctypes.data, data[hdr_off:].ctypes.data, data[st_off:].ctypes.data, n, arg, offset, 1, 1, 0, tag.ctypes.data)

    if datyp == 0:
        data[:(n * nbits + 31) // 32] = f.view(np.uint32)[:(n * nbits + 31) // 32]
    elif datyp in (1, 129):
        if datyp > 128 and nbits <= 16:
            cf(1, 5, nbits + 64 * max(16, nbits), 0)
            zl = O.orc_armn_compress(data[5:].ctypes.data, ni, nj, nk, nbits, 1)
            if zl < 0:
                out = 1; data[:] = 0; cf(0, 3, nbits, 24)
            else:
                words = 2 * (((16 + zl) * 8 + 63) // 64); data[0] = words
        else:
            cf(0, 3, nbits, 24)
    elif datyp in (2, 130, 4):
        op = 3 if datyp == 4 else 1
        if datyp == 130:
            tok = (f.view(np.int32) & ((1 << nbits) - 1)).astype(np.uint16)
            z = np.zeros(n + 64, np.uint32)
            zl = O.orc_armn_encode(z.ctypes.data, tok.ctypes.data, ni, nj, nbits)
            if zl >= 1 + 2 * n:
                zl = -1
            if zl < 0:
                out = 2
                assert O.orc_compact_integer(f.ctypes.data, None, data[1:].ctypes.data, n, nbits, 0, 1, 1)
            else:
                data[1:1 + (zl + 3) // 4] = z[:(zl + 3) // 4]
                words = 2 * (((4 + zl) * 8 + 63) // 64); data[0] = words
        else:
            assert O.orc_compact_integer(f.ctypes.data, None, data.ctypes.data, n, nbits, 0, 1, op)
    elif datyp == 5:
        data[:n] = f.view(np.uint32)
    elif datyp == 133:
        zl = O32.orc_armn_compress32(data[1:].ctypes.data, f.ctypes.data, ni, nj, nk, nbits)
        if zl < 0:
            out = 5; data[:] = 0; data[:n] = f.view(np.uint32)
        else:
            words = 2 * (((16 + zl) * 8 + 63) // 64); data[0] = words
    elif datyp in (6, 134):
        if datyp > 128 and nbits <= 16:
            assert O.orc_float_packer(f.ctypes.data, nbits, data[1:].ctypes.data, data[1 + hs:].ctypes.data, n) == 0
            zl = O.orc_armn_compress(data[1 + hs:].ctypes.data, ni, nj, nk, nbits, 1)
            if zl < 0:
                out = 6; data[:] = 0
                assert O.orc_float_packer(f.ctypes.data, nbits, data.ctypes.data, data[hs:].ctypes.data, n) == 0
            else:
                words = 2 * (((16 + 4 * hs + zl) * 8 + 63) // 64); data[0] = words
        else:
            assert O.orc_float_packer(f.ctypes.data, nbits, data.ctypes.data, data[hs:].ctypes.data, n) == 0
    else:
        raise ValueError(datyp)
    return words, out, data


def unpack(data, ni, nj, nk, datyp, nbits, dtype=np.float32):
    """c_fstluk's unpacking switch (fstd98.c:2270-2440) composed from the oracle's unpackers; works on a copy of the data part"""
    O = top.O(); O32 = ta32.O()
    n = ni * nj * nk
    d = data.copy()
    out = np.zeros(n, dtype)
    tag = np.array([99999.0], np.float64)
    hs = 3
    if datyp == 0:
        out.view(np.uint32)[:(n * nbits + 31) // 32] = d[:(n * nbits + 31) // 32]
    elif datyp == 1:
        assert O.orc_compact_float(out.ctypes.data, d.ctypes.data, d[3:].ctypes.data, n, nbits, 24, 1, 2, 0, tag.ctypes.data)
    elif datyp == 129:
        O.orc_armn_compress(d[5:].ctypes.data, ni, nj, nk, nbits, 2)
        assert O.orc_compact_float(out.ctypes.data, d[1:].ctypes.data, d[5:].ctypes.data, n, nbits + 64 * max(16, nbits), 0, 1, 2, 0, tag.ctypes.data)
    elif datyp in (2, 4):
        O.orc_compact_integer(out.ctypes.data, None, d.ctypes.data, n, nbits, 0, 1, 2 if datyp == 2 else 4)
    elif datyp == 130:
        tok = np.zeros(n, np.uint16)
        assert O.orc_armn_decode(tok.ctypes.data, d[1:].ctypes.data, ni, nj) == 0
        out[:] = (tok.astype(np.int16) & np.int16(-1 if nbits >= 16 else (1 << nbits) - 1)).astype(np.int32)
    elif datyp == 5:
        out.view(np.uint32)[:] = d[:n]
    elif datyp == 133:
        O32.orc_armn_uncompress32(out.ctypes.data, d[1:].ctypes.data, ni, nj, nk, nbits)
    elif datyp == 6:
        nb = ctypes.c_int(0)
        O.orc_float_unpacker(out.ctypes.data, d.ctypes.data, d[hs:].ctypes.data, n, ctypes.byref(nb))
    elif datyp == 134:
        nb = ctypes.c_int(0)
        O.orc_armn_compress(d[1 + hs:].ctypes.data, ni, nj, nk, nbits, 2)
        O.orc_float_unpacker(out.ctypes.data, d[1:].ctypes.data, d[1 + hs:].ctypes.data, n, ctypes.byref(nb))
    else:
        raise ValueError(datyp)
    return out
