"""CPU oracle of the IEEE-32 compressor (oracle/orc_armn32.c; reference src/compresseur/armn_compress_32.c, PARITY UNPINNED: the file
needs <App.h>).  Pins available here: compress -> uncompress round trips through the restated decoders, header words computed by
hand, stream-size bookkeeping."""
import ctypes
import numpy as np
import pytest

import oraclelib as ol
import ezcases as ec


def O():
    L = ol.oracle()
    L.orc_armn_compress32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.orc_armn_uncompress32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return L


def field32(ni, nj, kind, seed):
    base = ec.synth_field(ni, nj, seed=seed)                       # ~250..310, positive, one or two binades
    if kind == "positive":
        return base
    if kind == "negative":
        return (-base).astype(np.float32)
    if kind == "mixed":                                             # wind-like: both signs, many binades, exact zeros
        u, _ = ec.synth_wind(ni, nj, seed=seed)
        u[::97] = 0.0
        return u
    if kind == "one_binade":                                        # every exponent equal: no exponent stream
        return (np.float32(256.0) + np.float32(200.0) * ec.hash_uniform(seed, ni * nj)).astype(np.float32)
    if kind == "stripes":                                           # long runs of equal sign (RLE COUNT tokens and the 255-repeat token)
        z = base.reshape(nj, ni).copy(); z[nj // 3:2 * nj // 3] *= -1.0
        return np.ascontiguousarray(z.reshape(-1))
    if kind == "noise":                                             # incompressible mantissas
        return (ec.hash_uniform(seed, ni * nj) * np.float32(1000.0) - np.float32(500.0)).astype(np.float32)
    raise ValueError(kind)


def truncated(f, znbits):
    """what a round trip must return: sign and exponent kept, mantissa cut to znbits - 9 bits"""
    nb = znbits - 9
    u = f.view(np.uint32)
    keep = np.uint32(0xFFFFFFFF) if nb >= 23 else np.uint32((0xFFFFFFFF << (23 - nb)) & 0xFFFFFFFF)
    return (u & keep).view(np.float32)


SHAPES = [(16, 16), (17, 19), (64, 48), (100, 31), (301, 200)]
KINDS = ["positive", "negative", "mixed", "one_binade", "stripes"]


@pytest.mark.parametrize("ni,nj", SHAPES)
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("znbits", [32, 24, 16])
def test_round_trip(ni, nj, kind, znbits):
    f = field32(ni, nj, kind, seed=ni + nj)
    z = np.zeros(ni * nj + 1024, np.uint32)
    zlng = O().orc_armn_compress32(z.ctypes.data, f.ctypes.data, ni, nj, 1, znbits)
    if zlng < 0:
        pytest.skip("not compressible at this size")
    assert zlng % 4 == 0 and 8 < zlng <= ni * nj * znbits // 8 + 64
    back = np.zeros(ni * nj, np.float32)
    assert O().orc_armn_uncompress32(back.ctypes.data, z.ctypes.data, ni, nj, 1, znbits) == ni * nj
    assert np.array_equal(back.view(np.uint32), truncated(f, znbits).view(np.uint32)), (kind, int((back != truncated(f, znbits)).sum()))


def test_header_words_by_hand():
    """a positive field of one binade: no sign stream, no exponent stream.  _fstzip word = PARALLELOGRAM32 (5) | degree 1 << 4 | step 3 << 7 |
    nbits 23 << 10 | levels 1 << 15 | version 2 << 18 (zfstlib.h:49); zieee_info = exp_base << 16 | 0 << 8 | codes (0): 256..456 has biased exponent 135"""
    ni = nj = 32
    f = field32(ni, nj, "one_binade", seed=3)
    assert f.min() >= 256.0 and f.max() < 512.0
    z = np.zeros(ni * nj + 64, np.uint32)
    zlng = O().orc_armn_compress32(z.ctypes.data, f.ctypes.data, ni, nj, 1, 32)
    assert zlng > 0
    assert int(z[0]) == (5 | 1 << 4 | 3 << 7 | 23 << 10 | 1 << 15 | 2 << 18)
    assert int(z[1]) == 135 << 16
    assert int(z[2]) == 0                                           # the mantissa length slot receives lng_signe (= 0 here): armn_compress_32.c:237
    assert (int(z[3]) >> 29) == 5                                   # the mantissa stream opens with its 3-bit container size


def random_bits_field(ni, nj, seed):
    """random bit patterns (finite): random signs, exponents over many binades, random mantissas"""
    u = (ec.splitmix64(seed, ni * nj) >> np.uint64(32)).astype(np.uint32)
    u = (u & np.uint32(0x807FFFFF)) | (((u >> np.uint32(23)) % np.uint32(200) + np.uint32(20)) << np.uint32(23))
    return u.view(np.float32)


def test_incompressible_and_small_fields_are_refused():
    f = random_bits_field(64, 48, seed=9)
    z = np.zeros(2 * 64 * 48 + 1024, np.uint32)
    assert O().orc_armn_compress32(z.ctypes.data, f.ctypes.data, 64, 48, 1, 32) == -1      # exponent stream longer than ni * nj bytes (:200-204) or no room left for the mantissas (:546-549)
    g = field32(15, 40, "positive", seed=1)
    assert O().orc_armn_compress32(z.ctypes.data, g.ctypes.data, 15, 40, 1, 32) == -1      # ni < 16
