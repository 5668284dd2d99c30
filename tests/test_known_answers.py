"""Known-answer vectors DERIVED BY HAND from the reference's source text (not produced by any code of this repository), for the
pieces whose reference files cannot be compiled here (`use app` / `#include <App.h>`: the un-vendored App submodule):
  * the grid-descriptor codecs CIGAXG / CXGAIG (src/base/igaxg.f90:51-135, xgaig.f90:57-192), which oracle/_ref links from
    oracle/orc_igaxg.c;
  * one stream per branch of the packers (compact.tmplc header styles and missing-value token, compact_integer.c header /
    signed bias, float_packer.c rounding, c_zfstlib.c parallelogram edges).
Every expected value below carries its derivation.  They pin the CPU oracle here and, in the GPU suite
(tests/test_gpu_packers.py::test_known_answers_on_the_gpu), the HIP path through the C ABI."""
import ctypes
import numpy as np
import pytest

import oraclelib as ol
import test_oracle_packers as top

# ---------------------------------------------------------------------------------------------------------------
# CXGAIG: real -> integer descriptors.  NINT = round half away from zero on REAL*4 products.
# ---------------------------------------------------------------------------------------------------------------
CXGAIG = [
    # 'L' (xgaig.f90:129-139): ig1 = nint(dlat*100), ig2 = nint(dlon*100), ig3 = nint((90+lat0)*100), ig4 = nint(lon0*100) (+36000 while < 0)
    ("L", (-90.0, 0.0, 0.05, 0.05), (5, 5, 0, 0)),                 # cfg2's target: 5, 5, (90-90)*100, 0
    ("L", (-89.55, 0.0, 0.9, 0.9), (90, 90, 45, 0)),               # cfg1's source: 90, 90, nint(0.45*100 = 45.0003..) = 45
    ("L", (45.5, -10.0, 0.25, 0.5), (25, 50, 13550, 35000)),       # lon0 < 0: -1000 + 36000
    ("L", (-90.0, 359.99, 1.0, 1.0), (100, 100, 0, 35999)),
    # 'E' (xgaig.f90:141-168): j1 = nint((xlat1+90)*40), j2 = nint(xlat2*40) (+7201 if < 0), j3 = nint((xlon1'+90)*40) (-16384 if >= 16384),
    # j4 = nint(xlon2'*40), lon' = lon + 360 while < 0; ig1 = j1>>2, ig3 = j3<<2 | (j1&3), ig2 = j2>>2, ig4 = j4<<2 | (j2&3)
    ("E", (31.0, -90.0, -20.0, -20.0), (1210, 1600, 57600, 54401)),   # j1 = 4840, j2 = -800 + 7201 = 6401, j3 = 360*40 = 14400, j4 = 340*40 = 13600
    ("E", (0.0, 180.0, 0.0, 270.0), (900, 0, 43200, 43200)),          # Yin: j1 = 3600, j2 = 0, j3 = 10800, j4 = 10800
    ("E", (0.0, 0.0, 90.0, 0.0), (900, 900, 14400, 0)),               # Yang: j1 = 3600, j2 = 3600, j3 = 3600, j4 = 0
    ("E", (-45.25, 350.0, 10.1, 10.0), (447, 101, 4866, 1600)),       # j1 = nint(44.75*40 = 1790) -> 447 r 2; j2 = 404 -> 101 r 0; j3 = 17600 - 16384 = 1216 -> 4864 | 2; j4 = 400 -> 1600
    # 'N' / 'S', old style (xgaig.f90:57-66): ig1 = nint(pj*10), ig2 = nint(pi*10), ig3 = nint(dgrw*100) (+36000 while < 0), ig4 = nint(d60*.01)
    ("N", (50.5, 45.5, 100000.0, 21.0), (455, 505, 2100, 1000)),
    ("S", (40.0, 60.0, 150000.0, -10.0), (600, 400, 35000, 1500)),
    # 'A' / 'B' / 'G' (xgaig.f90:98-104): truncation of xg1, xg2
    ("G", (0.0, 1.0, 0.0, 0.0), (0, 1, 0, 0)),
    ("A", (2.0, 0.0, 0.0, 0.0), (2, 0, 0, 0)),
]

# CIGAXG: integer -> real descriptors
CIGAXG = [
    # 'L' (igaxg.f90:93-97): xg1 = ig3*0.01 - 90, xg2 = ig4*0.01, xg3 = ig1*0.01, xg4 = ig2*0.01, all REAL*4
    ("L", (5, 5, 0, 0), (np.float32(0) * np.float32(0.01) - np.float32(90), np.float32(0), np.float32(5) * np.float32(0.01), np.float32(5) * np.float32(0.01))),
    ("L", (90, 90, 45, 0), (np.float32(45) * np.float32(0.01) - np.float32(90), np.float32(0), np.float32(90) * np.float32(0.01), np.float32(90) * np.float32(0.01))),
    ("L", (25, 50, 13550, 35000), (np.float32(13550) * np.float32(0.01) - np.float32(90), np.float32(35000) * np.float32(0.01), np.float32(0.25), np.float32(0.5))),
    # 'E' (igaxg.f90:106-122): lg1 = ig1<<2 | (ig3&3), lg3 = ig3>>2 (+16384 if < 3559), lg2 = ig2<<2 | (ig4&3) (-7201 if > 3600), lg4 = ig4>>2;
    # xg1 = (lg1-3600)/40, xg2 = (lg3-3600)/40, xg3 = lg2/40, xg4 = lg4/40 (REAL*8, rounded to REAL*4)
    ("E", (1210, 1600, 57600, 54401), (31.0, 270.0, -20.0, 340.0)),      # lg1 = 4840, lg3 = 14400, lg2 = 6401 - 7201 = -800, lg4 = 13600
    ("E", (900, 0, 43200, 43200), (0.0, 180.0, 0.0, 270.0)),
    ("E", (447, 101, 4866, 1600), (-45.25, 350.0, 10.1, 10.0)),          # lg1 = 1790, lg3 = 1216 + 16384 = 17600, lg2 = 404, lg4 = 400
    # 'N' / 'S' old style (igaxg.f90:53-58): xg1 = ig2*0.1, xg2 = ig1*0.1, xg3 = ig4*100, xg4 = ig3*0.01
    ("N", (455, 505, 2100, 1000), (np.float32(505) * np.float32(0.1), np.float32(455) * np.float32(0.1), 100000.0, np.float32(2100) * np.float32(0.01))),
    ("S", (600, 400, 35000, 1500), (40.0, 60.0, 150000.0, np.float32(35000) * np.float32(0.01))),
    ("G", (0, 1, 0, 0), (0.0, 1.0, 0.0, 0.0)),
]


@pytest.mark.parametrize("grtyp,xg,want", CXGAIG)
def test_cxgaig_known_answers(grtyp, xg, want):
    assert ol.cxgaig(grtyp, *[float(np.float32(v)) for v in xg]) == want


@pytest.mark.parametrize("grtyp,ig,want", CIGAXG)
def test_cigaxg_known_answers(grtyp, ig, want):
    x = [ctypes.c_float() for _ in range(4)]
    ol.oracle().orc_cigaxg(grtyp.encode(), *[ctypes.byref(v) for v in x], *ig)
    got = np.array([v.value for v in x], np.float32)
    assert np.array_equal(got.view(np.uint32), np.array(want, np.float32).view(np.uint32)), (got, want)


def test_descriptor_codecs_reach_the_product_unchanged():
    """the product's own copies (ez_host.c h_cigaxg / h_cxgaig) through the host-only C ABI: c_ezqkdef + c_gdll of 'L' grids whose
    corner coordinates follow from the hand-decoded descriptors above (xg1 + (nj-1) xg3 etc., REAL*4 arithmetic of grll.f:57-64)"""
    from librmn_amd import ezscint as ez
    for ig, (ni, nj) in (((5, 5, 0, 0), (7200, 3601)), ((90, 90, 45, 0), (400, 200)), ((25, 50, 13550, 35000), (30, 20))):
        gd = ez.ezqkdef(ni, nj, "L", *ig)
        assert gd >= 0
        rc, lat, lon = ez.gdll(gd, ni * nj)
        assert rc == 0
        f = np.float32
        lat0 = f(ig[2]) * f(0.01) - f(90); lon0 = f(ig[3]) * f(0.01); dlat = f(ig[0]) * f(0.01); dlon = f(ig[1]) * f(0.01)
        assert lat[0] == lat0 and lat[-1] == f(lat0 + f(nj - 1) * dlat)
        assert lon[0] == np.fmod(lon0, f(360)) and lon[ni - 1] == np.fmod(f(lon0 + f(ni - 1) * dlon), f(360))


# ---------------------------------------------------------------------------------------------------------------
# packers: one hand-computed stream per branch.  PACK_VECTORS is shared with the GPU suite.
# ---------------------------------------------------------------------------------------------------------------
def _bits(fields):
    """MSB-first concatenation of (value, width) pairs -> list of 32-bit words (zero padded): the `stuff` macro of bitPacking.h:126-139"""
    acc = 0; n = 0
    for v, w in fields:
        assert 0 <= v < (1 << w)
        acc = (acc << w) | v; n += w
    pad = (-n) % 32
    acc <<= pad; n += pad
    return [(acc >> (n - 32 * (k + 1))) & 0xFFFFFFFF for k in range(n // 32)]


PACK_VECTORS = {
    # compact_float([1, 2, 3, 4], 4 bits), header style 2 (compact.tmplc:143-334):
    #   min 1, max 4; range = (4 - 1) * 2 = 6 with the mantissa zeroed = 4 = 2^2 -> tempExpo 2; scaledExpOfRange = 2 - 4 = -2 -> + 4096 = 0xFFE;
    #   min = 1.0: double exponent 1023 -> scaledExpOfMinFloat = 1023 - 1023 + 1024 - 48 = 976 = 0x3D0, sign 0 -> h[1] = 0x0FFE3D00;
    #   h[0] = 0xfff << 20 | 4; h[2] = hidden bit | 0 mantissa = 0x80000000; h[3] = 4 << 8;
    #   mulFactor = 2^4 / 2^2 = 4: tokens 0, 4, 8, 12 -> 0x048C....
    "cf_style2": dict(a=[1.0, 2.0, 3.0, 4.0], nbits=4, hdr=[0xFFF00004, 0x0FFE3D00, 0x80000000, 0x00000400], stream=[0x048C0000]),
    # the same values, header style 1 (stream starts at bit 24 of header word 3, :159-168): type 0x7ff; the first two tokens
    # (0, 4) overwrite the low byte of h[3]: 0x00000404; the stream continues with 8, C
    "cf_style1": dict(a=[1.0, 2.0, 3.0, 4.0], nbits=4, style1=True, image=[0x7FF00004, 0x0FFE3D00, 0x80000000, 0x00000404, 0x8C000000]),
    # missing value (:187-204, :218-226): [1, 2, tag, 4]: min / max skip the tag; tempInt = 3 * 16 / 4 = 12 != missingToken (15): no bump;
    # tokens 0, 4, F, C; header type 0xfef
    "cf_missing": dict(a=[1.0, 2.0, -999.0, 4.0], nbits=4, tag=-999.0, hdr=[0xFEF00004, 0x0FFE3D00, 0x80000000, 0x00000400], stream=[0x04FC0000]),
    # the bump (:223-225): [0, 1.875, tag]: range = 3.75 -> 2 (2^1); tempInt = 1.875 * 16 / 2 = 15 == missingToken -> exponent + 1: tempExpo 2,
    # scaledExpOfRange = -2 -> 0xFFE; min = 0: scaledExpOfMinFloat = (0 - 1023 + 1024 - 48) & 0x111 = 0xFFFFFFD1 & 0x111 = 0x111 (sic, :240-242),
    # h[1] = 0xFFE << 16 | (0x111 << 4 | 0) = 0x0FFE1110; h[2] = 0; mulFactor 4: tokens 0, int(7.5) = 7, F
    "cf_bump": dict(a=[0.0, 1.875, -999.0], nbits=4, tag=-999.0, hdr=[0xFEF00003, 0x0FFE1110, 0x00000000, 0x00000400], stream=[0x07F00000]),
    # 32-bit tokens, offset 0 (the direct-copy branch, compact.tmplc:302-312): range 4 = 2^2 -> tempExpo 2; scaledExpOfRange = 2 - 32 = -30 -> + 4096 = 0xFE2;
    #   h[1] = 0xFE2 << 16 | 0x3D0 << 4 = 0x0FE23D00; h[3] = 32 << 8; mulFactor = 2^32 / 2^2 = 2^30: tokens (x - 1) * 2^30, one per word, no shifting
    "cf_32bits": dict(a=[1.0, 2.0, 3.0, 4.0], nbits=32, hdr=[0xFFF00004, 0x0FE23D00, 0x80000000, 0x00002000], stream=[0x00000000, 0x40000000, 0x80000000, 0xC0000000]),
    # more than 16 bits per token (20): scaledExpOfRange = 2 - 20 = -18 -> 0xFEE; h[3] = 20 << 8; mulFactor = 2^20 / 4 = 2^18: tokens 0, 0x40000, 0x80000, 0xC0000
    #   in 20 bits each, MSB first: 00000 40000 80000 C0000 -> 0x00000400 0x0080000C 0x0000....
    "cf_20bits": dict(a=[1.0, 2.0, 3.0, 4.0], nbits=20, hdr=[0xFFF00004, 0x0FEE3D00, 0x80000000, 0x00001400], stream=[0x00000400, 0x0080000C, 0x00000000]),
    # compact_integer (compact_integer.c:325-570), unsigned, no header, 4 bits: tokens 1, 2, 3
    "ci_plain": dict(a=[1, 2, 3], nbits=4, op=1, stream=[0x12300000], rc=4),
    # with header (constructHeader :186-231), nbits = -1: min 1, max 3, bitRequired = bitlen(2) = 2, shift 0:
    # h = {0xFD000000 | 0 << 6 | 2, n = 3, min = 1, max = 3}; tokens (x - 1): 00 01 10 -> 0b000110 << 26
    "ci_header": dict(a=[1, 2, 3], nbits=-1, op=1, header=True, hdr=[0xFD000002, 3, 1, 3], stream=[0x18000000], rc=2),
    # signed, no header, 4 bits (:530): positiveMask = 1 << 3: tokens (x + 8) & 15 = 7, 8, 9
    "ci_signed": dict(a=[-1, 0, 1], nbits=4, op=3, stream=[0x78900000], rc=4),
    # header with fewer bits than the range needs: [0, 100], nbits 4: bitRequired = bitlen(100) = 7 -> shift 3: h[0] = 0xFD000000 | 3 << 6 | 4;
    # tokens (x >> 3) & 15: 0, 12
    "ci_shift": dict(a=[0, 100], nbits=4, op=1, header=True, hdr=[0xFD0000C4, 2, 0, 100], stream=[0x0C000000], rc=4),
    # c_float_packer([1, 2, 3], 8 bits) (float_packer.c:128-202): MaxExp = 128 (3.0); aligned mantissas 1.0 -> 0x800000 >> 1 = 0x400000 (= Minimum),
    # 2.0 -> 0x800000, 3.0 -> 0xC00000; range 0x800000: smallest shift with range >> s <= 255 is 16 (0x80), Round = 0x8000;
    # tokens (m - Min + Round) >> 16 = 0, 0x40, 0x80; h[0] = 0xEFF << 20 | (8 - 1) << 16 | 128 << 8 | 16
    "fp_round": dict(a=[1.0, 2.0, 3.0], nbits=8, hdr=[0xEFF78010, 0x00400000, 3], stream=[0x00000040, 0x00800000]),
    # both signs, 16 bits: [-1, 2]: MaxExp 128 (2.0); Maximum = 0x800000; Minimum: 1.0's mantissa 0x800000 >> (128 - 127) = 0x400000, negated (:160-168);
    # range 0xC00000: >> 8 = 0xC000 <= 0xFFFF (>> 7 = 0x18000 is not): Shift2 8, Round 0x80; tokens (-0x400000 + 0x400000 + 0x80) >> 8 = 0 and
    # (0x800000 + 0x400000 + 0x80) >> 8 = 0xC000, two per word; h[0] = 0xEFF << 20 | 15 << 16 | 128 << 8 | 8; h[1] = Minimum = -0x400000
    "fp_signed": dict(a=[-1.0, 2.0], nbits=16, hdr=[0xEFFF8008, 0xFFC00000, 2], stream=[0x0000C000]),
}


def armn_constant_16x16(v):
    """armn_compress of a constant 16 x 16 field of 16-bit tokens (c_zfstlib.c:646-789): every Lorenzo difference is 0, so the stream is the
    _fstzip word (predictor 4 | degree 1 << 4 | step 3 << 7 | nbits 16 << 10 | levels 1 << 15, zfstlib.h:49), the 3-bit container size (4),
    row 1 (16 tokens), column 1 rows 2..16 (15 tokens), 5 x 5 = 25 tiles of one 4-bit zero, and 32 terminator bits that complete the word"""
    words = [4 | 1 << 4 | 3 << 7 | 16 << 10 | 1 << 15] + _bits([(4, 3)] + [(v, 16)] * 31 + [(0, 4)] * 25)
    nbits = 3 + 31 * 16 + 25 * 4                        # 599 bits -> 19 words; zlng = 1 + 4 * (1 + 19)
    return words, 1 + 4 * (1 + (nbits + 31) // 32)


def armn_one_step_16x16():
    """a field that is 0 except u(i >= 2, j >= 2) = 5 (1-based): the only non-zero Lorenzo difference is d(2,2) = 5 - (0 + 0 - 0) = 5, in the first
    tile: width field = bitlen(5) = 3, its 9 differences in 3 + 1 = 4 bits each (5, then eight zeros); the other 24 tiles are 4-bit zeros"""
    u = np.zeros((16, 16), np.uint16); u[1:, 1:] = 5
    row1 = [(0, 16)] * 16; col1 = [(0, 16)] * 15
    tiles = [(3, 4)] + [(5, 4)] + [(0, 4)] * 8 + [(0, 4)] * 24
    words = [4 | 1 << 4 | 3 << 7 | 16 << 10 | 1 << 15] + _bits([(4, 3)] + row1 + col1 + tiles)
    nbits = 3 + 31 * 16 + 4 + 9 * 4 + 24 * 4
    return u.reshape(-1), words, 1 + 4 * (1 + (nbits + 31) // 32)


ARMN_HEADER_PG = 4 | 1 << 4 | 3 << 7 | 16 << 10 | 1 << 15        # PARALLELOGRAM, degree 1, step 3, nbits 16, levels 1 (zfstlib.h:49, c_zfstlib.c:222-236)
ARMN_HEADER_MIN = 3 | 0 << 4 | 5 << 7 | 16 << 10 | 1 << 15       # MINIMUM, degree 0, step 5 (c_zfstlib.c:132: c_fstzip(..., MINIMUM, 0, 5, nbits, 0))


def armn_wide_difference_16x16():
    """the 5-bit width field (c_zfstlib.c:701-711, :736-737, :758-766).  u(2,1) = u(1,2) = 65535, everything else 0 (1-based (i, j), i fastest):
       d(2,2) = 0 - (u(2,1) + u(1,2) - u(1,1)) = -131070: |d| > 65535 with nbits >= 15 -> the container of the width fields is 5 bits;
       d(3,2) = 0 - (u(3,1) + u(2,2) - u(2,1)) = +65535, d(2,3) = 0 - (u(2,2) + u(1,3) - u(1,2)) = +65535, d(3,3) = 0 - (0 + 0 - 0) = 0, the rest 0.
       First tile (i, j = 2..4): local_max = 131070 > 65535 -> nbits_needed = (int)(1 + log(131070.5) / log(2)) = (int)17.99999 = 17 (not 16: no 17-bit escape),
       its 9 differences in 17 + 1 = 18 bits, tile rows outer: -131070 & 0x3FFFF = 0x20002, 0xFFFF, 0 | 0xFFFF, 0, 0 | 0, 0, 0.  The other 24 tiles: 5-bit zeros."""
    u = np.zeros((16, 16), np.uint16); u[0, 1] = 65535; u[1, 0] = 65535            # [j - 1, i - 1]
    row1 = [(int(v), 16) for v in u[0, :]]; col1 = [(int(v), 16) for v in u[1:, 0]]
    tile1 = [(17, 5)] + [(t, 18) for t in (0x20002, 0xFFFF, 0, 0xFFFF, 0, 0, 0, 0, 0)]
    fields = [(5, 3)] + row1 + col1 + tile1 + [(0, 5)] * 24
    nbits = sum(w for _, w in fields)                        # 3 + 31 * 16 + 5 + 9 * 18 + 24 * 5 = 786
    assert nbits == 786
    return u.reshape(-1), [ARMN_HEADER_PG] + _bits(fields), 1 + 4 * (1 + (nbits + 31) // 32)


def armn_minimum_15x5():
    """MINIMUM (packTokensMinimum, c_zfstlib.c:469-588; taken because nj < 16, :132): tiles of 5 x 5 from (1, 1), a 4-bit width per tile.
       ni = 15, nj = 5: three tiles side by side.
       tile 1 (i = 1..5): values 0 everywhere but one 40000: max - min = 40000 >= 256 -> 8 + fastlog[40000 >> 8 = 156] = 8 + (int)(1 + log2(156.5) = 8.29) = 16 -> 15
               (the escape): width 15, then the 25 values RAW in 16 bits each, no minimum;
       tile 2 (i = 6..10): constant 0x1234: width 0, then the minimum in nbits = 16 bits;
       tile 3 (i = 11..15): 100 everywhere but one 103: max - min = 3 -> fastlog[3] = (int)(1 + log2(3.5) = 2.81) = 2: width 2, the minimum (16 bits), 25 x 2 bits."""
    u = np.zeros((5, 15), np.uint16)
    u[:, 0:5] = 0; u[2, 3] = 40000
    u[:, 5:10] = 0x1234
    u[:, 10:15] = 100; u[4, 14] = 103
    t1 = [(15, 4)] + [(int(u[n, m]), 16) for n in range(5) for m in range(0, 5)]
    t2 = [(0, 4), (0x1234, 16)]
    t3 = [(2, 4), (100, 16)] + [(int(u[n, m]) - 100, 2) for n in range(5) for m in range(10, 15)]
    fields = t1 + t2 + t3
    nbits = sum(w for _, w in fields)                        # 404 + 20 + 70 = 494
    assert nbits == 494
    return u.reshape(-1), [ARMN_HEADER_MIN] + _bits(fields), 1 + 4 * (1 + (nbits + 31) // 32)


def armn32_step_16x16(with_mantissa):
    """c_armn_compress32 (armn_compress_32.c:59-275) of a 16 x 16 REAL field, znbits = 32 (23 mantissa bits kept): 1.0 on the first row and column,
       2.0 elsewhere (and 3.0 at (2, 2) when with_mantissa).  All positive -> no sign stream (code 0x00); exponents 127 / 128: exp_base 127, range 1 ->
       nbits_needed = ((float)1 exponent field 127) - 126 = 1, code 0x08.
       word 0: PARALLELOGRAM32 (5) | degree 1 << 4 | step 3 << 7 | nbits 23 << 10 | levels 1 << 15 | version 2 << 18 = 0x0008DD95 (:96-101, zfstlib.h:49)
       word 1: exp_base << 16 | nbits_needed << 8 | codes = 0x007F0108 (:252-256)
       word 2: the exponent plane's byte count rounded up to 4 (:196-199); then the plane (packTokensParallelogram_8, :642-750): width container 4 in istep = 3
               bits, row 1 (16 x 1 bit, zeros), column 1 (15 x 1 bit, zeros); d(2,2) = 1 - (0 + 0 - 0) = 1, d(3,2) = 1 - (0 + 1 - 0) = 0, d(2,3) = 0, the rest 0:
               first tile width fastlog[1] = (int)(1 + log2(1.5)) = 1 in 4 bits, its 9 differences in 2 bits (1, then zeros); 24 tiles of 4-bit zeros: 152 bits ->
               5 words flushed -> 1 + 4 * 5 = 21 -> 24 bytes (the sixth word is never written: undefined)
       then the slot in front of the mantissa plane, which receives lng_signe = 0 (sic, :244), and the plane (packTokensParallelogram32, :456-565): container 5 in
               3 bits, row 1 and column 1 (31 x 23 bits), 25 tiles with 5-bit widths.  Mantissas all zero: 841 bits -> 27 words, first 0xA0000000.
               with 3.0 at (2, 2): mantissa 0x400000 there: d(2,2) = +X, d(3,2) = -X, d(2,3) = -X, d(3,3) = +X (X = 2^22): width = exponent field of (float)X - 126 =
               149 - 126 = 23, differences in 24 bits: X, -X & 0xFFFFFF = 0xC00000, 0 | 0xC00000, X, 0 | 0, 0, 0 -> 1057 bits -> 34 words.
       returns (field, [(first word index, words)...], zlng)"""
    f = np.full((16, 16), 2.0, np.float32); f[0, :] = 1.0; f[:, 0] = 1.0
    if with_mantissa:
        f[1, 1] = 3.0
    expo = _bits([(4, 3)] + [(0, 1)] * 31 + [(1, 4), (1, 2)] + [(0, 2)] * 8 + [(0, 4)] * 24)
    assert len(expo) == 5
    X = 1 << 22
    m11 = X if with_mantissa else 0
    mrow1 = [(0, 23)] * 16; mcol1 = [(0, 23)] * 15
    tile1 = ([(23, 5)] + [(t, 24) for t in (X, 0xC00000, 0, 0xC00000, X, 0, 0, 0, 0)]) if with_mantissa else [(0, 5)]
    mfields = [(5, 3)] + mrow1 + mcol1 + tile1 + [(0, 5)] * 24
    mbits = sum(w for _, w in mfields)
    assert mbits == (1057 if with_mantissa else 841) and m11 in (0, X)
    mant = _bits(mfields)
    nm = (mbits + 31) // 32                                  # words flushed by the terminator
    lng_m = 1 + 4 * nm; lng_m += (4 - lng_m % 4) % 4
    pieces = [(0, [0x0008DD95, 0x007F0108, 24]), (3, expo), (9, [0]), (10, mant[:nm])]
    return f.reshape(-1), pieces, 40 + lng_m


def armn32_signed(ni, nj):
    """the sign sub-stream of c_armn_compress32 (pack1bitRLE, armn_compress_32.c:827-905; code DIFF_SIGNE_PACKED 0x20, :171-176): the field of armn32_step_16x16
       (1.0 on the first row and column, 2.0 elsewhere) with its first 20 values negated.  Signs: a run of 20 ones, then ni * nj - 20 zeros.  Tokens are 8 bits:
       COUNT(1) | value | 6-bit count for runs of 8 .. 62 (:888-893); a longer run is cut into 62s (:862) -- and from the second piece on, while more than 256
       points remain, the byte 0xFF stands for 255 more points of the last value (:865-869); 32 zero bits end the stream (:902-903).
         16 x 16: 11|010100 (20 ones) = D4; 236 zeros = 62 + 62 + 62 + 50 (never more than 256 left): BE BE BE, 10|110010 = B2; + 32 zero bits = 72 bits: two words
                  flushed, lng = 1 + 4 * 2 = 9 -> 12 bytes (:183-185)
         32 x 32: D4; 1004 zeros = 62 (BE), then 942, 687, 432 left: FF FF FF (255 each), then 177 = 62 + 62 + 53: BE BE, 10|110101 = B5; + 32 zero bits = 96 bits.  `stuff` writes a word when the NEXT token
                  does not fit (bitPacking.h:126-139): the third, exactly full word is never flushed: two words, lng = 9 -> 12 bytes as well
       Layout (:160-245): [w0][w1: codes 0x20 | 0x08][lng_signe][sign words][lng_exposant][exponent plane][slot := lng_signe (sic, :244)][mantissa plane].  The planes
       are armn32_step_16x16's with ni + nj - 1 prefix entries and ((ni + 1) / 3) x ((nj + 1) / 3) tiles; mantissas all zero."""
    f = np.full((nj, ni), 2.0, np.float32); f[0, :] = 1.0; f[:, 0] = 1.0
    f = f.reshape(-1); f[:20] *= -1.0
    ntiles = ((ni - 1 + 2) // 3) * ((nj - 1 + 2) // 3)
    if (ni, nj) == (16, 16):
        sign = [0xD4BEBEBE, 0xB2000000]; lng_s = 12
    else:
        assert (ni, nj) == (32, 32)
        sign = [0xD4BEFFFF, 0xFFBEBEB5]; lng_s = 12
    efields = [(4, 3)] + [(0, 1)] * (ni + nj - 1) + [(1, 4), (1, 2)] + [(0, 2)] * 8 + [(0, 4)] * (ntiles - 1)
    ebits = sum(w for _, w in efields); ne = (ebits + 31) // 32
    lng_e = 1 + 4 * ne; lng_e += (4 - lng_e % 4) % 4
    mfields = [(5, 3)] + [(0, 23)] * (ni + nj - 1) + [(0, 5)] * ntiles
    mbits = sum(w for _, w in mfields); nm = (mbits + 31) // 32
    lng_m = 1 + 4 * nm; lng_m += (4 - lng_m % 4) % 4
    w_e = 3 + lng_s // 4                                     # the exponent plane's length word
    w_m = w_e + 1 + lng_e // 4                               # the slot in front of the mantissa plane
    pieces = [(0, [0x0008DD95, 0x007F0128, lng_s]), (3, sign), (w_e, [lng_e]), (w_e + 1, _bits(efields)[:ne]), (w_m, [lng_s]), (w_m + 1, _bits(mfields)[:nm])]
    return f, pieces, 4 * (w_m + 1) + lng_m


def check_pack_vector(name, v, cf_pack, ci_pack, fp_pack):
    """runs one vector through the three packer callables (oracle or HIP path) and compares with the hand-computed words"""
    if name.startswith("cf_"):
        a = np.array(v["a"], np.float32)
        img = cf_pack(a, v["nbits"], style1=v.get("style1", False), has_missing=1 if "tag" in v else 0, tag=v.get("tag", 0.0))
        if v.get("style1"):
            assert [int(x) for x in img[:5]] == v["image"], (name, [hex(int(x)) for x in img[:5]])
        else:
            assert [int(x) for x in img[:4]] == v["hdr"], (name, [hex(int(x)) for x in img[:4]])
            assert [int(x) for x in img[4:4 + len(v["stream"])]] == v["stream"], (name, [hex(int(x)) for x in img[4:6]])
    elif name.startswith("ci_"):
        a = np.array(v["a"], np.int32 if v["op"] == 3 else np.uint32)
        rc, hdr, out = ci_pack(a, v["nbits"], v["op"], v.get("header", False))
        assert rc == v["rc"], (name, rc)
        if v.get("header"):
            assert [int(x) for x in hdr] == v["hdr"], (name, [hex(int(x)) for x in hdr])
        assert [int(x) for x in out[:len(v["stream"])]] == v["stream"], (name, [hex(int(x)) for x in out[:2]])
    else:
        rc, hdr, st = fp_pack(np.array(v["a"], np.float32), v["nbits"])
        assert rc == 0 and [int(x) & 0xFFFFFFFF for x in hdr] == v["hdr"] and [int(x) & 0xFFFFFFFF for x in st] == v["stream"], (name, hdr, st)


def _orc_cf(a, nbits, style1=False, has_missing=0, tag=0.0):
    return top.pack_float(a, nbits, style2=not style1, has_missing=has_missing, tag=tag)


def _orc_ci(a, nbits, op, header):
    hdr = np.zeros(4, np.uint32); out = np.zeros(8, np.uint32)
    rc = top.O().orc_compact_integer(a.ctypes.data, hdr.ctypes.data if header else None, out.ctypes.data, a.size, nbits, 0, 1, op)
    return rc, hdr, out


def _orc_fp(a, nbits):
    hdr = np.zeros(3, np.int32); st = np.zeros((a.size + 1) // 2, np.int32)
    rc = top.O().orc_float_packer(a.ctypes.data, nbits, hdr.ctypes.data, st.ctypes.data, a.size)
    return rc, hdr, st


@pytest.mark.parametrize("name", sorted(PACK_VECTORS))
def test_packer_known_answers_oracle(name):
    check_pack_vector(name, PACK_VECTORS[name], _orc_cf, _orc_ci, _orc_fp)


def test_armn_known_answers_oracle():
    import packcases as pc
    O = top.O()
    for v in (0x1234, 0, 0xFFFF):
        words, zlng = armn_constant_16x16(v)
        z = np.zeros(16 * 16 + 64, np.uint32)
        tok = np.full(256, v, np.uint16)
        assert O.orc_armn_encode(z.ctypes.data, tok.ctypes.data, 16, 16, 16) == zlng == 81
        assert [int(x) for x in z[:len(words)]] == words
    tok, words, zlng = armn_one_step_16x16()
    z = np.zeros(16 * 16 + 64, np.uint32)
    assert O.orc_armn_encode(z.ctypes.data, tok.ctypes.data, 16, 16, 16) == zlng
    assert [int(x) for x in z[:len(words)]] == words
    tok, words, zlng = armn_wide_difference_16x16()
    z = np.zeros(16 * 16 + 64, np.uint32)
    assert O.orc_armn_encode(z.ctypes.data, tok.ctypes.data, 16, 16, 16) == zlng == 105
    assert [int(x) for x in z[:len(words)]] == words
    tok, words, zlng = armn_minimum_15x5()
    z = np.zeros(15 * 5 + 64, np.uint32)
    assert O.orc_armn_encode(z.ctypes.data, tok.ctypes.data, 15, 5, 16) == zlng == 69
    assert [int(x) for x in z[:len(words)]] == words


@pytest.mark.parametrize("with_mantissa", [False, True])
def test_armn32_known_answers_oracle(with_mantissa):
    import test_oracle_armn32 as ta32
    f, pieces, zlng = armn32_step_16x16(with_mantissa)
    z = np.zeros(16 * 16 * 2 + 64, np.uint32)
    assert ta32.O().orc_armn_compress32(z.ctypes.data, f.ctypes.data, 16, 16, 1, 32) == zlng == (180 if with_mantissa else 152)
    for w0, words in pieces:
        assert [int(x) for x in z[w0:w0 + len(words)]] == words, (w0, [hex(int(x)) for x in z[w0:w0 + len(words)]], [hex(x) for x in words])


@pytest.mark.parametrize("ni,nj,want", [(16, 16, 168), (32, 32, 372)])
def test_armn32_sign_runs_known_answers_oracle(ni, nj, want):
    import test_oracle_armn32 as ta32
    f, pieces, zlng = armn32_signed(ni, nj)
    z = np.zeros(ni * nj * 2 + 64, np.uint32)
    assert ta32.O().orc_armn_compress32(z.ctypes.data, f.ctypes.data, ni, nj, 1, 32) == zlng == want
    for w0, words in pieces:
        assert [int(x) for x in z[w0:w0 + len(words)]] == words, (w0, [hex(int(x)) for x in z[w0:w0 + len(words)]], [hex(x) for x in words])
    back = np.zeros(ni * nj, np.float32)
    assert ta32.O().orc_armn_uncompress32(back.ctypes.data, z.ctypes.data, ni, nj, 1, 32) == ni * nj
    assert np.array_equal(back.view(np.uint32), f.view(np.uint32))


def ci_offset_stride():
    """compact_integer with a bit offset and a stride (compact_integer.c:136-170, :104-125): a = [1, 99, 2, 99, 3], stride 2 -> the tokens 1, 2, 3 in 4 bits behind the
       first 8 bits of a word that holds 0xAB000FFF: lastWordShifted = 0xAB, then 0xAB123 with 12 bits of space left; the squeeze at the end (:165-169) keeps the
       word's last 12 bits: 0xAB123FFF.  Returns 4 (the token size).  Unpacking from bit 8 with stride 2 writes elements 0, 2, 4 and leaves 1, 3 alone."""
    return np.array([1, 99, 2, 99, 3], np.uint32), 0xAB000FFF, 0xAB123FFF


def test_ci_offset_stride_known_answer_oracle():
    a, before, after = ci_offset_stride()
    out = np.array([before, 0x55555555], np.uint32)
    assert top.O().orc_compact_integer(a.ctypes.data, None, out.ctypes.data, 3, 4, 8, 2, 1) == 4
    assert int(out[0]) == after and int(out[1]) == 0x55555555
    back = np.full(5, 7, np.uint32)
    assert top.O().orc_compact_integer(back.ctypes.data, None, out.ctypes.data, 3, 4, 8, 2, 2) == 4
    assert [int(x) for x in back] == [1, 7, 2, 7, 3]


def fst_data_part_vectors():
    """the data part of an FST record as c_fstecr lays it out (fstd98.c:1160-1352), composed by hand from the vectors above:
         datyp 2, 4 bits, [1, 2, 3] (unsigned)      compact_integer(field, NULL, &data[0], n, nbits, 0, 1, 1) (:1236): the plain stream 0x123.....
         datyp 4, 4 bits, [-1, 0, 1] (signed)       compact_integer(..., 3) (:1288): tokens (x + 8) & 15 = 7 8 9
         datyp 6, 8 bits, [1., 2., 3.]              c_float_packer(field, nbits, &data[0], &data[header_size], n) with header_size 3 (:1347, float_packer.c:305):
                                                    the three header words, then the 16-bit tokens
         datyp 133, 32 bits, armn32_step_16x16      data[0] = the record's length in words: nbytes = 16 + zlng = 168 -> (168 * 8 + 63) / 64 = 21 pairs = 42 words
                                                    (:1317-1321), the record of c_armn_compress32 from data[1] on
       -> list of (datyp, nbits, field, ni, nj, [(first word, words) ...], words of the data part or None)"""
    f133, pieces, zlng = armn32_step_16x16(False)
    nw = 2 * ((8 * (16 + zlng) + 63) // 64)
    return [
        (2, 4, np.array([1, 2, 3], np.uint32), 3, 1, [(0, [0x12300000])], None),
        (4, 4, np.array([-1, 0, 1], np.int32), 3, 1, [(0, [0x78900000])], None),
        (6, 8, np.array([1.0, 2.0, 3.0], np.float32), 3, 1, [(0, [0xEFF78010, 0x00400000, 3, 0x00000040, 0x00800000])], None),
        (133, 32, f133, 16, 16, [(0, [nw])] + [(1 + w0, words) for w0, words in pieces], nw),
    ]


def test_fst_data_part_known_answers_twin():
    import fst_twin
    for datyp, nbits, f, ni, nj, pieces, nw in fst_data_part_vectors():
        words, d_out, data = fst_twin.pack(f, ni, nj, 1, datyp, nbits)
        assert d_out == datyp, (datyp, d_out)
        if nw is not None:
            assert words == nw == 42
        for w0, want in pieces:
            assert [int(x) & 0xFFFFFFFF for x in data[w0:w0 + len(want)]] == want, (datyp, w0, [hex(int(x)) for x in data[w0:w0 + len(want)]])
