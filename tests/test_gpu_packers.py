"""GPU parity tests of the packers: HIP path (through the C ABI) vs the CPU oracle, BIT-EXACT
(every packed word, header words, return values)."""
import ctypes
import os
import numpy as np
import pytest

import oraclelib as ol
import packcases as pc
import test_oracle_packers as top

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
from librmn_amd import packers as pk   # noqa: E402


@pytest.mark.parametrize("nbits_arg", [1, 4, 8, 12, 15, 16, 17, 24, 31, 32, 16 + 64 * 16, 12 + 64 * 16])
@pytest.mark.parametrize("n,stride,offset", [(1, 1, 0), (2, 1, 0), (31, 1, 0), (33, 2, 5), (1000, 1, 0), (1000, 3, 37), (7200 * 17, 1, 0), (7200 * 17 + 5, 1, 0)])
def test_compact_float_pack_bit_exact(nbits_arg, n, stride, offset):
    a = pc.float_field(n * stride, seed=nbits_arg + n)
    want = top.pack_float(a, nbits_arg, offset=offset, stride=stride, prefill=0xDEADBEEF)
    got = pk.compact_float_pack(a, nbits_arg, offset=offset, stride=stride, prefill=0xDEADBEEF)
    assert got is not None
    m = got.size
    assert np.array_equal(got[:m], want[:m]), (np.nonzero(got[:m] != want[:m])[0][:5], [hex(int(x)) for x in got[:6]], [hex(int(x)) for x in want[:6]])


def test_compact_float_style1_and_missing_and_zero_min():
    a = np.array([0.0, 1.0, 2.0, 3.0, 3.5], np.float32)
    want = top.pack_float(a, 4, style2=False)
    got = pk.compact_float_pack(a, 4, style1=True)
    assert np.array_equal(got[:5], want[:5])
    b = pc.float_field(5000, seed=1); b[[3, 50, 4999]] = -999.0
    want = top.pack_float(b, 12, has_missing=1, tag=-999.0)
    got = pk.compact_float_pack(b, 12, has_missing=1, tag=-999.0)
    assert np.array_equal(got[:got.size - 1], want[:got.size - 1])
    c = np.full(100, 7.25, np.float32)                              # zero range
    assert np.array_equal(pk.compact_float_pack(c, 16)[:54], top.pack_float(c, 16)[:54])


@pytest.mark.parametrize("nbits", [4, 12, 16, 24])
def test_compact_float_unpack_bit_exact(nbits):
    n = 10007
    a = pc.float_field(n, seed=nbits)
    buf = top.pack_float(a, nbits)
    want = np.zeros(n, np.float32); tagv = np.array([0.0], np.float32)
    top.O().orc_compact_float(want.ctypes.data, buf[:4].ctypes.data, buf[4:].ctypes.data, n, nbits, 0, 1, 2, 0, tagv.ctypes.data)
    got = pk.compact_float_unpack(buf, n, nbits)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("nbits", [1, 2, 4, 8, 12, 15, 16, 17, 24, 31, 32, -1])
@pytest.mark.parametrize("op,header,stride,offset", [(1, False, 1, 0), (3, False, 1, 0), (1, True, 1, 0), (3, True, 2, 0), (1, False, 3, 37), (3, False, 1, 5)])
def test_compact_integer_bit_exact(nbits, op, header, stride, offset):
    if nbits == 32 and offset:
        pytest.skip("undefined in the reference")
    n = 3001
    rng = np.random.default_rng(abs(nbits) * 11 + op)
    width = 20 if nbits == -1 else nbits
    a = rng.integers(0, 2 ** min(width, 32), n * stride, dtype=np.uint64).astype(np.uint32)
    if op == 3:
        a = (a.astype(np.int64) - 2 ** (width - 1)).astype(np.int32)
    rc_w, hdr_w, out_w = top.pack_int(a, nbits, op, header, offset, stride, prefill=0x5A5A5A5A)
    rc_g, hdr_g, out_g = pk.compact_integer_pack(a, nbits, op, header, offset, stride, prefill=0x5A5A5A5A)
    assert rc_g == rc_w
    words = (offset + n * rc_w + 31) // 32
    assert np.array_equal(out_g[:words], out_w[:words])
    if header:
        assert np.array_equal(hdr_g, hdr_w)
    # unpack through the HIP path
    rc, back = pk.compact_integer_unpack(out_g, n, rc_w if nbits == -1 else nbits, op + 1, hdr_g if header else None, offset, stride, a.dtype)
    if header and (int(hdr_w[0]) >> 6) & 0x3F:
        sh = (int(hdr_w[0]) >> 6) & 0x3F
        ref = a[::stride].astype(np.int64); mn = ref.min()
        assert np.array_equal(back[::stride].astype(np.int64), ((ref - mn) >> sh << sh) + mn)
    else:
        assert np.array_equal(back[::stride], a[::stride])


@pytest.mark.parametrize("nbits", [1, 5, 12, 14, 16])
def test_float_packer_bit_exact(nbits):
    for n in (1, 2, 1001, 100000):
        a = (np.arange(n, dtype=np.float64) * 1.234 - 1123.123).astype(np.float32)
        hdr_w = np.zeros(3, np.int32); st_w = np.zeros((n + 1) // 2, np.int32)
        assert top.O().orc_float_packer(a.ctypes.data, nbits, hdr_w.ctypes.data, st_w.ctypes.data, n) == 0
        rc, hdr_g, st_g = pk.float_packer(a, nbits)
        assert rc == 0 and np.array_equal(hdr_g, hdr_w) and np.array_equal(st_g, st_w)
        rc, back, nb = pk.float_unpacker(hdr_g, st_g, n)
        want = np.zeros(n, np.float32); nbw = ctypes.c_int(0)
        top.O().orc_float_unpacker(want.ctypes.data, hdr_w.ctypes.data, st_w.ctypes.data, n, ctypes.byref(nbw))
        assert rc == 0 and nb == nbits and np.array_equal(back.view(np.uint32), want.view(np.uint32))


ARMN = [(16, 16), (17, 19), (64, 48), (7200, 17), (15, 40), (40, 9), (1000, 777)]


@pytest.mark.parametrize("ni,nj", ARMN)
@pytest.mark.parametrize("kind", ["smooth", "noisy", "constant", "bigdiff"])
@pytest.mark.parametrize("nbits", [16, 12, 4])
def test_armn_compress_bit_exact(ni, nj, kind, nbits):
    tok = pc.token_field(ni, nj, nbits, kind, seed=ni * 3 + nj)
    words = pc.tokens_to_words(tok)
    bw = np.zeros(words.size + 8, np.uint32); bw[:words.size] = words
    bg = bw.copy()
    zw = top.O().orc_armn_compress(bw.ctypes.data, ni, nj, 1, nbits, 1)
    zg = pk.armn_compress(bg, ni, nj, nbits)
    assert zg == zw, (zg, zw)
    if zw > 0:
        nfull = (zw - 1) // 4                 # the zlng-th byte is undefined in the reference
        assert np.array_equal(bg[:nfull], bw[:nfull]), np.nonzero(bg[:nfull] != bw[:nfull])[0][:5]
        back = np.zeros(ni * nj, np.uint16)   # and the reference's decoder (restated) recovers the tokens
        assert top.O().orc_armn_decode(back.ctypes.data, bg.ctypes.data, ni, nj) == 0
        assert np.array_equal(back, tok)
    else:
        assert np.array_equal(bg[:words.size], words)             # rejected: buffer untouched


def test_cfg5_pipeline_full_size_device_resident():
    """interp -> compact_float(16-bit slots) -> armn_compress on one full-size field, all on the device;
    compared with the oracle chain on the same interpolated field (bit-exact record)."""
    from librmn_amd import ezscint as ez
    import ezcases as ec
    ni, nj, no, mo = 4400, 2200, 7200, 3601
    gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
    ez.ezdefset(gdout, gdin)
    ez.ezsetopt("interp_degree", "cubic"); ez.ezsetopt("polar_correction", "yes")
    ez.use_stream(0)
    f = torch.from_numpy(ec.synth_field(ni, nj, seed=2)).cuda()
    z = torch.empty(no * mo, dtype=torch.float32, device="cuda")
    assert ez.ezsint_dev(z, f) == 0
    n = no * mo
    rec = torch.zeros(4 + n // 2 + 64, dtype=torch.int32, device="cuda")
    zlng = pk.pack16_compress_dev(rec, z, no, mo, 16)
    torch.cuda.synchronize()
    zh = z.cpu().numpy()
    want = top.pack_float(zh, 16 + 64 * 16)
    zw = top.O().orc_armn_compress(want[4:].ctypes.data, no, mo, 1, 16, 1)
    assert zlng == zw and zw > 0
    got = rec.cpu().numpy().view(np.uint32)
    nfull = 4 + (zw - 1) // 4
    assert np.array_equal(got[:nfull], want[:nfull])
    assert zw < 0.5 * n * 2                                  # a smooth field compresses


@pytest.mark.parametrize("degree", ["cubic", "nearest"])
@pytest.mark.parametrize("F", [1, 3, 5])        # 1: lone field; 3: pre-launched pole values; 5: pole sums inside the launch
def test_fused_interp_pack16_equals_separate_steps(degree, F):
    """ezhip_ezsint_pack16_batch_dev (compact_float's min/max pass fused into the interpolation kernel) leaves the same
    fields and bit-identical records as c_ezsint_batch_dev followed by compact_float_dev per field; and the records
    equal the CPU oracle's compact_float of the interpolated field."""
    import torch
    from librmn_amd import ezscint as ez
    import ezcases as ec
    ni, nj, no, mo, nbits = 360, 181, 520, 261, 16
    gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 69, 69, 0, 0)
    assert ez.ezdefset(gdout, gdin) == 1
    assert ez.ezsetopt("interp_degree", degree) == 0 and ez.ezsetopt("polar_correction", "yes") == 0
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    d_in = torch.stack([torch.from_numpy(ec.synth_field(ni, nj, seed=90 + f)) for f in range(F)]).cuda().contiguous()
    n = no * mo
    rs = 4 + (n + 1) // 2 + 8
    out_a = torch.empty((F, n), dtype=torch.float32, device="cuda"); out_b = torch.empty_like(out_a)
    rec_a = torch.zeros((F, rs), dtype=torch.int32, device="cuda"); rec_b = torch.zeros_like(rec_a)
    assert pk.ezsint_pack16_batch_dev(rec_a, rs, out_a, d_in, F, n, nbits) == 0
    assert ez.ezsint_batch_dev(out_b, d_in, F) == 0
    for f in range(F):
        assert pk.compact_float_pack_dev(out_b[f], rec_b[f], rec_b[f][4:], n, nbits + 64 * 16) != 0
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b)
    assert torch.equal(rec_a, rec_b)
    # against the oracle packer on the GPU-interpolated field
    z = out_a[F - 1].cpu().numpy()
    want = top.pack_float(z, nbits + 64 * 16)
    got = rec_a[F - 1].cpu().numpy().view(np.uint32)
    m = 4 + (n + 1) // 2
    assert np.array_equal(got[:m], want[:m])
    assert ez.ezsetopt("interp_degree", "cubic") == 0


def test_pack16_compress_batch_equals_field_by_field():
    """ezhip_pack16_compress_batch_dev (asynchronous per field, zlng and the commit on the device, one sync) ==
    ezhip_pack16_compress_dev field by field: same zlng, same first zlng - 1 bytes; one incompressible field keeps
    its plain pack and reports -1"""
    import torch
    ni, nj, F, nbits = 300, 170, 4, 16
    n = ni * nj
    fields = [pc.float_field(n, seed=300 + f) for f in range(F)]
    fields[2] = (np.frombuffer(np.random.default_rng(3).bytes(4 * n), dtype=np.uint32) % 60000).astype(np.float32)   # noise: not compressible
    d_f = torch.stack([torch.from_numpy(a) for a in fields]).cuda().contiguous()
    rs = 4 + (n + 1) // 2 + 40
    rec_a = torch.zeros((F, rs), dtype=torch.int32, device="cuda"); rec_b = torch.zeros_like(rec_a)
    rc, zl = pk.pack16_compress_batch_dev(rec_a, rs, d_f, n, F, ni, nj, nbits)
    assert rc == 0
    zb = [pk.pack16_compress_dev(rec_b[f], d_f[f], ni, nj, nbits) for f in range(F)]
    torch.cuda.synchronize()
    assert list(zl) == zb, (list(zl), zb)
    assert zl[2] == -1 and all(z > 0 for k, z in enumerate(zl) if k != 2)
    a = rec_a.cpu().numpy().view(np.uint8).reshape(F, -1); b = rec_b.cpu().numpy().view(np.uint8).reshape(F, -1)
    for f in range(F):
        m = 16 + (int(zl[f]) - 1 if zl[f] > 0 else 2 * n)          # header + stream bytes that are defined
        assert np.array_equal(a[f, :m], b[f, :m]), f


# ---------------------------------------------------------------------------------------------
# armn_compress UNCOMPRESS: the HIP decoder against the restated reference decoder
# ---------------------------------------------------------------------------------------------
DEC_SHAPES = ARMN + [(100, 31), (31, 100), (19, 16), (16, 19), (256, 256), (4000, 50), (50, 4000), (18, 18), (3001, 301)]


def _oracle_stream(tok, ni, nj, nbits, level):
    O = top.O()
    O.orc_armn_compress_setlevel(level)
    z = np.zeros(ni * nj + 64, np.uint32)
    zlng = O.orc_armn_encode(z.ctypes.data, tok.ctypes.data, ni, nj, nbits)
    O.orc_armn_compress_setlevel(-1)
    return z, zlng


@pytest.mark.parametrize("ni,nj", DEC_SHAPES)
@pytest.mark.parametrize("kind", ["smooth", "noisy", "constant", "bigdiff"])
@pytest.mark.parametrize("nbits,level", [(16, 1), (12, 1), (4, 1), (16, 0), (9, 0)])
def test_armn_uncompress_matches_oracle_decoder(ni, nj, kind, nbits, level):
    """streams written by the oracle's encoder (PARALLELOGRAM at level BEST, MINIMUM at FAST / small / <= 4 bits),
    compressible or not, decoded by the HIP path == the restated reference decoder == the original tokens"""
    tok = pc.token_field(ni, nj, nbits, kind, seed=ni * 7 + nj)
    z, zlng = _oracle_stream(tok, ni, nj, nbits, level)
    zwords = (zlng - 1) // 4 + 1
    want = np.zeros(ni * nj, np.uint16)
    assert top.O().orc_armn_decode(want.ctypes.data, z.ctypes.data, ni, nj) == 0
    assert np.array_equal(want, tok)
    d_z = torch.from_numpy(z[:zwords].view(np.int32).copy()).cuda()
    d_out = torch.full((1 + ni * nj // 2 + 4,), -1, dtype=torch.int32, device="cuda")
    assert pk.armn_uncompress_dev(d_out, d_z, zwords, ni, nj, nbits) == ni * nj * 2
    got = d_out.cpu().numpy().view(np.uint32)
    words = pc.tokens_to_words(tok)
    assert np.array_equal(got[:words.size], words), np.nonzero(got[:words.size] != words)[0][:5]
    assert np.all(got[1 + ni * nj // 2:] == 0xFFFFFFFF)            # nothing written past the (1 + n/2) words


SCAN_SHAPES = [(3100, 200), (3076, 64), (3076, 65), (3077, 64), (3074, 47), (3073, 46), (5200, 40), (1501, 300), (600, 601), (257, 1000)]      # odd last tile per row or not x last row of another height or not      # rows of >= 1024 tiles take the parallel form by default


@pytest.mark.parametrize("ni,nj", SCAN_SHAPES)
@pytest.mark.parametrize("kind", ["smooth", "noisy", "constant", "bigdiff"])
@pytest.mark.parametrize("mode", ["default", "chain_kernel", "scan_from_64_tiles_per_row"])
@pytest.mark.parametrize("level", [1, 0])
def test_armn_uncompress_parallel_form_and_chain_kernel(ni, nj, kind, mode, level, monkeypatch):
    """the chain between row ends resolved in parallel (k_dsc_*: the default for rows of >= 768 tiles), the serial chain kernel (EZHIP_DEC_SCAN=0) and the
    parallel form pushed onto short rows (where it gives up on some streams and hands them to the chain kernel): the oracle's tokens every time; shapes with
    and without a last row of another height, with and without an odd last tile per row"""
    if mode == "chain_kernel":
        monkeypatch.setenv("EZHIP_DEC_SCAN", "0")
    if mode == "scan_from_64_tiles_per_row":
        monkeypatch.setenv("EZHIP_DEC_SCAN_MIN_NTX", "64")
    nbits = 16
    tok = pc.token_field(ni, nj, nbits, kind, seed=ni + 3 * nj)
    z, zlng = _oracle_stream(tok, ni, nj, nbits, level)              # level 1: PARALLELOGRAM (tiles of 3 x 3), 0: MINIMUM (5 x 5)
    zwords = (zlng - 1) // 4 + 1
    d_z = torch.from_numpy(z[:zwords].view(np.int32).copy()).cuda()
    d_out = torch.full((1 + ni * nj // 2 + 4,), -1, dtype=torch.int32, device="cuda")
    assert pk.armn_uncompress_dev(d_out, d_z, zwords, ni, nj, nbits) == ni * nj * 2
    got = d_out.cpu().numpy().view(np.uint32)
    words = pc.tokens_to_words(tok)
    assert np.array_equal(got[:words.size], words), np.nonzero(got[:words.size] != words)[0][:5]
    assert np.all(got[1 + ni * nj // 2:] == 0xFFFFFFFF)


@pytest.mark.parametrize("damage", ["half_the_stream", "bit_flips", "zeroed_piece"])
def test_armn_uncompress_damaged_streams_both_forms_agree(damage, monkeypatch):
    """a truncated or corrupted record: the parallel form either follows the same (wrong) chain as the serial kernel or gives the field up to it -- the same
    return code and, where a result exists, the same words; nothing hangs, nothing is read or written outside the buffers"""
    ni, nj, nbits = 3100, 200, 16
    tok = pc.token_field(ni, nj, nbits, "smooth", seed=77)
    z, zlng = _oracle_stream(tok, ni, nj, nbits, 1)
    zwords = (zlng - 1) // 4 + 1
    zd = z[:zwords].copy()
    rng = np.random.default_rng(5)
    if damage == "half_the_stream":
        zd = zd[:zwords // 2].copy()
    elif damage == "bit_flips":
        for k in rng.integers(zwords // 8, zwords, 40):
            zd[k] ^= np.uint32(1 << int(rng.integers(0, 32)))
    else:
        zd[zwords // 3: zwords // 3 + 300] = 0
    res = {}
    for mode in ("default", "chain_kernel"):
        if mode == "chain_kernel":
            monkeypatch.setenv("EZHIP_DEC_SCAN", "0")
        else:
            monkeypatch.delenv("EZHIP_DEC_SCAN", raising=False)
        d_z = torch.from_numpy(zd.view(np.int32).copy()).cuda()
        d_out = torch.full((1 + ni * nj // 2 + 4,), -1, dtype=torch.int32, device="cuda")
        rc = pk.armn_uncompress_dev(d_out, d_z, zd.size, ni, nj, nbits)
        torch.cuda.synchronize()
        res[mode] = (rc, d_out.cpu().numpy().copy())
    assert res["default"][0] == res["chain_kernel"][0], (res["default"][0], res["chain_kernel"][0])
    if res["default"][0] > 0:
        assert np.array_equal(res["default"][1], res["chain_kernel"][1])
    assert np.all(res["default"][1][1 + ni * nj // 2:] == -1)


@pytest.mark.parametrize("swap", [1, 0])
def test_armn_uncompress_host_in_place_round_trip(swap):
    ni, nj, nbits = 301, 200, 16
    tok = pc.token_field(ni, nj, nbits, "smooth", seed=5)
    words = pc.tokens_to_words(tok)
    buf = np.zeros(1 + ni * nj // 2 + 2, np.uint32); buf[:words.size] = words
    zlng = pk.armn_compress(buf, ni, nj, nbits)
    assert 0 < zlng < ni * nj * 2
    buf[(zlng + 3) // 4:] = 0xDEADBEEF                              # whatever followed the stream in the record
    pk.armn_setswap(swap)
    try:
        assert pk.armn_uncompress(buf, ni, nj, nbits) == ni * nj * 2
    finally:
        pk.armn_setswap(1)
    if swap:
        assert np.array_equal(buf[:words.size], words)
    else:                                                            # fstluk's integer path (fstd98.c:2326-2330): natural ushort order
        assert np.array_equal(buf.view(np.uint16)[:ni * nj], tok)
    # the refusals of the reference (c_zfstlib.c:182-184): nothing touched, the odd byte count returned
    assert pk.armn_uncompress(buf, ni, nj, 17) == 1 + ni * nj * 17 // 8
    assert pk.armn_uncompress(buf, 1, nj, 16) == 1 + nj * 16 // 8


def test_armn_uncompress_rejects_unknown_header():
    ni, nj = 64, 48
    z = np.zeros(ni * nj, np.uint32); z[0] = 2 | 3 << 7 | 16 << 10         # SAMPLE predictor (deactivated since 2006)
    d_z = torch.from_numpy(z.view(np.int32)).cuda()
    d_out = torch.zeros(1 + ni * nj // 2, dtype=torch.int32, device="cuda")
    assert pk.armn_uncompress_dev(d_out, d_z, z.size, ni, nj, 16) == -1


def test_armn_uncompress_batch_and_full_size_record():
    """cfg5 read path at full size: 3 different records (one of them plain: not compressible) written by the HIP
    write path, decoded as a batch -> the original 16-bit tokens; then record -> floats == compact_float unpack"""
    import ezcases as ec
    no, mo, F = 7200, 3601, 3
    n = no * mo
    stride = 4 + n // 2 + 64
    fields = np.stack([ec.synth_field(no, mo, seed=31), ec.synth_field(no, mo, seed=32, noise=0.5),
                       ec.synth_field(no, mo, seed=33, base=1.0, amp=0.5, noise=1e-5)]).reshape(F, n)
    d_f = torch.from_numpy(fields).cuda()
    recs = torch.zeros(F * stride, dtype=torch.int32, device="cuda")
    plain = torch.zeros(F * stride, dtype=torch.int32, device="cuda")
    for f in range(F):                                                  # plain 16-bit-slot packs (header + tokens)
        assert pk.compact_float_pack_dev(d_f[f], plain[f * stride:], plain[f * stride + 4:], n, 16 + 64 * 16)
    rc, zl = pk.pack16_compress_batch_dev(recs, stride, d_f, n, F, no, mo, 16)
    assert rc == 0 and (zl > 0).sum() >= 2
    toks = torch.full((F, 1 + n // 2), -1, dtype=torch.int32, device="cuda")
    comp = [f for f in range(F) if zl[f] > 0]
    # batch decode of the compressed ones (contiguous selection: decode all, compare the compressed)
    assert pk.armn_uncompress_batch_dev(toks, 1 + n // 2, recs[4:], stride, stride - 4, no, mo, 16, F) in (n * 2, -1)
    torch.cuda.synchronize()
    P = plain.view(F, stride)
    for f in comp:
        assert torch.equal(toks[f, :n // 2], P[f, 4:4 + n // 2]), f
    out = torch.empty(n, dtype=torch.float32, device="cuda"); ref = torch.empty(n, dtype=torch.float32, device="cuda")
    tagv = np.array([0.0], np.float32)
    for f in range(F):
        assert pk.uncompress_unpack16_dev(out, recs[f * stride:], no, mo, 16, zl[f] > 0) == 0
        assert pk._lib().compact_float_dev(ref.data_ptr(), plain[f * stride:].data_ptr(), plain[f * stride + 4:].data_ptr(), n, 16 + 64 * 16, 0, 1, 2, 0, tagv.ctypes.data, 2)
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), f
        rng = float(d_f[f].max() - d_f[f].min())                     # tokens truncate to 2^ceil(log2(range)) / 65536
        assert float((out - d_f[f]).abs().max()) <= 2.0 ** np.ceil(np.log2(rng)) / 65536 * 1.01


# ---------------------------------------------------------------------------------------------
# round 2: one-pass encoder (k_armn_enc1), fused cfg5 pipeline, capacity bound, swap state
# ---------------------------------------------------------------------------------------------
ENC1_SHAPES = [(16, 16), (17, 16), (18, 33), (100, 31), (31, 100), (3073, 19), (3075, 22), (3100, 16), (6200, 20), (9300, 17), (2048, 64), (257, 513)]


@pytest.mark.parametrize("ni,nj", ENC1_SHAPES)
@pytest.mark.parametrize("kind", ["smooth", "noisy", "bigdiff"])
@pytest.mark.parametrize("nbits", [16, 15, 9, 5])
def test_armn_one_pass_encoder_chunk_geometries(ni, nj, kind, nbits):
    """shapes that exercise every chunk geometry of k_armn_enc1: several tile rows per chunk (narrow fields), exactly one,
    two / three / four segments per tile row, clipped last tiles in both directions, chunks shorter than one stream word;
    'bigdiff' at 15-16 bits takes the 5-bit container re-run (c_zfstlib.c:701-711).  Bit-exact against the oracle encoder."""
    tok = pc.token_field(ni, nj, nbits, kind, seed=ni + 7 * nj)
    words = pc.tokens_to_words(tok)
    n = ni * nj
    d_w = torch.from_numpy(words.view(np.int32)).cuda()
    cap = n // 2 + 16
    guard = 64
    d_z = torch.full((cap + guard,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    zg = pk.armn_compress_dev(d_z, d_w, ni, nj, nbits)
    torch.cuda.synchronize()
    z = np.zeros(n + 64, np.uint32)
    zw = top.O().orc_armn_encode(z.ctypes.data, tok.ctypes.data, ni, nj, nbits)
    if zw >= 1 + 2 * n:
        zw = -1
    assert zg == zw, (zg, zw)
    got = d_z.cpu().numpy().view(np.uint32)
    assert np.all(got[cap:] == 0x5A5A5A5A)                          # no store beyond the documented capacity, compressible or not
    if zw > 0:
        nfull = (zw - 1) // 4
        assert np.array_equal(got[:nfull], z[:nfull]), np.nonzero(got[:nfull] != z[:nfull])[0][:5]


def test_armn_compress_swap_state_zero():
    """c_armn_compress_setswap(0) (fstd98.c:1209-1211): COMPRESS then reads the 16-bit halves in memory order
    (c_zfstlib.c:119-126 skipped); compress -> uncompress under the same state is the identity on the words"""
    ni, nj, nbits = 301, 200, 16
    tok = pc.token_field(ni, nj, nbits, "smooth", seed=5)
    words = pc.tokens_to_words(tok)
    raw = words.view(np.uint16)[:ni * nj].copy()                    # what the reference's ushort pointer sees without the swap
    z = np.zeros(ni * nj + 64, np.uint32)
    zw = top.O().orc_armn_encode(z.ctypes.data, raw.ctypes.data, ni, nj, nbits)
    buf = np.zeros(1 + ni * nj // 2 + 2, np.uint32); buf[:words.size] = words
    pk.armn_setswap(0)
    try:
        zg = pk.armn_compress(buf, ni, nj, nbits)
        assert zg == zw and zw > 0
        assert np.array_equal(buf[:(zw - 1) // 4], z[:(zw - 1) // 4])
        assert pk.armn_uncompress(buf, ni, nj, nbits) == ni * nj * 2
        assert np.array_equal(buf[:words.size], words)
        pk.armn_setlevel(0)                                          # MINIMUM method (multi-kernel path) under the same state
        try:
            buf2 = np.zeros(1 + ni * nj // 2 + 2, np.uint32); buf2[:words.size] = words
            z2 = pk.armn_compress(buf2, ni, nj, nbits)
            assert z2 > 0 and pk.armn_uncompress(buf2, ni, nj, nbits) == ni * nj * 2
            assert np.array_equal(buf2[:words.size], words)
        finally:
            pk.armn_setlevel(1)
    finally:
        pk.armn_setswap(1)


@pytest.mark.parametrize("degree", ["cubic", "linear", "nearest"])
@pytest.mark.parametrize("F", [1, 2, 5])
def test_fused_cfg5_pipeline_equals_unfused(degree, F):
    """ezhip_ezsint_pack16_compress_batch_dev (interpolation twice -- min/max only, then straight to 16-bit tokens --, one-pass
    encoder writing in place) leaves the records and byte counts of the unfused chain c_ezsint_batch_dev -> compact_float ->
    armn_compress, bit for bit; one source field is pure noise (its interpolation is barely / not compressible)"""
    from librmn_amd import ezscint as ez
    import ezcases as ec
    ni, nj, no, mo, nbits = 360, 181, 520, 261, 16
    gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 69, 69, 0, 0)
    assert ez.ezdefset(gdout, gdin) == 1
    assert ez.ezsetopt("interp_degree", degree) == 0 and ez.ezsetopt("polar_correction", "yes") == 0
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    srcs = [ec.synth_field(ni, nj, seed=190 + f) for f in range(F)]
    if F > 1:
        srcs[1] = (ec.hash_uniform(5, ni * nj) * np.float32(1000.0)).astype(np.float32)
    d_in = torch.stack([torch.from_numpy(a) for a in srcs]).cuda().contiguous()
    n = no * mo
    rs = 4 + n // 2 + 16
    rec_a = torch.full((F, rs), 0x11111111, dtype=torch.int32, device="cuda"); rec_b = torch.zeros_like(rec_a)
    rc, zl_a = pk.ezsint_pack16_compress_batch_dev(rec_a, rs, d_in, F, no, mo, nbits)
    assert rc == 0, rc
    out = torch.empty((F, n), dtype=torch.float32, device="cuda")
    assert pk.ezsint_pack16_batch_dev(rec_b, rs, out, d_in, F, n, nbits) == 0
    rc, zl_b = pk.pack16_compress_batch_dev(rec_b, rs, None, 0, F, no, mo, nbits, prepacked=1)
    assert rc == 0
    torch.cuda.synchronize()
    assert list(zl_a) == list(zl_b), (list(zl_a), list(zl_b))
    a = rec_a.cpu().numpy().view(np.uint8).reshape(F, -1); b = rec_b.cpu().numpy().view(np.uint8).reshape(F, -1)
    for f in range(F):
        m = 16 + (int(zl_a[f]) - 1 if zl_a[f] > 0 else 2 * n)
        assert np.array_equal(a[f, :m], b[f, :m]), (f, int(zl_a[f]), np.nonzero(a[f, :m] != b[f, :m])[0][:5])
    # and against the CPU oracle chain on the GPU-interpolated field 0
    z = out[0].cpu().numpy()
    want = top.pack_float(z, nbits + 64 * 16)
    zw = top.O().orc_armn_compress(want[4:].ctypes.data, no, mo, 1, nbits, 1)
    assert zw == zl_a[0]
    if zw > 0:
        got = rec_a[0].cpu().numpy().view(np.uint32)
        assert np.array_equal(got[:4 + (zw - 1) // 4], want[:4 + (zw - 1) // 4])
    assert ez.ezsetopt("interp_degree", "cubic") == 0


SEPENC_SHAPES = [
    # target ni x nj of an L grid inside a G 360 x 181 source: strips of 255 columns (85 tiles), row groups of 15 rows (5 tile rows)
    (520, 261, "three strips, the last with 3 tiles; (nj - 1) % 3 == 2"),
    (512, 256, "(ni - 1) % 3 == 1: the last tile column holds one column; (nj - 1) % 3 == 0"),
    (766, 122, "ni - 1 = 3 x 255: exactly three full strips; the last tile row holds one row"),
    (768, 47, "one tile past three strips: a fourth strip of one clipped tile; the last row group holds one tile row"),
    (256, 16, "the smallest shape the one-launch form takes: one strip, one row group"),
    (1022, 333, "five strips, 23 row groups, both edges ragged"),
]


@pytest.mark.parametrize("no,mo,what", SEPENC_SHAPES, ids=[f"{a}x{b}" for a, b, _ in SEPENC_SHAPES])
@pytest.mark.parametrize("degree", ["cubic", "linear"])
def test_cfg5_one_launch_form_equals_the_two_kernels(no, mo, what, degree):
    """EZHIP_CFG5_FUSED=1: interpolation, quantisation and armn_compress in ONE launch (k_sepx_enc: the tokens never reach HBM) against the
    default two kernels (k_sepx<.., tokens> + k_armn_enc1) -- records and byte counts bit for bit, over shapes that put the strip / row-group /
    tile edges everywhere; one field of noise (not compressible: the one-launch form hands it back to the two-kernel path), one constant"""
    from librmn_amd import ezscint as ez
    import ezcases as ec
    nbits, F = 16, 4
    ni, nj = max(64, int(no * 0.7) // 2 * 2), max(24, int(mo * 0.7) // 2 * 2)             # a finer target, as in cfg5 (a strip of 256 target columns fits the staged window)
    gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0)
    gdout = ez.ezqkdef(no, mo, "L", max(1, 17000 // mo), max(1, 35900 // no), 0, 0)         # the whole globe, whatever the size
    assert ez.ezdefset(gdout, gdin) == 1
    assert ez.ezsetopt("interp_degree", degree) == 0 and ez.ezsetopt("polar_correction", "yes") == 0
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    srcs = [ec.synth_field(ni, nj, seed=300 + f) for f in range(F)]
    srcs[1] = (ec.hash_uniform(7, ni * nj) * np.float32(1000.0)).astype(np.float32)
    srcs[2] = np.full(ni * nj, 3.25, np.float32)
    d_in = torch.stack([torch.from_numpy(a) for a in srcs]).cuda().contiguous()
    n = no * mo
    rs = 4 + n // 2 + 16
    out = {}
    try:
        for fused in ("0", "1"):
            os.environ.pop("EZHIP_CFG5_FUSED", None)
            if fused == "1": os.environ["EZHIP_CFG5_FUSED"] = "1"
            rec = torch.full((F, rs), 0x22222222, dtype=torch.int32, device="cuda")
            rc, zl = pk.ezsint_pack16_compress_batch_dev(rec, rs, d_in, F, no, mo, nbits)
            if rc == -2:
                pytest.skip("grid pair not on the single-launch k_sepx path")
            assert rc == 0, (fused, rc)
            out[fused] = (list(zl), rec.cpu().numpy().view(np.uint8).reshape(F, -1))
    finally:
        os.environ.pop("EZHIP_CFG5_FUSED", None)
        ez.ezsetopt("interp_degree", "cubic")
    assert out["0"][0] == out["1"][0], (what, out["0"][0], out["1"][0])
    for f in range(F):
        zl = out["0"][0][f]
        m = 16 + (int(zl) - 1 if zl > 0 else 2 * n)
        a, b = out["0"][1][f, :m], out["1"][1][f, :m]
        assert np.array_equal(a, b), (what, f, int(zl), np.nonzero(a != b)[0][:5])
    assert out["0"][0][0] > 0                                                                  # the smooth field did compress


def test_fused_cfg5_pipeline_full_size():
    """the fused pipeline on full-size cfg5 fields (G 4400x2200 -> L 7200x3601 bicubic, 16 bits): records and byte counts equal
    the single-field chain (c_ezsint_dev -> ezhip_pack16_compress_dev), which test_cfg5_pipeline_full_size_device_resident
    pins against the oracle"""
    from librmn_amd import ezscint as ez
    import ezcases as ec
    ni, nj, no, mo = 4400, 2200, 7200, 3601
    gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
    assert ez.ezdefset(gdout, gdin) == 1
    ez.ezsetopt("interp_degree", "cubic"); ez.ezsetopt("polar_correction", "yes")
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    F = 3
    d_in = torch.stack([torch.from_numpy(ec.synth_field(ni, nj, seed=2 + 11 * f)) for f in range(F)]).cuda().contiguous()
    n = no * mo
    rs = 4 + n // 2 + 16
    rec = torch.zeros((F, rs), dtype=torch.int32, device="cuda")
    rc, zl = pk.ezsint_pack16_compress_batch_dev(rec, rs, d_in, F, no, mo, 16)
    assert rc == 0 and all(z > 0 for z in zl), (rc, list(zl))
    z = torch.empty(n, dtype=torch.float32, device="cuda")
    one = torch.zeros(4 + n // 2 + 64, dtype=torch.int32, device="cuda")
    for f in (0, F - 1):
        assert ez.ezsint_dev(z, d_in[f]) == 0
        zs = pk.pack16_compress_dev(one, z, no, mo, 16)
        torch.cuda.synchronize()
        assert zs == zl[f], (f, zs, int(zl[f]))
        m = 4 + (zs - 1) // 4
        assert torch.equal(one[:m], rec[f][:m]), f
    # ... and one field DIRECTLY against the oracle chain (compact_float + armn_compress of the float field c_ezsint_dev leaves), not only through the
    # unfused HIP chain: header words, byte count and every full word of the stream
    assert ez.ezsint_dev(z, d_in[0]) == 0
    torch.cuda.synchronize()
    want = top.pack_float(z.cpu().numpy(), 16 + 64 * 16)
    zw = top.O().orc_armn_compress(want[4:].ctypes.data, no, mo, 1, 16, 1)
    assert zw == zl[0] and zw > 0, (zw, int(zl[0]))
    got = rec[0].cpu().numpy().view(np.uint32)
    nfull = 4 + (zw - 1) // 4
    assert np.array_equal(got[:nfull], want[:nfull]), np.nonzero(got[:nfull] != want[:nfull])[0][:5]


def test_cfg5_record_against_reference_interpolated_floats():
    """cfg5 END TO END against the reference side: tests/golden/cfg5_record_golden.npz holds zlng, the header and sampled 16-bit tokens of the record the
    oracle's packers make of the field the REFERENCE's c_ezsint interpolated (tests/golden/make_cfg5_record.py).  The HIP pipeline's bicubic values are
    within 1 ulp of the reference's, not identical (the 1e-5 bar), so its record may differ where a value crosses a quantisation step: the header must be
    identical (the field's extrema are), at most 0.1 % of the sampled tokens may differ and then by one step, and the byte count within 1e-3."""
    from librmn_amd import ezscint as ez
    import ezcases as ec
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "cfg5_record_golden.npz"))
    ni, nj, no, mo = 4400, 2200, 7200, 3601
    gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
    assert ez.ezdefset(gdout, gdin) == 1
    ez.ezsetopt("interp_degree", "cubic"); ez.ezsetopt("polar_correction", "yes")
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    d_in = torch.from_numpy(ec.synth_field(ni, nj, seed=2)).cuda().reshape(1, -1).contiguous()
    n = no * mo
    rs = 4 + n // 2 + 16
    rec = torch.zeros((1, rs), dtype=torch.int32, device="cuda")
    rc, zl = pk.ezsint_pack16_compress_batch_dev(rec, rs, d_in, 1, no, mo, 16)
    assert rc == 0 and zl[0] > 0, (rc, list(zl))
    zlng, want_zlng = int(zl[0]), int(G["zlng"])
    assert zlng == want_zlng, (zlng, want_zlng)          # (round 6: exact -- the interpolated field under the record equals the reference's at every one of its 25.9 M points, tests/test_gpu_wind_pin.py)
    got_hdr = rec[0, :4].cpu().numpy().view(np.uint32)
    assert np.array_equal(got_hdr, G["header"]), ([hex(int(x)) for x in got_hdr], [hex(int(x)) for x in G["header"]])
    # the record's tokens back (the HIP decoder; bit-exact against the oracle's in the decoder tests)
    zwords = (zlng - 1) // 4 + 1
    d_tok = torch.full((1 + n // 2 + 4,), -1, dtype=torch.int32, device="cuda")
    assert pk.armn_uncompress_dev(d_tok, rec[0, 4:4 + zwords].contiguous(), zwords, no, mo, 16) == n * 2
    w = d_tok[:n // 2].cpu().numpy().view(np.uint32)
    tok = np.empty(n, np.uint16); tok[0::2] = (w >> 16).astype(np.uint16); tok[1::2] = (w & 0xFFFF).astype(np.uint16)
    tok = tok.reshape(mo, no)
    ndiff = 0; nsamp = 0; worst = 0
    for got, want in ((tok[G["rows"]], G["tok_rows"]), (tok[:, G["cols"]], G["tok_cols"])):
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        ndiff += int((d != 0).sum()); nsamp += d.size; worst = max(worst, int(d.max()))
    print(f"cfg5 end to end: zlng {zlng} (reference-side {want_zlng}, {abs(zlng - want_zlng) / want_zlng:.2e}), {ndiff} of {nsamp} sampled tokens differ (max {worst} step)")
    assert ndiff == 0, (ndiff, nsamp, worst)


# ---------------------------------------------------------------------------------------------
# compact_double, compact_short, compact_char (the sibling entry points c_fstecr / c_fstluk call)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nbits_arg", [4, 12, 16, 24, 31, 32, 16 + 64 * 16])
@pytest.mark.parametrize("n,stride,offset", [(1, 1, 0), (33, 2, 5), (1000, 1, 0), (7200 * 17 + 5, 1, 0)])
def test_compact_double_bit_exact(nbits_arg, n, stride, offset):
    a = pc.float_field(n * stride, seed=nbits_arg + n).astype(np.float64) * (1.0 + 1e-9 * np.arange(n * stride))    # real double content
    eff = (nbits_arg >> 6) if nbits_arg > 64 else nbits_arg
    want = np.full(4 + (offset + n * eff + 31) // 32 + 1, 0xDEADBEEF, np.uint32)
    tagv = np.array([0.0], np.float64)
    assert top.O().orc_compact_double(a.ctypes.data, want.ctypes.data, want[4:].ctypes.data, n, nbits_arg, offset, stride, 1, 0, tagv.ctypes.data)
    got = pk.compact_double_pack(a, nbits_arg, offset=offset, stride=stride, prefill=0xDEADBEEF)
    assert got is not None and np.array_equal(got, want), np.nonzero(got != want)[0][:5]
    back_w = np.zeros(n * stride, np.float64)
    top.O().orc_compact_double(back_w.ctypes.data, want.ctypes.data, want[4:].ctypes.data, n, nbits_arg, offset, stride, 2, 0, tagv.ctypes.data)
    back = pk.compact_double_unpack(got, n, nbits_arg, offset=offset, stride=stride)
    assert np.array_equal(back.view(np.uint64), back_w.view(np.uint64))


def test_compact_double_missing_values():
    a = pc.float_field(5000, seed=1).astype(np.float64); a[[3, 50, 4999]] = -999.0
    tagv = np.array([-999.0], np.float64)
    want = np.zeros(4 + (5000 * 12 + 31) // 32 + 1, np.uint32)
    assert top.O().orc_compact_double(a.ctypes.data, want.ctypes.data, want[4:].ctypes.data, 5000, 12, 0, 1, 1, 1, tagv.ctypes.data)
    got = pk.compact_double_pack(a, 12, has_missing=1, tag=-999.0)
    assert np.array_equal(got[:got.size - 1], want[:got.size - 1])
    back = pk.compact_double_unpack(got, 5000, 12, has_missing=1, tag=-999.0)
    assert back[3] == -999.0 and back[4999] == -999.0


@pytest.mark.parametrize("dtype,bits", [(np.uint16, [1, 7, 12, 16, -1]), (np.uint8, [1, 5, 8, -1])])
@pytest.mark.parametrize("header,stride,offset", [(False, 1, 0), (True, 1, 0), (False, 3, 37), (True, 2, 0)])
def test_compact_short_and_char_bit_exact(dtype, bits, header, stride, offset):
    n = 3001
    rng = np.random.default_rng(17 + stride + offset)
    O = top.O()
    for nbits in bits:
        width = (12 if dtype == np.uint16 else 6) if nbits == -1 else min(nbits, 8 * np.dtype(dtype).itemsize)
        a = rng.integers(0, 2 ** width, n * stride, dtype=np.uint64).astype(dtype)
        nb = 32 if nbits == -1 else nbits
        w_out = np.full((offset + n * nb + 31) // 32 + 1, 0x0F0F0F0F, np.uint32); w_hdr = np.zeros(4, np.uint32)
        fn = O.orc_compact_short if dtype == np.uint16 else O.orc_compact_char
        ops = (5, 6) if dtype == np.uint16 else (9, 10)
        rc_w = fn(a.ctypes.data, w_hdr.ctypes.data if header else None, w_out.ctypes.data, n, nbits, offset, stride, ops[0])
        rc, hdr, out = pk.compact_narrow_pack(a, nbits, header=header, offset=offset, stride=stride, prefill=0x0F0F0F0F)
        assert rc == rc_w and np.array_equal(hdr, w_hdr) and np.array_equal(out, w_out), (dtype, nbits)
        back_w = np.full(n * stride, 7, dtype)
        fn(back_w.ctypes.data, w_hdr.ctypes.data if header else None, w_out.ctypes.data, n, rc_w, offset, stride, ops[1])
        rc2, back = pk.compact_narrow_unpack(out, n, rc, dtype, hdr=hdr if header else None, offset=offset, stride=stride, fill=7)
        assert np.array_equal(back, back_w), (dtype, nbits)
        if not header:
            assert np.array_equal(back[::stride], a[::stride])


def test_known_answers_on_the_gpu():
    """the hand-derived known-answer vectors of tests/test_known_answers.py (one stream per packer branch, derived from the reference's
    source text, not from any code of this repository) through the HIP path"""
    import test_known_answers as ka

    def cf(a, nbits, style1=False, has_missing=0, tag=0.0):
        return pk.compact_float_pack(a, nbits, style1=style1, has_missing=has_missing, tag=tag)

    def ci(a, nbits, op, header):
        return pk.compact_integer_pack(a, nbits, op, header=header)

    for name, v in ka.PACK_VECTORS.items():
        ka.check_pack_vector(name, v, cf, ci, pk.float_packer)
    for val in (0x1234, 0, 0xFFFF):
        words, zlng = ka.armn_constant_16x16(val)
        buf = np.zeros(16 * 16 // 2 + 8, np.uint32); buf[:128] = pc.tokens_to_words(np.full(256, val, np.uint16))
        assert pk.armn_compress(buf, 16, 16, 16) == zlng
        assert [int(x) for x in buf[:len(words)]] == words
    tok, words, zlng = ka.armn_one_step_16x16()
    buf = np.zeros(16 * 16 // 2 + 8, np.uint32); buf[:128] = pc.tokens_to_words(tok)
    assert pk.armn_compress(buf, 16, 16, 16) == zlng
    assert [int(x) for x in buf[:len(words)]] == words
    # the 5-bit width field (a difference beyond 16 bits) and a MINIMUM stream with its three tile forms (raw escape, constant, minimum + offsets)
    for (tok, words, zlng), (ni, nj) in ((ka.armn_wide_difference_16x16(), (16, 16)), (ka.armn_minimum_15x5(), (15, 5))):
        buf = np.zeros(ni * nj // 2 + 8, np.uint32); buf[:(ni * nj + 1) // 2] = pc.tokens_to_words(tok)
        assert pk.armn_compress(buf, ni, nj, 16) == zlng
        assert [int(x) for x in buf[:len(words)]] == words, [hex(int(x)) for x in buf[:len(words)]]
        assert pk.armn_uncompress(buf, ni, nj, 16) == ni * nj * 2                      # and back, through the decoder
        assert np.array_equal(buf[:(ni * nj + 1) // 2], pc.tokens_to_words(tok))
    # c_armn_compress32: the exponent plane (packTokensParallelogram_8) and the mantissa plane (packTokensParallelogram32) of a hand-derived record
    for with_mantissa in (False, True):
        f, pieces, zlng = ka.armn32_step_16x16(with_mantissa)
        zl, z = pk.armn_compress32(f, 16, 16, 32)
        assert zl == zlng
        for w0, words in pieces:
            assert [int(x) for x in z[w0:w0 + len(words)]] == words, (w0, [hex(int(x)) for x in z[w0:w0 + len(words)]])
        rc, back = pk.armn_uncompress32(z, 16, 16, 32)
        assert rc == 256 and np.array_equal(back.view(np.uint32), f.view(np.uint32))
    # the data part of an FST record (c_fstecr's packing switch) composed by hand from the vectors above
    for datyp, nbits, f, ni, nj, pieces, nw in ka.fst_data_part_vectors():
        w, d_out, b_out, buf = pk.fst_pack_data(f, ni, nj, 1, datyp, nbits)
        assert d_out == datyp and b_out == nbits and w > 0, (datyp, d_out, b_out, w)
        if nw is not None:
            assert w == nw
        for w0, want in pieces:
            assert [int(x) for x in buf[w0:w0 + len(want)]] == want, (datyp, w0, [hex(int(x)) for x in buf[w0:w0 + len(want)]])
        rc, back = pk.fst_unpack_data(buf, ni, nj, 1, d_out, b_out, dtype=f.dtype)
        assert rc == 0 or rc == ni * nj, rc
        if datyp != 6:
            assert np.array_equal(back.view(np.uint32), f.view(np.uint32)), datyp
    # compact_integer with a bit offset and a stride: the words around the tokens keep their bits
    a, before, after = ka.ci_offset_stride()
    out = np.array([before, 0x55555555], np.uint32)
    assert pk._lib().compact_integer(a.ctypes.data, None, out.ctypes.data, 3, 4, 8, 2, 1) == 4
    assert int(out[0]) == after and int(out[1]) == 0x55555555
    back = np.full(5, 7, np.uint32)
    assert pk._lib().compact_integer(back.ctypes.data, None, out.ctypes.data, 3, 4, 8, 2, 2) == 4
    assert [int(x) for x in back] == [1, 7, 2, 7, 3]
    # its sign sub-stream (pack1bitRLE): counted runs, the cut into 62s, the 255-repeat byte; decoded by the host walk and by the device kernels (k_rle_*)
    for ni, nj in ((16, 16), (32, 32)):
        f, pieces, zlng = ka.armn32_signed(ni, nj)
        zl, z = pk.armn_compress32(f, ni, nj, 32)
        assert zl == zlng
        for w0, words in pieces:
            assert [int(x) for x in z[w0:w0 + len(words)]] == words, (w0, [hex(int(x)) for x in z[w0:w0 + len(words)]])
        for rc, back in (pk.armn_uncompress32(z, ni, nj, 32), pk.armn_uncompress32_lng(z, 4 * ((zl + 3) // 4), ni, nj, 32)):
            assert rc == ni * nj and np.array_equal(back.view(np.uint32), f.view(np.uint32))


# ---------------------------------------------------------------------------------------------
# c_armn_compress32 / c_armn_uncompress32 (datyp 133): the IEEE-32 compressor
# ---------------------------------------------------------------------------------------------
import test_oracle_armn32 as ta32   # noqa: E402

A32_SHAPES = [(16, 16), (17, 19), (64, 48), (100, 31), (301, 200), (1000, 777), (3073, 40)]


@pytest.mark.parametrize("ni,nj", A32_SHAPES)
@pytest.mark.parametrize("kind", ta32.KINDS)
@pytest.mark.parametrize("znbits", [32, 24, 16])
def test_armn_compress32_bit_exact(ni, nj, kind, znbits):
    """the HIP encoder against the oracle's restatement of c_armn_compress32: same byte count, same bytes; then the HIP decoder (and the
    oracle's) bring back the field with its mantissas cut to znbits - 9 bits"""
    f = ta32.field32(ni, nj, kind, seed=ni + nj)
    zw = np.zeros(ni * nj * max(znbits, 8) // 32 + 1024, np.uint32)
    want = ta32.O().orc_armn_compress32(zw.ctypes.data, f.ctypes.data, ni, nj, 1, znbits)
    got, zg = pk.armn_compress32(f, ni, nj, znbits)
    assert got == want, (got, want)
    if want < 0:
        return
    assert np.array_equal(zg[:want // 4], zw[:want // 4]), np.nonzero(zg[:want // 4] != zw[:want // 4])[0][:5]
    rc, back = pk.armn_uncompress32(zg, ni, nj, znbits)
    assert rc == ni * nj
    assert np.array_equal(back.view(np.uint32), ta32.truncated(f, znbits).view(np.uint32)), int((back != ta32.truncated(f, znbits)).sum())
    back_o = np.zeros(ni * nj, np.float32)
    ta32.O().orc_armn_uncompress32(back_o.ctypes.data, zg.ctypes.data, ni, nj, 1, znbits)
    assert np.array_equal(back_o.view(np.uint32), back.view(np.uint32))
    # the stream's length given (the FST record's case): both tile chains are followed on the device -- the exact length, and an upper bound with zero words behind
    for device_walk in ("0", "1"):
        os.environ["EZHIP_A32_DEVICE_WALK"] = device_walk         # 1: the chains followed by the kernels of armn_compress UNCOMPRESS (opt-in: slower than the host walks at full size)
        try:
            for zbytes in (4 * ((want + 3) // 4), 4 * ((want + 3) // 4) + 4 * 37):
                rc2, back2 = pk.armn_uncompress32_lng(zg, zbytes, ni, nj, znbits)
                assert rc2 == ni * nj
                assert np.array_equal(back2.view(np.uint32), back.view(np.uint32)), (device_walk, zbytes, int((back2.view(np.uint32) != back.view(np.uint32)).sum()))
        finally:
            del os.environ["EZHIP_A32_DEVICE_WALK"]


def _signs_from_runs(n, runs):
    """+-1 per point: consecutive runs of the given lengths with alternating signs, the last one stretched / cut to n points"""
    s = np.ones(n, np.float32); pos = 0; sg = 1.0
    for r in runs:
        if pos >= n:
            break
        s[pos:pos + r] = sg; pos += r; sg = -sg
    s[pos:] = sg
    return s


RLE_PATTERNS = {
    # every branch of pack1bitRLE (armn_compress_32.c:827-901): raw tokens, count tokens 8 .. 62, the 62 + remainder split, remainders of 1 .. 7 points that leave as a
    # raw token reaching into the following runs, the 0xFF repeat (more than 256 points left after a 62), runs swallowed by a raw token, the field's end inside a raw token
    "short": [1, 2, 3, 4, 5, 6, 7] * 40,
    "around8": [7, 8, 9, 7, 8, 15, 14, 16, 6, 8] * 30,
    "around62": [61, 62, 63, 64, 65, 69, 70, 71, 62, 1, 62, 7, 63] * 12,
    "tails": [62 + k for k in range(0, 20)] + [124 + k for k in range(0, 20)] + [3, 1, 2] * 5 + [62 + 62 + k for k in range(1, 9)],
    "repeat255": [62 + 256, 62 + 257, 62 + 257 + 255, 62 + 256 + 255 * 3 + 5, 62 + 258 + 255 * 2, 1, 62 + 1000, 2, 3, 62 + 511, 62 + 512, 62 + 513],
    "swallow": [70, 1, 1, 1, 1, 1, 1, 1, 1, 9, 67, 2, 2, 2, 30, 66, 6, 8, 65, 3, 3, 9] * 15,
    "alternating": [1] * 3000,
    "one_change": [5000],
    "end_raw": [100] * 5 + [3],
}


@pytest.mark.parametrize("pattern", sorted(RLE_PATTERNS))
@pytest.mark.parametrize("ni,nj", [(64, 48), (301, 200), (1000, 777)])
def test_armn_compress32_sign_run_coder_on_the_device(pattern, ni, nj):
    """c_armn_compress32 of fields whose SIGNS follow crafted run lengths (the magnitudes are a smooth positive field): the device run-length coder (a prefix scan
    of seven-state maps) against the oracle's sequential restatement of pack1bitRLE, byte for byte; the host coder (EZHIP_A32_RLE_ENC_HOST=1) gives the same bytes"""
    n = ni * nj
    runs = list(RLE_PATTERNS[pattern])
    reps = 1
    while sum(runs) * reps < n and pattern not in ("one_change",):
        reps += 1
    sg = _signs_from_runs(n, runs * reps)
    if pattern == "end_raw":                                   # the last points: a short run that ends the field inside a raw token
        sg[-3:] = -sg[-4]
    f = (ta32.field32(ni, nj, "positive", seed=ni + nj) * sg).astype(np.float32)
    zw = np.zeros(n + 1024, np.uint32)
    want = ta32.O().orc_armn_compress32(zw.ctypes.data, f.ctypes.data, ni, nj, 1, 32)
    got, zg = pk.armn_compress32(f, ni, nj, 32)
    assert got == want, (pattern, got, want)
    if want > 0:
        assert np.array_equal(zg[:want // 4], zw[:want // 4]), (pattern, np.nonzero(zg[:want // 4] != zw[:want // 4])[0][:5])
        os.environ["EZHIP_A32_RLE_ENC_HOST"] = "1"
        try:
            got_h, zh = pk.armn_compress32(f, ni, nj, 32)
        finally:
            del os.environ["EZHIP_A32_RLE_ENC_HOST"]
        assert got_h == want and np.array_equal(zh[:want // 4], zw[:want // 4])
        rc, back = pk.armn_uncompress32(zg, ni, nj, 32)
        assert rc == n and np.array_equal(back.view(np.uint32), f.view(np.uint32))


def test_armn_compress32_refusals_and_full_size():
    f = ta32.random_bits_field(64, 48, seed=9)
    assert pk.armn_compress32(f, 64, 48, 32)[0] == -1                 # incompressible: as the oracle says
    assert pk.armn_compress32(ta32.field32(15, 40, "positive", 1), 15, 40, 32)[0] == -1
    # a full-size cfg2 output-like field (7200 x 3601), positive and mixed signs: HIP == oracle, round trip
    import ezcases as ec
    ni, nj = 7200, 3601
    for kind in ("positive", "mixed"):
        f = ta32.field32(ni, nj, kind, seed=5)
        zw = np.zeros(ni * nj + 1024, np.uint32)
        want = ta32.O().orc_armn_compress32(zw.ctypes.data, f.ctypes.data, ni, nj, 1, 32)
        got, zg = pk.armn_compress32(f, ni, nj, 32)
        assert got == want and want > 0
        assert np.array_equal(zg[:want // 4], zw[:want // 4])
        rc, back = pk.armn_uncompress32(zg, ni, nj, 32)
        assert rc == ni * nj and np.array_equal(back.view(np.uint32), f.view(np.uint32))
        for device_walk in ("0", "1"):
            os.environ["EZHIP_A32_DEVICE_WALK"] = device_walk
            try:
                rc, back = pk.armn_uncompress32_lng(zg, 4 * ((want + 3) // 4), ni, nj, 32)
                assert rc == ni * nj and np.array_equal(back.view(np.uint32), f.view(np.uint32)), device_walk
                rc, _ = pk.armn_uncompress32_lng(zg, 4 * (want // 8), ni, nj, 32)      # half the record: the chain leaves the stream, refused
                assert rc == -1, device_walk
            finally:
                del os.environ["EZHIP_A32_DEVICE_WALK"]


# ---------------------------------------------------------------------------------------------
# FST record framing: the data part of a record as c_fstecr builds it
# ---------------------------------------------------------------------------------------------
import fst_twin   # noqa: E402

FST_CASES = [(1, 12), (1, 16), (1, 24), (129, 12), (129, 16), (6, 12), (6, 16), (6, 20), (6, 28), (134, 12), (134, 16), (133, 32), (133, 24), (5, 32), (0, 32),
             (129, 24), (129, 32), (134, 20), (134, 24)]      # turbo types beyond 16 bits lose the flag (fstd98.c:934)


@pytest.mark.parametrize("datyp,nbits", FST_CASES)
@pytest.mark.parametrize("kind", ["smooth", "rough"])
def test_fst_record_data_part_float(datyp, nbits, kind):
    """ezhip_fst_pack_data against the test twin of c_fstecr's switch built from the oracle's packers: same final datyp, same word count, same
    words (the stream's last, undefined byte aside); ezhip_fst_unpack_data then returns what the oracle's unpackers return"""
    ni, nj, nk = 120, 75, 1
    n = ni * nj
    f = pc.float_field(n, seed=3) if kind == "smooth" else (ec_hash(n) * np.float32(1000.0)).astype(np.float32)
    w, d_out, b_out, got = pk.fst_pack_data(f, ni, nj, nk, datyp, nbits)
    ww, dw, want = fst_twin.pack(f, ni, nj, nk, datyp, nbits)
    assert (w, d_out) == (ww, dw), (w, ww, d_out, dw)
    # the words the reference DEFINES: the byte after an armn stream is undefined, and behind a refused compression the tail of the
    # record keeps what the first attempt left there
    m = {0: n * b_out // 32, 5: n, 1: (120 + n * b_out) // 32, 6: 3 + n // 2}.get(d_out, w - 2)
    assert np.array_equal(got[:m], want[:m]), (datyp, nbits, np.nonzero(got[:m] != want[:m])[0][:5])
    rc, back = pk.fst_unpack_data(got.copy(), ni, nj, nk, d_out, b_out)
    assert rc == 0
    back_w = fst_twin.unpack(want, ni, nj, nk, d_out, b_out)
    assert np.array_equal(back.view(np.uint32), back_w.view(np.uint32)), int((back != back_w).sum())
    if d_out in (0, 5) or (d_out == 133 and nbits == 32):
        assert np.array_equal(back.view(np.uint32), f.view(np.uint32))
    elif not (d_out == 129 and nbits < 16):      # (the reference reads a 129 record with fewer than 16 bits as contiguous tokens although they sit in 16-bit slots)
        scale = float(np.abs(f).max())
        mant = b_out - 9 if d_out == 133 else min(b_out, 23)
        assert np.abs(back - f).max() <= scale * 2.0 ** (-(mant - 1)) * 1.01, float(np.abs(back - f).max())


def ec_hash(n):
    import ezcases as ec
    return ec.hash_uniform(12, n)


@pytest.mark.parametrize("datyp,nbits", [(2, 12), (2, 16), (2, 31), (130, 12), (130, 16), (130, 24), (4, 12), (4, 24)])
def test_fst_record_data_part_integer(datyp, nbits):
    ni, nj, nk = 100, 60, 1
    n = ni * nj
    rng = np.random.default_rng(nbits)
    if datyp == 4:
        f = rng.integers(-(1 << (nbits - 1)) + 1, (1 << (nbits - 1)) - 1, n).astype(np.int32)
    else:
        i = np.arange(n) % ni; j = np.arange(n) // ni
        f = ((i * 7 + j * 3) % (1 << min(nbits, 15))).astype(np.int32)          # smooth enough for armn_compress
    w, d_out, b_out, got = pk.fst_pack_data(f, ni, nj, nk, datyp, nbits)
    ww, dw, want = fst_twin.pack(f, ni, nj, nk, datyp, nbits)
    assert (w, d_out) == (ww, dw)
    m = w - 2 if d_out > 128 else (n * nbits + 31) // 32
    assert np.array_equal(got[:m], want[:m]), np.nonzero(got[:m] != want[:m])[0][:5]
    rc, back = pk.fst_unpack_data(got.copy(), ni, nj, nk, d_out, b_out, dtype=np.int32)
    assert rc == 0 and np.array_equal(back, f)


# ---- element sizes other than 4 bytes and the missing-value flag (fstd98.c:808-826, :1133-1145, :1198-1231, :1263-1285; fst_missing.c) ----
def _fst_field(dtype, n, ni, seed):
    i = np.arange(n) % ni; j = np.arange(n) // ni
    if np.dtype(dtype).kind == "f":
        return (pc.float_field(n, seed=seed).astype(np.float64) * (1.0 + 1e-9 * np.arange(n))).astype(dtype)
    bits = {1: 6, 2: 11, 4: 13}[np.dtype(dtype).itemsize]
    base = ((i * 5 + j * 3) % (1 << bits)).astype(np.int64)
    if np.dtype(dtype).kind == "i":
        base = base - (1 << (bits - 1))
    return base.astype(dtype)


FST_EX_CASES = [(1, 16, np.float64), (1, 24, np.float64), (129, 16, np.float64), (129, 12, np.float64), (5, 64, np.float64), (5, 16, np.float32), (5, 24, np.float32),
                (2, 12, np.uint16), (2, 7, np.uint8), (130, 12, np.uint16), (130, 7, np.uint8), (130, 16, np.uint16), (4, 12, np.int16), (4, 7, np.int8), (4, 16, np.int32)]


@pytest.mark.parametrize("datyp,nbits,dtype", FST_EX_CASES)
@pytest.mark.parametrize("missing", [0, 64])
def test_fst_record_element_sizes_and_missing_values(datyp, nbits, dtype, missing):
    """ezhip_fst_pack_data_ex / ezhip_fst_unpack_data_ex on REAL*8, 16-bit and 8-bit arrays, with and without the missing-value flag, against the
    twin of c_fstecr / c_fstluk built from the oracle's packers and a numpy restatement of fst_missing.c"""
    ni, nj, nk = 96, 50, 1
    n = ni * nj
    f = _fst_field(dtype, n, ni, seed=nbits)
    if missing:
        f = f.copy()
        f[[5, 77, n // 2, n - 1]] = fst_twin.MAGIC[np.dtype(dtype)]
    assert pk.fst_force_missing_value_usage(bool(missing)) == (1 if missing else 0)
    try:
        w, d_out, b_out, got = pk.fst_pack_data(f, ni, nj, nk, datyp | missing, nbits)
        ww, dw, want = fst_twin.pack(f, ni, nj, nk, datyp | missing, nbits)
        assert w > 0 and (w, d_out) == (ww, dw), (w, ww, d_out, dw)
        base = d_out & ~64
        m = {0: n * b_out // 32, 1: (120 + n * b_out) // 32, 2: (n * b_out) // 32, 4: (n * b_out) // 32}.get(base, w - 2)
        if base == 5:
            m = n * b_out // 32
        assert np.array_equal(got[:m], want[:m]), (datyp, nbits, np.nonzero(got[:m] != want[:m])[0][:5])
        rc, back = pk.fst_unpack_data(got.copy(), ni, nj, nk, d_out, b_out, dtype=dtype)
        assert rc == 0
        back_w = fst_twin.unpack(want, ni, nj, nk, d_out, b_out, dtype=dtype)
        assert np.array_equal(back.view(np.uint8), back_w.view(np.uint8)), int((back != back_w).sum())
        # where the reference itself reads back what it wrote (not: signed 16- / 8-bit stand-ins, which stay -1, fst_missing.c:1037-1050; byte arrays with fewer
        # than 8 bits, fstd98.c:2321; datyp 129 below 16 bits, read as contiguous tokens)
        sane = not (np.dtype(dtype) in (np.dtype(np.int8), np.dtype(np.int16)) or (dtype == np.uint8 and nbits < 8) or ((d_out & ~64) == 129 and nbits < 16))
        if missing and (d_out & 64) and sane:
            mv = fst_twin.MAGIC[np.dtype(dtype)]
            assert (back[[5, 77, n // 2, n - 1]] == mv).all()          # the magic values come back where they were
            keep = np.ones(n, bool); keep[[5, 77, n // 2, n - 1]] = False
            if np.dtype(dtype).kind != "f" and not (dtype == np.uint8 and datyp == 2 and nbits < 8):
                assert np.array_equal(back[keep], f[keep])
        elif np.dtype(dtype).kind != "f" and not missing and not (dtype == np.uint8 and datyp == 2 and nbits < 8):
            # (a byte array written with fewer than 8 bits does not read back in the reference either: c_fstluk unpacks 8-bit tokens whatever the record
            # says, fstd98.c:2321 against :1222)
            assert np.array_equal(back, f)
    finally:
        pk.fst_force_missing_value_usage(False)


def test_fst_missing_value_encoders_against_the_restatement():
    """ezhip_fst_encode_missing_value / decode on every element type: stand-in values and counts as fst_missing.c computes them (numpy twin)"""
    import ctypes
    from librmn_amd.lib import load_library
    L = load_library()
    L.ezhip_fst_encode_missing_value.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 6
    L.ezhip_fst_decode_missing_value.argtypes = [ctypes.c_void_p] + [ctypes.c_int32] * 5
    L.ezhip_fst_decode_missing_value.restype = None
    pk.fst_force_missing_value_usage(True)
    try:
        for dtype, kind in [(np.float32, 1), (np.float64, 1), (np.int32, 4), (np.int16, 4), (np.int8, 4), (np.uint32, 2), (np.uint16, 2), (np.uint8, 2)]:
            for nbits in (6, 8, 12, 16):
                for variant in ("some", "first", "constant", "none"):
                    n = 500
                    a = _fst_field(dtype, n, 25, seed=nbits + 1)
                    mv = fst_twin.MAGIC[np.dtype(dtype)]
                    if variant == "constant":
                        a[:] = a[3]
                    if variant in ("some", "constant"):
                        a[[7, 100, 499]] = mv
                    if variant == "first":
                        a[[0, 1, 250]] = mv
                    eb = a.itemsize
                    dst = np.zeros_like(a)
                    cnt = L.ezhip_fst_encode_missing_value(dst.ctypes.data, a.ctypes.data, n, kind, nbits, int(eb == 1), int(eb == 2), int(eb == 8))
                    want, wc = fst_twin.mv_encode(a, nbits)
                    assert cnt == wc, (dtype, nbits, variant, cnt, wc)
                    if cnt:
                        assert np.array_equal(dst.view(np.uint8), want.view(np.uint8)), (dtype, nbits, variant)
                        back = dst.copy()
                        L.ezhip_fst_decode_missing_value(back.ctypes.data, n, kind, int(eb == 1), int(eb == 2), int(eb == 8))
                        assert np.array_equal(back.view(np.uint8), fst_twin.mv_decode(want).view(np.uint8)), (dtype, nbits, variant)
    finally:
        pk.fst_force_missing_value_usage(False)


@pytest.mark.parametrize("kind", ["smooth", "rough"])
@pytest.mark.parametrize("nbits", [16, 12])
@pytest.mark.parametrize("in_place", [False, True])
def test_fst_frame_of_a_device_resident_record(kind, nbits, in_place):
    """ezhip_fst_frame_record_dev: the record compact_float(16-bit slots) + armn_compress left in HBM becomes the data part c_fstecr(datyp 129) writes --
    length word + header + stream, or the re-packed datyp 1 form when compression did not pay -- without the field or the tokens visiting the host;
    against ezhip_fst_pack_data on the same field (which the twin of c_fstecr pins)"""
    ni, nj, nk = 120, 75, 1
    n = ni * nj
    f = pc.float_field(n, seed=5) if kind == "smooth" else (ec_hash(n) * np.float32(1000.0)).astype(np.float32)
    d_f = torch.from_numpy(f).cuda()
    front = 1
    buf = torch.zeros(front + 4 + n // 2 + 64, dtype=torch.int32, device="cuda")
    rec = buf[front:]
    zl = pk.pack16_compress_dev(rec, d_f, ni, nj, nbits)
    torch.cuda.synchronize()
    assert zl > 0 or kind == "rough", zl                     # (12-bit noise in 16-bit slots still compresses; 16-bit noise does not)
    w_h, d_h, b_h, want = pk.fst_pack_data(f, ni, nj, nk, 129, nbits)
    if in_place and zl < 0:
        w, d_out = pk.fst_frame_record_dev(buf, buf.numel(), rec, zl, ni, nj, nk, nbits)
        assert w == -1                                         # the datyp 1 form cannot be built over its own tokens
        return
    out = buf if in_place else torch.full((n + 256,), 0x55555555, dtype=torch.int32, device="cuda")
    w, d_out = pk.fst_frame_record_dev(out, out.numel(), rec, zl, ni, nj, nk, nbits)
    torch.cuda.synchronize()
    assert (w, d_out) == (w_h, d_h), (w, w_h, d_out, d_h)
    got = out.cpu().numpy().view(np.uint32)
    m = w - 2 if d_out == 129 else (120 + n * nbits) // 32
    assert np.array_equal(got[:m], want[:m]), np.nonzero(got[:m] != want[:m])[0][:5]
    if d_out == 129:
        assert not got[5 + (zl + 3) // 4:w].any()              # zero padding up to the length word's count


def test_fst_frame_rejects_overlapping_ranges():
    """only d_data + 1 == d_record frames in place: d_data == d_record (or any other overlap) would shift words while other threads still read them"""
    ni, nj, nk, nbits = 120, 75, 1, 16
    n = ni * nj
    d_f = torch.from_numpy(pc.float_field(n, seed=5)).cuda()
    buf = torch.zeros(8 + 4 + n // 2 + 64, dtype=torch.int32, device="cuda")
    rec = buf[2:]
    zl = pk.pack16_compress_dev(rec, d_f, ni, nj, nbits)
    assert zl > 0
    for data in (buf[2:], buf[0:], buf[3:]):                    # same start, two words in front, one word behind
        w, _ = pk.fst_frame_record_dev(data, data.numel(), rec, zl, ni, nj, nk, nbits)
        assert w == -1
    w, d_out = pk.fst_frame_record_dev(buf[1:], buf.numel() - 1, rec, zl, ni, nj, nk, nbits)      # the in-place form still works
    assert w > 0 and d_out == 129


@pytest.mark.parametrize("datyp,nbits", [(129, 16), (133, 32)])
def test_fst_unpack_damaged_length_word_is_refused(datyp, nbits):
    """data[0] comes from the file: a length word beyond the documented size of the data part (or too small to hold a record) is refused before anything
    is copied; a damaged sign sub-stream of a datyp 133 record stops at its own length instead of running through the buffer"""
    ni, nj, nk = 120, 75, 1
    f = pc.float_field(ni * nj, seed=7)
    if datyp == 133:
        f = (f - np.float32(f.mean())).astype(np.float32)        # both signs: the record carries a sign sub-stream
    w, d_out, b_out, got = pk.fst_pack_data(f, ni, nj, nk, datyp, nbits)
    assert d_out == datyp
    rc, back = pk.fst_unpack_data(got.copy(), ni, nj, nk, d_out, b_out)
    assert rc == 0
    for bad in (0, 3, got.size + 1000, 0x7FFFFFF0, 0xFFFFFFFF):
        dmg = got.copy(); dmg[0] = np.uint32(bad)
        rc, _ = pk.fst_unpack_data(dmg, ni, nj, nk, d_out, b_out)
        assert rc == -1, bad
    if datyp == 133:
        dmg = got.copy()
        lng_s = int(dmg[3]) >> 2                                  # [lng][w0][info][lng_s][sign runs ...]
        assert 0 < lng_s < w
        dmg[4:4 + lng_s] = 0                                      # all-zero sign stream = raw 7-bit groups only: needs 8/7 n bits, more than the sub-stream holds
        rc, _ = pk.fst_unpack_data(dmg, ni, nj, nk, d_out, b_out)
        assert rc == -1


DMIN_SHAPES = [(600, 601), (3100, 203), (1000, 777), (5200, 44), (50, 4000), (1280, 640), (355, 1002)]      # ni a multiple of 5: every tile of a row holds 25 points; nj with and without a last row of another height


@pytest.mark.parametrize("ni,nj", DMIN_SHAPES)
@pytest.mark.parametrize("kind", ["smooth", "noisy", "constant", "bigdiff"])
@pytest.mark.parametrize("nbits", [16, 9, 5])
def test_armn_uncompress_minimum_streams_by_composition(ni, nj, kind, nbits, monkeypatch):
    """MINIMUM streams (level FAST) whose rows hold whole tiles: the chain of tile headers by composition of the windows' maps (k_dmin_*, the default) against
    the serial chain kernel (EZHIP_DEC_NO_DMIN=1) and the oracle's tokens; widths 16 / 9 / 5 bits (tiles of 20 to 404 bits)"""
    tok = pc.token_field(ni, nj, nbits, kind, seed=ni + 5 * nj + nbits)
    z, zlng = _oracle_stream(tok, ni, nj, nbits, 0)
    assert int(z[0] & 15) == 3                                          # MINIMUM
    zwords = (zlng - 1) // 4 + 1
    d_z = torch.from_numpy(z[:zwords].view(np.int32).copy()).cuda()
    words = pc.tokens_to_words(tok)
    outs = []
    for no_dmin in (False, True):
        if no_dmin:
            monkeypatch.setenv("EZHIP_DEC_NO_DMIN", "1")
        d_out = torch.full((1 + ni * nj // 2 + 4,), -1, dtype=torch.int32, device="cuda")
        assert pk.armn_uncompress_dev(d_out, d_z, zwords, ni, nj, nbits) == ni * nj * 2
        got = d_out.cpu().numpy().view(np.uint32)
        assert np.array_equal(got[:words.size], words), (no_dmin, np.nonzero(got[:words.size] != words)[0][:5])
        assert np.all(got[1 + ni * nj // 2:] == 0xFFFFFFFF)
        outs.append(got)
    assert np.array_equal(outs[0], outs[1])


RAGGED_MIN_SHAPES = [(4001, 203, None), (3903, 77, None), (7201, 61, None), (5204, 96, None),           # rows of >= 768 tiles: the form's own threshold
                     (403, 300, 16), (1001, 160, 16), (603, 200, 16), (1604, 111, 16)]                    # narrow rows forced through it: most stretches outlast their row


@pytest.mark.parametrize("ni,nj,min_ntx", RAGGED_MIN_SHAPES)
@pytest.mark.parametrize("kind", ["smooth", "noisy"])
@pytest.mark.parametrize("nbits", [16, 11, 9])
def test_armn_uncompress_ragged_minimum_streams(ni, nj, min_ntx, kind, nbits, monkeypatch, capfd):
    """MINIMUM streams whose rows end on a narrower tile (ni not a multiple of 5; c_zfstlib.c:592-643): the canonical chain by composition, the row recurrence on
    top of it and -- where a stretch does not rejoin the canonical chain before its row ends (16- and 11-bit tokens: tile lengths are multiples of 5 but the raw
    tile's) -- rows walked explicitly through the composed maps (k_dsc_rows / dsc_walk_row); against the serial chain kernel and the oracle's tokens"""
    tok = pc.token_field(ni, nj, nbits, kind, seed=3 * ni + nj + nbits)
    z, zlng = _oracle_stream(tok, ni, nj, nbits, 0)
    assert int(z[0] & 15) == 3                                          # MINIMUM
    zwords = (zlng - 1) // 4 + 1
    d_z = torch.from_numpy(z[:zwords].view(np.int32).copy()).cuda()
    words = pc.tokens_to_words(tok)
    if min_ntx is not None:
        monkeypatch.setenv("EZHIP_DEC_RAGGED_MIN_NTX", str(min_ntx))
    outs = []
    for scan in ("2", "0"):
        monkeypatch.setenv("EZHIP_DEC_SCAN", scan)
        d_out = torch.full((1 + ni * nj // 2 + 4,), -1, dtype=torch.int32, device="cuda")
        capfd.readouterr()
        assert pk.armn_uncompress_dev(d_out, d_z, zwords, ni, nj, nbits) == ni * nj * 2
        torch.cuda.synchronize()
        if scan == "2" and zwords * 32 >= 64 * 2048 + 17 * 2048:        # (streams of fewer than 64 windows stay with the serial kernel)
            err = capfd.readouterr().err
            assert "scan form, field 0: ok 1" in err or "composed ragged form, field 0: ok 1" in err, err[-600:]      # a parallel form resolved the chain, not the serial kernel
        got = d_out.cpu().numpy().view(np.uint32)
        assert np.array_equal(got[:words.size], words), (scan, np.nonzero(got[:words.size] != words)[0][:5])
        assert np.all(got[1 + ni * nj // 2:] == 0xFFFFFFFF)
        outs.append(got)
    assert np.array_equal(outs[0], outs[1])


def test_armn_uncompress_minimum_batch_and_damaged(monkeypatch):
    """a batch of MINIMUM streams through the composed form, one of them cut short: the damaged one is reported (-1), the others decode"""
    ni, nj, nbits, F = 1000, 303, 16, 5
    toks = [pc.token_field(ni, nj, nbits, "smooth" if f % 2 else "noisy", seed=50 + f) for f in range(F)]
    streams = [_oracle_stream(t, ni, nj, nbits, 0) for t in toks]
    zw = max((zl - 1) // 4 + 1 for _, zl in streams)
    stride = zw + 64
    buf = np.zeros((F, stride), np.uint32)
    for f, (z, zl) in enumerate(streams):
        buf[f, :(zl - 1) // 4 + 1] = z[:(zl - 1) // 4 + 1]
    d_z = torch.from_numpy(buf.view(np.int32)).cuda()
    ostride = 1 + ni * nj // 2
    d_out = torch.zeros((F, ostride), dtype=torch.int32, device="cuda")
    assert pk.armn_uncompress_batch_dev(d_out, ostride, d_z, stride, zw, ni, nj, nbits, F) == ni * nj * 2
    got = d_out.cpu().numpy().view(np.uint32)
    for f in range(F):
        w = pc.tokens_to_words(toks[f])
        assert np.array_equal(got[f, :w.size], w), f
    # field 2 loses the second half of its stream (zeros): the chain ends long before the last tile's data is there -- values differ, nothing crashes,
    # both forms agree on what they return
    dmg = buf.copy(); zl2 = streams[2][1]; dmg[2, ((zl2 - 1) // 4 + 1) // 2:] = 0
    res = []
    for no_dmin in (False, True):
        if no_dmin:
            monkeypatch.setenv("EZHIP_DEC_NO_DMIN", "1")
        d_o = torch.zeros((F, ostride), dtype=torch.int32, device="cuda")
        rc = pk.armn_uncompress_batch_dev(d_o, ostride, torch.from_numpy(dmg.view(np.int32)).cuda(), stride, zw, ni, nj, nbits, F)
        g = d_o.cpu().numpy().view(np.uint32)
        for f in (0, 1, 3, 4):
            w = pc.tokens_to_words(toks[f])
            assert np.array_equal(g[f, :w.size], w), (no_dmin, f)
        res.append(rc)
    assert res[0] == res[1]


@pytest.mark.parametrize("ni,nj", [(1600, 801), (2560, 1280)])
def test_armn_uncompress32_whole_rows_take_the_device_walk(ni, nj):
    """ni - 1 a multiple of 3: every row of tiles of the exponent / mantissa planes is whole, the chains resolve by composition on the device (the default
    route of the length-aware entry for such shapes); same bits as the host walks and as the field"""
    for kind in ("positive", "mixed"):
        f = ta32.field32(ni, nj, kind, seed=ni + nj)
        zl, z = pk.armn_compress32(f, ni, nj, 32)
        assert zl > 0
        res = {}
        for route in ("default", "0", "1"):
            if route != "default":
                os.environ["EZHIP_A32_DEVICE_WALK"] = route
            try:
                rc, back = pk.armn_uncompress32_lng(z, 4 * ((zl + 3) // 4), ni, nj, 32)
            finally:
                os.environ.pop("EZHIP_A32_DEVICE_WALK", None)
            assert rc == ni * nj, (kind, route)
            res[route] = back
        for route in res:
            assert np.array_equal(res[route].view(np.uint32), f.view(np.uint32)), (kind, route)
        rc, _ = pk.armn_uncompress32_lng(z, 4 * (zl // 8), ni, nj, 32)          # half the record: refused on the default (device) route too
        assert rc == -1


@pytest.mark.parametrize("ni,nj", [(64, 48), (301, 200), (1000, 777), (2560, 1280)])
@pytest.mark.parametrize("kind", ["mixed", "stripes"])
def test_armn_uncompress32_sign_runs_on_the_device(ni, nj, kind):
    """the sign run lengths (unpack1bitRLE: raw groups of seven, counted runs, the 255-repeat token) decoded by the device kernels (the default) against the
    host thread (EZHIP_A32_RLE_HOST=1) and the field; whole and ragged rows of tiles (device and host walks of the planes)"""
    f = ta32.field32(ni, nj, kind, seed=3 * ni + nj)
    zl, z = pk.armn_compress32(f, ni, nj, 32)
    if zl < 0:
        pytest.skip("not compressible at this size")
    outs = []
    for host in (False, True):
        if host:
            os.environ["EZHIP_A32_RLE_HOST"] = "1"
        try:
            rc, back = pk.armn_uncompress32(z, ni, nj, 32)
            rc2, back2 = pk.armn_uncompress32_lng(z, 4 * ((zl + 3) // 4), ni, nj, 32)
        finally:
            os.environ.pop("EZHIP_A32_RLE_HOST", None)
        assert rc == ni * nj and rc2 == ni * nj
        assert np.array_equal(back.view(np.uint32), f.view(np.uint32)), (host, int((back.view(np.uint32) != f.view(np.uint32)).sum()))
        assert np.array_equal(back2.view(np.uint32), f.view(np.uint32)), host
        outs.append(back)
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


@pytest.mark.parametrize("ni,nj", [(301, 220), (300, 220), (1600, 801), (2560, 1280)])
@pytest.mark.parametrize("kind", ["positive", "mixed"])
def test_armn_uncompress32_record_and_field_on_the_device(ni, nj, kind):
    """c_armn_uncompress32_zdev: the record stays where c_armn_compress32_dev wrote it (HBM), the field comes back in HBM.  Whole-tile rows resolve on the
    device, ragged rows send the record down for the host's walk: both give the field's bits, and the device-written record equals the host entry's"""
    import torch
    f = ta32.field32(ni, nj, kind, seed=5 * ni + nj)
    zl, z = pk.armn_compress32(f, ni, nj, 32)
    d_f = torch.from_numpy(f).cuda()
    d_z = torch.zeros(ni * nj + 64, dtype=torch.int32, device="cuda")
    zl_dev = pk.armn_compress32_dev(d_z, d_f, ni, nj, 32)
    assert zl_dev == zl
    if zl < 0:
        pytest.skip("not compressible at this size")
    nw = (zl + 3) // 4
    assert np.array_equal(d_z[:nw].cpu().numpy().view(np.uint32), z[:nw])
    d_back = torch.full((ni * nj,), -7.0, dtype=torch.float32, device="cuda")
    before = d_z.clone()
    assert pk.armn_uncompress32_zdev(d_back, d_z, 4 * nw, ni, nj, 32) == ni * nj
    assert np.array_equal(d_back.cpu().numpy().view(np.uint32), f.view(np.uint32))
    assert torch.equal(before, d_z)                                   # the caller's record is read only
    assert pk.armn_uncompress32_zdev(d_back, d_z, 4 * (nw // 2), ni, nj, 32) == -1        # half a record
    d_bad = d_z.clone(); d_bad[0] = 0x7                               # not a PARALLELOGRAM32 record
    assert pk.armn_uncompress32_zdev(d_back, d_bad, 4 * nw, ni, nj, 32) == -1


RAGGED_SHAPES = [(902, 400), (1502, 300), (1001, 298), (3077, 130), (3002, 700)]      # (ni - 1) % 3 != 0: a narrower last tile per row; with and without a last row of another height


@pytest.mark.parametrize("ni,nj", RAGGED_SHAPES)
@pytest.mark.parametrize("kind", ["smooth", "noisy", "bigdiff"])
@pytest.mark.parametrize("mode", ["composed_only", "default", "ragged_off"])
def test_armn_uncompress_ragged_rows_by_composition(ni, nj, kind, mode, monkeypatch, capfd):
    """PARALLELOGRAM streams with ragged rows of tiles through the composed ragged form (k_drg_*: the canonical chain from the composition of the windows' maps, then
    the row recurrence): pushed onto every stream (the merged-exit form switched off through its row-length threshold), the default order of the forms, and
    the form switched off (merged-exit form or serial chain kernel): the oracle's tokens every time.  EZHIP_DEC_SCAN=2 prints each form's verdict: in the
    first mode the composed form has to be the one that resolved the chain"""
    if mode == "composed_only":
        monkeypatch.setenv("EZHIP_DEC_SCAN_MIN_NTX", "1000000"); monkeypatch.setenv("EZHIP_DEC_SCAN", "2"); monkeypatch.setenv("EZHIP_DEC_RAGGED_MIN_NTX", "96")
    if mode == "ragged_off":
        monkeypatch.setenv("EZHIP_DEC_NO_RAGGED", "1")
    nbits = 16
    tok = pc.token_field(ni, nj, nbits, kind, seed=2 * ni + nj)
    z, zlng = _oracle_stream(tok, ni, nj, nbits, 1)
    zwords = (zlng - 1) // 4 + 1
    d_z = torch.from_numpy(z[:zwords].view(np.int32).copy()).cuda()
    d_out = torch.full((1 + ni * nj // 2 + 4,), -1, dtype=torch.int32, device="cuda")
    assert pk.armn_uncompress_dev(d_out, d_z, zwords, ni, nj, nbits) == ni * nj * 2
    got = d_out.cpu().numpy().view(np.uint32)
    words = pc.tokens_to_words(tok)
    assert np.array_equal(got[:words.size], words), np.nonzero(got[:words.size] != words)[0][:5]
    assert np.all(got[1 + ni * nj // 2:] == 0xFFFFFFFF)
    if mode == "composed_only":
        err = capfd.readouterr().err
        assert "composed ragged form, field 0:" in err, err[-600:]
        # rows of tiles shorter than a stretch, long runs of empty tiles, escape tiles: the form may give up (the serial chain kernel then runs); these it resolves
        if (ni, nj, kind) in ((3077, 130, "smooth"), (3077, 130, "noisy"), (1502, 300, "smooth"), (3002, 700, "smooth")):
            assert "composed ragged form, field 0: ok 1" in err, err[-600:]


@pytest.mark.parametrize("ni,nj", [(1001, 600), (2000, 1000), (2561, 1281), (3002, 700)])
@pytest.mark.parametrize("kind", ["positive", "mixed"])
def test_armn_uncompress32_ragged_rows_on_the_device(ni, nj, kind, capfd):
    """the exponent and mantissa planes of a field whose rows of tiles end on a narrower tile: chains by composition + row recurrence on the device (the default
    for rows of >= 256 tiles), forced on, and the host's walk: the field's bits every time; record in host memory and in HBM.  With EZHIP_DEC_SCAN=2 the
    device's verdict is printed: the composed ragged form resolves both planes"""
    import torch
    f = ta32.field32(ni, nj, kind, seed=ni + 2 * nj)
    zl, z = pk.armn_compress32(f, ni, nj, 32)
    assert zl > 0
    nw = (zl + 3) // 4
    for route in ("default", "0", "1"):
        if route != "default":
            os.environ["EZHIP_A32_DEVICE_WALK"] = route
        try:
            rc, back = pk.armn_uncompress32_lng(z, 4 * nw, ni, nj, 32)
        finally:
            os.environ.pop("EZHIP_A32_DEVICE_WALK", None)
        assert rc == ni * nj, (kind, route)
        assert np.array_equal(back.view(np.uint32), f.view(np.uint32)), (kind, route)
    d_z = torch.from_numpy(z[:nw + 64].view(np.int32).copy()).cuda()
    d_back = torch.zeros(ni * nj, dtype=torch.float32, device="cuda")
    capfd.readouterr()
    os.environ["EZHIP_A32_DEVICE_WALK"] = "1"; os.environ["EZHIP_DEC_SCAN"] = "2"
    try:
        assert pk.armn_uncompress32_zdev(d_back, d_z, 4 * nw, ni, nj, 32) == ni * nj
    finally:
        os.environ.pop("EZHIP_A32_DEVICE_WALK", None); os.environ.pop("EZHIP_DEC_SCAN", None)
    assert np.array_equal(d_back.cpu().numpy().view(np.uint32), f.view(np.uint32))
    err = capfd.readouterr().err
    assert "composed ragged form, field 0:" in err, err[-800:]      # (each plane's verdict; a plane it leaves open is walked on the host)
    rc, _ = pk.armn_uncompress32_lng(z, 4 * (nw // 2), ni, nj, 32)          # half the record
    assert rc == -1


@pytest.mark.parametrize("ni,nj", [(2000, 1000), (1600, 801)])
def test_armn_uncompress32_damaged_records_device_and_host_routes_agree(ni, nj):
    """bit flips and a zeroed piece inside the planes of a record (ragged rows: composition + row recurrence; whole rows: composition): the device route and the
    host's walk read the same damaged stream the same way -- same return code, same bits when one comes back -- and nothing faults"""
    f = ta32.field32(ni, nj, "mixed", seed=77)
    zl, z = pk.armn_compress32(f, ni, nj, 32)
    assert zl > 0
    nw = (zl + 3) // 4
    rng = np.random.default_rng(5)
    for damage in ("bit_flips", "zeroed_piece", "late_flip"):
        d = z.copy()
        if damage == "bit_flips":
            for w in rng.integers(nw // 8, nw - 8, 6):
                d[w] ^= np.uint32(1 << int(rng.integers(0, 32)))
        elif damage == "zeroed_piece":
            a = int(nw * 0.6); d[a:a + 300] = 0
        else:
            d[nw - 40] ^= np.uint32(0x00010000)
        res = []
        for route in ("0", "1"):
            os.environ["EZHIP_A32_DEVICE_WALK"] = route
            try:
                rc, back = pk.armn_uncompress32_lng(d, 4 * nw, ni, nj, 32)
            finally:
                os.environ.pop("EZHIP_A32_DEVICE_WALK", None)
            res.append((rc, back))
        assert res[0][0] == res[1][0], (damage, res[0][0], res[1][0])
        if res[0][0] > 0:
            assert np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32)), damage
