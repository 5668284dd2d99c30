"""Shared inputs for the packer tests (deterministic, libm-free)."""
import numpy as np
import ezcases as ec


def float_field(n, seed, lo=250.0, span=60.0, noise=1e-3):
    t = np.arange(n, dtype=np.float32) / np.float32(max(n, 1))
    z = np.float32(lo) + np.float32(span) * (ec.tri(t * 3 + 0.2) * 0.7 + 0.3 * ec.tri(t * 17))
    z = z * (np.float32(1) + np.float32(noise) * (ec.hash_uniform(seed, n) - np.float32(0.5)))
    return np.ascontiguousarray(z.astype(np.float32))


def token_field(ni, nj, nbits, kind, seed):
    """16-bit token fields for armn_compress: smooth / noisy / constant / bigdiff (|Lorenzo diff| > 65535)"""
    i = np.arange(ni, dtype=np.float64)[None, :] / ni
    j = np.arange(nj, dtype=np.float64)[:, None] / nj
    top = (1 << nbits) - 1
    if kind == "smooth":
        v = 0.5 * top * (ec.tri(i * 2 + 0.1).astype(np.float64) * ec.tri(j * 1.5 + 0.3).astype(np.float64)) + 0.2 * top
        v = v + (ec.hash_uniform(seed, ni * nj).reshape(nj, ni).astype(np.float64) - 0.5) * max(1.0, top / 4096.0)
    elif kind == "noisy":
        v = ec.hash_uniform(seed, ni * nj).reshape(nj, ni).astype(np.float64) * top
    elif kind == "constant":
        v = np.full((nj, ni), top // 3, dtype=np.float64)
    elif kind == "bigdiff":
        v = np.where(((np.arange(ni)[None, :] + np.arange(nj)[:, None]) % 2) == 0, 0.0, float(top))
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(np.clip(np.rint(v), 0, top).astype(np.uint16).reshape(-1))


def tokens_to_words(tok):
    """two 16-bit tokens per 32-bit word, first in the high half (compact_float with 16-bit slots)"""
    n = tok.size
    t = np.zeros(n + (n & 1), np.uint32)
    t[:n] = tok
    return ((t[0::2] << np.uint32(16)) | t[1::2]).astype(np.uint32)
