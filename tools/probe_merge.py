"""How fast do the speculative walks of a window merge?  (CPU, oracle stream; the measurement behind DESIGN_LOG.md 9 item 4's plan.)
For windows of DW bits of a PARALLELOGRAM stream: the walk from every entry offset e < ext under the interior point count; per window the number of
distinct exits, and the same for the composition of 8 windows.  python tools/probe_merge.py [noise] [ni nj]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ezcases as ec
import test_oracle_packers as top
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-5
ni, nj = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2400, 1201)
DW = int(os.environ.get("DW", "2048"))
full = ec.synth_field(7200, 3601, seed=40, noise=noise).reshape(3601, 7200).astype(np.float64)
f = full[1000:1000 + nj, 2000:2000 + ni]      # a piece of the cfg5 field: its smoothness per grid step, quantised over the whole field's range
tok = np.round((f - full.min()) / (full.max() - full.min()) * 65535.0).astype(np.uint16).reshape(nj, ni)
z = np.zeros(ni * nj + 64, np.uint32)
top.O().orc_armn_compress_setlevel(-1)
zl = top.O().orc_armn_encode(z.ctypes.data, np.ascontiguousarray(tok).ctypes.data, ni, nj, 16)
print("zlng", zl, "ratio", zl / (2.0 * ni * nj), "header %08x" % z[0])
assert (z[0] & 15) == 4
z1 = z[1:(zl + 3) // 4 + 40]
bits = np.unpackbits(z1.byteswap().view(np.uint8))            # MSB-first bit array
C = int(z1[0] >> 29); nbits = 16
body = 3 + (ni + nj - 1) * nbits
nb = bits.size - 64
# header value at every bit position
w = np.zeros(nb, np.int64)
for k in range(C): w = (w << 1) | bits[k:k + nb]
width = np.where((w == 15) | (w == 16), 17, w + 1)
step = C + np.where(w == 0, 0, 9 * width)
nx = np.arange(nb) + step
ext = C + 9 * (32 if C == 5 else 17)
nwin = (int(zl) * 8 - 32) // DW - 9
res1, res8 = [], []
E = np.arange(ext)
# the true chain (interior everywhere: row ends ignored for this statistic) for reference
exits = np.zeros((nwin + 9, ext), np.int64)
for wdx in range(nwin + 8):
    p = wdx * DW + E; end = (wdx + 1) * DW
    p = np.minimum(p, nb - 1)
    for _ in range(DW // C + 2):
        act = p < end
        if not act.any(): break
        p = np.where(act, nx[np.minimum(p, nb - 1)], p)
    exits[wdx] = p - end
first = body // DW + 1
d1 = np.array([len(np.unique(exits[wdx])) for wdx in range(first, nwin)])
comp = []
for wdx in range(first, nwin):
    e = E.copy()
    ok = np.ones(ext, bool)
    for k in range(8):
        ok &= e < ext
        e = np.where(ok, exits[wdx + k][np.minimum(e, ext - 1)], e)
    comp.append(len(np.unique(e[ok])) + (0 if ok.all() else 1000))
d8 = np.array(comp)
print(f"windows of {DW} bits: {d1.size}; distinct exits per window: 1 in {np.mean(d1 == 1) * 100:.1f} %, <= 2 in {np.mean(d1 <= 2) * 100:.1f} %, median {int(np.median(d1))}, max {d1.max()}")
print(f"eight windows composed: distinct exits 1 in {np.mean(d8 == 1) * 100:.1f} %, <= 2 in {np.mean(d8 <= 2) * 100:.1f} %, entries leaving the table in {np.mean(d8 >= 1000) * 100:.1f} % of the windows")
