import sys, os, time
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, os.path.join(_R, "tests")); sys.path.insert(0, _R)
import numpy as np, torch
import ezcases as ec, test_oracle_packers as top
from librmn_amd import packers as pk
ni, nj = (int(sys.argv[2]) if len(sys.argv) > 2 else 7200), (int(sys.argv[3]) if len(sys.argv) > 3 else 3601)
f = ec.synth_field(ni, nj, seed=40, noise=float(sys.argv[1]) if len(sys.argv) > 1 else 1e-5).astype(np.float64)
tok = np.round((f - f.min()) / (f.max() - f.min()) * 65535.0).astype(np.uint16)
O = top.O(); O.orc_armn_compress_setlevel(0)
z = np.zeros(ni * nj + 64, np.uint32)
zl = O.orc_armn_encode(z.ctypes.data, tok.ctypes.data, ni, nj, 16); O.orc_armn_compress_setlevel(-1)
zw = (zl - 1) // 4 + 1
print("MINIMUM stream: zlng", zl, "ratio", zl / (2.0 * ni * nj), "method", int(z[0] & 15))
d_z = torch.from_numpy(z[:zw].view(np.int32).copy()).cuda()
outs = {}
for mode in ("1", "0", "2"):
    os.environ["EZHIP_DEC_SCAN"] = mode
    d_o = torch.zeros(1 + ni * nj // 2, dtype=torch.int32, device="cuda")
    for rep in range(1 if mode == "2" else 4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = pk.armn_uncompress_dev(d_o, d_z, zw, ni, nj, 16)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    outs[mode] = d_o
    print(f"EZHIP_DEC_SCAN={mode}: {dt * 1e3:.2f} ms rc {rc}; equal to the first: {bool(torch.equal(d_o, outs['1']))}")
import packcases as pc
print("tokens right:", bool(np.array_equal(outs["1"].cpu().numpy().view(np.uint32)[:ni * nj // 2], pc.tokens_to_words(tok)[:ni * nj // 2])))
