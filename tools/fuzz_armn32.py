"""Random shapes / field kinds / precisions of c_armn_compress32 + c_armn_uncompress32 against the oracle, bit for bit.
python tools/fuzz_armn32.py [ncases] [seed] [ni_min ni_max nj_min nj_max]   (ni >= 768: the device route of the length-aware entries, ragged rows included)"""
import os, sys
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
import test_oracle_armn32 as ta32
from librmn_amd import packers as pk
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lo_i, hi_i, lo_j, hi_j = (int(a) for a in sys.argv[3:7]) if len(sys.argv) > 6 else (16, 1400, 16, 500)
bad = 0
for k in range(ncases):
    ni = int(rng.integers(lo_i, hi_i)); nj = int(rng.integers(lo_j, hi_j))
    kind = ta32.KINDS[int(rng.integers(len(ta32.KINDS)))]
    znbits = int(rng.choice([32, 28, 24, 20, 16, 12]))
    f = ta32.field32(ni, nj, kind, seed=int(rng.integers(1 << 30)))
    if rng.random() < 0.2:
        f[rng.integers(0, f.size, 5)] = 0.0                      # exact zeros (exponent 0) in the planes
    zw = np.zeros(ni * nj * max(znbits, 8) // 32 + 1024, np.uint32)
    want = ta32.O().orc_armn_compress32(zw.ctypes.data, f.ctypes.data, ni, nj, 1, znbits)
    got, zg = pk.armn_compress32(f, ni, nj, znbits)
    ok = got == want
    if ok and want > 0:
        ok = np.array_equal(zg[:want // 4], zw[:want // 4])
        if ok:
            rc, back = pk.armn_uncompress32(zg, ni, nj, znbits)
            back_o = np.zeros(ni * nj, np.float32)
            ta32.O().orc_armn_uncompress32(back_o.ctypes.data, zg.ctypes.data, ni, nj, 1, znbits)
            ok = rc == ni * nj and np.array_equal(back.view(np.uint32), back_o.view(np.uint32))
            if ok:                                               # the length-aware entry (host walk held to the length; EZHIP_A32_DEVICE_WALK=1: the device walk)
                rc2, back2 = pk.armn_uncompress32_lng(zg, 4 * ((want + 3) // 4) + 4 * int(rng.integers(0, 3)), ni, nj, znbits)
                ok = rc2 == ni * nj and np.array_equal(back2.view(np.uint32), back_o.view(np.uint32))
            if ok:                                               # record and field in HBM
                nw = (want + 3) // 4
                d_z = torch.from_numpy(zg[:nw + 64].view(np.int32).copy()).cuda(); d_b = torch.zeros(ni * nj, dtype=torch.float32, device="cuda")
                rc3 = pk.armn_uncompress32_zdev(d_b, d_z, 4 * nw, ni, nj, znbits)
                ok = rc3 == ni * nj and np.array_equal(d_b.cpu().numpy().view(np.uint32), back_o.view(np.uint32))
    if not ok:
        bad += 1
        print("MISMATCH", dict(ni=ni, nj=nj, kind=kind, znbits=znbits, got=got, want=want), flush=True)
print(f"armn32: {ncases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
