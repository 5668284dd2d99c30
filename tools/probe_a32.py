"""c_armn_uncompress32 of one 7200 x 3601 field: the plain entry point (tile chains walked on the host) against the one with the stream's length (walked on the device)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import test_oracle_armn32 as ta32
from librmn_amd import packers as pk
ni, nj = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7200, 3601)      # 7201 3601: every row of tiles whole (ni - 1 a multiple of 3)
for kind in ("positive", "mixed"):
    f = ta32.field32(ni, nj, kind, seed=5)
    zl, z = pk.armn_compress32(f, ni, nj, 32)
    print(kind, "zlng", zl, "ratio", zl / (4.0 * ni * nj))
    for name, fn in (("host walk", lambda: pk.armn_uncompress32(z, ni, nj, 32)), ("length given (default route)", lambda: pk.armn_uncompress32_lng(z, 4 * ((zl + 3) // 4), ni, nj, 32)),
                     ("length given, host walk", lambda: (os.environ.__setitem__("EZHIP_A32_DEVICE_WALK", "0"), pk.armn_uncompress32_lng(z, 4 * ((zl + 3) // 4), ni, nj, 32), os.environ.__delitem__("EZHIP_A32_DEVICE_WALK"))[1]), ("device walk", lambda: (os.environ.__setitem__("EZHIP_A32_DEVICE_WALK", "1"), pk.armn_uncompress32_lng(z, 4 * ((zl + 3) // 4), ni, nj, 32), os.environ.__delitem__("EZHIP_A32_DEVICE_WALK"))[1])):
        ts = []
        for rep in range(4):
            t0 = time.perf_counter(); rc, back = fn(); ts.append(time.perf_counter() - t0)
        ok = rc == ni * nj and np.array_equal(back.view(np.uint32), f.view(np.uint32))
        print(f"  {name:30s} {min(ts) * 1e3:7.2f} ms   ok={ok}")
