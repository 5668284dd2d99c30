"""c_armn_compress32 / c_armn_uncompress32 at the cfg2 output size through the host API (includes the PCIe copies), with the oracle's time beside it."""
import os, sys, time
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np
import test_oracle_armn32 as ta32
from librmn_amd import packers as pk
ni, nj = 7200, 3601
f = ta32.field32(ni, nj, "mixed", seed=5)
for znbits in (32, 24):
    n, z = pk.armn_compress32(f, ni, nj, znbits)
    t0 = time.perf_counter()
    for _ in range(3): n, z = pk.armn_compress32(f, ni, nj, znbits)
    tc = (time.perf_counter() - t0) / 3
    rc, back = pk.armn_uncompress32(z, ni, nj, znbits)
    t0 = time.perf_counter()
    for _ in range(3): rc, back = pk.armn_uncompress32(z, ni, nj, znbits)
    tu = (time.perf_counter() - t0) / 3
    zw = np.zeros(ni * nj + 1024, np.uint32)
    t0 = time.perf_counter(); ta32.O().orc_armn_compress32(zw.ctypes.data, f.ctypes.data, ni, nj, 1, znbits); to = time.perf_counter() - t0
    bo = np.zeros(ni * nj, np.float32)
    t0 = time.perf_counter(); ta32.O().orc_armn_uncompress32(bo.ctypes.data, z.ctypes.data, ni, nj, 1, znbits); tuo = time.perf_counter() - t0
    print(f"znbits {znbits}: stream {n} bytes ({n / (4 * ni * nj):.3f} of the field); compress {tc * 1e3:.1f} ms (oracle, one core: {to * 1e3:.0f} ms), uncompress {tu * 1e3:.1f} ms (oracle {tuo * 1e3:.0f} ms)")
