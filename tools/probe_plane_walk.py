"""development: the device's tile walk of the planes of a c_armn_compress32 record with the forms' verdicts (EZHIP_DEC_SCAN=2); python tools/probe_plane_walk.py ni nj [kind]"""
import os, sys
os.environ["EZHIP_DEC_SCAN"] = "2"; os.environ["EZHIP_A32_DEVICE_WALK"] = "1"
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from librmn_amd import packers as pk
import test_oracle_armn32 as ta32
ni, nj = int(sys.argv[1]), int(sys.argv[2]); kind = sys.argv[3] if len(sys.argv) > 3 else "positive"
f = ta32.field32(ni, nj, kind, seed=5)
zl, z = pk.armn_compress32(f, ni, nj, 32)
print(f"{ni} x {nj} {kind}: record {zl} bytes", flush=True)
rc, back = pk.armn_uncompress32_lng(z, 4 * ((zl + 3) // 4), ni, nj, 32)
print("rc", rc, "bit-identical", bool(np.array_equal(back.view(np.uint32), f.view(np.uint32))))
