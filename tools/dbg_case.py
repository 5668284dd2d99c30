import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, ctypes
from librmn_amd import ezscint as ez, packers as pk, load_library
import ezcases as ec
no, mo = 766, 122
ni, nj = max(64, int(no * 0.7) // 2 * 2), max(24, int(mo * 0.7) // 2 * 2)
gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0)
gdout = ez.ezqkdef(no, mo, "L", max(1, 17000 // mo), max(1, 35900 // no), 0, 0)
print("defset", ez.ezdefset(gdout, gdin))
ez.ezsetopt("interp_degree", "cubic"); ez.ezsetopt("polar_correction", "yes")
ez.use_stream(torch.cuda.current_stream().cuda_stream)
F = 4
srcs = [ec.synth_field(ni, nj, seed=300 + f) for f in range(F)]
srcs[1] = (ec.hash_uniform(7, ni * nj) * np.float32(1000.0)).astype(np.float32)
srcs[2] = np.full(ni * nj, 3.25, np.float32)
d_in = torch.stack([torch.from_numpy(a) for a in srcs]).cuda().contiguous()
n = no * mo; rs = 4 + n // 2 + 16
rec = torch.full((F, rs), 0x22222222, dtype=torch.int32, device="cuda")
for k in range(F):
    rc, zl = pk.ezsint_pack16_compress_batch_dev(rec[k:k+1], rs, d_in[k:k+1], 1, no, mo, 16)
    L = load_library(); L.ezhip_last_error.restype = ctypes.c_char_p
    print("field", k, "rc", rc, list(zl), L.ezhip_last_error())
for env in ({}, {"EZHIP_CFG5_NO_BB": "1"}):
    for k, v in env.items(): os.environ[k] = v
    rc, zl = pk.ezsint_pack16_compress_batch_dev(rec, rs, d_in, F, no, mo, 16)
    print("batch", env, "rc", rc, list(zl), L.ezhip_last_error())
    for fsub in (2, 3):
        rc, zl = pk.ezsint_pack16_compress_batch_dev(rec[:fsub], rs, d_in[:fsub], fsub, no, mo, 16)
        print("  batch of", fsub, "rc", rc, list(zl), L.ezhip_last_error())
