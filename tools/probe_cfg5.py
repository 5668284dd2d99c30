"""cfg5 pipeline probe (MI355X): the fused batch call, and armn_compress_dev on one full-size token field.
   python tools/probe_cfg5.py [F]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from librmn_amd import ezscint as ez, packers as pk
import ezcases as ec

F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ni, nj, no, mo = 4400, 2200, 7200, 3601
gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
ez.ezdefset(gdout, gdin)
stream = torch.cuda.current_stream(); ez.use_stream(stream.cuda_stream)
base = torch.from_numpy(ec.synth_field(ni, nj, seed=1000)).cuda()
gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
d_in = torch.empty((F, ni * nj), dtype=torch.float32, device="cuda")
for f in range(F):
    d_in[f] = base * (1.0 + 1e-3 * (torch.rand(ni * nj, device="cuda", generator=gen) - 0.5)) + 0.01 * f
n = no * mo
rs = 4 + n // 2 + 16
recs = torch.zeros((F, rs), dtype=torch.int32, device="cuda")

def run():
    rc, zl = pk.ezsint_pack16_compress_batch_dev(recs, rs, d_in, F, no, mo, 16)
    assert rc == 0
    return zl
for _ in range(3):
    zl = run()
t0 = time.perf_counter()
R = 5
for _ in range(R):
    zl = run()
dt = (time.perf_counter() - t0) / R
print(f"fused cfg5 pipeline, batch of {F}: {dt * 1e6 / F:.1f} us per field (wall), zlng[0] = {zl[0]}")

# one-pass encoder alone on the tokens of field 0 (plain 16-bit pack of the interpolated field)
z = torch.empty(n, dtype=torch.float32, device="cuda")
ez.ezsint_dev(z, d_in[0])
rec = torch.zeros(4 + n // 2 + 64, dtype=torch.int32, device="cuda")
assert pk.compact_float_pack_dev(z, rec, rec[4:], n, 16 + 64 * 16) != 0
d_z = torch.zeros(n // 2 + 32, dtype=torch.int32, device="cuda")
tok = rec[4:].clone()
for _ in range(3):
    zs = pk.armn_compress_dev(d_z, tok, no, mo, 16)
t0 = time.perf_counter()
for _ in range(10):
    zs = pk.armn_compress_dev(d_z, tok, no, mo, 16)
print(f"armn_compress_dev (one field, incl. sync): {(time.perf_counter() - t0) / 10 * 1e6:.1f} us, zlng = {zs}")
