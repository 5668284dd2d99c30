"""Times c_ezsint_dev from a Yin-Yang 'U' source (two Z-on-E subgrids) to the cfg2 target (development tool)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import ezcases as ec
from librmn_amd import ezscint as ez

ni, nj = 2881, 961            # 0.09375-degree Yin-Yang subgrids: lon 45-3dx .. 315+3dx, lat -45-3dy .. 45+3dy
dx = 270.0 / (ni - 7)
ax = (45.0 - 3 * dx + dx * np.arange(ni, dtype=np.float64)).astype(np.float32)
dy = 90.0 / (nj - 7)
ay = (-45.0 - 3 * dy + dy * np.arange(nj, dtype=np.float64)).astype(np.float32)
gy = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.YIN_IG, ax, ay); ga = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.YAN_IG, ax, ay)
gu = ez.ezgdef_supergrid(ni, 2 * nj, "U", "F", 1, [gy, ga])
no, mo = 7200, 3601
go = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
assert ez.ezdefset(go, gu) == 1
ez.use_stream(torch.cuda.current_stream().cuda_stream)
for deg in ("cubic", "linear", "nearest"):
    ez.ezsetopt("interp_degree", deg)
    t0 = time.perf_counter(); assert ez.prepare_set() == 0; torch.cuda.synchronize(); tp = time.perf_counter() - t0
    d_in = torch.rand(2 * ni * nj, dtype=torch.float32, device="cuda"); d_out = torch.empty(no * mo, dtype=torch.float32, device="cuda")
    for _ in range(5): ez.ezsint_dev(d_out, d_in)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ez.ezsint_dev(d_out, d_in)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{deg}: plan {tp * 1e3:.0f} ms (first degree only), {dt * 1e6:.0f} us per field = {no * mo / dt / 1e9:.1f} Gpoints/s, "
          f"{(8 * ni * nj + 4 * no * mo) / dt / 1e9:.0f} GB/s algorithmic")
