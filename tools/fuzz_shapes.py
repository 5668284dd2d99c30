import os, sys, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from librmn_amd import ezscint as ez
import oraclelib as ol, ezcases as ec
O = ol.oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ez.use_stream(torch.cuda.current_stream().cuda_stream)
DEGN = {0: "nearest", 1: "linear", 3: "cubic"}
bad = 0; ncase = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    st = rng.choice(["G", "L", "A", "B"])
    nj = int(rng.integers(12, 120)) * 2
    ni = int(rng.integers(24, 300))
    if st == "G": ni = 2 * nj if rng.random() < 0.5 else ni
    if st == "L":
        dlat = 180.0 / nj; 
        # global L on the 0.01 degree lattice
        k = int(rng.integers(20, 300)); dl = k / 100.0
        ni = int(360.0 / dl); nj = int(180.0 / dl) + 1
        if ni < 16 or nj < 12: continue
        src = (ni, nj, "L", (k, k, 0, 0))
    elif st == "B": src = (ni + 1 - (ni % 2), nj + 1, "B", (0, 0, 0, 0))
    else: src = (ni, nj, st, (0, 0, 0, 0))
    k2 = int(rng.integers(4, 40)) if os.environ.get("FUZZ_FINE") else int(rng.integers(15, 400)); dl2 = k2 / 100.0
    no = int(rng.integers(5, int(360.0 / dl2) + 1)); mo = int(rng.integers(3, int(180.0 / dl2) + 2))
    lat0 = int(rng.integers(0, max(1, int((180.0 - (mo - 1) * dl2) * 100)))) ; lon0 = int(rng.integers(0, 36000))
    dst = (no, mo, "L", (k2, k2, lat0, lon0))
    try:
        gi_h = ez.ezqkdef(src[0], src[1], src[2], *src[3]); go_h = ez.ezqkdef(dst[0], dst[1], dst[2], *dst[3])
        if gi_h < 0 or go_h < 0: continue
        assert ez.ezdefset(go_h, gi_h) == 1
        gi = ol.grid_define(src[0], src[1], src[2], src[3]); go = ol.grid_define(dst[0], dst[1], dst[2], dst[3])
        gs = O.orc_defset(go, gi)
    except Exception as e:
        print("skip", src, dst, e); continue
    F = int(rng.integers(1, 4))
    fields = [ec.synth_field(src[0], src[1], seed=200 + it * 7 + f) for f in range(F)]
    d_in = torch.stack([torch.from_numpy(f) for f in fields]).cuda().contiguous()
    for degree in (0, 1, 3):
        for polar in (1, 0):
            ez.ezsetopt("interp_degree", DEGN[degree]); ez.ezsetopt("polar_correction", "yes" if polar else "no")
            mode = ez.set_mode()
            d_out = torch.full((F, no * mo), -7.0, dtype=torch.float32, device="cuda")
            rc = ez.ezsint_batch_dev(d_out, d_in, F) if F > 1 else ez.ezsint_dev(d_out[0], d_in[0])
            torch.cuda.synchronize()
            got = d_out.cpu().numpy()
            opts = ol.default_opts(); opts.degre_interp = degree; opts.polar_correction = polar
            for f in range(F):
                want = np.zeros(no * mo, np.float32)
                rco = O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(want), ol.fptr(fields[f]))
                scale = np.maximum(np.abs(want), 1e-30)          # (round 6: the pure relative error, no floor)
                err = np.abs(got[f].astype(np.float64) - want) / scale
                ncase += 1
                if not (err.max() <= 1e-5) or rc != rco:
                    bad += 1
                    print("MISMATCH", src, dst, "deg", degree, "polar", polar, "mode", mode, "F", F, "f", f, "rc", rc, rco, "err", float(err.max()), "at", int(np.argmax(err)), flush=True)
print("cases", ncase, "bad", bad)
