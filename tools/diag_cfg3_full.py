"""cfg3 full field against the reference child: where the largest differences sit, per development variant.  python tools/diag_cfg3_full.py REF.npy "VAR=1" ..."""
import os, sys
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from librmn_amd import ezscint as ez
import ezcases as ec
ref = np.load(sys.argv[1]); ur, vr = ref[0].astype(np.float64), ref[1].astype(np.float64)
ni, nj, no, mo = 2560, 1280, 4000, 2000
ax, ay = ec.ze_axes(ni, nj)
uu, vv = ec.synth_wind(ni, nj, seed=3)
for a in (uu, vv):
    a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
V = np.maximum(np.hypot(ur, vr), 1e-3)
for var in (sys.argv[2:] or [""]):
    for kv in filter(None, var.split(",")):
        k, val = kv.split("="); os.environ[k] = val
    g_in = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); g_out = ez.ezqkdef(no, mo, "L", 9, 9, 0, 0)
    assert ez.ezdefset(g_out, g_in) == 1
    ez.ezsetopt("interp_degree", os.environ.get("DIAG_DEGREE", "cubic")); ez.ezsetopt("polar_correction", "yes")
    d_u = torch.from_numpy(uu).cuda(); d_v = torch.from_numpy(vv).cuda()
    o_u = torch.zeros(no * mo, device="cuda"); o_v = torch.zeros_like(o_u)
    for call in range(2):
        assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) >= 0
        torch.cuda.synchronize()
        u = o_u.cpu().numpy().astype(np.float64); v = o_v.cpu().numpy().astype(np.float64)
        e = np.maximum(np.abs(u - ur), np.abs(v - vr)) / V
        bad = np.nonzero(e > 5e-6)[0]
        print(f"[{var or 'defaults'}] call {call}: max {e.max():.3g} |V|; {bad.size} points above 5e-6; 99.99 pct {np.quantile(e, 0.9999):.3g}; values that differ at all: {int(np.count_nonzero(u != ur) + np.count_nonzero(v != vr))}", flush=True)
        if call == 1 and bad.size:
            d_x = torch.empty(no * mo, device="cuda"); d_y = torch.empty_like(d_x); ez.set_xy_dev(d_x, d_y); torch.cuda.synchronize()
            x = d_x.cpu().numpy(); y = d_y.cpu().numpy()
            o = bad[np.argsort(-e[bad])][:12]
            for k in o:
                print(f"   pt {k} (row {k // no}, col {k % no}) x {x[k]:.4f} y {y[k]:.4f} ref ({ur[k]:.5f}, {vr[k]:.5f}) got ({u[k]:.5f}, {v[k]:.5f}) err {e[k]:.3g}")
            print("   rows of bad points: min %d max %d; x range %.1f .. %.1f; y range %.1f .. %.1f" % (bad.min() // no, bad.max() // no, x[bad].min(), x[bad].max(), y[bad].min(), y[bad].max()))
            hist, edges = np.histogram(y[bad], bins=16, range=(1, nj)); print("   y histogram of bad points:", hist.tolist())
    ez.gdrls(g_out); ez.gdrls(g_in)
    for kv in filter(None, var.split(",")):
        os.environ.pop(kv.split("=")[0], None)
