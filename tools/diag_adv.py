"""one adversarial case of tests/test_gpu_wind_pin.py under development variants: python tools/diag_adv.py kind ni nj no mo "VAR=1" ..."""
import os, sys, subprocess
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
os.environ["EZHIP_TESTS_NO_STDERR_LOG"] = "1"
import numpy as np
kind = sys.argv[1]; ni, nj, no, mo = (int(a) for a in sys.argv[2:6])
import ezcases as ec, oraclelib as ol
import re
src = open(os.path.join(_R, "tests", "test_gpu_wind_pin.py")).read()
ns = {"np": np, "ec": ec}
for name in ("_meff", "_vortex", "_jets", "_outer_rows"):
    exec(re.search(r"^def %s\(.*?(?=^def |^@pytest)" % name, src, re.S | re.M).group(0), ns)
ax, ay = ec.ze_axes(ni, nj)
u, v = ns["_" + kind](ni, nj, 11 + ni); u[:, -1] = u[:, 0]; v[:, -1] = v[:, 0]
uu = np.ascontiguousarray(u.astype(np.float32).reshape(-1)); vv = np.ascontiguousarray(v.astype(np.float32).reshape(-1))
dst_ig = ol.cxgaig("L", -90.0, 0.0, 180.0 / (mo - 1), 360.0 / no)
np.savez("/tmp/case.npz", src_ni=ni, src_nj=nj, src_grtyp="Z", src_grref="E", src_ig=np.array(ec.E_IG), src_ax=ax, src_ay=ay, dst_ni=no, dst_nj=mo, dst_grtyp="L", dst_grref=" ",
         dst_ig=np.array(dst_ig), dst_ax=np.zeros(0, np.float32), dst_ay=np.zeros(0, np.float32), degree=3, polar=1, uu=uu, vv=vv)
r = subprocess.run([sys.executable, os.path.join(_R, "tests", "ref_child.py"), "uvint_case", "/tmp/case.npz", "/tmp/out.npz"], capture_output=True, text=True); assert r.returncode == 0, r.stderr
R = np.load("/tmp/out.npz"); ur, vr = R["ur"].astype(np.float64), R["vr"].astype(np.float64)
import torch
from librmn_amd import ezscint as ez
V = np.maximum(np.hypot(ur, vr), 1e-3)
for var in (sys.argv[6:] or [""]):
    for kv in filter(None, var.split(",")):
        k, val = kv.split("="); os.environ[k] = val
    gdin = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); gdout = ez.ezqkdef(no, mo, "L", *dst_ig)
    assert ez.ezdefset(gdout, gdin) == 1
    ez.ezsetopt("interp_degree", "cubic"); ez.ezsetopt("polar_correction", "yes")
    d_u = torch.from_numpy(uu).cuda(); d_v = torch.from_numpy(vv).cuda()
    o_u = torch.zeros(no * mo, device="cuda"); o_v = torch.zeros_like(o_u)
    for call in range(2):
        assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) >= 0
        torch.cuda.synchronize()
    pu = o_u.cpu().numpy().astype(np.float64); pv = o_v.cpu().numpy().astype(np.float64)
    d_x = torch.empty(no * mo, device="cuda"); d_y = torch.empty_like(d_x); ez.set_xy_dev(d_x, d_y); torch.cuda.synchronize()
    x = d_x.cpu().numpy(); y = d_y.cpu().numpy()
    meff = ns["_meff"](x, y, uu, vv, ni, nj, ay)
    e = np.maximum(np.abs(pu - ur), np.abs(pv - vr))
    rel = e / V
    bad = np.nonzero(rel > 1e-5)[0]
    okm = np.isfinite(meff) & (meff > 0)
    print(f"[{var or 'defaults'}] max {rel.max():.3g} |V|, {bad.size} points above 1e-5; max e / M_eff {np.max(e[okm] / meff[okm]):.3g}; max (e - 5e-7 |V|) / M_eff {np.max((e[okm] - 5e-7 * V[okm]) / meff[okm]):.3g}")
    for k in bad[np.argsort(-rel[bad])][:6]:
        print(f"   pt {k} x {x[k]:.4f} y {y[k]:.4f} ref ({ur[k]:.8g}, {vr[k]:.8g}) got ({pu[k]:.8g}, {pv[k]:.8g}) |V| {V[k]:.4g} M_eff {meff[k]:.5g} e {e[k]:.3g}")
    ez.gdrls(gdout); ez.gdrls(gdin)
    for kv in filter(None, var.split(",")):
        os.environ.pop(kv.split("=")[0], None)
