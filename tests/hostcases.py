"""Inputs of the host set-up routines' tests (ez_nwtncof, ez_xpncof): deterministic, libm-free; shared by tests/golden/make_host_tables.py (which runs the
reference build on them) and tests/test_host_tables.py."""
import numpy as np
import ezcases as ec


def nwtncof_cases():
    """name -> (ax, ay, extension)"""
    out = {}
    for name, ni, nj, ext in (("reg0", 37, 21, 0), ("wrap1", 49, 25, 1), ("wrap2", 48, 24, 2), ("cfg3", 2560, 1280, 1)):
        if name == "cfg3":
            ax, ay = ec.ze_axes(ni, nj)
        else:
            u = ec.hash_uniform(100 + ni, ni).astype(np.float64); v = ec.hash_uniform(200 + nj, nj).astype(np.float64)
            if ext == 2:
                ax = (np.arange(ni) * (360.0 / ni) + u * 3.0).astype(np.float32)
            elif ext == 1:
                ax = (np.arange(ni) * (360.0 / (ni - 1))).astype(np.float32); ax[1:-1] += ((u[1:-1] - 0.5) * 4.0).astype(np.float32)
            else:
                ax = np.cumsum(0.5 + 1.5 * u).astype(np.float32)
            ay = (-80.0 + np.cumsum(2.0 + 4.0 * v)).astype(np.float32)
        out[name] = (np.ascontiguousarray(ax), np.ascontiguousarray(ay), ext)
    return out


def xpncof_cases():
    """name -> (ni, nj, grtyp, grref, ig1..ig4, ax, ay): every branch of ez_xpncof.c:48-226 the product supports"""
    out = {}
    for t in "ABG":
        for ig1 in (0, 1, 2):
            out[f"{t}{ig1}"] = (48 + (t == "B"), 24 + (t == "B"), t, " ", ig1, 0, 0, 0, None, None)
    # 'L' grids: global with a repeated meridian, global without, regional, a band that misses a pole; ig from the 0.01 degree encoding of cxgaig
    for name, ni, nj, ig in (("L_repeat", 361, 181, (100, 100, 0, 0)), ("L_norepeat", 360, 181, (100, 100, 0, 0)), ("L_regional", 80, 40, (50, 50, 2000, 3000)),
                             ("L_band", 360, 100, (100, 100, 0, 0)), ("L_cfg2", 7200, 3601, (5, 5, 0, 0))):
        out[name] = (ni, nj, "L", " ", *ig, None, None)
    ax, ay = ec.ze_axes(65, 32); out["ZE_global"] = (65, 32, "Z", "E", *ec.E_IG, ax, ay)
    ax, ay = ec.zereg_axes(51, 41); out["ZE_regional"] = (51, 41, "Z", "E", *ec.E_IG, ax, ay)
    for name, ni, nj, lon_span, lat0, lat1 in (("ZL_wrap1", 73, 37, 360.0, -90.0, 90.0), ("ZL_wrap2", 72, 37, 355.0, -90.0, 90.0), ("ZL_regional", 40, 30, 100.0, 10.0, 60.0),
                                              ("ZL_nosouth", 73, 30, 360.0, -40.0, 90.0)):
        ax = (np.arange(ni) * (lon_span / (ni - 1))).astype(np.float32)
        ay = (lat0 + np.arange(nj) * ((lat1 - lat0) / (nj - 1))).astype(np.float32)
        out[name] = (ni, nj, "Z", "L", 100, 100, 9000, 0, ax, ay)
    return out
