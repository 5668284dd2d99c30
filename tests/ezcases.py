"""Shared grid/field case definitions for the parity tests (deterministic, libm-free inputs)."""
import numpy as np


def splitmix64(seed, n):
    """splitmix64 stream -> uint64[n] (SURVEY.md section 8d generator)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(seed, n):
    """float32 uniform in [0,1) from the top 24 bits."""
    return ((splitmix64(seed, n) >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def tri(t):
    """triangle wave in [0,1], period 1 (piecewise-polynomial, no libm)."""
    t = t - np.floor(t)
    return (1.0 - np.abs(2.0 * t - 1.0)).astype(np.float32)


def synth_field(ni, nj, seed, base=250.0, amp=60.0, noise=1e-3):
    """Smooth 'temperature-like' field + small hash noise, float32 [nj, ni] flattened (i fastest)."""
    i = np.arange(ni, dtype=np.float32)[None, :] / np.float32(ni)
    j = np.arange(nj, dtype=np.float32)[:, None] / np.float32(nj)
    smooth = tri(i * 2 + 0.13) * tri(j + 0.31) + 0.25 * tri(i * 5 + j * 3)
    z = np.float32(base) + np.float32(amp) * smooth.astype(np.float32)
    z = z * (np.float32(1.0) + np.float32(noise) * (hash_uniform(seed, ni * nj).reshape(nj, ni) - np.float32(0.5)))
    return np.ascontiguousarray(z.astype(np.float32).reshape(-1))


def synth_wind(ni, nj, seed):
    i = np.arange(ni, dtype=np.float32)[None, :] / np.float32(ni)
    j = np.arange(nj, dtype=np.float32)[:, None] / np.float32(nj)
    u = 30.0 * (tri(i * 3 + j) - 0.5) * 2 + 2.0 * (hash_uniform(seed, ni * nj).reshape(nj, ni) - 0.5)
    v = 30.0 * (tri(i + j * 2 + 0.4) - 0.5) * 2 + 2.0 * (hash_uniform(seed + 7, ni * nj).reshape(nj, ni) - 0.5)
    return (np.ascontiguousarray(u.astype(np.float32).reshape(-1)),
            np.ascontiguousarray(v.astype(np.float32).reshape(-1)))


def ze_axes(ni, nj):
    """Z-on-E axes of a global GEM-like grid: column ni duplicates column 1."""
    ax = (np.arange(ni, dtype=np.float64) * 360.0 / (ni - 1)).astype(np.float32)
    ay = (-90.0 + (np.arange(nj, dtype=np.float64) + 0.5) * 180.0 / nj).astype(np.float32)
    return ax, ay


def cloud_axes(ni, nj):
    """'Y' grid: ni*nj scattered points, ax = longitudes, ay = latitudes (hash-generated, libm-free)"""
    n = ni * nj
    lon = (hash_uniform(77, n).astype(np.float64) * 360.0).astype(np.float32)
    lat = (hash_uniform(78, n).astype(np.float64) * 178.0 - 89.0).astype(np.float32)
    lat[:4] = np.float32([90.0, -90.0, 89.97, -89.97]); lon[4] = np.float32(0.0); lon[5] = np.float32(359.99)
    return lon, lat


def zereg_axes(ni, nj):
    """regional Z-on-E axes (rotated frame): longitudes 150..200, latitudes -20..20, slightly stretched"""
    ax = (150.0 + np.arange(ni, dtype=np.float64) * (50.0 / (ni - 1)) + 0.2 * tri(np.arange(ni) / ni).astype(np.float64)).astype(np.float32)
    ay = (-20.0 + np.arange(nj, dtype=np.float64) * (40.0 / (nj - 1))).astype(np.float32)
    return ax, ay


def zl_axes(ni, nj):
    """Z-on-L axes (reference grid cxgaig('L', 0, 0, 1, 1)): longitudes 0..351 step 9 stretched a little, latitudes -90..90"""
    ax = (np.arange(ni, dtype=np.float64) * (360.0 / ni) + 0.3 * tri(np.arange(ni) / ni).astype(np.float64)).astype(np.float32)
    ay = (-90.0 + np.arange(nj, dtype=np.float64) * (180.0 / (nj - 1))).astype(np.float32)
    return ax, ay


E_IG = (1210, 1600, 57600, 54401)   # cxgaig('E', 31, -90, -20, -20)
N_IG = (455, 505, 2100, 1000)       # cxgaig('N', 50.5, 45.5, 100000, 21)
S_IG = (600, 400, 10000, 1500)      # cxgaig('S', 40, 60, 150000, 100)
N2_IG = (255, 305, 800, 1200)       # cxgaig('N', 30.5, 25.5, 120000, 8)

# name -> dict(src=(ni,nj,grtyp,ig,grref,axes_fn), dst=(ni,nj,grtyp,ig))
def scalar_cases():
    cases = {}
    cases["G_to_L"] = dict(src=(64, 32, "G", (0, 0, 0, 0), " ", None), dst=(90, 46, "L", (400, 400, 0, 0)))
    cases["G_to_L_odd"] = dict(src=(50, 26, "G", (0, 0, 0, 0), " ", None), dst=(72, 37, "L", (500, 500, 0, 0)))
    # L global, ni*dlon = 360 -> extension 2 (cfg1 shape): lat0=-89.55 dlat=.9 ; 40x200 -> use 9 deg
    cases["Lglobal_to_L"] = dict(src=(40, 20, "L", (900, 900, 450, 0), " ", None), dst=(80, 40, "L", (450, 450, 0, 0)))
    # L with duplicated last column -> extension 1
    cases["Lrepeat_to_L"] = dict(src=(41, 21, "L", (900, 900, 0, 0), " ", None), dst=(60, 31, "L", (600, 600, 0, 0)))
    # L regional -> extension 0, points outside -> DEHORS / extrapolation
    cases["Lregional_to_L"] = dict(src=(30, 20, "L", (200, 200, 11000, 20000), " ", None), dst=(40, 30, "L", (200, 200, 10400, 19000)))
    cases["A_to_L"] = dict(src=(48, 24, "A", (0, 0, 0, 0), " ", None), dst=(60, 31, "L", (600, 600, 0, 0)))
    cases["B_to_L"] = dict(src=(49, 25, "B", (0, 0, 0, 0), " ", None), dst=(60, 31, "L", (600, 600, 0, 0)))
    cases["ZE_to_L"] = dict(src=(65, 32, "Z", E_IG, "E", ze_axes), dst=(80, 41, "L", (450, 450, 0, 0)))
    cases["G_to_G"] = dict(src=(64, 32, "G", (0, 0, 0, 0), " ", None), dst=(96, 48, "G", (0, 0, 0, 0)))
    cases["L_to_G"] = dict(src=(40, 20, "L", (900, 900, 450, 0), " ", None), dst=(64, 32, "G", (0, 0, 0, 0)))
    # polar stereographic grids: N_IG = cxgaig('N', pi=50.5, pj=45.5, d60=100 km, dgrw=21), S_IG = cxgaig('S', 40, 60, 150 km, 100)
    cases["N_to_L"] = dict(src=(101, 91, "N", N_IG, " ", None), dst=(60, 40, "L", (100, 100, 12000, 20000)))      # partly outside the source: extrapolation
    cases["L_to_N"] = dict(src=(360, 180, "L", (100, 100, 50, 0), " ", None), dst=(101, 91, "N", N_IG))
    cases["S_to_L"] = dict(src=(81, 121, "S", S_IG, " ", None), dst=(90, 30, "L", (100, 100, 100, 0)))
    cases["G_to_S"] = dict(src=(128, 64, "G", (0, 0, 0, 0), " ", None), dst=(81, 121, "S", S_IG))
    # rotated Z-on-E TARGET (winds: c_ezgfwfllw)
    cases["L_to_ZE"] = dict(src=(40, 20, "L", (900, 900, 450, 0), " ", None), dst=(65, 32, "Z", E_IG), dst_ref="E", dst_axes=ze_axes)
    cases["G_to_ZE"] = dict(src=(64, 32, "G", (0, 0, 0, 0), " ", None), dst=(33, 16, "Z", E_IG), dst_ref="E", dst_axes=ze_axes)
    # Z-on-L SOURCE (global, irregular lat-lon axes) and a regional rotated Z-on-E SOURCE (a LAM grid: points outside -> extrapolation)
    cases["ZL_to_L"] = dict(src=(40, 21, "Z", (100, 100, 9000, 0), "L", zl_axes), dst=(60, 31, "L", (600, 600, 0, 0)))
    cases["ZEreg_to_L"] = dict(src=(51, 41, "Z", E_IG, "E", zereg_axes), dst=(50, 40, "L", (100, 100, 9000, 24000)))
    # Z-on-L TARGET (irregular lat-lon axes)
    cases["G_to_ZL"] = dict(src=(64, 32, "G", (0, 0, 0, 0), " ", None), dst=(40, 21, "Z", (100, 100, 9000, 0)), dst_ref="L", dst_axes=zl_axes)
    # 'Y' target: a cloud of points (incl. both poles and the seam)
    cases["G_to_Y"] = dict(src=(64, 32, "G", (0, 0, 0, 0), " ", None), dst=(37, 11, "Y", (0, 0, 0, 0)), dst_ref="L", dst_axes=cloud_axes)
    cases["ZE_to_Y"] = dict(src=(65, 32, "Z", E_IG, "E", ze_axes), dst=(37, 11, "Y", (0, 0, 0, 0)), dst_ref="L", dst_axes=cloud_axes)
    cases["N_to_N"] = dict(src=(101, 91, "N", N_IG, " ", None), dst=(60, 50, "N", N2_IG))
    return cases


def dst_spec(case):
    """target grid spec in the 6-tuple form of the source: (ni, nj, grtyp, ig, grref, axes_fn)"""
    return case["dst"] + (case.get("dst_ref", " "), case.get("dst_axes"))


# Yin-Yang 'U' grid: two Z-on-E subgrids of the same shape.  Yin's frame is the geographic one
# shifted (centre 0N 180E, equator on the equator); Yang's centre is 0N 0E and its equator runs through the poles.
YIN_IG = (900, 0, 43200, 43200)     # cxgaig('E', 0, 180, 0, 270)
YAN_IG = (900, 900, 14400, 0)       # cxgaig('E', 0, 0, 90, 0)


def yy_axes(ni, nj):
    """3-degree Yin-Yang subgrid axes: longitudes 36..324, latitudes -54..54 (overlap of 3 points beyond 45..315 / -45..45)"""
    ax = (36.0 + 3.0 * np.arange(ni, dtype=np.float64)).astype(np.float32)
    ay = (-54.0 + 3.0 * np.arange(nj, dtype=np.float64)).astype(np.float32)
    return ax, ay


YY_NI, YY_NJ = 97, 37


def yy_targets():
    return {"L": (72, 37, "L", (500, 500, 0, 0)), "G": (48, 24, "G", (0, 0, 0, 0)), "N": (41, 35, "N", (205, 175, 4000, 1000))}


def yy_fields():
    """scalar field and wind pair on the U grid: [Yin; Yang] concatenated, as the reference expects them"""
    z = np.concatenate([synth_field(YY_NI, YY_NJ, seed=3), synth_field(YY_NI, YY_NJ, seed=4)])
    (u0, v0), (u1, v1) = synth_wind(YY_NI, YY_NJ, seed=5), synth_wind(YY_NI, YY_NJ, seed=6)
    return z, np.concatenate([u0, u1]), np.concatenate([v0, v1])


YYT_NI, YYT_NJ = 117, 45


def yyt_axes(ni, nj):
    """a second Yin-Yang grid (2.5 degrees, same frames) used as a TARGET: longitudes 35..325, latitudes -55..55"""
    ax = (35.0 + 2.5 * np.arange(ni, dtype=np.float64)).astype(np.float32)
    ay = (-55.0 + 2.5 * np.arange(nj, dtype=np.float64)).astype(np.float32)
    return ax, ay


def hemi_cases():
    """hemispheric A / B grids (ig1 = 1 north, 2 south): scalar interpolation only"""
    c = {}
    c["Anord_to_L"] = dict(src=(48, 12, "A", (1, 0, 0, 0), " ", None), dst=(60, 31, "L", (600, 600, 0, 0)))
    c["Asud_to_L"] = dict(src=(48, 12, "A", (2, 0, 0, 0), " ", None), dst=(60, 31, "L", (600, 600, 0, 0)))
    c["Bnord_to_L"] = dict(src=(49, 13, "B", (1, 0, 0, 0), " ", None), dst=(60, 31, "L", (600, 600, 0, 0)))
    c["Bsud_to_L"] = dict(src=(49, 13, "B", (2, 0, 0, 0), " ", None), dst=(60, 31, "L", (600, 600, 0, 0)))
    c["L_to_Anord"] = dict(src=(40, 20, "L", (900, 900, 450, 0), " ", None), dst=(48, 12, "A", (1, 0, 0, 0)))
    c["G_to_Bsud"] = dict(src=(64, 32, "G", (0, 0, 0, 0), " ", None), dst=(49, 13, "B", (2, 0, 0, 0)))
    # y-inverted grids (ig2 = 1): the rows run from north to south
    c["Ainv_to_L"] = dict(src=(48, 24, "A", (0, 1, 0, 0), " ", None), dst=(60, 31, "L", (600, 600, 0, 0)))
    c["Ginv_to_L"] = dict(src=(64, 32, "G", (0, 1, 0, 0), " ", None), dst=(60, 31, "L", (600, 600, 0, 0)))
    c["AnordInv_to_L"] = dict(src=(48, 12, "A", (1, 1, 0, 0), " ", None), dst=(60, 31, "L", (600, 600, 0, 0)))
    c["L_to_Binv"] = dict(src=(40, 20, "L", (900, 900, 450, 0), " ", None), dst=(49, 25, "B", (0, 1, 0, 0)))
    c["G_to_Ginv"] = dict(src=(64, 32, "G", (0, 0, 0, 0), " ", None), dst=(48, 24, "G", (0, 1, 0, 0)))
    return c


def e_cases():
    """regular rotated 'E' grids.  As a SOURCE only without polar correction: the reference dereferences a NULL ay there
    (SURVEY appendix D.1); as a scalar TARGET without restriction."""
    c = {}
    c["E_to_L"] = dict(src=(41, 20, "E", E_IG, " ", None), dst=(60, 31, "L", (600, 600, 0, 0)), polar=(0,))
    c["L_to_E"] = dict(src=(40, 20, "L", (900, 900, 450, 0), " ", None), dst=(41, 20, "E", E_IG), polar=(1, 0))
    c["G_to_E"] = dict(src=(64, 32, "G", (0, 0, 0, 0), " ", None), dst=(33, 16, "E", E_IG), polar=(1, 0))
    return c
