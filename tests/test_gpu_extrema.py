"""compact_float's first pass in the cfg5 pipeline: the minimum and the maximum of the field c_ezsint WOULD produce (compact.tmplc:173-204 on the
output of ezsint.c), found from bounds of the source windows and the exact evaluation of the few windows that can hold an extremum
(ezhip_ezsint_batch_minmax_bb_dev, k_bb_* in ez_kernels.hip) -- against the pass that interpolates every point (ezhip_ezsint_batch_minmax_only_dev),
and that one against the stored field of c_ezsint_batch_dev.  Bit for bit."""
import os
import numpy as np
import pytest
import torch

import ezcases as ec
from librmn_amd import ezscint as ez

pytestmark = pytest.mark.gpu


def _fields(ni, nj):
    """source fields that stress the pruning: where the extremum sits, how many windows can hold it"""
    i = np.arange(ni, dtype=np.float32)[None, :]; j = np.arange(nj, dtype=np.float32)[:, None]
    smooth = ec.synth_field(ni, nj, seed=41).reshape(nj, ni)
    out = {"smooth": smooth}
    out["noisy"] = smooth * (1.0 + 1e-3 * (ec.hash_uniform(3, ni * nj).reshape(nj, ni) - 0.5))
    out["constant"] = np.full((nj, ni), 271.25, np.float32)
    blob = np.zeros((nj, ni), np.float32)                                   # precipitation-like: zero almost everywhere, overshoot below zero at the rims
    blob[nj // 3: nj // 3 + 9, ni // 5: ni // 5 + 14] = 37.5
    blob[nj // 2, (ni * 3) // 4] = 1.0e3
    out["blob"] = blob
    out["ramp_to_pole"] = (j * np.float32(1.5) + np.float32(0.0) * i).astype(np.float32)          # extrema in the first / last source rows (polar strips)
    seam = smooth.copy(); seam[nj // 2, 0] = 900.0; seam[nj // 2 + 5, ni - 1] = -900.0             # extrema on either side of the longitude seam
    out["seam"] = seam
    spike = smooth.copy(); spike[3, ni // 2] = 5.0e4; spike[nj - 4, ni // 3] = -5.0e4
    out["spikes_near_poles"] = spike
    nan = smooth.copy(); nan[nj // 2, ni // 2] = np.nan; nan[nj // 4, ni // 4] = np.inf
    out["nonfinite"] = nan
    out["noise"] = (ec.hash_uniform(5, ni * nj).reshape(nj, ni) * np.float32(1000.0)).astype(np.float32)   # no extremum stands out: expected to be flagged
    return {k: np.ascontiguousarray(v, np.float32).reshape(-1) for k, v in out.items()}


PAIRS = [
    ("G 360x181 -> L 520x261 global", (360, 181, "G", 0, 0, 0, 0), (520, 261, "L", 69, 69, 0, 0)),
    ("L 256x128 -> L regional", (256, 128, "L", 140, 140, 0, 0), (300, 140, "L", 50, 50, 30, 40)),
    ("A 288x144 -> G 400x200", (288, 144, "A", 0, 0, 0, 0), (400, 200, "G", 0, 0, 0, 0)),
    ("B 361x181 -> L 600x301 odd sizes", (361, 181, "B", 0, 0, 0, 0), (600, 301, "L", 60, 60, 0, 0)),
]


@pytest.mark.parametrize("degree", ["cubic", "linear", "nearest"])
@pytest.mark.parametrize("polar", ["yes", "no"])
@pytest.mark.parametrize("name,src,dst", PAIRS, ids=[p[0] for p in PAIRS])
def test_extrema_from_bounds_equal_the_interpolating_pass(name, src, dst, degree, polar):
    gi = ez.ezqkdef(src[0], src[1], src[2], *src[3:]); go = ez.ezqkdef(dst[0], dst[1], dst[2], *dst[3:])
    assert ez.ezdefset(go, gi) == 1
    assert ez.ezsetopt("interp_degree", degree) == 0 and ez.ezsetopt("polar_correction", polar) == 0
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    try:
        fl = _fields(src[0], src[1])
        names = list(fl)
        d_in = torch.stack([torch.from_numpy(fl[k]) for k in names]).cuda().contiguous()
        F = len(names)
        rc_i, mn_i, mx_i, _ = ez.ezsint_batch_extrema_dev(d_in, F, "interp")
        if rc_i == -2:
            pytest.skip("grid pair not on the single-launch k_sepx path")
        assert rc_i == 0
        # the interpolating pass itself against the stored field
        out = torch.empty((F, dst[0] * dst[1]), dtype=torch.float32, device="cuda")
        assert ez.ezsint_batch_dev(out, d_in, F) == 0
        o = out.cpu().numpy()
        for f, k in enumerate(names):
            assert np.nanmin(o[f]) == mn_i[f], (k, np.nanmin(o[f]), mn_i[f])
            assert np.nanmax(o[f]) == mx_i[f], (k, np.nanmax(o[f]), mx_i[f])
        for force in (False, True):
            if force:
                os.environ["EZHIP_BB_FORCE_ALL"] = "1"
            try:
                rc_b, mn_b, mx_b, flags = ez.ezsint_batch_extrema_dev(d_in, F, "bounds")
            finally:
                os.environ.pop("EZHIP_BB_FORCE_ALL", None)
            assert rc_b == 0, rc_b
            for f, k in enumerate(names):
                if flags[f]:
                    # only fields without a clear extremum may be handed back ("constant": when the plan's weights do not sum to 1 to REAL*8 rounding --
                    # the regular-grid cubic of A / B / L sources -- a window of one value is not known to interpolate to that value)
                    assert k in ("noise", "noisy", "nonfinite", "constant") and not force, (k, force)
                    continue
                assert mn_b[f] == mn_i[f] and mx_b[f] == mx_i[f], (k, force, mn_b[f], mn_i[f], mx_b[f], mx_i[f])
            if not force:
                assert not flags[names.index("smooth")] and not flags[names.index("blob")]
                if src[2] == "G":
                    assert not flags[names.index("constant")]
    finally:
        ez.ezsetopt("interp_degree", "cubic"); ez.ezsetopt("polar_correction", "yes")


def test_extrema_from_bounds_full_size_cfg2():
    """BASELINE cfg2 / cfg5 shape: G 4400x2200 -> L 7200x3601 bicubic, polar correction on"""
    ni, nj, no, mo = 4400, 2200, 7200, 3601
    gi = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); go = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
    assert ez.ezdefset(go, gi) == 1
    ez.ezsetopt("interp_degree", "cubic"); ez.ezsetopt("polar_correction", "yes")
    ez.use_stream(torch.cuda.current_stream().cuda_stream)
    base = torch.from_numpy(ec.synth_field(ni, nj, seed=1000)).cuda()
    gen = torch.Generator(device="cuda"); gen.manual_seed(7)
    F = 4
    d_in = torch.empty((F, ni * nj), dtype=torch.float32, device="cuda")
    for f in range(F):
        d_in[f] = base * (1.0 + 1e-3 * (torch.rand(ni * nj, device="cuda", generator=gen) - 0.5)) + 0.01 * f
    d_in[3] = torch.from_numpy(ec.synth_field(ni, nj, seed=2)).cuda()
    rc_i, mn_i, mx_i, _ = ez.ezsint_batch_extrema_dev(d_in, F, "interp")
    rc_b, mn_b, mx_b, flags = ez.ezsint_batch_extrema_dev(d_in, F, "bounds")
    assert rc_i == 0 and rc_b == 0
    assert not flags.any(), flags
    assert np.array_equal(mn_i.view(np.uint32), mn_b.view(np.uint32)) and np.array_equal(mx_i.view(np.uint32), mx_b.view(np.uint32)), (mn_i, mn_b, mx_i, mx_b)
