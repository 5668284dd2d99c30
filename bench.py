#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native EZ interpolation hot path.

Workload (BASELINE.json configs[1] / configs[3]): c_ezsint bicubic, 4400x2200 global Gaussian 'G'
-> 7200x3601 0.05-degree lat-lon 'L', default options (cubic, polar_correction=yes), synthetic
"temperature-like" fields (tests/ezcases.py generator, libm-free).  One STEP = one pass of the hot
path over one batch of FIELDS_PER_STEP distinct device-resident source fields per GPU (fields shard
by record across ranks, no data-path collective: weak scaling; 8 GPUs x 32 fields = the 256-field
batch of configs[3]).  Inputs are resident in HBM when the timed region starts.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     -- k_sepx<3,16>: one launch interpolates the step's whole batch (blockIdx.z = field), so
                  algorithmic bytes per launch = fields x (4*ni_s*nj_s + 4*npts_out = 142.43 MB)
                  / average launch duration, HIP events on the launch stream
  cpu_baseline -- the reference's own c_ezsint (oracle/_ref/libezref.so, 1 thread) or, if that
                  build is absent, the oracle port, on a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np   # noqa: E402

NI_S, NJ_S, NI_D, NJ_D = 4400, 2200, 7200, 3601
L_IG = (5, 5, 0, 0)                       # 'L' lat0=-90 lon0=0 dlat=dlon=0.05
NPTS_OUT = NI_D * NJ_D
ALGO_BYTES = 4 * NI_S * NJ_S + 4 * NPTS_OUT   # 142,428,800 B per field
HBM_PEAK_GBPS = 8000.0                        # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(sample_fields=4):
    """reference c_ezsint steady state (x,y cached), 1 thread, on `sample_fields` fields"""
    import ezcases as ec
    zin = ec.synth_field(NI_S, NJ_S, seed=2)
    zout = np.zeros(NPTS_OUT, np.float32)
    try:
        from reflib import ref, have_ref, fptr
        if have_ref():
            L = ref()
            gdin = L.c_ezqkdef(NI_S, NJ_S, b"G", 0, 0, 0, 0, 0)
            gdout = L.c_ezqkdef(NI_D, NJ_D, b"L", L_IG[0], L_IG[1], L_IG[2], L_IG[3], 0)
            L.c_ezdefset(gdout, gdin)
            t0 = time.perf_counter()
            L.c_ezsint(fptr(zout), fptr(zin))              # first call: lat/lon + locate + zones
            first = time.perf_counter() - t0
            t0 = time.perf_counter()
            for _ in range(sample_fields):
                L.c_ezsint(fptr(zout), fptr(zin))
            dt = (time.perf_counter() - t0) / sample_fields
            return {"value": NPTS_OUT / dt / 1e6, "unit": "Mpoints/s", "cores": 1, "kind": "reference", "s_per_field": dt, "first_call_s": first,
                    "sample": f"{sample_fields} fields cfg2 steady-state c_ezsint, oracle/_ref/libezref.so"}
    except Exception as e:   # noqa: BLE001
        sys.stderr.write(f"cpu_baseline: reference build unusable ({e}); timing the oracle port\n")
    import oraclelib as ol
    O = ol.oracle()
    gi = ol.grid_define(NI_S, NJ_S, "G"); go = ol.grid_define(NI_D, NJ_D, "L", L_IG)
    gs = O.orc_defset(go, gi)
    opts = ol.default_opts()
    O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(zout), ol.fptr(zin))
    t0 = time.perf_counter()
    for _ in range(sample_fields):
        O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(zout), ol.fptr(zin))
    dt = (time.perf_counter() - t0) / sample_fields
    return {"value": NPTS_OUT / dt / 1e6, "unit": "Mpoints/s", "cores": 1, "kind": "port", "s_per_field": dt,
            "sample": f"{sample_fields} fields cfg2 steady-state orc_ezsint"}


def cpu_baseline_all_cores(per_thread_fields=3):
    """the CPU restatement over fields on ALL host cores (one field per thread at a time; the reference itself is single
    threaded -- its OpenMP pragmas are disabled -- and not re-entrant on one grid set, the oracle port is after its first call)"""
    import threading
    import ezcases as ec, oraclelib as ol
    nthreads = max(1, min(os.cpu_count() or 1, 32))
    O = ol.oracle()
    gi = ol.grid_define(NI_S, NJ_S, "G"); go = ol.grid_define(NI_D, NJ_D, "L", L_IG)
    gs = O.orc_defset(go, gi)
    opts = ol.default_opts()
    zin = ec.synth_field(NI_S, NJ_S, seed=2)
    outs = [np.zeros(NPTS_OUT, np.float32) for _ in range(nthreads)]
    O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(outs[0]), ol.fptr(zin))            # first call: locate + zones (cached in the set)

    def work(k):
        for _ in range(per_thread_fields):
            O.orc_ezsint(gs, ctypes.byref(opts), ol.fptr(outs[k]), ol.fptr(zin))
    ths = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    nf = nthreads * per_thread_fields
    return {"value": NPTS_OUT * nf / dt / 1e6, "unit": "Mpoints/s", "cores": nthreads, "nproc": os.cpu_count(), "kind": "port",
            "sample": f"{nf} fields cfg2 orc_ezsint, {dt:.1f} s"}


def pack_cpu_baseline(field, ref_interp_s):
    """the pack half of the metric on the CPU, 1 thread, ONE 7200 x 3601 field (the timed batch's checked output, copied to the host): the oracle's
    compact_float (compact.tmplc:143-334; 16-bit tokens in 16-bit slots) and armn_compress (c_zfstlib.c:67-203) -- the packers' sources include the
    un-vendored App.h, so the CPU side is the oracle port, not a reference build -- and cfg5's chain (fstd98.c:1170-1172) = the reference's c_ezsint
    (`ref_interp_s`, from cpu_baseline) + those two."""
    import oraclelib as ol
    O = ol.oracle()
    n = field.size
    buf = np.zeros(4 + n // 2 + 64, np.uint32)
    tag = np.array([9.9e30], np.float32)
    O.orc_compact_float.restype = ctypes.c_void_p
    args = (ctypes.c_void_p(field.ctypes.data), ctypes.c_void_p(buf.ctypes.data), ctypes.c_void_p(buf[4:].ctypes.data), n, 16 + 64 * 16, 0, 1, 1, 0, ctypes.c_void_p(tag.ctypes.data))
    O.orc_compact_float(*args)                          # untimed: page faults of the record
    t0 = time.perf_counter(); O.orc_compact_float(*args); cf = time.perf_counter() - t0
    t0 = time.perf_counter(); zl = O.orc_armn_compress(ctypes.c_void_p(buf[4:].ctypes.data), NI_D, NJ_D, 1, 16, 1); ac = time.perf_counter() - t0
    r = {"kind": "port", "cores": 1, "sample": "1 field 7200x3601 (orc_compact_float 16-bit + orc_armn_compress)",
         "compact_float_s_per_field": cf, "compact_float_GBps": 4.0 * n / cf / 1e9,
         "armn_compress_s_per_field": ac, "armn_compress_GBps": 2.0 * n / ac / 1e9, "zlng_bytes": int(zl), "unit": "GB/s of input"}
    if ref_interp_s:
        r["cfg5_chain_s_per_field"] = ref_interp_s + cf + ac
        r["cfg5_chain_fields_per_s"] = 1.0 / (ref_interp_s + cf + ac)
    return r


CFG3_REF_FILE = None


def start_cfg3_reference_child():
    """the reference's c_ezuvint on cfg3 (tests/ref_child.py: oracle/_ref, 1 thread, a stack of its own for the Fortran automatic arrays) in a FRESH process,
    started before this process imports torch or touches a GPU; collected before the warm-up starts (nothing of it runs beside a timed region)."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "_ref", "libezref.so")
    if not os.path.exists(so):
        return None
    try:
        import tempfile
        global CFG3_REF_FILE
        CFG3_REF_FILE = os.path.join(tempfile.mkdtemp(prefix="ezbench_"), "cfg3_ref.npy")      # the reference's (u, v): extras compares the exact-winds mode with it bit for bit
        return subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ref_child.py"), "cfg3_uvint", "--reps", "2", "--out", CFG3_REF_FILE],
                                stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT)
    except OSError:
        return None


def collect_cfg3_reference_child(proc):
    if proc is None:
        return None
    try:
        out, _ = proc.communicate(timeout=180)
        r = json.loads(out.strip().splitlines()[-1])
        return {"value": r["points"] / r["s_per_pair"] / 1e6, "unit": "Mpoint-pairs/s", "cores": 1, "kind": "reference", "s_per_pair": r["s_per_pair"],
                "first_call_s": r["first_s"], "sample": "%d pairs cfg3 steady-state c_ezuvint, fresh child process, oracle/_ref/libezref.so" % r["reps"]}
    except Exception as e:   # noqa: BLE001
        try:
            proc.kill()
        except Exception:   # noqa: BLE001
            pass
        return {"error": repr(e)[:120]}


def slim(o, sig=5):
    """floats to `sig` significant digits (the line has to fit the driver's 8 KB tail; every figure keeps more digits than its run-to-run spread)"""
    if isinstance(o, float):
        return float("%.*g" % (sig, o)) if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: slim(v, sig) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [slim(v, sig) for v in o]
    return o


def check_outputs(ez, torch, d_out, d_in, check_f):
    """the timed launch's outputs, after the timed loop: field `check_f` (input = the fixture's 'synth' field) against the
    reference's own full-size run (sampled rows / columns within 1e-5 relative, float64 sum), and the first / last field of
    the batch against a single-field c_ezsint_dev call, bit for bit.  Returns a dict for the JSON line."""
    res = {"ok": False}
    try:
        G = np.load(os.path.join(ROOT, "tests", "golden", "cfg2_full_golden.npz"))
        o2 = d_out[check_f].view(NJ_D, NI_D)
        rows = torch.from_numpy(G["rows"]).cuda(); cols = torch.from_numpy(G["cols"]).cuda()
        worst = 0.0; worst_rel = 0.0; worst_abs = 0.0
        for got, want in ((o2[rows].cpu().numpy(), G["synth/d3_p1/rows"]), (o2[:, cols].cpu().numpy(), G["synth/d3_p1/cols"])):
            # the PURE relative error at every point (round 6: no floor); also reported apart: where |want| >= 1e-3 max|want| / the absolute error elsewhere
            d = np.abs(got.astype(np.float64) - want); aw = np.abs(want.astype(np.float64)); big = aw >= aw.max() * 1e-3
            worst = max(worst, float((d / np.maximum(aw, 1e-30)).max()))
            if big.any():
                worst_rel = max(worst_rel, float((d[big] / aw[big]).max()))
            if (~big).any():
                worst_abs = max(worst_abs, float(d[~big].max()))
        s = float(d_out[check_f].double().sum().item())
        sum_rel = abs(s - float(G["synth/d3_p1/sum"])) / abs(s)
        one = torch.empty(NPTS_OUT, dtype=torch.float32, device="cuda")
        same = True
        for f in (0, d_out.shape[0] - 1):
            assert ez.ezsint_dev(one, d_in[f]) == 0
            torch.cuda.synchronize()
            same = same and bool(torch.equal(one, d_out[f]))
        res = {"ok": bool(worst <= 1e-5 and sum_rel <= 1e-8 and same), "field": check_f,
               "max_rel_err_vs_reference_run": worst, "max_pure_rel_err": worst_rel, "max_abs_err_near_zero": worst_abs, "sum_rel_diff": sum_rel, "batch_equals_single_calls_bitwise": same,
               "against": "tests/golden/cfg2_full_golden.npz + c_ezsint_dev"}
    except Exception as e:   # noqa: BLE001
        res["error"] = repr(e)[:200]
    return res


def profile_value(key):
    """(value, file) of a figure the round's profiling passes left under profiles/ (newest round first): HBM / fabric traffic per unit from the PMC passes of
    tools/prof_round.sh (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE: MI355X_MICROARCH.md), instruction counts from the SQ passes (tools/pmc_sq.sh).  Counters need their
    own rocprofv3 runs: this run does not collect them, and every figure taken from here is labelled with its file in the line."""
    import glob, re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")), key=lambda f: int(re.search(r"r(\d\d)_", f).group(1)), reverse=True)
    for f in files:
        try:
            with open(f) as fh:
                v = json.load(fh).get(key)
            if v is not None:
                return float(v), "profiles/" + os.path.basename(f)
        except Exception:   # noqa: BLE001
            pass
    return None


profile_traffic = profile_value


def roofline_cfg5(pipe_us, zl_mean):
    """the cfg5 pipeline is bound by instruction ISSUE, not by bytes.  Its VALU wave-instructions per field come from the newest SQ-counter profile under profiles/
    (labelled; never a literal in this file); without one `achieved` and `frac` are omitted.  peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction
    (MI355X_MICROARCH.md: SIMD-32, two or more waves); peak_one_wave = the same at the 4.7 cycles one wave's stream alone sustains (tools/irate.hip).  DESIGN.md 5."""
    v = profile_value("cfg5_valu_wave_instructions_per_field")
    peak = 1024 * 2.4 / 2.0
    r = {"bound": "issue", "peak": peak, "unit": "G VALU wave-instr/s", "peak_one_wave": 1024 * 2.4 / 4.7,
         "kernel": "k_bb_* + k_sepx<3,16,3> + k_armn_enc1"}
    if v:
        r.update({"achieved": v[0] / (pipe_us * 1e-6) / 1e9, "frac": v[0] / (pipe_us * 1e-6) / 1e9 / peak,
                  "valu_wave_instr_per_field": v[0], "valu_source": v[1]})
    return r


def extras(ez, torch, stream, d_out, d_in):
    """secondary measurements next to the headline (never part of `value`): one field per launch, the host-pointer
    ABI, and BASELINE configs[2] (c_ezuvint, Z-on-E 2560x1280 -> L 4000x2000).  Best effort: {} on any failure."""
    import ezcases as ec
    ex = {}
    try:
        def ev_time(fn, reps, bursts=5):
            """median over `bursts` bursts of `reps` back-to-back calls (HIP events on the launch stream); ten untimed calls first.  One burst now and then
            runs 15 - 20 % long on a fresh box (clocks): the median of five does not move with it"""
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(bursts):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(reps):
                    fn()
                e1.record(stream); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / reps)
            return sorted(ts)[len(ts) // 2]
        ex["single_field_launch_us"] = ev_time(lambda: ez.ezsint_dev(d_out[0], d_in[0]), 40)      # median of 5 bursts of 40 back-to-back c_ezsint_dev calls, HIP events
        # the same calls with the CALLER alternating two streams (ezhip_use_stream between calls; separate output arrays): consecutive launches of one stream cannot
        # overlap, those of two streams do -- the fill of one launch runs beside the drain of the other.  Nothing in the library changes: the drop-in call, one field each
        try:
            s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
            o2 = [d_out[0], d_out[1]]

            def burst(reps):
                for k in range(reps):
                    ez.use_stream(s2[k & 1].cuda_stream)
                    ez.ezsint_dev(o2[k & 1], d_in[k % d_in.shape[0]])
            burst(20); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                burst(80)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 80 * 1e6)
            ex["single_field_two_streams_us"] = sorted(ts)[2]       # median of 5 bursts of 80 calls alternating over two caller streams, wall clock incl. the launch calls
        finally:
            ez.use_stream(stream.cuda_stream)
        zin_h = np.ascontiguousarray(d_in[0].cpu().numpy()); zout_h = np.zeros(NPTS_OUT, np.float32)     # pageable, touched
        cez = ez._lib().c_ezsint
        cez(zout_h.ctypes.data, zin_h.ctypes.data)
        t0 = time.perf_counter()
        for _ in range(3):
            cez(zout_h.ctypes.data, zin_h.ctypes.data)
        ex["host_pointer_abi_ms_per_field"] = (time.perf_counter() - t0) / 3 * 1e3
        # the same call between arrays the caller page-locked once (ezhip_register_host_buffer): source rows up / result rows down overlap
        import ctypes as _ct
        _L = ez._lib()
        _L.ezhip_register_host_buffer.argtypes = [_ct.c_void_p, _ct.c_size_t]; _L.ezhip_unregister_host_buffer.argtypes = [_ct.c_void_p]
        if _L.ezhip_register_host_buffer(zin_h.ctypes.data, zin_h.nbytes) == 0 and _L.ezhip_register_host_buffer(zout_h.ctypes.data, zout_h.nbytes) == 0:
            cez(zout_h.ctypes.data, zin_h.ctypes.data)
            t0 = time.perf_counter()
            for _ in range(3):
                cez(zout_h.ctypes.data, zin_h.ctypes.data)
            ex["host_pointer_abi_registered_ms_per_field"] = (time.perf_counter() - t0) / 3 * 1e3
            _L.ezhip_unregister_host_buffer(zin_h.ctypes.data); _L.ezhip_unregister_host_buffer(zout_h.ctypes.data)
        # winds between the headline's own (unrotated) grids: no per-point matrix there, the reference's chain runs per call (k_wind_rotate with the C library's REAL trig) and
        # the result equals the reference's bit for bit in the default mode (tools/fuzz_vs_ref.py: 423 of 423 random pairs)
        try:
            uu2, vv2 = ec.synth_wind(NI_S, NJ_S, seed=4)
            d_u2 = torch.from_numpy(uu2).cuda(); d_v2 = torch.from_numpy(vv2).cuda()
            assert ez.ezuvint_dev(d_out[0], d_out[1], d_u2, d_v2) >= 0
            us2 = ev_time(lambda: ez.ezuvint_dev(d_out[0], d_out[1], d_u2, d_v2), 5, bursts=3)
            ex["cfg2_uvint"] = {"workload": "c_ezuvint_dev bicubic G 4400x2200 -> L 7200x3601", "dtype": "f64", "us_per_pair": us2, "Mpoint_pairs_per_s": NPTS_OUT / us2}
            del d_u2, d_v2
        except Exception as e:   # noqa: BLE001
            ex["cfg2_uvint"] = {"error": repr(e)[:120]}
        # ---- BASELINE configs[2]: c_ezuvint, Z-on-E 2560x1280 (rotated global) -> L 4000x2000, bicubic, polar_correction=yes
        ni, nj, no, mo = 2560, 1280, 4000, 2000
        ax, ay = ec.ze_axes(ni, nj)
        g_in = ez.ezgdef_fmem(ni, nj, "Z", "E", *ec.E_IG, ax, ay); g_out = ez.ezqkdef(no, mo, "L", 9, 9, 0, 0)
        assert ez.ezdefset(g_out, g_in) == 1
        uu, vv = ec.synth_wind(ni, nj, seed=3)
        for a in (uu, vv):
            a2 = a.reshape(nj, ni); a2[:, -1] = a2[:, 0]
        d_u = torch.from_numpy(uu).cuda(); d_v = torch.from_numpy(vv).cuda()
        o_u = torch.empty(no * mo, dtype=torch.float32, device="cuda"); o_v = torch.empty_like(o_u)
        # the set's FIRST call: locate of the 8 M target points in the rotated source (k_locate, bit-equal to the C library's REAL trig), Newton
        # tables, the wind matrix of the grid pair, the special points' list, the tile table and the tile-ordered stream copy
        torch.cuda.synchronize(); t0 = time.perf_counter()
        assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) >= 0
        torch.cuda.synchronize(); first3 = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) >= 0
        torch.cuda.synchronize(); second3 = (time.perf_counter() - t0) * 1e3      # (builds the staged-tile caches behind it)
        us = ev_time(lambda: ez.ezuvint_dev(o_u, o_v, d_u, d_v), 20)
        algo3 = 2 * 4 * ni * nj + 2 * 4 * no * mo                      # both source components in, both target components out
        t3 = profile_traffic("cfg3_traffic_MB_per_pair")
        ex["cfg3_uvint"] = {"workload": "c_ezuvint_dev bicubic Z-on-E 2560x1280 -> L 4000x2000", "dtype": "f32 + f64 second pass",
                            "us_per_pair": us, "Mpoint_pairs_per_s": no * mo / us, "first_call_ms": first3, "second_call_ms": second3,
                            "roofline": {"bound": "hbm", "achieved": algo3 / us / 1e3, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": algo3 / us / 1e3 / HBM_PEAK_GBPS,
                                         "traffic": t3[0] * 1e6 if t3 else None, "traffic_source": t3[1] if t3 else None,
                                         "kernel": "k_uvt<32,32>", "algorithmic_bytes_per_launch": algo3}}
        # the exact-winds mode (ezhip_set_wind_exact: the reference's chain as written on every call, scalar kernels in front): time, and every one of its 16 M values
        # against the reference child's own result, bit for bit
        try:
            ez.set_wind_exact(1)
            assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) >= 0
            usx = ev_time(lambda: ez.ezuvint_dev(o_u, o_v, d_u, d_v), 5, bursts=3)
            ex["cfg3_uvint_exact"] = {"workload": "the same with ezhip_set_wind_exact(1)", "dtype": "f64", "us_per_pair": usx}
            if CFG3_REF_FILE and os.path.exists(CFG3_REF_FILE):
                ref = np.load(CFG3_REF_FILE)
                nd = int(np.count_nonzero(o_u.cpu().numpy().view(np.uint32) != ref[0].view(np.uint32)) + np.count_nonzero(o_v.cpu().numpy().view(np.uint32) != ref[1].view(np.uint32)))
                ex["cfg3_uvint_exact"]["values_differing_from_reference_bits"] = nd
                ex["cfg3_uvint_exact"]["values_compared"] = int(2 * no * mo)
                del ref
        finally:
            ez.set_wind_exact(0)
        assert ez.ezuvint_dev(o_u, o_v, d_u, d_v) >= 0
        if CFG3_REF_FILE and os.path.exists(CFG3_REF_FILE):      # ... and the default (fast) mode's distance from the same field
            ref = np.load(CFG3_REF_FILE)
            gu = o_u.cpu().numpy().astype(np.float64); gv = o_v.cpu().numpy().astype(np.float64)
            ex["cfg3_uvint"]["max_err_vs_reference_in_V"] = float((np.maximum(np.abs(gu - ref[0]), np.abs(gv - ref[1])) / np.maximum(np.hypot(ref[0].astype(np.float64), ref[1].astype(np.float64)), 1e-3)).max())
            del ref, gu, gv
            try:
                os.remove(CFG3_REF_FILE); os.rmdir(os.path.dirname(CFG3_REF_FILE))
            except OSError:
                pass
        # several pairs of the same grid set at once (wind levels): c_ezuvint_batch_dev reads x, y and the rotation of a point once per batch (12 of a pair's 34 bytes
        # per point) -- one launch of k_uvt's batch form, the special points' kernel once with a pair index; results equal to single calls bit for bit
        KB = 8
        d_ub = torch.stack([d_u * (1.0 + 0.01 * f) for f in range(KB)]).contiguous(); d_vb = torch.stack([d_v * (1.0 - 0.01 * f) for f in range(KB)]).contiguous()
        o_ub = torch.empty((KB, no * mo), dtype=torch.float32, device="cuda"); o_vb = torch.empty_like(o_ub)
        usb = ev_time(lambda: ez.ezuvint_batch_dev(o_ub, o_vb, d_ub, d_vb, KB), 10) / KB
        r_u = torch.empty_like(o_u); r_v = torch.empty_like(o_v)
        same_b = True
        for f in (0, KB - 1):
            assert ez.ezuvint_dev(r_u, r_v, d_ub[f], d_vb[f]) >= 0
            torch.cuda.synchronize()
            same_b = same_b and bool(torch.equal(r_u, o_ub[f]) and torch.equal(r_v, o_vb[f]))
        tb = profile_value("cfg3_batch_traffic_MB_per_pair")
        ex["cfg3_uvint_batch"] = {"workload": "c_ezuvint_batch_dev, %d pairs per call, same grids" % KB, "dtype": "f32 + f64 second pass",
                                  "us_per_pair": usb, "Mpoint_pairs_per_s": no * mo / usb, "frac_of_hbm_peak": algo3 / usb / 1e3 / HBM_PEAK_GBPS,
                                  "equal_to_single_calls_bitwise": same_b, "traffic_MB_per_pair": tb[0] if tb else None, "traffic_source": tb[1] if tb else None,
                                  "kernel": "k_uvt<32,32,false,true>"}
        del d_ub, d_vb, o_ub, o_vb, r_u, r_v
        # the scalar twin on the same grid pair: c_ezsint from the rotated source (k_st: stencil windows staged in LDS, the literal REAL*8 form of the reference)
        us1 = ev_time(lambda: ez.ezsint_dev(o_u, d_u), 20)
        algo1 = 4 * ni * nj + 4 * no * mo
        t1 = profile_value("cfg3_sint_traffic_MB_per_field"); v1 = profile_value("cfg3_sint_valu_wave_instructions_per_field")
        valu_peak = 1024 * 2.4 / 2.0                # G wave64 VALU instructions / s: 2 cycles each on a SIMD-32 with two or more waves (MI355X_MICROARCH.md, constants table)
        ex["cfg3_sint"] = {"workload": "c_ezsint_dev bicubic, same grids", "dtype": "f64", "us_per_field": us1, "Mpoints_per_s": no * mo / us1,
                           "roofline": {"bound": "valu", "achieved": (v1[0] / us1 / 1e3) if v1 else None, "peak": valu_peak, "unit": "G VALU wave-instr/s",
                                        "frac": (v1[0] / us1 / 1e3 / valu_peak) if v1 else None, "valu_source": v1[1] if v1 else None,
                                        "hbm_frac_of_algorithmic_bytes": algo1 / us1 / 1e3 / HBM_PEAK_GBPS,
                                        "traffic": t1[0] * 1e6 if t1 else None, "kernel": "k_st<32,32>"}}
        del d_u, d_v, o_u, o_v
        # the step after the horizontal one: vertical interpolation of device-resident profiles, search + linear + lapse-rate in one pass
        from librmn_amd import interpv as V
        ncol, ns, nd = 7200 * 3601 // 16, 80, 60
        ps = 1.0 + 0.05 * torch.sin(torch.arange(ncol, device="cuda", dtype=torch.float32) * 1e-3)
        vls = torch.linspace(1, ns, ns, device="cuda")[:, None] * ps[None, :]
        vld = (torch.linspace(1.5, ns - 0.5, nd, device="cuda")[:, None] + torch.zeros((1, ncol), device="cuda")).contiguous()
        ss = torch.sin(vls * 0.1); sd = torch.empty((nd, ncol), device="cuda")
        us = ev_time(lambda: V.column_dev(V.LINEAR, V.X_LAPSERATE, ncol, vls, ss, ss, None, vld, sd, sd, True, True, 0.1, 0.1), 20)
        ex["interpv_column"] = {"workload": f"FindPos + Linear + LapseRate fused, {ncol} columns, {ns} -> {nd} levels", "dtype": "f32",
                                "us": us, "frac_of_hbm_peak": 4.0 * ncol * (2 * ns + 2 * nd) / us / 1e3 / HBM_PEAK_GBPS}
        del vls, vld, ss, sd
        # the read side: armn_compress UNCOMPRESS of cfg5 records in HBM -- one stream alone, and 16 / 32 decoded concurrently
        from librmn_amd import packers as pk
        n = NPTS_OUT
        Fd = 32
        stride = 4 + n // 2 + 64
        recs = torch.zeros(Fd * stride, dtype=torch.int32, device="cuda")
        rc_, zl_ = pk.pack16_compress_batch_dev(recs, stride, d_out[:Fd].contiguous(), n, Fd, NI_D, NJ_D, 16)
        toks = torch.zeros((Fd, 1 + n // 2), dtype=torch.int32, device="cuda")
        cap = int(max(zl_) + 3) // 4 + 1
        def dec(nb):
            for _ in range(2):
                torch.cuda.synchronize(); t0_ = time.perf_counter()
                pk.armn_uncompress_batch_dev(toks, 1 + n // 2, recs[4:], stride, cap, NI_D, NJ_D, 16, nb)
                torch.cuda.synchronize(); dt_ = time.perf_counter() - t0_
            return dt_ * 1e3
        one_ms, batch16_ms, batch_ms = dec(1), dec(16), dec(Fd)
        zmean = float(np.mean([z for z in zl_ if z > 0]))
        ex["armn_uncompress"] = {"workload": "UNCOMPRESS of 7200x3601 16-bit records in HBM, ratio %.2f" % (zmean / (2.0 * n)), "dtype": "u16",
                                 "single_stream_ms": one_ms, "batch_of_16_ms_per_field": batch16_ms / 16, "batch_of_32_ms_per_field": batch_ms / Fd,
                                 "single_stream_frac_of_hbm_peak": (zmean + 2.0 * n) / (one_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                 "batch_frac_of_hbm_peak": (zmean + 2.0 * n) * Fd / (batch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        # the same field as a MINIMUM stream (c_armn_compress_setlevel(FAST)): its chain of tile headers is resolved by composition of the windows' maps
        # (k_dmin_*: rows of whole 5 x 5 tiles); ... and with rows that end on a narrower tile (the same tokens read as a 7201 x 3600 field): the composed ragged form
        try:
            pk.armn_setlevel(0)
            d_zm = torch.zeros(n // 2 + 64, dtype=torch.int32, device="cuda")
            for key, nim, njm in (("minimum_ms", NI_D, NJ_D), ("minimum_ragged_ms", NI_D + 1, NJ_D - 1)):
                pk.armn_setlevel(0)
                nm_ = nim * njm
                zlm = pk.armn_compress_dev(d_zm, toks[0], nim, njm, 16)
                pk.armn_setlevel(1)
                if zlm > 0:
                    zwm = (zlm - 1) // 4 + 1
                    tk = torch.zeros(1 + nm_ // 2, dtype=torch.int32, device="cuda")
                    best = 1e9
                    for _ in range(4):
                        torch.cuda.synchronize(); t0_ = time.perf_counter()
                        rcm = pk.armn_uncompress_dev(tk, d_zm, zwm, nim, njm, 16)
                        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0_)
                    ex["armn_uncompress"][key] = best * 1e3
                    ex["armn_uncompress"][key.replace("_ms", "_tokens_equal")] = bool(rcm == 2 * nm_ and torch.equal(tk[:nm_ // 2], toks[0][:nm_ // 2]))
        finally:
            pk.armn_setlevel(1)
        del recs, toks
        # the IEEE-32 compressor's read side (datyp 133): c_armn_uncompress32 through its three entries (the reference's signature without a length, _lng with the
        # record's length, _zdev record and field in HBM); 7200 columns = ragged rows, 7201 = rows of whole tiles
        import ezcases as _ec
        for key, ni32 in (("armn_uncompress32", NI_D), ("armn_uncompress32_whole_tile_rows", NI_D + 1)):
            f32 = _ec.synth_field(ni32, NJ_D, seed=5); n32 = ni32 * NJ_D
            zl32, z32 = pk.armn_compress32(f32, ni32, NJ_D, 32)
            if zl32 <= 0: continue
            best = bestl = 1e9
            out_a = np.zeros(n32, np.float32); out_b = np.zeros(n32, np.float32)          # the caller's field buffers, kept across calls (touched: no first-touch page faults in the clock)
            out_a.fill(1.0); out_b.fill(1.0)
            for _ in range(3):
                t0_ = time.perf_counter(); rc32, back32 = pk.armn_uncompress32(z32, ni32, NJ_D, 32, out=out_a); best = min(best, time.perf_counter() - t0_)
                t0_ = time.perf_counter(); rcl, backl = pk.armn_uncompress32_lng(z32, 4 * ((zl32 + 3) // 4), ni32, NJ_D, 32, out=out_b); bestl = min(bestl, time.perf_counter() - t0_)
            ex[key] = {"columns": ni32, "dtype": "u32 planes of f32", "ratio": zl32 / (4.0 * n32), "host_arrays_no_length_ms": best * 1e3, "host_arrays_length_given_ms": bestl * 1e3,
                       "bit_identical": bool(rc32 == n32 and np.array_equal(back32.view(np.uint32), f32.view(np.uint32)) and rcl == n32 and np.array_equal(backl.view(np.uint32), f32.view(np.uint32)))}
            # record and field both in HBM (c_armn_compress32_dev -> c_armn_uncompress32_zdev)
            d_f32 = torch.from_numpy(f32).cuda(); d_z32 = torch.zeros(n32 + 64, dtype=torch.int32, device="cuda"); d_b32 = torch.empty(n32, dtype=torch.float32, device="cuda")
            pk.armn_compress32_dev(d_z32, d_f32, ni32, NJ_D, 32); torch.cuda.synchronize()
            t0_ = time.perf_counter()
            zl32_dev = pk.armn_compress32_dev(d_z32, d_f32, ni32, NJ_D, 32)
            torch.cuda.synchronize(); ex[key]["compress32_in_hbm_ms"] = (time.perf_counter() - t0_) * 1e3
            if zl32_dev == zl32:
                bestd = 1e9
                for _ in range(4):
                    torch.cuda.synchronize(); t0_ = time.perf_counter()
                    rcd = pk.armn_uncompress32_zdev(d_b32, d_z32, 4 * ((zl32 + 3) // 4), ni32, NJ_D, 32)
                    torch.cuda.synchronize(); bestd = min(bestd, time.perf_counter() - t0_)
                ex[key]["in_hbm_ms"] = bestd * 1e3
                ex[key]["in_hbm_bit_identical"] = bool(rcd == n32 and torch.equal(d_b32.view(torch.int32), d_f32.view(torch.int32)))
            del d_f32, d_z32, d_b32
    except Exception as e:   # noqa: BLE001
        ex["error"] = repr(e)[:300]
    return ex


def free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def visible_gpus_without_runtime():
    """GPUs this process would see, counted WITHOUT the HIP runtime (torch.cuda.device_count() falls back to hipGetDeviceCount -- which initialises the
    runtime -- when amdsmi discovery fails; forking ranks from a GPU-initialised parent is the fragile pattern on this pool): KFD topology nodes with
    SIMDs, cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when they hold a plain index list.  None: unknown (no sysfs)."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for d in os.listdir(base):
            try:
                props = open(os.path.join(base, d, "properties")).read()
            except OSError:
                continue
            for line in props.splitlines():
                f = line.split()
                if len(f) == 2 and f[0] == "simd_count" and int(f[1]) > 0:
                    n += 1
    except OSError:
        return None
    if n == 0:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and all(t.strip().isdigit() for t in v.split(",") if t.strip()):
            n = min(n, len([t for t in v.split(",") if t.strip()]))
    return n


def launch_ranks(n, argv, stub):
    """`python bench.py --gpus N` without a launcher's WORLD_SIZE in the environment: start N FRESH rank processes (this file again, one per
    LOCAL_RANK, rendezvous on 127.0.0.1) BEFORE anything in this process touches a GPU, wait for them, return the worst exit code.  The
    parent never initialises HIP -- it does not even import torch: the devices are counted from the KFD topology in sysfs, and where that cannot be read
    the ranks' own refusal (LOCAL_RANK >= device count -> exit 2) is the check -- and never exec()s: children are ordinary subprocesses."""
    import subprocess
    if not stub:
        have = visible_gpus_without_runtime()
        if have is not None and have < n:
            sys.stderr.write(f"bench.py: --gpus {n} asked for, {have} GPU(s) visible on this node: refusing to report an N-GPU line from fewer devices\n")
            return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_LAUNCHED_BY="bench.py")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    import signal

    def stop_children(signum=None, frame=None):     # end exactly the processes started here (never by pattern)
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        if signum is not None:
            raise SystemExit(128 + signum)
    for sg in (signal.SIGTERM, signal.SIGINT):      # the driver's own timeout must not leave ranks behind on the GPUs
        signal.signal(sg, stop_children)
    deadline = time.time() + float(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "1800"))
    worst = 0
    pending = list(procs)
    while pending:
        for pr in list(pending):
            rc = pr.poll()
            if rc is None:
                continue
            pending.remove(pr)
            if rc != 0:
                worst = worst or rc
                for other in pending:          # a dead rank leaves the others in a collective
                    other.terminate()
        if time.time() > deadline:
            sys.stderr.write("bench.py: the ranks did not finish within BENCH_LAUNCH_TIMEOUT_S; stopping them\n")
            stop_children()
            return 124
        time.sleep(0.05)
    return worst


def timed_region(step, steps, dist, sync, stream=None, torch=None):
    """the contract's timed region: barrier + synchronize, EXACTLY `steps` steps, synchronize + barrier; returns (wall seconds of this
    rank, HIP-event ms on the launch stream or None).  The barrier in front goes in BEHIND whatever the caller queued as warm-up: draining
    the device first and then running the collective left the GPU idle for the length of a barrier and the timed steps started on dropped
    clocks (+9 % per step, measured with one rank under torchrun against the same box without a process group)."""
    if dist:
        dist.barrier()
    sync()
    ev0 = ev1 = None
    if torch is not None and stream is not None:
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if ev0:
        ev0.record(stream)
    for _ in range(steps):
        step()
    if ev1:
        ev1.record(stream)
    sync()
    elapsed = time.perf_counter() - t0
    if dist:
        dist.barrier()
    sync()
    return elapsed, (ev0.elapsed_time(ev1) if ev0 else None)


def ranks_seen(dist, device):
    """how many ranks the collective layer really connected: an all_reduce(SUM) of ones (RCCL on the GPU box)"""
    if not dist:
        return 1
    import torch
    one = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(one.item())


def stub_rank(args, world, rank):
    """TEST ONLY (--stub-step, tests/test_bench_launcher.py): the launcher, the rendezvous, the barriers, the max-over-ranks time and the
    rank count of the real bench, over gloo on the CPU with a sleep as the step.  No GPU, no library: the line says "stub": true and
    carries no throughput -- nothing in it can be mistaken for a measurement."""
    import torch.distributed as dist
    from librmn_amd import sharding as sh
    d = None
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        d = dist
    F = args.fields_per_step
    mine = sh.fields_of_rank(F * world, rank, world)
    for _ in range(args.warmup):
        time.sleep(0.001)
    elapsed, _ = timed_region(lambda: time.sleep(0.002 * (1 + rank)), args.steps, d, lambda: None)
    elapsed = sh.max_over_ranks(elapsed)
    n = ranks_seen(d, "cpu")
    owned = sh.sum_over_ranks(float(len(mine)))
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test (no GPU work)", "stub": True, "value": None, "n_gpus": n, "world_size_env": world,
                          "gpus_flag": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
                          "fields_owned_by_all_ranks": int(owned), "fields_per_step_per_gpu": F, "feed": args.feed,
                          "launched_by": os.environ.get("BENCH_LAUNCHED_BY", "external launcher")}), flush=True)
    if d:
        d.destroy_process_group()


def host_fed(ez, torch, stream, d_in, d_out, F, steps, dist, sh):
    """SURVEY 8e "bench both": the same batch step with its F source fields starting in (page-locked) HOST memory -- every step uploads
    F x 38.72 MB inside the timed region; upload of step k+1 (copy stream) overlaps the interpolation of step k (two device buffers).
    Results stay in HBM as in the headline.  Returns the object for the JSON line."""
    nin = NI_S * NJ_S
    h_in = torch.empty((F, nin), dtype=torch.float32, pin_memory=True)
    h_in.copy_(d_in)
    bufs = [d_in, torch.empty_like(d_in)]
    copy_stream = torch.cuda.Stream()
    up = [torch.cuda.Event() for _ in range(2)]; done = [torch.cuda.Event() for _ in range(2)]
    k = [0]

    def step():
        b = k[0] & 1; k[0] += 1
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(done[b])              # the launch that last read this buffer
            bufs[b].copy_(h_in, non_blocking=True)
            up[b].record(copy_stream)
        stream.wait_event(up[b])
        rc = ez.ezsint_batch_dev(d_out, bufs[b], F)
        assert rc == 0, rc
        done[b].record(stream)
    for _ in range(2):
        step()
    elapsed, _ = timed_region(step, steps, dist, torch.cuda.synchronize)
    elapsed = sh.max_over_ranks(elapsed, device="cuda")
    world = dist.get_world_size() if dist else 1
    up_bytes = 4.0 * nin * F
    return {"workload": f"the headline step, its {F} sources uploaded from page-locked host memory inside the timed region",
            "steps": steps, "ms_per_step": elapsed * 1e3 / steps, "value": float(NPTS_OUT) * F * steps * world / elapsed / 1e6, "unit": "Mpoints/s",
            "upload_GBps_per_gpu": up_bytes * steps / elapsed / 1e9, "upload_bytes_per_step_per_gpu": up_bytes}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node.  Under an external launcher (WORLD_SIZE set) it must agree with it; "
                                                           "alone with N > 1, bench.py starts the N rank processes itself")
    # defaults: a step is ~1 ms.  After an idle period the launch durations run 0.92, 0.95, 1.08, 1.18, 1.13, 1.06 ... ms
    # and settle at 0.92-0.94 ms only after ~40 launches (power management; profiles/r01_launch_series.txt): warm up
    # past that, then time 60 steps
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--fields-per-step", type=int, default=32)
    ap.add_argument("--feed", choices=["device", "host"], default="device",
                    help="device (default, the contract's `value`): sources resident in HBM when the timed region starts.  host: every step uploads its "
                         "sources from page-locked host memory inside the timed region (SURVEY 8e); the line then says config.feed = host")
    ap.add_argument("--host-fed-steps", type=int, default=4, help="steps of the host-fed secondary measurement reported next to a device-fed headline (0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stub-step", action="store_true", help=argparse.SUPPRESS)      # tests only: see stub_rank
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus is not None and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], args.stub_step))
    world = int(env_world or "1")
    if args.gpus is not None and args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE is {world}: refusing to print a line whose n_gpus is ambiguous")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.stub_step:
        return stub_rank(args, world, rank)

    # the reference's c_ezuvint on cfg3 as a child of its own, started BEFORE anything here touches a GPU (extras.cfg3_uvint.cpu_baseline)
    cfg3_child = start_cfg3_reference_child() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (0 GPU(s) visible): the MI355X hot path has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):      # BENCH_FORCE_DIST: exercise the RCCL path with one rank
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist.barrier()      # the communicator is built by the first collective (hundreds of ms of idle GPU): here, not between the warm-up and the timed steps

    from librmn_amd import ezscint as ez
    from librmn_amd import sharding as sh
    import ezcases as ec

    F = args.fields_per_step
    my_fields = sh.fields_of_rank(F * world, rank, world)      # record sharding: field f -> rank f mod world
    # the HIP runtime initialises itself lazily on a process's first allocation / launch (~150 ms on these boxes: it used to be counted as the library's first call)
    torch.zeros(8, device="cuda").add_(1.0); torch.cuda.synchronize()
    t_prep = time.perf_counter()
    gdin = ez.ezqkdef(NI_S, NJ_S, "G", 0, 0, 0, 0)
    t_def = time.perf_counter()
    gdout = ez.ezqkdef(NI_D, NJ_D, "L", *L_IG)
    assert ez.ezdefset(gdout, gdin) == 1
    stream = torch.cuda.current_stream()
    ez.use_stream(stream.cuda_stream)
    assert ez.prepare_set() == 0 and ez.set_mode() == 1
    torch.cuda.synchronize()
    # what the reference does inside its grid definitions and its first c_ezsint: Gaussian latitudes, lat/lon, locate, zones (+ the k_sepx plan and its uploads)
    first_call_ms = (time.perf_counter() - t_prep) * 1e3
    gauss_ms = (t_def - t_prep) * 1e3

    # F distinct synthetic source fields, resident in HBM (seed per global field index)
    base = torch.from_numpy(ec.synth_field(NI_S, NJ_S, seed=1000 + my_fields[0])).cuda()
    d_in = torch.empty((F, NI_S * NJ_S), dtype=torch.float32, device="cuda")
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234 + rank)
    for f in range(F):
        d_in[f] = base * (1.0 + 1e-3 * (torch.rand(NI_S * NJ_S, device="cuda", generator=gen) - 0.5)) + 0.01 * my_fields[f]
    # one field of rank 0's batch is the reference fixture's input (tests/golden/cfg2_full_golden.npz, 'synth'): after the
    # timed loop its output is compared with the reference's own full-size run (`checked` in the JSON line)
    CHECK_F = F // 2
    if rank == 0:
        d_in[CHECK_F] = torch.from_numpy(ec.synth_field(NI_S, NJ_S, seed=2)).cuda()
    d_out = torch.empty((F, NPTS_OUT), dtype=torch.float32, device="cuda")

    def step():
        rc = ez.ezsint_batch_dev(d_out, d_in, F)
        assert rc == 0, rc

    cfg3_cpu = collect_cfg3_reference_child(cfg3_child)      # it has been running beside the imports and the set-up; nothing of it is left when the warm-up starts
    # the launch durations settle only after ~40 launches from idle (see the defaults above): when the caller asks for
    # a shorter warm-up, the difference runs first as untimed initialisation (reported as config.prewarm_steps)
    prewarm = max(0, 40 - args.warmup)
    for _ in range(prewarm + args.warmup):
        step()
    hf = None
    if args.feed == "host":
        # the headline region itself is host-fed: `value` then includes the uploads and config.feed says so
        hf = host_fed(ez, torch, stream, d_in, d_out, F, args.steps, dist, sh)
        elapsed, ev_ms = hf["ms_per_step"] * args.steps / 1e3, None
        for _ in range(10):
            step()
        _, ev_ms = timed_region(step, args.steps, dist, torch.cuda.synchronize, stream, torch)      # the kernel's own duration for `roofline`
    else:
        elapsed, ev_ms = timed_region(step, args.steps, dist, torch.cuda.synchronize, stream, torch)
        elapsed = sh.max_over_ranks(elapsed, device="cuda")        # the batch takes as long as its slowest rank
    n_seen = ranks_seen(dist, "cuda")
    checked = check_outputs(ez, torch, d_out, d_in, CHECK_F) if rank == 0 else None
    if args.feed == "device" and args.host_fed_steps > 0:
        hf = host_fed(ez, torch, stream, d_in, d_out, F, args.host_fed_steps, dist, sh)      # every rank: it carries barriers

    # ---- second timed region: the packers on the interpolated fields (device-resident) ------------------
    from librmn_amd import packers as pk
    npk = min(F, 8)
    rec = torch.zeros((npk, 4 + NPTS_OUT // 2 + 64), dtype=torch.int32, device="cuda")
    def pack_step():
        for f in range(npk):
            assert pk.compact_float_pack_dev(d_out[f], rec[f], rec[f][4:], NPTS_OUT, 16 + 64 * 16) != 0
    pack_step(); torch.cuda.synchronize()
    pe0 = torch.cuda.Event(enable_timing=True); pe1 = torch.cuda.Event(enable_timing=True)
    pe0.record(stream)
    for _ in range(3):
        pack_step()
    pe1.record(stream); torch.cuda.synchronize()
    pack_us = pe0.elapsed_time(pe1) * 1e3 / (3 * npk)
    # fused front half of cfg5: the batch interpolation with compact_float's min/max pass folded into k_sepx, then header +
    # token kernel per field (ezhip_ezsint_pack16_batch_dev): the pack cost of a field = fused step - plain step
    rs = 4 + NPTS_OUT // 2 + 64
    recs = torch.zeros((F, rs), dtype=torch.int32, device="cuda")
    def fused_step():
        assert pk.ezsint_pack16_batch_dev(recs, rs, d_out, d_in, F, NPTS_OUT, 16) == 0
    for _ in range(3):
        fused_step()
    torch.cuda.synchronize()
    fe0 = torch.cuda.Event(enable_timing=True); fe1 = torch.cuda.Event(enable_timing=True)
    nfused = max(3, min(20, args.steps))
    fe0.record(stream)
    for _ in range(nfused):
        fused_step()
    fe1.record(stream); torch.cuda.synchronize()
    fused_us = fe0.elapsed_time(fe1) * 1e3 / (nfused * F)          # interp + pack16 per field
    # fused cfg5 step: compact_float(16-bit slots) + armn_compress per field (returns zlng: one sync per field)
    t1 = time.perf_counter()
    zl = [pk.pack16_compress_dev(rec[f], d_out[f], NI_D, NJ_D, 16) for f in range(npk)]
    torch.cuda.synchronize()
    comp_us = (time.perf_counter() - t1) * 1e6 / npk

    # the whole cfg5 pipeline on the batch: interpolation + 16-bit pack (fused min/max) + armn_compress of every field,
    # one synchronisation per batch (ezhip_ezsint_pack16_batch_dev + ezhip_pack16_compress_batch_dev(prepacked))
    def pipeline_step():
        assert pk.ezsint_pack16_batch_dev(recs, rs, d_out, d_in, F, NPTS_OUT, 16) == 0
        rc_, zl_ = pk.pack16_compress_batch_dev(recs, rs, None, 0, F, NI_D, NJ_D, 16, prepacked=1)
        assert rc_ == 0
        return zl_
    pipeline_step()
    t2 = time.perf_counter()
    npipe = 3
    for _ in range(npipe):
        zl_batch = pipeline_step()
    pipe_unfused_us = (time.perf_counter() - t2) * 1e6 / (npipe * F)
    # round 2: the same records from the FUSED pipeline -- interpolation twice (min/max only, then straight to 16-bit
    # tokens: the float fields are never stored), one-pass armn encoder writing each stream in place, one sync per batch
    recs2 = torch.zeros((F, rs), dtype=torch.int32, device="cuda")
    def fused_pipeline_step():
        rc_, zl_ = pk.ezsint_pack16_compress_batch_dev(recs2, rs, d_in, F, NI_D, NJ_D, 16)
        assert rc_ == 0, rc_
        return zl_
    zl_f = fused_pipeline_step()
    pipe_checked = bool(list(zl_f) == list(zl_batch)) and all(
        bool(torch.equal(recs2[f][:4 + (int(zl_f[f]) - 1) // 4], recs[f][:4 + (int(zl_f[f]) - 1) // 4])) for f in (0, F - 1) if zl_f[f] > 0)
    t2 = time.perf_counter()
    npipe = 5
    for _ in range(npipe):
        zl_f = fused_pipeline_step()
    pipe_us = (time.perf_counter() - t2) * 1e6 / (npipe * F)
    zl_mean = float(np.mean([z for z in zl_f if z > 0])) if any(z > 0 for z in zl_f) else 0.0
    # one k_sepx launch per step covers the F fields of the batch (the pole sums run inside the same launch)
    kern_us = ev_ms * 1e3 / args.steps            # average launch-to-launch duration on the stream
    achieved = F * ALGO_BYTES / (kern_us * 1e-6) / 1e9
    # HBM/fabric traffic of the dominant kernel: PMC counters need their own rocprofv3 passes (FETCH_SIZE x2 on gfx950,
    # WRITE_SIZE exact: MI355X_MICROARCH.md); the per-field figure measured by tools/pmc_traffic.sh is kept in profiles/
    tr = profile_traffic("traffic_MB_per_field")
    traffic = tr[0] * 1e6 * F if tr else None
    tr5 = profile_traffic("cfg5_fused_pipeline_total_MB_per_field")
    out = None
    if rank == 0:
        total_pts = float(NPTS_OUT) * F * args.steps * world
        cf_tr = profile_value("compact_float_traffic_bytes_per_value")
        fused_cost = fused_us - ev_ms * 1e3 / (args.steps * F)
        # key order: contract fields, roofline, cpu_baseline, pack, check, then the secondary objects -- prose lives in DESIGN.md section 5, not in the line
        out = {
            "metric": "interp Mpoints/s + pack GB/s, 4400x2200->7200x3601 bicubic, 1/2/4/8 GPU",
            "value": total_pts / elapsed / 1e6,
            "unit": "Mpoints/s",
            "n_gpus": n_seen,                      # what an all_reduce(SUM) of ones over the process group returned, not the flag
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "cfg2/cfg4: c_ezsint bicubic G 4400x2200 -> L 7200x3601, polar_correction=yes, "
                                   f"{F} device-resident fields per step per GPU (sharded by record, no collective)",
                       "fields_per_step_per_gpu": F, "points_per_field": NPTS_OUT, "prewarm_steps": prewarm, "feed": args.feed,
                       "world_size": world, "launched_by": os.environ.get("BENCH_LAUNCHED_BY", "external launcher" if env_world else "single process"),
                       "develop_build": ez.develop_build(), "env_overrides": sorted(k for k in os.environ if k.startswith(("EZHIP_", "INTERPV_HIP_")))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": tr[1] if tr else None,
                         "kernel": "k_sepx<3,16>", "avg_launch_us": kern_us, "fields_per_launch": F,
                         "us_per_field": kern_us / F, "algorithmic_bytes_per_launch": F * ALGO_BYTES},
            "cpu_baseline": None,
            "pack": {"unit": "GB/s of float input", "dtype": "f32 -> u16 tokens (f64 quantisation)",
                     "compact_float": {"us_per_field": pack_us, "GBps": 4.0 * NPTS_OUT / (pack_us * 1e-6) / 1e9,
                                       "fused_into_interp_cost_us_per_field": fused_cost,
                                       "roofline": {"bound": "hbm", "achieved": 6.0 * NPTS_OUT / (pack_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                                    "frac": 6.0 * NPTS_OUT / (pack_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                                    "traffic": cf_tr[0] * NPTS_OUT if cf_tr else None, "traffic_source": cf_tr[1] if cf_tr else None,
                                                    "kernel": "k_stats + k_cf_header + k_cf_pack16", "algorithmic_bytes_per_launch": 6.0 * NPTS_OUT}},
                     "pack16_plus_armn_compress_us_per_field": comp_us,
                     "cfg5": {"pipeline_us_per_field": pipe_us,       # fused: extrema from bounds + interpolation straight to tokens + one-pass armn_compress, batch of F, one sync (host wall clock)
                              "fields_per_s": 1e6 / pipe_us, "GBps": 4.0 * NPTS_OUT / (pipe_us * 1e-6) / 1e9,
                              "unfused_us_per_field": pipe_unfused_us,     # interp + pack16 + armn_compress as separate steps
                              "records_equal_unfused": pipe_checked, "zlng_bytes": int(zl[0]), "compression_ratio": float(zl[0]) / (2.0 * NPTS_OUT),
                              # read the source once + write zlng
                              "hbm_frac_of_algorithmic_bytes": (4.0 * NI_S * NJ_S + zl_mean) / (pipe_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                              "traffic_MB_per_field": tr5[0] if tr5 else None, "traffic_source": tr5[1] if tr5 else None,
                              "roofline": roofline_cfg5(pipe_us, zl_mean)},
                     "cpu_baseline": None},
            "checked": bool(checked and checked.get("ok")),
            "check": checked,
        }
        if world == 1:
            ex = extras(ez, torch, stream, d_out, d_in)
            ex["first_call_setup_ms"] = first_call_ms     # c_ezqkdef x 2 + c_ezdefset + ezhip_prepare_set, once per grid pair; steady-state numbers exclude it
            ex["gauss_latitudes_ms"] = gauss_ms
            if "cfg3_uvint" in ex:
                ex["cfg3_uvint"]["cpu_baseline"] = cfg3_cpu
            sf = ex.get("single_field_launch_us")
            if sf:      # north_star words its 60 % target on "a field": the lone-field launch next to the batch launch (a property of the batch launch: DESIGN.md 10.4)
                t1f = profile_value("single_field_traffic_MB")
                s2u = ex.get("single_field_two_streams_us")
                out["roofline_single_field"] = {"bound": "hbm", "achieved": ALGO_BYTES / (sf * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                                "frac": ALGO_BYTES / (sf * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                                "traffic": t1f[0] * 1e6 if t1f else None, "avg_launch_us": sf,
                                                "two_streams_us_per_field": s2u, "two_streams_frac": (ALGO_BYTES / (s2u * 1e-6) / 1e9 / HBM_PEAK_GBPS) if s2u else None}
            out["extras"] = ex
        if hf is not None:
            out["host_fed"] = hf
        check_field_host = d_out[CHECK_F].cpu().numpy() if not args.no_cpu_baseline else None
    if dist:
        dist.destroy_process_group()      # before rank 0's CPU legs: the other ranks leave, nobody waits in a collective
    if rank == 0:
        if not args.no_cpu_baseline:
            # the reference's CPU path on this box's host cores, in the same run, at every N (a SCALE line carries it too).  Bounded
            # sample; at N > 1 the all-cores leg is skipped (the other ranks' processes are ending on the same cores)
            out["cpu_baseline"] = cpu_baseline()
            try:
                out["pack"]["cpu_baseline"] = pack_cpu_baseline(check_field_host, out["cpu_baseline"].get("s_per_field"))
            except Exception as e:   # noqa: BLE001
                out["pack"]["cpu_baseline"] = {"error": repr(e)[:200]}
            if world == 1:
                try:
                    out["cpu_baseline_all_cores"] = cpu_baseline_all_cores()
                except Exception as e:   # noqa: BLE001
                    out["cpu_baseline_all_cores"] = {"error": repr(e)[:200]}
        line = json.dumps(slim(out), separators=(",", ":"))
        if len(line) > 7600:
            sys.stderr.write("bench.py: the line is %d bytes; the driver keeps an 8 KB tail\n" % len(line))
        print(line, flush=True)


if __name__ == "__main__":
    main()
