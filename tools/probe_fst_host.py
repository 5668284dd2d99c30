"""ezhip_fst_pack_data / ezhip_fst_unpack_data of one REAL 7200 x 3601 field as datyp 129, 16 bits, between host arrays: one PCIe trip each way (default) against the
four of compact_float + armn_compress on host arrays (EZHIP_FST_FOUR_TRIPS=1); ms per call and the data parts compared"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from librmn_amd import packers as pk
import ezcases as ec
ni, nj = 7200, 3601
f = ec.synth_field(ni, nj, seed=3, noise=1e-4)
ref = None
for variant in ("", "1", "", "1"):
    os.environ.pop("EZHIP_FST_FOUR_TRIPS", None)
    if variant: os.environ["EZHIP_FST_FOUR_TRIPS"] = "1"
    for rep in range(3):
        t0 = time.perf_counter(); out = pk.fst_pack_data(f, ni, nj, 1, 129, 16); t1 = time.perf_counter()
    words, datyp_out, nbits_out, data = out
    full = data; data = data[:words].copy()          # (the four-trip form expands the stream IN PLACE: it needs the whole buffer)
    for rep in range(3):
        t2 = time.perf_counter(); rc, back = pk.fst_unpack_data(full.copy() if variant else data, ni, nj, 1, datyp_out, nbits_out); t3 = time.perf_counter()
    same = None
    if ref is None: ref = (data.copy(), back.copy())
    else: same = bool(np.array_equal(ref[0], data) and np.array_equal(ref[1].view(np.uint32), back.view(np.uint32)))
    print(f"[{'four trips' if variant else 'one trip each way'}] pack {1e3 * (t1 - t0):.2f} ms, unpack {1e3 * (t3 - t2):.2f} ms, words {data.size}, datyp {datyp_out}, equal to the first: {same}", flush=True)
