"""compact_float on device-resident cfg2 fields (25 927 200 floats, 16-bit tokens), 8 fields in rotation:
   FLOAT_PACK (k_stats + k_cf_header + k_cf_pack16) and FLOAT_UNPACK (k_cf_unpack16), us per field and the rate of
   the algorithmic bytes (6 B per point).      python tools/probe_pack.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from librmn_amd import packers as pk
from librmn_amd import ezscint as ez

N = 7200 * 3601
NB = 8
stream = torch.cuda.current_stream(); ez.use_stream(stream.cuda_stream)
g = torch.Generator(device="cuda"); g.manual_seed(7)
fields = [(torch.rand(N, device="cuda", generator=g) * 80.0 + 230.0) for _ in range(NB)]
recs = [torch.zeros(4 + N // 2 + 64, dtype=torch.int32, device="cuda") for _ in range(NB)]
outs = [torch.empty(N, dtype=torch.float32, device="cuda") for _ in range(NB)]
tagv = np.array([0.0], np.float32)
lib = pk._lib()


def pack(b):
    assert pk.compact_float_pack_dev(fields[b], recs[b], recs[b][4:], N, 16 + 64 * 16) != 0


def unpack(b):
    assert lib.compact_float_dev(outs[b].data_ptr(), recs[b].data_ptr(), recs[b][4:].data_ptr(), N, 16 + 64 * 16, 0, 1, pk.FLOAT_UNPACK, 0, tagv.ctypes.data, 2) != 0


def timeit(f, reps=48):
    for i in range(8):
        f(i % NB)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for i in range(reps):
        f(i % NB)
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


tp = timeit(pack)
tu = timeit(unpack)
err = max(float((outs[b] - fields[b]).abs().max()) for b in range(NB))
print(f"compact_float FLOAT_PACK   16 bit: {tp:7.2f} us per field, {6.0 * N / tp / 1e6:5.2f} TB/s of 6 B per point = {6.0 * N / tp / 1e6 / 8:.3f} of 8 TB/s")
print(f"compact_float FLOAT_UNPACK 16 bit: {tu:7.2f} us per field, {6.0 * N / tu / 1e6:5.2f} TB/s of 6 B per point = {6.0 * N / tu / 1e6 / 8:.3f} of 8 TB/s (includes the header's D2H)")
print(f"round trip max |error| {err:.3e} (quantum {80.0 / 65536 * 2:.3e})")
