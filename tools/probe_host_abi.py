"""c_ezsint on host arrays (cfg2 shape), output array reused: ms per call between ordinary arrays, and between arrays registered with
ezhip_register_host_buffer for 1..16 row ranges; bitwise equality of the results."""
import ctypes, os, sys, time
import numpy as np
_R = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, __import__("os").path.join(_R, "tests"))
import torch
from librmn_amd import ezscint as ez, load_library
import ezcases as ec
L = load_library()
L.ezhip_register_host_buffer.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
L.ezhip_unregister_host_buffer.argtypes = [ctypes.c_void_p]
L.c_ezsint.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
ni, nj, no, mo = 4400, 2200, 7200, 3601
gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
ez.ezdefset(gdout, gdin)
zin = ec.synth_field(ni, nj, seed=2)
zout = np.zeros(no * mo, np.float32)
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
def timed(label):
    assert L.c_ezsint(p(zout), p(zin)) == 0
    t0 = time.perf_counter()
    for _ in range(10):
        L.c_ezsint(p(zout), p(zin))
    print(f"{label:42s} {(time.perf_counter() - t0) / 10 * 1e3:7.3f} ms per c_ezsint", flush=True)
    return zout.copy()
ref = timed("ordinary arrays")
t0 = time.perf_counter()
assert L.ezhip_register_host_buffer(p(zin), zin.nbytes) == 0 and L.ezhip_register_host_buffer(p(zout), zout.nbytes) == 0
print(f"registering both arrays: {(time.perf_counter() - t0) * 1e3:.2f} ms")
for k in ("1", "2", "4", "6", "8", "12", "16"):
    os.environ["EZHIP_HOST_CHUNKS"] = k
    z = timed(f"registered arrays, {k} row ranges")
    assert np.array_equal(ref.view(np.uint32), z.view(np.uint32)), "result differs"
os.environ["EZHIP_HOST_NO_CHUNKS"] = "1"
timed("registered arrays, whole copies")
L.ezhip_unregister_host_buffer(p(zin)); L.ezhip_unregister_host_buffer(p(zout))
