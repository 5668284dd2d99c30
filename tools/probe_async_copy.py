"""Does hipMemcpyAsync on PAGEABLE host memory return before the copy is done, and do an H2D and a D2H on two streams overlap?
Development probe for the host-pointer ABI (DESIGN_LOG.md section 5)."""
import ctypes, time, sys
import numpy as np
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
import torch
from librmn_amd import load_library
L = load_library()
L.ezhip_malloc.restype = ctypes.c_void_p; L.ezhip_malloc.argtypes = [ctypes.c_size_t]
L.ezhip_h2d.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
L.ezhip_d2h.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
L.ezhip_set_stream.argtypes = [ctypes.c_void_p]
n_in, n_out = 4400 * 2200 * 4, 7200 * 3601 * 4
hin = np.ones(n_in // 4, np.float32); hout = np.zeros(n_out // 4, np.float32)
din = L.ezhip_malloc(n_in); dout = L.ezhip_malloc(n_out)
s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()
pin = lambda a: a.ctypes.data_as(ctypes.c_void_p)
for rep in range(3):
    L.ezhip_set_stream(ctypes.c_void_p(s1.cuda_stream))
    t0 = time.perf_counter(); L.ezhip_h2d(din, pin(hin), n_in); t1 = time.perf_counter(); L.ezhip_sync(); t2 = time.perf_counter()
    print(f"H2D {n_in/1e6:.1f} MB: call returns after {1e3*(t1-t0):.3f} ms, done after {1e3*(t2-t0):.3f} ms ({n_in/(t2-t0)/1e9:.1f} GB/s)")
    t0 = time.perf_counter(); L.ezhip_d2h(pin(hout), dout, n_out); t1 = time.perf_counter(); L.ezhip_sync(); t2 = time.perf_counter()
    print(f"D2H {n_out/1e6:.1f} MB: call returns after {1e3*(t1-t0):.3f} ms, done after {1e3*(t2-t0):.3f} ms ({n_out/(t2-t0)/1e9:.1f} GB/s)")
    # both directions at once on two streams
    t0 = time.perf_counter()
    L.ezhip_set_stream(ctypes.c_void_p(s2.cuda_stream)); L.ezhip_d2h(pin(hout), dout, n_out); ta = time.perf_counter()
    L.ezhip_set_stream(ctypes.c_void_p(s1.cuda_stream)); L.ezhip_h2d(din, pin(hin), n_in); tb = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"D2H on stream 2 + H2D on stream 1: calls return after {1e3*(ta-t0):.3f} / {1e3*(tb-t0):.3f} ms, both done after {1e3*(t2-t0):.3f} ms")
# chunked D2H: 8 pieces
    t0 = time.perf_counter()
    step = n_out // 8 // 4 * 4
    for k in range(8):
        L.ezhip_d2h(ctypes.c_void_p(hout.ctypes.data + k * step), ctypes.c_void_p(dout + k * step), step)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"D2H in 8 pieces: calls return after {1e3*(t1-t0):.3f} ms, done after {1e3*(t2-t0):.3f} ms")

# two host threads, one direction each (ctypes drops the GIL inside the calls)
import threading
def worker(stream, fn, *a):
    L.ezhip_set_stream(ctypes.c_void_p(stream.cuda_stream)); fn(*a); L.ezhip_sync()
for rep in range(3):
    ta = threading.Thread(target=worker, args=(s1, L.ezhip_h2d, din, pin(hin), n_in))
    tb = threading.Thread(target=worker, args=(s2, L.ezhip_d2h, pin(hout), dout, n_out))
    t0 = time.perf_counter(); tb.start(); ta.start(); ta.join(); t1 = time.perf_counter(); tb.join(); t2 = time.perf_counter()
    print(f"two threads: H2D thread done after {1e3*(t1-t0):.3f} ms, D2H thread done after {1e3*(t2-t0):.3f} ms")

# page-locked arrays: asynchronous copies, per stream and both directions at once
L.ezhip_register_host_buffer.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.ezhip_register_host_buffer(pin(hin), n_in) == 0 and L.ezhip_register_host_buffer(pin(hout), n_out) == 0
s3 = torch.cuda.Stream(); s4 = torch.cuda.Stream()
streams = {"null": None, "s1": s1, "s2": s2, "s3": s3, "s4": s4}
def setst(s):
    L.ezhip_set_stream(ctypes.c_void_p(s.cuda_stream if s is not None else 0))
for name, st in streams.items():
    setst(st)
    for rep in range(2):
        t0 = time.perf_counter(); L.ezhip_d2h(pin(hout), dout, n_out); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    d = t2 - t0
    t0 = time.perf_counter(); L.ezhip_h2d(din, pin(hin), n_in); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"registered, stream {name}: D2H call returns after {1e3*(t1-t0):.3f} ms, done {1e3*d:.3f} ms ({n_out/d/1e9:.1f} GB/s); H2D {1e3*(t3-t0):.3f} ms")
for a, b in (("s1", "s2"), ("s2", "s1"), ("null", "s2"), ("s3", "s4")):
    for rep in range(2):
        t0 = time.perf_counter()
        setst(streams[a]); L.ezhip_d2h(pin(hout), dout, n_out)
        setst(streams[b]); L.ezhip_h2d(din, pin(hin), n_in)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"registered, D2H on {a} + H2D on {b}: both done after {1e3*(t2-t0):.3f} ms")
