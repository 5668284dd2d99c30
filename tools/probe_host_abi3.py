"""c_ezsint on ordinary host arrays (cfg2 shape), ms per call: upload then download on the calling thread (default) against the uploader thread
(EZHIP_HOST_UPLOADER=1: source rows go up range by range on a second thread while the caller fetches finished rows); results compared"""
import ctypes, os, subprocess, sys, time
if len(sys.argv) > 1 and sys.argv[1] == "child":
    _R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
    import numpy as np, zlib
    from librmn_amd import ezscint as ez, load_library
    import ezcases as ec
    L = load_library()
    L.c_ezsint.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    ni, nj, no, mo = 4400, 2200, 7200, 3601
    gdin = ez.ezqkdef(ni, nj, "G", 0, 0, 0, 0); gdout = ez.ezqkdef(no, mo, "L", 5, 5, 0, 0)
    ez.ezdefset(gdout, gdin)
    zin = ec.synth_field(ni, nj, seed=2); zout = np.zeros(no * mo, np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for _ in range(3):
        assert L.c_ezsint(p(zout), p(zin)) == 0
    ts = []
    for _ in range(16):
        t0 = time.perf_counter(); L.c_ezsint(p(zout), p(zin)); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"[{sys.argv[2]}] min {min(ts):.3f} median {sorted(ts)[8]:.3f} ms per c_ezsint, crc32 of the result {zlib.crc32(zout.tobytes()):08x}", flush=True)
else:
    for tag, extra in [("caller uploads", {}), ("uploader thread", {"EZHIP_HOST_UPLOADER": "1"}), ("uploader thread, 12 ranges", {"EZHIP_HOST_UPLOADER": "1", "EZHIP_HOST_CHUNKS": "12"}),
                       ("uploader thread, 3 ranges", {"EZHIP_HOST_UPLOADER": "1", "EZHIP_HOST_CHUNKS": "3"}), ("caller uploads", {})]:
        env = dict(os.environ); env.update(extra)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", tag], env=env)
