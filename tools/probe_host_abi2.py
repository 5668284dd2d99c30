"""c_ezsint on ordinary host arrays (cfg2 shape): ms per call for copy-thread counts / bounce sizes (each setting needs a fresh process: read at first use)"""
import ctypes, os, subprocess, sys, time
if len(sys.argv) > 1 and sys.argv[1] == "child":
    _R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
    import numpy as np
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
    for _ in range(12):
        t0 = time.perf_counter(); L.c_ezsint(p(zout), p(zin)); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{os.environ.get('EZHIP_COPY_THREADS', '4'):>3s} copy threads, bounce {os.environ.get('EZHIP_BOUNCE_MB', '16'):>2s} MB: min {min(ts):.3f} median {sorted(ts)[6]:.3f} ms per c_ezsint", flush=True)
else:
    for th, mb in [(4, 16), (8, 16), (16, 16), (8, 8), (16, 8), (16, 4), (12, 32)]:
        env = dict(os.environ); env["EZHIP_COPY_THREADS"] = str(th); env["EZHIP_BOUNCE_MB"] = str(mb)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
