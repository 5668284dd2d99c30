"""Generates tests/golden/host_tables_golden.npz from the reference's own ez_nwtncof (f_ezscint.F90 / ez_nwtncof.inc) and ez_xpncof (ez_xpncof.c) in
oracle/_ref/libezref.so on the inputs of tests/hostcases.py.  Runs only in the build container.

    python tests/golden/make_host_tables.py
"""
import ctypes, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from reflib import ref, fptr          # noqa: E402
import hostcases as hc                # noqa: E402


def main():
    L = ref()
    I = lambda v: ctypes.byref(ctypes.c_int32(v))
    out = {}
    for name, (ax, ay, ext) in hc.nwtncof_cases().items():
        ni, nj = ax.size, ay.size
        cx = np.zeros(6 * ni, np.float32); cy = np.zeros(6 * nj, np.float32)
        L.ez_nwtncof_(fptr(cx), fptr(cy), fptr(ax), fptr(ay), I(ni), I(nj), I(1), I(ni), I(1), I(nj), I(ext))
        out[f"nwtncof/{name}/cx"] = cx; out[f"nwtncof/{name}/cy"] = cy
    L.ez_xpncof.restype = None
    for name, (ni, nj, grtyp, grref, ig1, ig2, ig3, ig4, ax, ay) in hc.xpncof_cases().items():
        v = [ctypes.c_int32(-99) for _ in range(5)]
        L.ez_xpncof(*[ctypes.byref(x) for x in v], ni, nj, ctypes.c_char(grtyp.encode()), ctypes.c_char(grref.encode()), ig1, ig2, ig3, ig4, 0,
                    fptr(ax) if ax is not None else None, fptr(ay) if ay is not None else None)
        out[f"xpncof/{name}"] = np.array([x.value for x in v], np.int32)
    np.savez_compressed(os.path.join(HERE, "host_tables_golden.npz"), **out)
    for k in sorted(out):
        if k.startswith("xpncof"):
            print(k, out[k])


if __name__ == "__main__":
    main()
