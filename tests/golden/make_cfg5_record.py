"""Generates tests/golden/cfg5_record_golden.npz: BASELINE cfg5 end to end on the REFERENCE side -- the reference's own c_ezsint
(oracle/_ref/libezref.so) of tests/ezcases.synth_field(seed=2) at full size (G 4400x2200 -> L 7200x3601, bicubic, polar correction),
then the oracle's compact_float (16 bits in 16-bit slots) and armn_compress of THAT field (the reference's packers cannot be built here:
oracle/orc_pack.c restates them, PARITY UNPINNED as everywhere for the packers).  Kept: the record's byte count zlng, its four header
words, the 16-bit tokens on sampled rows and columns, a hash of all tokens.  tests/test_gpu_packers.py::test_cfg5_record_against_reference_interpolated_floats
bounds how far the HIP pipeline's record (its bicubic values are within 1 ulp of the reference's, not identical) lies from this one.
Runs only in the build container.

    ulimit -s unlimited; python tests/golden/make_cfg5_record.py
"""
import ctypes, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from reflib import ref, fptr          # noqa: E402
import ezcases as ec                  # noqa: E402
import test_oracle_packers as top     # noqa: E402

NI, NJ, NO, MO = 4400, 2200, 7200, 3601
ROWS = np.array([0, 1, 2, 3, 4, 17, 900, 1800, 1801, 2700, 3596, 3597, 3598, 3599, 3600])
COLS = np.array([0, 1, 2, 3, 100, 3599, 3600, 7196, 7197, 7198, 7199])


def main():
    L = ref()
    gdin = L.c_ezqkdef(NI, NJ, b"G", 0, 0, 0, 0, 0)
    gdout = L.c_ezqkdef(NO, MO, b"L", 5, 5, 0, 0, 0)
    assert L.c_ezdefset(gdout, gdin) == 1
    L.c_ezsetopt(b"interp_degree", b"cubic"); L.c_ezsetopt(b"polar_correction", b"yes")
    zin = ec.synth_field(NI, NJ, seed=2)
    z = np.zeros((MO, NO), np.float32)
    assert L.c_ezsint(fptr(z), fptr(zin)) == 0
    rec = top.pack_float(z.ravel(), 16 + 64 * 16)
    w = rec[4:4 + NO * MO // 2]
    tok = np.empty(NO * MO, np.uint16); tok[0::2] = (w >> 16).astype(np.uint16); tok[1::2] = (w & 0xFFFF).astype(np.uint16)
    tok = tok.reshape(MO, NO)
    zlng = top.O().orc_armn_compress(rec[4:].ctypes.data, NO, MO, 1, 16, 1)
    assert zlng > 0
    out = {"rows": ROWS, "cols": COLS, "zlng": np.int64(zlng), "header": rec[:4].copy(),
           "tok_rows": tok[ROWS].copy(), "tok_cols": tok[:, COLS].copy(),
           "tok_hash": np.array([int(tok.astype(np.uint64).sum()) & 0xFFFFFFFF, int(np.bitwise_xor.reduce(tok.ravel()))], np.uint32),
           "zmin": np.float32(z.min()), "zmax": np.float32(z.max())}
    np.savez_compressed(os.path.join(HERE, "cfg5_record_golden.npz"), **out)
    print("zlng", zlng, "header", [hex(int(x)) for x in rec[:4]], "min/max", z.min(), z.max())


if __name__ == "__main__":
    main()
