"""librmn_amd/csrc/libm_exact.h evaluated ON THE DEVICE against this machine's C library over EVERY REAL argument (2^32 bit patterns) of sinf, cosf, asinf, atanf, and
atan2f over `pairs` million random pairs: prints the count of differing results per function (two NaNs count as equal).  tools/check_libm_exact.c is the same
comparison for the host compilation of the header."""
import ctypes, os, sys, time
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import numpy as np, torch
from librmn_amd import ezscint as ez
import oraclelib as ol
O = ol.oracle()
O.orc_libm_apply.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
CH = 1 << 26
bad = [0] * 5
t0 = time.time()
want = np.empty(CH, np.float32); d_o = torch.empty(CH, dtype=torch.float32, device="cuda")
for c in range((1 << 32) // CH):
    a = (np.arange(CH, dtype=np.uint64) + np.uint64(c * CH)).astype(np.uint32).view(np.float32)
    d_a = torch.from_numpy(a).cuda()
    for fn in range(4):
        with np.errstate(all="ignore"):
            O.orc_libm_apply(fn, a.ctypes.data, None, want.ctypes.data, CH)
        assert ez.libm_exact_probe(fn, d_a, None, d_o) == 0
        torch.cuda.synchronize()
        got = d_o.cpu().numpy()
        bad[fn] += int(((got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))).sum())
rng = np.random.default_rng(9)
done = 0
while done < pairs * 1000000:
    if (done // CH) & 1:
        a = rng.integers(0, 2 ** 32, CH, dtype=np.uint64).astype(np.uint32).view(np.float32); b = rng.integers(0, 2 ** 32, CH, dtype=np.uint64).astype(np.uint32).view(np.float32)
    else:
        ang = rng.uniform(0, 2 * np.pi, CH); r = rng.uniform(0, 1, CH)
        a = (np.sin(ang) * r).astype(np.float32); b = (np.cos(ang) * r).astype(np.float32)
    d_a = torch.from_numpy(a).cuda(); d_b = torch.from_numpy(b).cuda()
    with np.errstate(all="ignore"):
        O.orc_libm_apply(4, a.ctypes.data, b.ctypes.data, want.ctypes.data, CH)
    assert ez.libm_exact_probe(4, d_a, d_b, d_o) == 0
    torch.cuda.synchronize()
    got = d_o.cpu().numpy()
    bad[4] += int(((got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))).sum())
    done += CH
print(f"device libm_exact.h against the C library: arguments with a differing result: sinf {bad[0]}, cosf {bad[1]}, asinf {bad[2]}, atanf {bad[3]} of 4294967296 each; "
      f"atan2f {bad[4]} of {done} pairs; {time.time() - t0:.0f} s, {os.cpu_count()} host threads")
sys.exit(1 if any(bad) else 0)
