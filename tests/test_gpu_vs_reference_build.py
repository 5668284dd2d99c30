"""The product on the GPU against the reference's OWN sources (oracle/_ref/libezref.so, built where they lie and shipped with the snapshot)
on random grid pairs -- no oracle in between.  tools/fuzz_vs_ref.py: c_ezsint nearest / bilinear bit for bit, bicubic within 1e-5, c_ezuvint
within 1e-5 |V|, global / hemispheric / regional sources, every extrapolation mode; tools/fuzz_average.py: the averaging degrees bit for bit."""
import os, subprocess, sys
import pytest
import reflib

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


@pytest.mark.skipif(not reflib.have_ref(), reason="oracle/_ref/libezref.so not built")
@pytest.mark.parametrize("tool,ncases,seed", [("fuzz_vs_ref.py", 400, 1), ("fuzz_vs_ref.py", 400, 2), ("fuzz_average.py", 200, 1)])
def test_random_grid_pairs_against_the_reference_build(tool, ncases, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(ncases), str(seed)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout
