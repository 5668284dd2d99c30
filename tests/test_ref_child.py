"""tests/ref_child.py (the reference build in a process of its own: the GPU suite's whole-field comparisons and bench.py's CPU baselines rest on it) against the committed
reference-run fixtures: the child's full cfg3 wind and scalar fields contain the fixtures' rows and columns bit for bit and reproduce their float64 sums.  CPU only."""
import json, os, subprocess, sys
import numpy as np
import pytest
import reflib

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not reflib.have_ref(), reason="oracle/_ref/libezref.so not built")


def _run(args, tmp_path):
    out = str(tmp_path / "out.npy")
    r = subprocess.run([sys.executable, os.path.join(HERE, "ref_child.py")] + args + ["--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    return json.loads(r.stdout.strip().splitlines()[-1]), np.load(out)


@pytest.mark.parametrize("degree,polar", [(3, 1), (1, 0)])
def test_cfg3_winds_of_the_child_are_the_fixture(degree, polar, tmp_path):
    G = np.load(os.path.join(HERE, "golden", "cfg3_full_golden.npz"))
    info, uv = _run(["cfg3_uvint", "--reps", "0", "--degree", str(degree), "--polar", str(polar)], tmp_path)
    assert info["rc"] == 0 and info["points"] == 8000000
    key = f"d{degree}_p{polar}"
    for k, nm in enumerate(("u", "v")):
        a = uv[k].reshape(2000, 4000)
        assert np.array_equal(a[G["rows"]].view(np.uint32), G[f"{key}/{nm}/rows"].view(np.uint32))
        assert np.array_equal(a[:, G["cols"]].view(np.uint32), G[f"{key}/{nm}/cols"].view(np.uint32))
        assert float(a.astype(np.float64).sum()) == float(G[f"{key}/{nm}/sum"])


def test_cfg3_scalar_of_the_child_is_the_fixture(tmp_path):
    G = np.load(os.path.join(HERE, "golden", "cfg3_full_golden.npz"))
    info, z = _run(["cfg3_sint"], tmp_path)
    a = z.reshape(2000, 4000)
    assert info["rc"] == 0
    assert np.array_equal(a[G["rows"]].view(np.uint32), G["d3_p1/z/rows"].view(np.uint32)) and np.array_equal(a[:, G["cols"]].view(np.uint32), G["d3_p1/z/cols"].view(np.uint32))
    assert float(a.astype(np.float64).sum()) == float(G["d3_p1/z/sum"])
