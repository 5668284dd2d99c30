"""The product's N-rank programs on the GPU box, each as a FRESH child process with a real RCCL process group (world size 1: the box has
one GPU; the same code path as N ranks -- init_process_group("nccl"), barriers, all_reduce / broadcast on device tensors):
  * tools/run_sharded.py --case cfg4 | cfg5 | targets --verify   (librmn_amd.batch_driver.HipBackend: BASELINE configs[3], configs[4] and
    the one-source -> many-target-grids broadcast); rank 0 re-computes its records through the plain single-field entry points;
  * bench.py under BENCH_FORCE_DIST=1 (the driver's scaling instrument): `checked` = the timed launch's outputs against the reference's own
    full-size run.
SURVEY.md section 8e.  The 8-GPU scaling curve is the driver's job; this pins the code path it runs."""
import json, os, socket, subprocess, sys
from conftest import run_child
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _child(argv, extra_env, timeout=900):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra_env)
    r = run_child([sys.executable] + argv, cwd=ROOT, env=env, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, (r.stdout[-1500:], r.stderr[-1500:])
    return json.loads(lines[-1])


@pytest.mark.parametrize("case,size,fields", [("cfg4", "small", 24), ("cfg5", "small", 24), ("cfg4", "full", 8), ("cfg5", "full", 8)])
def test_run_sharded_with_an_rccl_process_group(case, size, fields):
    out = _child([os.path.join(ROOT, "tools", "run_sharded.py"), "--case", case, "--size", size, "--fields", str(fields), "--chunk", "8", "--verify"],
                 {"RUN_SHARDED_FORCE_DIST": "1"})
    assert out["world"] == 1 and out["fields"] == fields
    assert out["verified_against_single_field_calls"] is True
    assert len(out["checksums"]) == fields and all(c != 0 for c in out["checksums"])
    if case == "cfg5":
        assert len(out["zlng"]) == fields and all(z > 0 for z in out["zlng"])


def test_many_targets_broadcast_with_an_rccl_process_group():
    dist = _child([os.path.join(ROOT, "tools", "run_sharded.py"), "--case", "targets", "--size", "small"], {"RUN_SHARDED_FORCE_DIST": "1"})
    plain = _child([os.path.join(ROOT, "tools", "run_sharded.py"), "--case", "targets", "--size", "small"], {})
    assert dist["targets"] == plain["targets"] == 8
    assert dist["checksums"] == plain["checksums"] and all(c != 0 for c in dist["checksums"])


def test_bench_under_a_process_group_checks_its_outputs():
    out = _child([os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline"], {"BENCH_FORCE_DIST": "1"})
    assert out["n_gpus"] == 1 and out["steps"] == 3
    assert out["checked"] is True, out.get("check")
    assert out["roofline"]["frac"] > 0.3 and out["pack"]["cfg5"]["records_equal_unfused"] is True
