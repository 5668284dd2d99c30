"""bench.py's own N-rank launcher (the driver runs `python bench.py --gpus N`, or torchrun with WORLD_SIZE set).

CPU tests: the launcher, the rendezvous on 127.0.0.1, the barriers, the max-over-ranks time and the rank count run over gloo with
`--stub-step` (a sleep instead of the GPU step: the line says "stub": true and carries no throughput).  What they pin:
  * `--gpus 2` with no WORLD_SIZE starts 2 fresh rank processes and reports n_gpus = what an all_reduce of ones returned;
  * the time is the slowest rank's;
  * `--gpus N` with fewer than N devices exits non-zero (no false N-GPU line), so does a --gpus / WORLD_SIZE disagreement."""
import json
from conftest import run_child
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env.update(extra)
    return env


def _run(args, env=None, timeout=180):
    return run_child([sys.executable, BENCH] + args, env=env or _clean_env(), timeout=timeout)


def _line(proc):
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (proc.stdout, proc.stderr)          # rank 0 prints ONE line, the other ranks none
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_and_counts_them():
    p = _run(["--gpus", "2", "--stub-step", "--steps", "5", "--warmup", "1", "--fields-per-step", "4"])
    assert p.returncode == 0, p.stderr
    out = _line(p)
    assert out["stub"] is True and out["value"] is None
    assert out["n_gpus"] == 2 and out["world_size_env"] == 2 and out["gpus_flag"] == 2
    assert out["launched_by"] == "bench.py"
    assert out["fields_owned_by_all_ranks"] == 8                 # 4 per rank, every record owned exactly once
    assert out["ms_per_step"] >= 4.0                             # rank 1 sleeps 4 ms per step, rank 0 2 ms: the slowest rank's time


def test_external_launcher_env_is_honoured():
    """the driver's other form: torchrun sets WORLD_SIZE / RANK / LOCAL_RANK / MASTER_*; bench.py must not spawn again"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = _clean_env(WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--stub-step", "--steps", "3", "--warmup", "0"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["launched_by"] == "external launcher"
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]


def test_single_process_default():
    out = _line(_run(["--stub-step", "--steps", "2", "--warmup", "0"]))
    assert out["n_gpus"] == 1 and out["gpus_flag"] is None


def test_flag_and_launcher_must_agree():
    p = _run(["--gpus", "2", "--stub-step"], env=_clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_more_gpus_than_devices_is_refused():
    """no stub: the real path.  This container has no GPU, the round-end box has one: `--gpus 9` can never be satisfied on one node"""
    p = _run(["--gpus", "9", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0
    assert "GPU(s) visible" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_gpus_2_on_a_one_gpu_box_is_refused():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0 and "GPU(s) visible" in p.stderr


@pytest.mark.gpu
def test_rank_body_with_a_real_process_group_of_one():
    """the N-rank code path of bench.py on the GPU box's one device: RCCL process group (BENCH_FORCE_DIST), barriers around the timed steps, the rank
    count from an all_reduce, host-fed object, checked outputs -- what the driver's `torch.distributed.run ... bench.py --gpus N` exercises per rank"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = _clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_FORCE_DIST="1")
    p = _run(["--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--host-fed-steps", "2"], env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = _line(p)
    assert out["n_gpus"] == 1 and out["config"]["launched_by"] == "external launcher" and out["config"]["feed"] == "device"
    assert out["checked"] is True and out["config"]["develop_build"] is False
    assert out["roofline"]["bound"] == "hbm" and 0.3 < out["roofline"]["frac"] < 1.0
    assert out["host_fed"]["steps"] == 2 and out["host_fed"]["value"] < out["value"]
    assert out["pack"]["cfg5"]["roofline"]["bound"] == "issue" and out["pack"]["cfg5"]["records_equal_unfused"] is True
    assert len(p.stdout.strip().splitlines()[-1]) < 7600          # the driver keeps an 8 KB tail of the line
