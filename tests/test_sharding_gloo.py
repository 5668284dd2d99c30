"""N > 1 path on CPU: world_size-2 gloo processes exercise the record sharding, the max-over-ranks timing
reduction, the source-field broadcast and the record-length gather that bench.py / a batch driver use."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from librmn_amd import sharding as sh


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, nfields, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sh.fields_of_rank(nfields, rank, world)
    # every rank "processes" its fields: the result of field f is a deterministic function of f
    lengths = [1000 + 7 * f for f in mine]
    t = sh.max_over_ranks(0.5 + rank)                       # slowest rank wins
    tot = sh.sum_over_ranks(float(len(mine)))
    src = torch.full((64,), float(rank + 1))
    sh.broadcast_source_field(src, root=0)
    full = sh.gather_record_lengths(lengths, nfields)
    out[rank] = (mine, t, tot, src.tolist(), full)
    dist.destroy_process_group()


def test_world2_record_sharding_and_reductions():
    world, nfields = 2, 9
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), nfields, out), nprocs=world, join=True)
    all_fields = sorted(out[0][0] + out[1][0])
    assert all_fields == list(range(nfields))               # every record exactly once
    assert out[0][0] == [0, 2, 4, 6, 8] and out[1][0] == [1, 3, 5, 7]
    for r in range(world):
        assert out[r][1] == 1.5                             # max over ranks of (0.5, 1.5)
        assert out[r][2] == float(nfields)
        assert out[r][3] == [1.0] * 64                      # root's source field everywhere
        assert out[r][4] == [1000 + 7 * f for f in range(nfields)]


def test_single_process_degenerates_cleanly():
    assert sh.fields_of_rank(5, 0, 1) == [0, 1, 2, 3, 4]
    assert sh.max_over_ranks(2.5) == 2.5 and sh.sum_over_ranks(3.0) == 3.0
    assert sh.targets_of_rank(4, 1, 2) == [1, 3] and sh.owner_of_field(7, 4) == 3
