"""The record-sharded batch drivers (librmn_amd/batch_driver.py) with world_size 1 and 2 over gloo on CPU.  The device work is
replaced by a stand-in backend defined HERE (a test double: the product has no CPU path); what is under test is the
orchestration -- ownership of records, chunking, what is gathered -- and the property the GPU box cannot show with one GPU:
two ranks return exactly the records (byte counts, checksums) that one rank returns."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from librmn_amd import batch_driver as bd


class FakeBackend:
    """deterministic stand-in: 'interpolation' = a fixed linear map of the field, 'record' = the field's words behind a 4-word header"""

    def define_set(self, src, dst):
        return {"nin": src[0] * src[1], "nout": dst[0] * dst[1], "ni_out": dst[0], "nj_out": dst[1], "k": dst[0] + 3 * dst[1]}

    def to_device(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def interp(self, h, fields):
        return [torch.roll(f, h["k"] % f.numel())[: h["nout"]] * 2.0 + 1.0 for f in fields]

    def interp_pack(self, h, fields, nbits=16):
        outs = self.interp(h, fields)
        recs, zl = [], []
        for o in outs:
            body = (o * 100.0).to(torch.int32)
            recs.append(torch.cat([torch.tensor([1, 2, 3, nbits], dtype=torch.int32), body]))
            zl.append(1 + 4 * int(body.numel()) - 8 * int(body[0].item() % 3))
        return recs, zl

    def checksum(self, x, nbytes=-1):
        w = x.contiguous().view(torch.int32)
        if nbytes >= 0:
            w = w[: nbytes // 4]
        u = w.to(torch.int64) & 0xFFFFFFFF
        idx = torch.arange(u.numel(), dtype=torch.int64)
        return int(((u * ((idx % 8191) + 1)).sum() & 0x3FFFFFFFFFFFFFFF).item())

    def sync(self):
        pass


SRC, DST = (40, 20, "G", 0, 0, 0, 0), (30, 10, "L", 100, 100, 0, 0)


def _field(f):
    return torch.arange(800, dtype=torch.float32) * 0.5 + float(f)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, nfields, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    be = FakeBackend(); h = be.define_set(SRC, DST)
    a = bd.run_interp_batch(be, h, nfields, _field, rank, world, chunk=3)
    b = bd.run_pack_batch(be, h, nfields, _field, rank, world, chunk=4)
    targets = [(10 + t, 5 + t, "L", 1, 1, 0, 0) for t in range(5)]
    c = bd.run_many_targets(be, SRC, targets, (lambda: _field(7) if rank == 0 else torch.zeros(800)), rank, world)
    out[rank] = (a["checksums"], a["records_of_rank"], b["zlng"], b["checksums"], c["checksums"], c["targets_of_rank"])
    dist.destroy_process_group()


def _single(nfields):
    be = FakeBackend(); h = be.define_set(SRC, DST)
    a = bd.run_interp_batch(be, h, nfields, _field, 0, 1, chunk=5)
    b = bd.run_pack_batch(be, h, nfields, _field, 0, 1, chunk=5)
    targets = [(10 + t, 5 + t, "L", 1, 1, 0, 0) for t in range(5)]
    c = bd.run_many_targets(be, SRC, targets, lambda: _field(7), 0, 1)
    return a, b, c


def test_two_ranks_return_the_records_of_one_rank():
    nfields = 11
    a1, b1, c1 = _single(nfields)
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), nfields, out), nprocs=2, join=True)
    for r in (0, 1):
        assert out[r][0] == a1["checksums"]                 # every record, identical to the single-rank run, on every rank
        assert out[r][2] == b1["zlng"] and out[r][3] == b1["checksums"]
        assert out[r][4] == c1["checksums"]                 # the broadcast source reached the non-root rank
    assert out[0][1] == [0, 2, 4, 6, 8, 10] and out[1][1] == [1, 3, 5, 7, 9]
    assert out[0][5] == [0, 2, 4] and out[1][5] == [1, 3]
    assert len(set(a1["checksums"])) == nfields             # the checksum tells records apart
