"""Record-sharded batch drivers over the C ABI (BASELINE configs[3], configs[4] and the
one-source -> many-target-grids case).  One process per GPU; `torch.distributed` only carries bookkeeping (record lengths,
checksums, timing) and, for the many-targets case, ONE broadcast of the source field -- the data path has no collective.

Every driver takes a *backend*: the object that owns the device work.  `HipBackend` is the product (librmn_ez_hip.so through
ctypes, device tensors); the multi-process CPU tests pass a stand-in with the same four methods to exercise the orchestration
(who owns which record, what is gathered, that N ranks return exactly the records one rank returns)."""
import time
from typing import Callable, Dict, List, Sequence

from . import sharding as sh


def record_stride_words(npts_out: int) -> int:
    """words of one cfg5 record buffer: 4 compact_float header words + the 16-bit tokens (or the armn_compress stream that
    replaces them) + slack.  The one definition the drivers and their verifiers share."""
    return 4 + (npts_out + 1) // 2 + 64


class HipBackend:
    """the MI355X path: fields are torch CUDA tensors, calls go through the C ABI on the current stream"""

    def __init__(self):
        import torch
        from . import ezscint as ez, packers as pk
        if not torch.cuda.is_available():
            raise RuntimeError("HipBackend needs a GPU: the hot path has no CPU fallback")
        self.torch, self.ez, self.pk = torch, ez, pk
        self.device = torch.device("cuda", torch.cuda.current_device())
        ez.use_stream(torch.cuda.current_stream().cuda_stream)

    def define_set(self, src, dst):
        """src / dst: (ni, nj, grtyp, ig1..ig4); returns an opaque handle"""
        gi = self.ez.ezqkdef(src[0], src[1], src[2], *src[3:7]); go = self.ez.ezqkdef(dst[0], dst[1], dst[2], *dst[3:7])
        assert gi >= 0 and go >= 0
        return {"gdin": gi, "gdout": go, "nin": src[0] * src[1], "nout": dst[0] * dst[1], "ni_out": dst[0], "nj_out": dst[1]}

    def to_device(self, host_field):
        return self.torch.from_numpy(host_field).to(self.device)

    def interp(self, h, fields: Sequence) -> List:
        """c_ezsint_batch_dev on a list of device source fields -> list of device output fields"""
        t = self.torch
        assert self.ez.ezdefset(h["gdout"], h["gdin"]) == 1
        d_in = t.stack(list(fields)).contiguous()
        d_out = t.empty((len(fields), h["nout"]), dtype=t.float32, device=self.device)
        rc = self.ez.ezsint_batch_dev(d_out, d_in, len(fields))
        assert rc in (0, 2), rc
        return [d_out[k] for k in range(len(fields))]

    def interp_pack(self, h, fields: Sequence, nbits: int = 16):
        """the fused cfg5 pipeline -> (list of record tensors [4 header words | stream], zlng list)"""
        t = self.torch
        assert self.ez.ezdefset(h["gdout"], h["gdin"]) == 1
        d_in = t.stack(list(fields)).contiguous()
        rs = record_stride_words(h["nout"])
        rec = t.zeros((len(fields), rs), dtype=t.int32, device=self.device)
        rc, zl = self.pk.ezsint_pack16_compress_batch_dev(rec, rs, d_in, len(fields), h["ni_out"], h["nj_out"], nbits)
        if rc == -2:
            # the fused call does not apply to this grid pair / shape (odd ni, nbits < 5, a set off the single-launch path, ...):
            # the unfused entry points produce the same records (include/packers_hip.h)
            d_out = t.empty((len(fields), h["nout"]), dtype=t.float32, device=self.device)
            rc = self.ez.ezsint_batch_dev(d_out, d_in, len(fields))
            if rc < 0:
                raise RuntimeError(f"c_ezsint_batch_dev failed ({rc}) for grid set {h['gdin']} -> {h['gdout']}")
            rc, zl = self.pk.pack16_compress_batch_dev(rec, rs, d_out, h["nout"], len(fields), h["ni_out"], h["nj_out"], nbits, prepacked=0)
        if rc < 0:
            raise RuntimeError(f"interp + pack16 + armn_compress failed ({rc}) for grid set {h['gdin']} -> {h['gdout']}, nbits {nbits}")
        return [rec[k] for k in range(len(fields))], [int(z) for z in zl]

    def checksum(self, x, nbytes: int = -1) -> int:
        """order-independent 62-bit checksum of a device tensor's first nbytes (whole words)"""
        t = self.torch
        w = x.view(t.int32)
        if nbytes >= 0:
            w = w[: nbytes // 4]
        u = w.to(t.int64) & 0xFFFFFFFF
        idx = t.arange(u.numel(), device=u.device, dtype=t.int64)
        return int(((u * ((idx % 8191) + 1)).sum() & 0x3FFFFFFFFFFFFFFF).item())

    def sync(self):
        self.torch.cuda.synchronize()


def _gather(values: Dict[int, int], total: int, device="cpu") -> List[int]:
    """every rank contributes the values of the records it owns; everyone gets the full per-record list"""
    import torch
    import torch.distributed as dist
    full = torch.zeros(total, dtype=torch.int64, device=device)
    for f, v in values.items():
        full[f] = int(v)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
    return [int(v) for v in full.tolist()]


def run_interp_batch(backend, handle, nfields: int, make_field: Callable[[int], object], rank: int, world: int,
                     chunk: int = 32, comm_device="cpu") -> Dict:
    """BASELINE configs[3]: nfields records, record f on rank f mod world, `chunk` records per c_ezsint_batch_dev launch.
    Returns {"checksums": per-record output checksum (all records, gathered), "seconds": slowest rank's time}."""
    mine = sh.fields_of_rank(nfields, rank, world)
    sums: Dict[int, int] = {}
    t0 = time.perf_counter()
    for k0 in range(0, len(mine), chunk):
        ids = mine[k0:k0 + chunk]
        outs = backend.interp(handle, [make_field(f) for f in ids])
        for f, o in zip(ids, outs):
            sums[f] = backend.checksum(o)
    backend.sync()
    dt = sh.max_over_ranks(time.perf_counter() - t0, device=comm_device)
    return {"checksums": _gather(sums, nfields, comm_device), "seconds": dt, "records_of_rank": mine}


def run_pack_batch(backend, handle, nfields: int, make_field: Callable[[int], object], rank: int, world: int,
                   chunk: int = 32, nbits: int = 16, comm_device="cpu") -> Dict:
    """BASELINE configs[4]: interpolation + compact_float(16) + armn_compress of every record, sharded by record.  The byte counts
    are gathered (sharding.gather_record_lengths: what a writer needs to lay the records out in one file), so are the checksums
    of the records' defined bytes."""
    mine = sh.fields_of_rank(nfields, rank, world)
    zl_local: List[int] = []
    sums: Dict[int, int] = {}
    t0 = time.perf_counter()
    for k0 in range(0, len(mine), chunk):
        ids = mine[k0:k0 + chunk]
        recs, zl = backend.interp_pack(handle, [make_field(f) for f in ids], nbits)
        for f, r, z in zip(ids, recs, zl):
            zl_local.append(z)
            sums[f] = backend.checksum(r, 16 + ((z - 1) // 4) * 4 if z > 0 else 16 + 2 * handle["nout"])
    backend.sync()
    dt = sh.max_over_ranks(time.perf_counter() - t0, device=comm_device)
    return {"zlng": sh.gather_record_lengths(zl_local, nfields, device=comm_device), "checksums": _gather(sums, nfields, comm_device),
            "seconds": dt, "records_of_rank": mine}


def run_many_targets(backend, src_spec, targets: Sequence, make_source: Callable[[], object], rank: int, world: int,
                     comm_device="cpu") -> Dict:
    """one source field -> many target grids: the root builds the source ON ITS DEVICE, one broadcast (RCCL over xGMI when the
    backend is the GPU one) hands it to every rank, target grid t is interpolated by rank t mod world"""
    src = make_source()                      # every rank allocates; only the root's content matters
    sh.broadcast_source_field(src, root=0)
    mine = sh.targets_of_rank(len(targets), rank, world)
    sums: Dict[int, int] = {}
    for t in mine:
        h = backend.define_set(src_spec, targets[t])
        sums[t] = backend.checksum(backend.interp(h, [src])[0])
    backend.sync()
    return {"checksums": _gather(sums, len(targets), comm_device), "targets_of_rank": mine}
