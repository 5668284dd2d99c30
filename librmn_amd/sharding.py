"""Record-level sharding of a batch of fields across the GPUs of one node.

Every field is independent given the (source, target) grid pair, so a batch shards by record with NO
data-path collective: field f belongs to rank f mod world.  Grid descriptors are tiny and replicated; each
rank rebuilds its own plan.  The only collectives are (a) the timing / bookkeeping reductions of a batch
driver, and (b) the optional broadcast of ONE source field for the one-source -> many-target-grids case
(ranks then shard by target grid).  One process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests)."""
from typing import List


def fields_of_rank(nfields: int, rank: int, world: int) -> List[int]:
    """global field indices owned by `rank` (round-robin by record)"""
    return list(range(rank, nfields, world))


def owner_of_field(f: int, world: int) -> int:
    return f % world


def targets_of_rank(ntargets: int, rank: int, world: int) -> List[int]:
    """many-target-grids case: target grid t belongs to rank t mod world"""
    return list(range(rank, ntargets, world))


def max_over_ranks(value: float, device="cpu") -> float:
    """the batch time is the slowest rank's time"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device="cpu") -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def broadcast_source_field(field, root: int = 0):
    """one-source -> many-target-grids: the root's source field (ni*nj float32, 38.72 MB at cfg2) is broadcast
    once; every rank then interpolates it onto its own target grids.  In place; returns the tensor."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(field, src=root)
    return field


def gather_record_lengths(local_lengths, world_total: int, device="cpu"):
    """lay out a packed output file: every rank contributes zlng of its fields; returns the full per-field list"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(local_lengths)
    world, rank = dist.get_world_size(), dist.get_rank()
    full = torch.zeros(world_total, dtype=torch.int64, device=device)
    for k, f in enumerate(fields_of_rank(world_total, rank, world)):
        full[f] = int(local_lengths[k])
    dist.all_reduce(full, op=dist.ReduceOp.SUM)
    return [int(v) for v in full.tolist()]
