"""Host-side logic of a multi-GPU run: which rank takes which block of the input, and how the ranks'
measurements combine. One process per GPU (torch.distributed: RCCL on GPUs, gloo in CPU tests).

Input sharding: the synthetic LUBM input is cut by university; rank r generates universities
[r*U, (r+1)*U). What the ranks do with their blocks (side by side, or one partitioning of the union with
exchanges) is decided by the caller; the functions here are host logic that CPU tests cover with gloo.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def university_block(rank: int, world: int, per_gpu: int) -> tuple[int, int]:
    """[begin, end) of the universities rank `rank` generates; blocks tile [0, world*per_gpu)."""
    if not 0 <= rank < world or per_gpu < 0:
        raise ValueError("rank %d outside world %d" % (rank, world))
    return rank * per_gpu, (rank + 1) * per_gpu


def partition_owner(partition: int, world: int) -> int:
    """GPU that holds first-level partition `partition` once partitions are exchanged
    (the reference's CustomPartitioner routes by the partition id itself, DAP.scala:541-551)."""
    return partition % world


def combine_timings(elapsed_s: float, partition_s: float, n_triples: int, device: torch.device):
    """(max elapsed, max partition time, total triples) over all ranks: whole-job throughput is the
    sum of the work divided by the slowest rank's time."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return elapsed_s, partition_s, n_triples
    t = torch.tensor([elapsed_s, partition_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = torch.tensor([n_triples], dtype=torch.int64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t[0]), float(t[1]), int(n[0])


def combine_counts(counts: dict, device: torch.device) -> dict:
    """Per-query result counts summed over ranks (each rank answers over its own shard)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(counts)
    names = sorted(counts)
    t = torch.tensor([counts[n] for n in names], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {n: int(v) for n, v in zip(names, t.tolist())}


def barrier(device: torch.device):
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
