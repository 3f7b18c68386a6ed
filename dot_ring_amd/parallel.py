"""Multi-GPU sharding helpers (one process per GPU, torch.distributed: RCCL on GPUs, gloo in CPU tests).

The Ring-VRF hot path shards two ways: across independent proofs (no collective at all) and across
the bases of one large MSM.  Only the second needs an exchange: each rank reduces its shard of (base, scalar)
pairs to ONE G1 point, the points are all-gathered (96 bytes per rank — latency-bound on xGMI, nowhere near the
per-link bandwidth) and every rank adds them up.  Point addition is not an RCCL reduction op, so this is an
all-gather followed by a local group-law fold, not an all-reduce.
"""
from __future__ import annotations

from . import _native


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [start, stop) slice of n items owned by `rank` (sizes differ by at most one)."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank / world size")
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_gather_points(point: bytes | None, group=None) -> list:
    """All-gather one affine G1 point (96-byte BE record, None = infinity) from every rank."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [point]
    world = dist.get_world_size(group)
    on_gpu = dist.get_backend(group) == "nccl"
    mine = torch.frombuffer(bytearray(point if point is not None else bytes(96)), dtype=torch.uint8)
    if on_gpu:
        mine = mine.cuda()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    out = []
    for g in gathered:
        raw = bytes(g.cpu().numpy().tobytes())
        out.append(None if raw == bytes(96) else raw)
    return out


def combine_partials(point: bytes | None, group=None) -> bytes | None:
    """Sum of every rank's partial MSM result; identical on all ranks."""
    return _native.g1_sum(all_gather_points(point, group))
