"""CPU, world_size > 1: the base-sharded MSM exchange (all-gather of one point per rank + local fold through the library's
dr_g1_sum) over gloo and over the launcher-independent TCP communicator.  The per-rank partials stand in for each rank's GPU
MSM here (no GPU in this suite); tests/test_gpu_sharded_msm.py runs the same exchange with real GPU shards."""
import os
import random
import socket
import sys

import multiprocessing as mp

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _case(n, zero_shard_of=None, world=2):
    """seeded points / scalars; the scalars of rank `zero_shard_of`'s shard are all zero (its partial is the point at infinity)"""
    from dot_ring_amd import parallel
    from oracle import coracle
    from oracle.pyref import kzg

    rng = random.Random(11 + n)
    pts = [coracle.g1_mul(kzg.G1_GEN, rng.randrange(1, coracle.FR_P)) for _ in range(n)]
    ks = [rng.randrange(coracle.FR_P) for _ in range(n)]
    if zero_shard_of is not None:
        lo, hi = parallel.shard_range(n, zero_shard_of, world)
        for i in range(lo, hi):
            ks[i] = 0
    return pts, ks


def _partial(pts, ks, lo, hi):
    from oracle import coracle
    from oracle.pyref import kzg

    if hi <= lo:
        return None
    part = coracle.g1_msm(pts[lo:hi], ks[lo:hi])
    return None if part is None else kzg.serialize(part)


def _want(pts, ks):
    from oracle import coracle
    from oracle.pyref import kzg

    total = coracle.g1_msm(pts, ks)
    return None if total is None else kzg.serialize(total)


def _gloo_worker(rank, world, port, n, zero_shard_of, out_q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from dot_ring_amd import parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pts, ks = _case(n, zero_shard_of, world)
    lo, hi = parallel.shard_range(n, rank, world)
    part = _partial(pts, ks, lo, hi)
    total = parallel.combine_partials(part)                  # torch.distributed default group (gloo)
    out_q.put((rank, total == _want(pts, ks), (lo, hi), part is None))
    dist.barrier()
    dist.destroy_process_group()


def _socket_worker(rank, world, port, n, zero_shard_of, out_q):
    sys.path.insert(0, ROOT)
    from dot_ring_amd import parallel

    comm = parallel.SocketComm(rank, world, "127.0.0.1", port)
    pts, ks = _case(n, zero_shard_of, world)
    lo, hi = parallel.shard_range(n, rank, world)
    part = _partial(pts, ks, lo, hi)
    total = parallel.combine_partials(part, comm)
    gathered = comm.all_gather(bytes([rank]) * 3)
    comm.barrier()
    ok = total == _want(pts, ks) and gathered == [bytes([r]) * 3 for r in range(world)]
    out_q.put((rank, ok, (lo, hi), part is None))
    comm.close()


def _run(worker, world, n, zero_shard_of):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, n, zero_shard_of, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in results)
    spans = [span for _, _, span, _ in results]
    assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    return results


@pytest.mark.parametrize("n,zero_shard_of", [(1, None), (37, None), (37, 1), (6, 0)])
def test_sharded_msm_all_gather_fold_gloo(n, zero_shard_of):
    """world 2 over gloo: ragged shards (37 = 19 + 18; 1 = 1 + 0: an empty shard), a shard whose scalars are all zero (its
    partial is the point at infinity), the fold consuming the library's own dr_g1_sum"""
    results = _run(_gloo_worker, 2, n, zero_shard_of)
    if zero_shard_of is not None:
        assert results[zero_shard_of][3]                     # that rank really contributed infinity


@pytest.mark.parametrize("world,n,zero_shard_of", [(2, 37, 1), (3, 2, None), (3, 40, 2)])
def test_sharded_msm_all_gather_fold_socket(world, n, zero_shard_of):
    """the launcher-independent TCP communicator (what bootstraps RCCL, and what N-ranks-on-one-GPU rehearsals use):
    world 2 and 3, an empty shard (n = 2 over 3 ranks), an all-zero shard"""
    _run(_socket_worker, world, n, zero_shard_of)


def test_shard_range_covers_everything():
    from dot_ring_amd import parallel

    for n in (0, 1, 7, 8, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    with pytest.raises(ValueError):
        parallel.shard_range(4, 2, 2)


def test_single_rank_communicators_are_trivial():
    from dot_ring_amd import parallel

    c = parallel.SocketComm(0, 1)
    assert c.all_gather(b"abc") == [b"abc"]
    c.barrier()
    c.close()
    assert parallel.combine_partials(None, c) is None
