"""CPU, world_size 2 over gloo: the base-sharded MSM exchange (all-gather of one point per rank + local fold)."""
import os
import random
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, out_q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from dot_ring_amd import parallel
    from oracle import coracle
    from oracle.pyref import kzg

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = random.Random(11)
    pts = [coracle.g1_mul(kzg.G1_GEN, rng.randrange(1, coracle.FR_P)) for _ in range(n)]
    ks = [rng.randrange(coracle.FR_P) for _ in range(n)]
    lo, hi = parallel.shard_range(n, rank, world)
    # the per-rank partial stands in for this rank's GPU MSM over its shard of the bases
    part = coracle.g1_msm(pts[lo:hi], ks[lo:hi]) if hi > lo else None
    total = parallel.combine_partials(None if part is None else kzg.serialize(part))
    want = coracle.g1_msm(pts, ks)
    out_q.put((rank, total == kzg.serialize(want), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1, 37])
def test_sharded_msm_all_gather_fold(n):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results)
    assert results[0][2][1] == results[1][2][0]          # contiguous, disjoint shards
    assert results[1][2][1] == n


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
