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


# ------------------------------------------------------------------------------------------------ proof-sharded batches
# The exchange of parallel.prove_batch_sharded / batch_verify_sharded without a GPU: the scheme's two GPU calls are replaced by a
# stand-in that derives 784 "proof" bytes from the arguments of each proof, everything else — shard ranges, the status header,
# the variable-size gather, lazily decoded proof objects over the gathered string, the AND of the verdicts, a failing shard — is
# the product code.  tests/test_gpu_sharded_prove.py runs the same calls with real proofs on the GPU.
def _fake_scheme():
    import hashlib

    import dot_ring_amd as d

    base = d.RingVRF[d.Bandersnatch]

    class FakeRingVRF(base):
        @staticmethod
        def _one(alpha, ad, sk, pk, salt):
            return hashlib.shake_256(b"|".join((bytes(alpha), bytes(ad), bytes(sk), bytes(pk), bytes(salt)))).digest(784)

        @classmethod
        def prove_batch(cls, alphas, additional_data, secret_keys, producer_keys, ring, ring_root=None, salts=None):
            if any(bytes(a) == b"not-in-ring" for a in alphas):
                raise ValueError("producer key is not in ring")
            if any(bytes(a) == b"device-lost" for a in alphas):
                raise RuntimeError("hipErrorLaunchFailure")
            salts = salts or [b""] * len(alphas)
            blob = b"".join(cls._one(*args) for args in zip(alphas, additional_data, secret_keys, producer_keys, salts))
            return cls._from_encoded(blob, len(alphas))

        @classmethod
        def batch_verify(cls, proofs, inputs, additional_data, ring, ring_root):
            sk, pk = ring
            return all(p.encode() == cls._one(a, ad, sk, pk, b"") for p, a, ad in zip(proofs, inputs, additional_data))

    return FakeRingVRF


def _sharded_case(count, poison=None):
    alphas = [b"in-%d" % i for i in range(count)]
    ads = [b"ad-%d" % (i * 7) for i in range(count)]
    if poison is not None:
        alphas[poison[0]] = poison[1]
    return alphas, ads, [b"\x05" * 32] * count, [b"\x06" * 32] * count


def _check_sharded(comm, count, dst, poison):
    from dot_ring_amd import parallel

    vrf = _fake_scheme()
    alphas, ads, sks, pks = _sharded_case(count, poison)
    ring = (sks[0] if count else b"", pks[0] if count else b"")
    if poison is not None:
        kind = ValueError if poison[1] == b"not-in-ring" else parallel.DotRingShardError
        with pytest.raises(kind) as err:
            parallel.prove_batch_sharded(comm, vrf, alphas, ads, sks, pks, ring, None, dst=dst)
        lo_rank = [r for r in range(comm.world) if parallel.shard_range(count, r, comm.world)[0] <= poison[0] < parallel.shard_range(count, r, comm.world)[1]][0]
        return f"rank {lo_rank}" in str(err.value)                       # every rank names the rank that failed
    proofs = parallel.prove_batch_sharded(comm, vrf, alphas, ads, sks, pks, ring, None, dst=dst)
    want = [vrf._one(a, ad, sk, pk, b"") for a, ad, sk, pk in zip(alphas, ads, sks, pks)]
    ok = True
    if dst is None or comm.rank == dst:
        ok = ok and len(proofs) == count and [p.encode() for p in proofs] == want and vrf.encode_batch(proofs) == b"".join(want)
    else:
        ok = ok and proofs is None
    # verification: from the gathered list everywhere, or from rank 0's copy alone
    ok = ok and parallel.batch_verify_sharded(comm, vrf, proofs, alphas, ads, ring, None) is True
    if count:
        bad_ads = list(ads)
        bad_ads[count - 1] = b"tampered"
        ok = ok and parallel.batch_verify_sharded(comm, vrf, proofs, alphas, bad_ads, ring, None) is False
    return ok


def _sharded_socket_worker(rank, world, port, count, dst, poison, out_q):
    sys.path.insert(0, ROOT)
    from dot_ring_amd import parallel

    comm = parallel.SocketComm(rank, world, "127.0.0.1", port)
    try:
        out_q.put((rank, _check_sharded(comm, count, dst, poison)))
        comm.barrier()
    finally:
        comm.close()


def _sharded_gloo_worker(rank, world, port, count, dst, poison, out_q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from dot_ring_amd import parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out_q.put((rank, _check_sharded(parallel.TorchComm(), count, dst, poison)))
    dist.barrier()
    dist.destroy_process_group()


def _run_sharded(worker, world, count, dst, poison):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, count, dst, poison, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r for r, _ in results] == list(range(world)) and all(ok for _, ok in results), results


@pytest.mark.parametrize("world,count,dst", [(2, 37, None), (3, 2, None), (3, 40, 0), (8, 67, 0), (8, 5, None), (2, 0, None)])
def test_prove_batch_sharded_exchange_socket(world, count, dst):
    """the TCP star: ragged splits (37 = 19 + 18), empty shards (2 proofs over 3 ranks, 5 over 8), gather to every rank and to rank 0
    alone (batch_verify_sharded then starts from rank 0's copy), the world-8 port layout of BASELINE configs[4], an empty batch"""
    _run_sharded(_sharded_socket_worker, world, count, dst, None)


@pytest.mark.parametrize("count,dst", [(37, None), (1, 0)])
def test_prove_batch_sharded_exchange_gloo(count, dst):
    """the same over a torch.distributed group (gloo, world 2): the all_gather-only communicators pad the shards to one size"""
    _run_sharded(_sharded_gloo_worker, 2, count, dst, None)


@pytest.mark.parametrize("world,poison", [(3, (10, b"not-in-ring")), (2, (0, b"device-lost"))])
def test_prove_batch_sharded_failing_shard_raises_on_every_rank(world, poison):
    """a shard that raises (the reference's ValueError for a producer key outside the ring; a device error) is announced in the
    status header before anyone waits for its proofs: every rank raises — the same exception type, naming the failing rank"""
    _run_sharded(_sharded_socket_worker, world, 12, None, poison)
