"""GPU: the base-sharded G1 MSM end to end (BASELINE configs[4]'s MSM leg; SURVEY 8(e), second mode).

Every rank builds ITS shard of the synthetic bases (first + i) * G on its GPU, runs the MSM kernels over it, exchanges one
point per rank and folds — and the result must equal the closed form [sum k_i (first + i)] * G, one oracle scalar
multiplication.  On the one-GPU test box the ranks share the card, so the multi-rank cases use the TCP communicator (RCCL
refuses two ranks on one device); the RCCL path itself (dlopen, ncclCommInitRank, ncclAllGather through dr_comm_*) runs
with world size 1."""
import multiprocessing as mp
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _closed_form(vals):
    import bench
    from oracle import coracle

    expect = sum(k * (1 + i) for i, k in enumerate(vals)) % coracle.FR_P
    if expect == 0:
        return None
    want = bytes(coracle.g1_msm_raw(bench.be_to_le_points(bench.G1_BE), expect.to_bytes(32, "little"), 1))
    return want[:48][::-1] + want[48:][::-1]


def _shard_msm(ctx, comm, n, rank, world, zero_rank, table):
    import bench
    from dot_ring_amd import parallel

    vals, raw = bench.seeded_scalars(n, b"shard")
    lo, hi = parallel.shard_range(n, rank, world)
    if zero_rank is not None:
        zlo, zhi = parallel.shard_range(n, zero_rank, world)
        vals = [0 if zlo <= i < zhi else v for i, v in enumerate(vals)]
        raw = b"".join(v.to_bytes(32, "little") for v in vals)
    cnt = hi - lo
    srs = d_scalars = None
    if cnt:
        srs = ctx.srs_synthetic(bench.G1_BE, cnt, first=lo + 1)           # base i of the whole MSM is (1 + i) * G
        if table:
            srs.precompute(table)
        d_scalars = ctx.alloc(32 * cnt).upload(raw[32 * lo : 32 * hi])
    got = parallel.g1_msm_sharded(ctx, comm, srs, d_scalars, cnt)
    if cnt:
        d_scalars.free()
        srs.close()
    return got, _closed_form(vals)


def test_rccl_communicator_world_1(ctx):
    """the native path on hardware: librccl through dlopen, ncclCommInitRank, ncclAllGather, the fused dr_g1_msm_sharded_dev"""
    from dot_ring_amd import parallel

    comm = parallel.RcclComm(ctx, 0, 1)
    assert comm.rccl_ranks() == 1                                        # ncclCommCount, not the caller's idea of the world
    assert comm.all_gather(b"\x01\x02\x03") == [b"\x01\x02\x03"]
    comm.barrier()
    got, want = _shard_msm(ctx, comm, 5000, 0, 1, None, 12)
    assert got == want
    got, want = _shard_msm(ctx, comm, 300, 0, 1, 0, 0)                   # all-zero scalars: infinity
    assert got is None and want is None
    comm.close()


def _worker(rank, world, port, n, zero_rank, table, out_q):
    sys.path.insert(0, ROOT)
    from dot_ring_amd import _native, parallel

    comm = parallel.SocketComm(rank, world, "127.0.0.1", port)
    ctx = _native.Context(0)                                             # every rank on the box's one GPU
    try:
        got, want = _shard_msm(ctx, comm, n, rank, world, zero_rank, table)
        out_q.put((rank, got == want, got))
    finally:
        comm.barrier()
        ctx.close()
        comm.close()


@pytest.mark.parametrize("world,n,zero_rank,table", [(2, (1 << 16) + 7, None, 12), (2, 4099, 1, 0), (3, 2, None, 0)])
def test_sharded_msm_real_gpu_shards(world, n, zero_rank, table):
    """ragged shards (2^16 + 7 over 2 ranks), a shard whose scalars are all zero (that rank contributes infinity), an empty
    shard (2 pairs over 3 ranks); every rank must hold the closed-form result"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpx = mp.get_context("spawn")
    q = mpx.Queue()
    procs = [mpx.Process(target=_worker, args=(r, world, port, n, zero_rank, table, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results)
    assert len({got for _, _, got in results}) == 1                      # identical on every rank


def _rccl_worker(rank, world, port, n, out_q):
    sys.path.insert(0, ROOT)
    from dot_ring_amd import _native, parallel

    boot = parallel.SocketComm(rank, world, "127.0.0.1", port)
    ctx = _native.Context(rank)                                          # one GPU per rank
    comm = parallel.RcclComm(ctx, rank, world, bootstrap=boot)
    try:
        got, want = _shard_msm(ctx, comm, n, rank, world, None, 12)
        out_q.put((rank, got == want, got, comm.rccl_ranks()))
    finally:
        comm.barrier()
        comm.close()
        ctx.close()
        boot.close()


def test_rccl_two_ranks_on_two_gpus():
    """ncclCommInitRank / ncclAllGather across two processes on two devices, the fused dr_g1_msm_sharded_dev end to end.
    Needs two GPUs: skipped on the one-GPU test boxes (there the multi-rank cases above use the TCP communicator)."""
    from dot_ring_amd import _native

    if _native.lib().dr_device_count() < 2:
        pytest.skip("needs two GPUs")
    world, n = 2, (1 << 16) + 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpx = mp.get_context("spawn")
    q = mpx.Queue()
    procs = [mpx.Process(target=_rccl_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in results)
    assert len({got for _, _, got, _ in results}) == 1
    assert all(cnt == 2 for _, _, _, cnt in results)
