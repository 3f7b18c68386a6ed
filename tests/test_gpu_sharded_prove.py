"""GPU: ONE batch of Ring-VRF proofs sharded over several ranks (BASELINE configs[4] as a library call; SURVEY 8(e), first mode).

parallel.prove_batch_sharded gives rank g proofs [g B / G, (g + 1) B / G), gathers the 784-byte proofs and hands the caller all B of
them in order; in deterministic mode (test_vectors=True) they must equal, byte for byte, what ONE process proves for the whole
batch with RingVRF.prove_batch — and the CPU oracle's proofs.  parallel.batch_verify_sharded verifies the slices and ANDs the
verdicts.  The one-GPU test box has every rank on the same card, so the exchange runs over the TCP communicator (RCCL refuses two
ranks on one device); the RCCL communicator runs the same calls with world size 1."""
import hashlib
import multiprocessing as mp
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RING_KEYS = 24


def _case(count):
    """ring of 24 members (domain 512), proof i signed by member i mod 24, inputs of different lengths"""
    import dot_ring_amd as d

    cv = d.Bandersnatch
    sks = [hashlib.sha256(b"shard-member-%d" % i).digest() for i in range(RING_KEYS)]
    sks = [(int.from_bytes(s, "little") % cv.curve.params.subgroup_order).to_bytes(32, "little") for s in sks]
    keys = [cv.public_key_from_secret(sk) for sk in sks]
    alphas = [b"sharded-input-" + bytes([i % 251]) * (i % 5) + i.to_bytes(4, "little") for i in range(count)]
    ads = [b"" if i % 3 == 0 else b"ad-%d" % i for i in range(count)]
    return d, cv, keys, alphas, ads, [sks[i % RING_KEYS] for i in range(count)], [keys[i % RING_KEYS] for i in range(count)]


def _ring(d, keys, test_vectors):
    params = d.RingProofParams.from_ring_size(len(keys), test_vectors=test_vectors)
    ring = d.Ring(keys, params)
    return ring, d.RingRoot.from_ring(ring, params)


def _worker(rank, world, port, count, dst, out_q):
    sys.path.insert(0, ROOT)
    os.environ["DOTRING_DEVICE"] = "0"                                    # every rank on the box's one GPU
    os.environ.setdefault("DOTRING_HOST_THREADS", "4")
    from dot_ring_amd import parallel

    comm = parallel.SocketComm(rank, world, "127.0.0.1", port)
    try:
        d, cv, keys, alphas, ads, sks, pks = _case(count)
        vrf = d.RingVRF[cv]
        ring, root = _ring(d, keys, True)
        proofs = parallel.prove_batch_sharded(comm, vrf, alphas, ads, sks, pks, ring, root, dst=dst)
        blob = None if proofs is None else vrf.encode_batch(proofs)
        ok = parallel.batch_verify_sharded(comm, vrf, proofs, alphas, ads, ring, root)
        bad = None
        if count:
            wrong = list(alphas)
            wrong[count // 2] = b"another input"
            bad = parallel.batch_verify_sharded(comm, vrf, proofs, wrong, ads, ring, root)
        # production mode (random hidden rows): nothing to compare bytes with — the gathered proofs must verify, on one rank alone too
        zring, zroot = _ring(d, keys, False)
        zproofs = parallel.prove_batch_sharded(comm, vrf, alphas, ads, sks, pks, zring, zroot)
        zk_ok = len(zproofs) == count and (count == 0 or vrf.batch_verify(zproofs, alphas, ads, zring, zroot))
        # a producer key that is not in the ring, in the LAST shard: every rank raises the reference's ValueError
        raised = None
        if count >= world:
            outsider = cv.public_key_from_secret(b"\x07" * 32)
            try:
                parallel.prove_batch_sharded(comm, vrf, alphas, ads, sks[:-1] + [b"\x07" * 32], pks[:-1] + [outsider], ring, root)
                raised = False
            except ValueError as exc:
                raised = f"rank {world - 1}" in str(exc)
        out_q.put((rank, blob, ok, bad, zk_ok, raised))
        comm.barrier()
    finally:
        comm.close()


@pytest.mark.parametrize("world,count,dst", [(2, 37, None), (3, 2, None), (3, 40, 0)])
def test_prove_batch_sharded_equals_one_process(world, count, dst):
    """2 and 3 ranks on the one GPU: a ragged split (37 = 19 + 18), an empty shard (2 proofs over 3 ranks), gather to rank 0 only
    (the verifier then starts from rank 0's copy).  The gathered deterministic proofs = the single-process prove_batch of the
    whole batch = the oracle's proofs for the first three."""
    from oracle.pyref import ring as oring

    d, cv, keys, alphas, ads, sks, pks = _case(count)
    vrf = d.RingVRF[cv]
    ring, root = _ring(d, keys, True)
    want = vrf.encode_batch(vrf.prove_batch(alphas, ads, sks, pks, ring, root))
    o_ring = oring.Ring(keys, oring.Params.from_ring_size(len(keys), test_vectors=True))
    o_root = oring.RingRoot(o_ring)
    for i in range(min(3, count)):
        assert want[784 * i : 784 * i + 784] == oring.ring_vrf_prove(o_ring, o_root, alphas[i], ads[i], sks[i])

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpx = mp.get_context("spawn")
    q = mpx.Queue()
    procs = [mpx.Process(target=_worker, args=(r, world, port, count, dst, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, blob, ok, bad, zk_ok, raised in results:
        if dst is None or rank == dst:
            assert blob == want, f"rank {rank}: gathered proofs differ from the single-process batch"
        else:
            assert blob is None
        assert ok is True and bad in (False, None) and zk_ok
        assert raised in (True, None), f"rank {rank}: the failing shard was not reported as the reference's ValueError"
    assert any(raised for *_, raised in results) or count < world


def test_prove_batch_sharded_over_the_rccl_communicator_world_1(ctx):
    """the same calls over RcclComm (ncclAllGather through dr_comm_all_gather) — one rank here; two ranks need two GPUs
    (tests/test_gpu_sharded_msm.py::test_rccl_two_ranks_on_two_gpus is the place that would run them)"""
    from dot_ring_amd import parallel

    d, cv, keys, alphas, ads, sks, pks = _case(5)
    vrf = d.RingVRF[cv]
    ring, root = _ring(d, keys, True)
    comm = parallel.RcclComm(ctx, 0, 1)
    try:
        proofs = parallel.prove_batch_sharded(comm, vrf, alphas, ads, sks, pks, ring, root)
        assert vrf.encode_batch(proofs) == vrf.encode_batch(vrf.prove_batch(alphas, ads, sks, pks, ring, root))
        assert parallel.batch_verify_sharded(comm, vrf, proofs, alphas, ads, ring, root)
        assert proofs[3].pedersen_proof.verify(alphas[3], ads[3])          # a gathered proof decodes on demand
        assert not parallel.batch_verify_sharded(comm, vrf, proofs, alphas[::-1], ads, ring, root)
    finally:
        comm.close()
