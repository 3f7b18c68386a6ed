"""CPU: the library's host-side hashing (SHA-512, SHAKE128/256, hash_to_field) against hashlib and the oracle.
These entry points need no GPU."""
import hashlib
import random

from dot_ring_amd import _native
from oracle.pyref import bandersnatch as obsn


def test_host_hashes_match_hashlib():
    rng = random.Random(1)
    for ln in [0, 1, 55, 111, 112, 113, 127, 128, 129, 135, 136, 137, 167, 168, 169, 200, 335, 336, 337, 1000, 5000]:
        data = bytes(rng.randrange(256) for _ in range(ln))
        assert _native.host_hash(0, data, 64) == hashlib.sha512(data).digest()
        for ol in (1, 16, 48, 64, 135, 136, 137, 167, 168, 169, 400):
            assert _native.host_hash(1, data, ol) == hashlib.shake_128(data).digest(ol)
            assert _native.host_hash(2, data, ol) == hashlib.shake_256(data).digest(ol)


def test_four_way_shake128_matches_hashlib():
    """the four-in-lockstep SHAKE128 sponge of the batch transcripts (hosthash.hpp: Shake4, Keccak-f on 4 x 64-bit vectors) against
    hashlib: message lengths around the 168-byte rate, four different messages per call, every digest length up to a block"""
    rng = random.Random(4)
    for ln in [0, 1, 7, 8, 9, 55, 166, 167, 168, 169, 170, 335, 336, 337, 500, 1000, 1700]:
        msgs = [bytes(rng.randrange(256) for _ in range(ln)) for _ in range(4)]
        for ol in (1, 16, 48, 64, 167, 168):
            got = _native.host_hash(3, b"".join(msgs), 4 * ol)
            assert [got[ol * k : ol * (k + 1)] for k in range(4)] == [hashlib.shake_128(m).digest(ol) for m in msgs], (ln, ol)


def test_four_way_shake128_without_avx2():
    """the same sponge through the code path for hosts without AVX2 (the vector type split by the compiler)"""
    import os
    import subprocess
    import sys

    code = ("import hashlib; from dot_ring_amd import _native; m=[bytes([k])*300 for k in range(4)]; "
            "g=_native.host_hash(3, b''.join(m), 4*64); assert [g[64*k:64*k+64] for k in range(4)] == [hashlib.shake_128(x).digest(64) for x in m]")
    env = dict(os.environ, DOTRING_KECCAK_GENERIC="1")
    subprocess.run([sys.executable, "-c", code], check=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_hash_to_field_batch_matches_oracle():
    le = lambda v: int(v).to_bytes(32, "little")
    msgs = [b"", b"abc", b"\x00" * 300] + [hashlib.sha256(bytes([i])).digest() * (i % 5) for i in range(40)]
    for suite in (obsn.SHA512, obsn.SHAKE128):
        s = _native.vrf_suite(suite.suite_id, suite.xof, le(obsn.G[0]) + le(obsn.G[1]), le(suite.blinding_base[0]) + le(suite.blinding_base[1]))
        got = _native.hash_to_field_batch(s, msgs)
        for i, m in enumerate(msgs):
            u0, u1 = obsn.hash_to_field(suite, m, 2)
            assert got[64 * i : 64 * i + 64] == le(u0) + le(u1)
    assert _native.hash_to_field_batch(s, []) == b""


def test_pairing_fast_final_exponentiation_matches_reference(srs_bytes):
    """The pairing check's fast final exponentiation (Frobenius maps + x-chain) equals the cube of the plain
    square-and-multiply one on real Miller-loop products, and the check accepts/rejects KZG-shaped equations."""
    import ctypes

    from dot_ring_amd.ring_proof.pcs import SRS

    srs = SRS.default()
    g2 = srs.g2_raw
    g, t1, t2 = (srs_bytes[96 * i : 96 * i + 96] for i in range(3))           # G, tau*G, tau^2*G
    assert _native.pairing_check([(t1, g2[0]), (_native.g1_neg(g), g2[1])])    # e(tau G, H) = e(G, tau H)
    assert _native.pairing_check([(t2, g2[0]), (_native.g1_neg(t1), g2[1])])
    assert not _native.pairing_check([(t2, g2[0]), (_native.g1_neg(g), g2[1])])
    assert _native.pairing_check([(None, g2[0]), (None, g2[1])])
    lib = _native.lib()
    for pairs in ([(t1, g2[0]), (_native.g1_neg(g), g2[1])], [(t2, g2[0]), (_native.g1_neg(g), g2[1])], [(t2, g2[1])], [(g, g2[0]), (t1, g2[1]), (t2, g2[0])]):
        ok = ctypes.c_int(0)
        assert lib.dr_pairing_selfcheck(b"".join(a for a, _ in pairs), b"".join(b for _, b in pairs), len(pairs), ctypes.byref(ok)) == 0
        assert ok.value == 1


def test_worker_pool_serves_concurrent_callers():
    """The library's persistent worker pool (hostproto.hpp: WorkerPool) with six threads inside parallel loops at once —
    the situation of the prover's main thread and its helper threads: every call returns the right bytes, none hangs."""
    import threading

    import dot_ring_amd as d

    suite = d.Bandersnatch.point_type._suite_struct()
    rng = random.Random(5)
    msgs = [bytes(rng.randrange(256) for _ in range(rng.randrange(0, 100))) for _ in range(3000)]
    want = b"".join(u.to_bytes(32, "little") for m in msgs[:64] for u in obsn.hash_to_field(obsn.SHA512, m, 2))
    wrong = []

    def work():
        for _ in range(25):
            if _native.hash_to_field_batch(suite, msgs)[: 64 * 64] != want:
                wrong.append(1)

    threads = [threading.Thread(target=work) for _ in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads) and not wrong
