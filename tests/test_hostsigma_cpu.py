"""CPU: the small-call host route of the Tiny / Thin / Pedersen VRFs (dot_ring_amd/csrc/hostsigma.hpp, hostsmall.hpp) compiled with plain
g++ behind a line-oriented harness (tests/native/hostsigma_check.cpp) and run over the reference's own vectors for the Bandersnatch
suites — every Tiny, Thin and Pedersen KAT of tests/golden/{ark-vrf,dot-ring} byte for byte, proofs and verdicts — plus seeded cases
against the oracle (salts, long inputs), the verdict codes for a bad public key / a malformed proof, the point decoder against the
oracle's dec_point, and the three scalar multiplications (fixed schedule for secrets, Straus, window table) against the oracle's.
tests/test_gpu_api.py runs the same vectors through the library's entry points on both routes (host cores / kernels)."""
import json
import os
import random
import subprocess

import pytest

from oracle import coracle
from oracle.pyref import bandersnatch as bsn
from oracle.pyref import vrf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = {"sha512": bsn.SHA512, "shake128": bsn.SHAKE128}
TINY = [("sha512", "ark-vrf/bandersnatch_sha-512_ell2_tiny.json"), ("sha512", "ark-vrf/bandersnatch_ed_sha512_ell2_ietf.json"),
        ("sha512", "dot-ring/bandersnatch_sha-512_ell2_tiny.json"), ("shake128", "ark-vrf/bandersnatch_shake128_ell2_tiny.json"),
        ("shake128", "dot-ring/bandersnatch_shake128_ell2_tiny.json")]
THIN = [("sha512", "ark-vrf/bandersnatch_sha-512_ell2_thin.json"), ("sha512", "dot-ring/bandersnatch_sha-512_ell2_thin.json"),
        ("shake128", "ark-vrf/bandersnatch_shake128_ell2_thin.json"), ("shake128", "dot-ring/bandersnatch_shake128_ell2_thin.json")]
PEDERSEN = [("sha512", "ark-vrf/bandersnatch_sha-512_ell2_pedersen.json"), ("sha512", "ark-vrf/bandersnatch_ed_sha512_ell2_pedersen.json"),
            ("sha512", "dot-ring/bandersnatch_sha-512_ell2_pedersen.json"), ("shake128", "ark-vrf/bandersnatch_shake128_ell2_pedersen.json"),
            ("shake128", "dot-ring/bandersnatch_shake128_ell2_pedersen.json")]


class Harness:
    def __init__(self, exe):
        self.proc = subprocess.Popen([exe], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)

    def ask(self, *fields) -> list:
        self.proc.stdin.write(" ".join((f.hex() or "-") if isinstance(f, (bytes, bytearray)) else str(f) for f in fields) + "\n")
        self.proc.stdin.flush()
        return self.proc.stdout.readline().split()

    def suite(self, s):
        le = lambda v: int(v).to_bytes(32, "little")
        assert self.ask("suite", s.suite_id, 1 if s.xof else 0, le(bsn.G[0]) + le(bsn.G[1]), le(s.blinding_base[0]) + le(s.blinding_base[1])) == ["ok"]

    def prove(self, scheme, alpha, ad, salt, sk) -> bytes:
        out = self.ask("prove", scheme, alpha, ad, salt, sk)
        assert out[0] == "proof", out
        return bytes.fromhex(out[1])

    def verify(self, scheme, proof, pk, inp, ad, salt=b"") -> int:
        out = self.ask("verify", scheme, proof, pk, inp, ad, salt)
        assert out[0] == "verdict", out
        return int(out[1])

    def close(self):
        self.proc.stdin.close()
        self.proc.wait(timeout=30)


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = tmp_path_factory.mktemp("hostsigma") / "hostsigma_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-Wno-psabi", "-I", os.path.join(ROOT, "dot_ring_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "hostsigma_check.cpp"), "-o", str(exe)], check=True)
    h = Harness(str(exe))
    yield h
    h.close()


def _load(golden_dir, rel):
    with open(os.path.join(golden_dir, rel)) as f:
        return json.load(f)


@pytest.mark.parametrize("suite,rel", TINY)
def test_host_tiny_kats(harness, golden_dir, suite, rel):
    harness.suite(SUITES[suite])
    for v in _load(golden_dir, rel):
        sk, al, ad, pk = (bytes.fromhex(v[k]) for k in ("sk", "alpha", "ad", "pk"))
        proof = harness.prove("tiny", al, ad, b"", sk)
        assert proof.hex() == v["gamma"] + v["proof_c"] + v["proof_s"]
        assert harness.verify("tiny", proof, pk, al, ad) == 1
        assert harness.verify("tiny", proof, pk, al + b"x", ad) == 0
        assert harness.verify("tiny", proof, pk, al, ad + b"x") == 0
        assert harness.verify("tiny", proof[:79] + bytes([proof[79] ^ 1]), pk, al, ad) in (0, 3)


@pytest.mark.parametrize("suite,rel", THIN)
def test_host_thin_kats(harness, golden_dir, suite, rel):
    harness.suite(SUITES[suite])
    for v in _load(golden_dir, rel):
        sk, al, ad, pk = (bytes.fromhex(v[k]) for k in ("sk", "alpha", "ad", "pk"))
        proof = harness.prove("thin", al, ad, b"", sk)
        assert proof.hex() == v["gamma"] + v["proof_r"] + v["proof_s"]
        assert harness.verify("thin", proof, pk, al, ad) == 1
        assert harness.verify("thin", proof, pk, al + b"x", ad) == 0
        other = bytes.fromhex(_load(golden_dir, rel)[0]["pk"])
        if other != pk:
            assert harness.verify("thin", proof, other, al, ad) == 0


@pytest.mark.parametrize("suite,rel", PEDERSEN)
def test_host_pedersen_kats(harness, golden_dir, suite, rel):
    harness.suite(SUITES[suite])
    for v in _load(golden_dir, rel):
        sk, al, ad = (bytes.fromhex(v[k]) for k in ("sk", "alpha", "ad"))
        out = harness.ask("prove", "pedersen", al, ad, b"", sk)
        proof, aux = bytes.fromhex(out[1]), bytes.fromhex(out[3])
        assert proof.hex() == v["gamma"] + v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"]
        assert aux[256:288].hex() == v["blinding"]
        assert harness.verify("pedersen", proof, b"", al, ad) == 1
        assert harness.verify("pedersen", proof, b"", al, ad + b"x") == 0
        assert harness.verify("pedersen", proof[:128] + proof[160:] + proof[128:160], b"", al, ad) in (0, 3)      # s and s_b swapped


def test_host_routes_match_the_oracle_on_seeded_cases(harness):
    """salts, empty and long inputs, secret keys as 32 random bytes (reduced mod n like the reference's dec_scalar_mod)"""
    rng = random.Random(64)
    for name, s in SUITES.items():
        harness.suite(s)
        with bsn.using(s):
            for i in range(12):
                sk = bytes(rng.randrange(256) for _ in range(32))
                al = bytes(rng.randrange(256) for _ in range((0, 1, 31, 200, 1000)[i % 5]))
                ad = bytes(rng.randrange(256) for _ in range((0, 5, 64)[i % 3]))
                salt = b"" if i % 2 else bytes(rng.randrange(256) for _ in range(7))
                pk = bsn.public_key_from_secret(sk)
                tiny = harness.prove("tiny", al, ad, salt, sk)
                assert tiny == vrf.tiny_prove(s, al, sk, ad, salt), (name, i)
                assert harness.verify("tiny", tiny, pk, al, ad, salt) == 1 and vrf.tiny_verify(s, tiny, pk, al, ad, salt)
                assert harness.verify("tiny", tiny, pk, al, ad, salt + b"s") == 0
                ped = harness.prove("pedersen", al, ad, salt, sk)
                assert ped == vrf.pedersen_prove(s, al, sk, ad, salt)[0], (name, i)
                assert harness.verify("pedersen", ped, b"", al, ad, salt) == 1
                thin = harness.prove("thin", al, ad, salt, sk)
                assert thin[:32] == tiny[:32] and harness.verify("thin", thin, pk, al, ad, salt) == 1


def test_host_verifier_verdict_codes(harness):
    """2 = the public key is not a prime-order point (the reference raises ValueError("Invalid public key")), 3 = a proof point that
    does not decode / a scalar >= n; identity, low-order and off-curve encodings; wrong lengths are refused by the caller"""
    s = bsn.SHA512
    harness.suite(s)
    with bsn.using(s):
        sk = (12345).to_bytes(32, "little")
        pk = bsn.public_key_from_secret(sk)
        tiny, thin = harness.prove("tiny", b"in", b"ad", b"", sk), harness.prove("thin", b"in", b"ad", b"", sk)
        identity = (1).to_bytes(32, "little")
        low_order = (bsn.P - 1).to_bytes(32, "little")                                    # (0, -1): order 2
        not_on_curve = next(e for e in ((k).to_bytes(32, "little") for k in range(2, 400)) if _dec_or_none(e) is None)
        for bad_pk in (identity, low_order, not_on_curve, b"\xff" * 32):
            assert harness.verify("tiny", tiny, bad_pk, b"in", b"ad") == 2
            assert harness.verify("thin", thin, bad_pk, b"in", b"ad") == 2
        for bad_pt in (identity, low_order, not_on_curve):
            assert harness.verify("tiny", bad_pt + tiny[32:], pk, b"in", b"ad") == 3
            assert harness.verify("thin", thin[:32] + bad_pt + thin[64:], pk, b"in", b"ad") == 3
        big_s = bsn.N.to_bytes(32, "little")                                              # s = n: not canonical
        assert harness.verify("tiny", tiny[:48] + big_s, pk, b"in", b"ad") == 3
        assert harness.verify("thin", thin[:64] + big_s, pk, b"in", b"ad") == 3
        ped = harness.prove("pedersen", b"in", b"ad", b"", sk)
        for k in range(4):
            assert harness.verify("pedersen", ped[: 32 * k] + identity + ped[32 * k + 32 :], b"", b"in", b"ad") == 3
        assert harness.verify("pedersen", ped[:128] + big_s + ped[160:], b"", b"in", b"ad") == 3
        assert harness.verify("pedersen", ped[:160] + big_s, b"", b"in", b"ad") == 3
        # a point of the full group that is NOT in the prime-order subgroup: P + (0, -1)
        p_plus_t = bsn.enc_point(bsn.te_add(bsn.dec_point(pk), (0, bsn.P - 1))) if hasattr(bsn, "te_add") else None
        if p_plus_t is not None:
            assert harness.verify("tiny", tiny, p_plus_t, b"in", b"ad") == 2


def _dec_or_none(enc):
    try:
        return bsn.dec_point(enc)
    except ValueError:
        return None


def test_host_point_decoder_matches_the_oracle(harness):
    rng = random.Random(32)
    cases = [bsn.enc_point(coracle.te_mul(bsn.G, rng.randrange(1, bsn.N))) for _ in range(30)]
    cases += [bytes(rng.randrange(256) for _ in range(32)) for _ in range(60)]
    cases += [(1).to_bytes(32, "little"), (0).to_bytes(32, "little"), bsn.P.to_bytes(32, "little"), (bsn.P - 1).to_bytes(32, "little"), b"\xff" * 32]
    valid = 0
    for enc in cases:
        want = _dec_or_none(enc)
        out = harness.ask("decode", enc)
        if want is None:
            assert out == ["invalid"], enc.hex()
        else:
            valid += 1
            assert out[0] == "point" and bytes.fromhex(out[1]) == want[0].to_bytes(32, "little") + want[1].to_bytes(32, "little")
    assert valid >= 30


def test_host_scalar_multiplications_match_the_oracle(harness):
    """te_mul_secret (fixed schedule, masked table picks), te_msm_public (Straus) and te_mul_fixed (window table; secret and indexed picks
    must agree) on edge scalars — 0, 1, n - 1, n, 2^256 - 1, single nibbles, all-ones — and random ones"""
    rng = random.Random(16)
    edge = [0, 1, 2, 15, 16, bsn.N - 1, bsn.N, bsn.N + 1, (1 << 256) - 1, 1 << 255, 0xF << 252, int("1" * 64, 16), int("f0" * 32, 16)]
    ks = edge + [rng.randrange(1 << 256) for _ in range(12)]
    pt = coracle.te_mul(bsn.G, 987654321)
    raw = pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little")
    for k in ks:
        want = coracle.te_mul(pt, k % bsn.N)
        want_raw = want[0].to_bytes(32, "little") + want[1].to_bytes(32, "little")
        for how in ("secret", "public", "fixed"):
            out = harness.ask("mul", how, raw, k.to_bytes(32, "little"))
            assert out[0] == "point" and bytes.fromhex(out[1]) == want_raw, (how, hex(k))
