"""CPU: host-side logic of the API mirror that needs no GPU — parameters, codecs, transcripts, hash-to-field, SRS
parsing, host G1 codecs — checked against the oracle restatement and the reference's documented behaviour."""
import random

import pytest

import dot_ring_amd as d
from dot_ring_amd.ring_proof import transcript as tr
from dot_ring_amd.ring_proof.params import RingProofParams
from dot_ring_amd.ring_proof.pcs import KZG, SRS, synthetic_div_with_eval
from dot_ring_amd.ring_proof.poly import fold, poly_divide_by_vanishing, poly_evaluate_single, poly_mul_small
from dot_ring_amd.vrf import codec, primitives
from oracle.pyref import bandersnatch as obsn
from oracle.pyref import kzg as okzg
from oracle.pyref import ring as oring
from oracle.pyref import vrf as ovrf


def test_params_defaults_capacity_and_errors():
    p = RingProofParams()
    assert (p.domain_size, p.max_ring_size, p.padding_rows, p.radix_domain_size) == (512, 255, 4, 2048)
    assert p.omega == oring.Params().omega and p.radix_omega == oring.Params().radix_omega
    assert p.last_index == 508 and p.required_srs_degree == 1536 and p.radix_shift == 4
    for size, dom in ((1, 512), (255, 512), (256, 1024), (1024, 2048), (1791, 2048), (1792, 4096), (3839, 4096)):
        q = RingProofParams.from_ring_size(size)
        assert q.domain_size == dom and q.max_ring_size == dom - 257
        assert q.omega == oring.Params.from_ring_size(size).omega          # incl. the Tonelli-Shanks root at N = 4096
    for bad in (0, -3, 3840):
        with pytest.raises(ValueError):
            RingProofParams.from_ring_size(bad)
    with pytest.raises(ValueError):
        RingProofParams(domain_size=500)
    with pytest.raises(ValueError):
        RingProofParams(domain_size=8192)
    with pytest.raises(ValueError):
        RingProofParams(padding_rows=3)
    with pytest.raises(ValueError):
        RingProofParams(domain_size=512, max_ring_size=256)
    with pytest.raises(ValueError):
        RingProofParams(domain_size=256)


def test_scalar_and_point_codecs():
    cv = d.Bandersnatch
    n = cv.curve.params.subgroup_order
    assert codec.enc_scalar(cv, n + 5) == (5).to_bytes(32, "little")
    assert codec.dec_scalar(cv, (n - 1).to_bytes(32, "little")) == n - 1
    with pytest.raises(ValueError, match="not canonical"):
        codec.dec_scalar(cv, n.to_bytes(32, "little"))
    with pytest.raises(ValueError):
        codec.dec_scalar(cv, bytes(31))
    assert codec.dec_scalar_mod(cv, b"\xff" * 48) == int.from_bytes(b"\xff" * 48, "little") % n
    with pytest.raises(ValueError):
        codec.enc_64(1 << 64)
    g = cv.point_type.generator_point()
    enc = g.point_to_string()
    assert enc == obsn.enc_point(obsn.G)
    assert cv.point_type.string_to_point(enc) == g
    assert (-g).point_to_string()[31] ^ enc[31] == 0x80
    with pytest.raises(ValueError):
        cv.point_type.string_to_point(b"")
    with pytest.raises(ValueError):
        cv.point_type.string_to_point((obsn.P).to_bytes(32, "little"))
    with pytest.raises(ValueError):
        cv.point_type(1, 2)                                     # not on the curve
    assert (g + cv.point_type.identity()) == g and (g - g).is_identity()
    assert (g + g) == g.double() and ((g + g).x, (g + g).y) == obsn.add(obsn.G, obsn.G)


@pytest.mark.parametrize("cv,osuite", [(d.Bandersnatch, obsn.SHA512), (d.Bandersnatch_SHAKE128, obsn.SHAKE128)])
def test_vrf_transcript_nonce_challenge_match_oracle(cv, osuite):
    rng = random.Random(1)
    t = primitives.new_transcript(cv)
    o = ovrf.Transcript(osuite)
    for _ in range(3):
        blob = rng.randbytes(rng.randrange(40))
        t.absorb(blob)
        o.absorb(blob)
    assert primitives.nonce(cv, 12345, t) == ovrf.nonce(osuite, 12345, o)
    pts = [obsn.mul_py(obsn.G, k) for k in (3, 77)]
    api_pts = [cv.point_type(*p) for p in pts]
    assert primitives.challenge(cv, api_pts, t) == ovrf.challenge(osuite, pts, o)
    assert primitives.point_to_hash(cv, api_pts[0]) == ovrf.point_to_hash(osuite, pts[0])
    a, b = t.squeeze(10), t.squeeze(70)
    assert a + b == o.squeeze(80)
    with pytest.raises(ValueError):
        t.absorb(b"x")
    assert cv.curve.hash_to_field(b"foo", 2) == obsn.hash_to_field(osuite, b"foo", 2)
    with pytest.raises(ValueError):
        primitives.secret_from_seed_scalar(cv, bytes(31))


def test_fiat_shamir_transcript_matches_oracle():
    rng = random.Random(2)
    a = tr.FiatShamirTranscript(obsn.P, b"Bandersnatch-SHA512-ELL2-v1")
    o = oring.FsTranscript(b"Bandersnatch-SHA512-ELL2-v1")
    a.absorb_labeled(b"vk", b"\x01" * 768)
    o.absorb(b"vk", b"\x01" * 768)
    assert a.copy().challenges(b"constraints_aggregation", 7) == o.fork().challenges(b"constraints_aggregation", 7)
    blob = rng.randbytes(64)
    a.absorb_labeled(b"instance", blob)
    o.absorb(b"instance", blob)
    assert a.challenge(b"evaluation_point") == o.challenges(b"evaluation_point", 1)[0]
    assert a.challenges(b"kzg_aggregation", 8) == o.challenges(b"kzg_aggregation", 8)
    assert a.challenges(b"x", 0) == []


def test_poly_helpers():
    p = obsn.P
    rng = random.Random(3)
    f = [rng.randrange(p) for _ in range(9)]
    x = rng.randrange(p)
    q, y = synthetic_div_with_eval(f, x)
    assert y == poly_evaluate_single(f, x, p) == oring.horner(f, x)
    assert poly_mul_small(q, [(-x) % p, 1], p) == [(c - (y if i == 0 else 0)) % p for i, c in enumerate(f)]
    n = 4
    quot = [rng.randrange(p) for _ in range(7)]
    prod = poly_mul_small(quot, [p - 1] + [0] * (n - 1) + [1], p)          # quot * (X^4 - 1)
    assert poly_divide_by_vanishing(prod, n, p) == quot
    assert poly_divide_by_vanishing([1, 2], 4, p) == [0]
    with pytest.raises(ValueError):
        poly_divide_by_vanishing([1], 0, p)
    assert fold([1, 2, 3, 4, 5], 2, p) == [9, 6]


def test_srs_file_and_kzg_host_codecs():
    srs = SRS.default()
    o = okzg.default_srs()
    assert srs.count == 6145 and srs.g1_points[0] == okzg.G1_GEN and srs.g1_points[6144] == o.g1[6144]
    assert len(srs.g1_points) == 6145 and srs.g1_points[-1] == o.g1[-1] and srs.g1_points[1:3] == o.g1[1:3]
    assert srs.g2_raw == o.g2_raw
    (x1, x0), (y1, y0) = srs.g2_points[0]
    assert x0.to_bytes(48, "big") + x1.to_bytes(48, "big") + y0.to_bytes(48, "big") + y1.to_bytes(48, "big") == srs.g2_raw[0]
    with pytest.raises(ValueError, match="no BLS12-381 SRS file"):
        SRS.from_loaded(12288)                                 # domain 4096 needs 12289 points; only 6145 ship
    pt = o.g1[5]
    assert KZG.compress_g1(pt) == okzg.compress(pt) == KZG.compress_g1(okzg.serialize(pt))
    assert KZG.decompress_g1(okzg.compress(pt)) == okzg.serialize(pt)
    assert KZG.serialize_g1_uncompressed(None) == b"\x40" + bytes(95) and KZG.compress_g1(None) == b"\xc0" + bytes(47)
    assert KZG.normalize_g1(okzg.serialize(pt)) == pt
    with pytest.raises(ValueError):
        KZG.decompress_g1(bytes(47))
    with pytest.raises(ValueError):
        KZG.decompress_g1(bytes(48))


def test_scheme_specialisation_and_lengths():
    t = d.TinyVRF[d.Bandersnatch]
    assert t.cv is d.Bandersnatch and t.__name__ == "TinyVRF[Bandersnatch]" and d.TinyVRF["x"] is d.TinyVRF
    assert d.PedersenVRF[d.Bandersnatch].proof_len() == 192
    assert d.RingVRF[d.Bandersnatch].proof_len() == 784
    assert d.RingRoot.encoded_len() == 144
    with pytest.raises(ValueError):
        d.RingVRF[d.Bandersnatch].parse_keys(bytes(33))
    assert d.RingVRF[d.Bandersnatch].parse_keys(bytes(64)) == [bytes(32), bytes(32)]
    with pytest.raises(NotImplementedError):
        d.TinyVRF.batch_verify()


# /root/reference/tests/test_coverage/test_params.py:7-88 — same cases, same message fragments
def test_params_reference_cases():
    params = RingProofParams()
    assert pow(params.omega, params.domain_size, params.prime) == 1 and pow(params.omega, params.domain_size // 2, params.prime) != 1
    assert len(params.domain) == params.domain_size and len(params.radix_domain) == params.radix_domain_size
    assert params.max_effective_ring_size == params.domain_size - params.scalar_bits - params.padding_rows
    ext = RingProofParams(domain_size=512, radix_domain_size=4096, base_root_size=2048, padding_rows=4, max_ring_size=1)
    assert (ext.base_root_size, ext.radix_domain_size) == (4096, 4096)
    jub = RingProofParams(cv=d.JubJub)                       # 252-bit order: one more key fits the default domain
    assert (jub.scalar_bits, jub.max_ring_size, jub.row_overhead) == (252, 256, 256)
    assert RingProofParams.from_ring_size(256, cv=d.JubJub).domain_size == 512
    assert RingProofParams.from_ring_size(257, cv=d.JubJub).domain_size == 1024


@pytest.mark.parametrize("kwargs,match", [
    ({"domain_size": 3}, "domain_size must be a power of two"),
    ({"domain_size": 4, "radix_domain_size": 6}, "radix_domain_size must be a power of two"),
    ({"domain_size": 8, "radix_domain_size": 4}, "must divide radix_domain_size"),
    ({"domain_size": 4, "radix_domain_size": 8, "base_root_size": 12, "padding_rows": 1, "max_ring_size": 1}, "must divide base_root_size"),
    ({"domain_size": 512, "padding_rows": 0, "max_ring_size": 1}, "padding_rows must be >= 1"),
    ({"domain_size": 512, "padding_rows": 512, "max_ring_size": 1}, "padding_rows must be less than domain_size"),
    ({"domain_size": 512, "padding_rows": 4, "max_ring_size": 256}, "exceeds supported size"),
    ({"domain_size": 256, "padding_rows": 4, "max_ring_size": 1}, "domain_size is too small"),
])
def test_params_validation_errors_reference_table(kwargs, match):
    with pytest.raises(ValueError, match=match):
        RingProofParams(**kwargs)


@pytest.mark.parametrize("ring_size,domain_size,max_ring_size", [
    (1, 512, 255), (254, 512, 255), (255, 512, 255), (256, 1024, 767), (767, 1024, 767), (768, 2048, 1791), (1791, 2048, 1791),
    (1792, 4096, 3839), (2047, 4096, 3839)])
def test_from_ring_size_matches_spec_capacity(ring_size, domain_size, max_ring_size):
    params = RingProofParams.from_ring_size(ring_size)
    assert (params.domain_size, params.max_ring_size, params.max_effective_ring_size) == (domain_size, max_ring_size, max_ring_size)
    assert pow(params.omega, params.domain_size, params.prime) == 1 and pow(params.omega, params.domain_size // 2, params.prime) != 1
    assert params.required_srs_degree == max(params.domain_size - 1, params.radix_domain_size - params.domain_size)


def test_sqrt_root_choice_matches_the_reference_schedule():
    """RingProofParams extends the shipped 2048-th root of unity by repeated square roots for 4N > 2048 (params.py:108-115), and WHICH
    root each step returns fixes omega for N = 4096.  The closed form in dot_ring_amd/ring_proof/params.py must return the root of the
    reference's Tonelli-Shanks loop (restated in oracle/pyref/ring.py): the three extension steps, random squares, the error for
    non-residues, and other primes (p = 3 mod 4, other 2-adicities) as plain square roots."""
    import inspect
    import random

    from dot_ring_amd.ring_proof.params import ROOT_OF_UNITY_2048, _sqrt_mod_prime
    from oracle.pyref import ring as oring

    P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    ref = oring._sqrt_mod_prime
    ref_call = (lambda v: ref(v, P)) if len(inspect.signature(ref).parameters) == 2 else ref
    root = ROOT_OF_UNITY_2048
    for _ in range(3):
        nxt = _sqrt_mod_prime(root, P)
        assert nxt == ref_call(root) and nxt * nxt % P == root
        root = nxt
    rng = random.Random(11)
    for _ in range(200):
        sq = pow(rng.randrange(1, P), 2, P)
        assert _sqrt_mod_prime(sq, P) == ref_call(sq)
    with pytest.raises(ValueError, match="No square root"):
        _sqrt_mod_prime(5, P)                                   # 5 is the non-residue of this field
    assert _sqrt_mod_prime(0, P) == 0
    for pr in (13, 17, 97, 193, 257, 7681, 12289, 65537, 1000003):
        for v in range(1, min(pr, 300)):
            assert pow(_sqrt_mod_prime(v * v % pr, pr), 2, pr) == v * v % pr
