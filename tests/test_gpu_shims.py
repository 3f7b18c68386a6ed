"""GPU: the reference-signature shims (dot_ring_amd/shims) called the way the reference calls its native modules —
projective integer tuples, GLV-sized and negative scalars, twiddle / bit-reversal lists built as polynomial/fft.py:14-55
builds them, the PCS protocol of pcs/protocol.py:10-40 — and compared with the oracle."""
import random
from types import SimpleNamespace

import pytest

from oracle import coracle
from oracle.pyref import bandersnatch as bsn

pytestmark = pytest.mark.gpu

P, N = bsn.P, bsn.N
A, D = bsn.A, bsn.D


def _proj(pt, rng):
    """affine -> extended projective (X, Y, Z, T) with a random Z, as glv.py hands points over"""
    z = rng.randrange(1, P)
    x, y = pt
    return x * z % P, y * z % P, z, x * y % P * z % P


def _norm(res):
    x, y, z, t = res
    zi = pow(z, -1, P)
    assert t * z % P == x * y % P                        # a consistent extended point
    return x * zi % P, y * zi % P


def _points(rng, count):
    return [coracle.te_mul(bsn.G, rng.randrange(1, N)) for _ in range(count)]


def test_scalar_mult_2_4_6_match_oracle(ctx):
    from dot_ring_amd.shims import bandersnatch_te_hip as te

    rng = random.Random(480)
    for terms, fn in ((2, te.scalar_mult_windowed_native_w2_cy), (4, te.scalar_mult_4_native_w2_cy), (6, te.scalar_mult_6_native_w2_cy)):
        for trial in range(6):
            pts = _points(rng, terms)
            # GLV halves (< 2^128), full-width scalars, and the edge values 0 / 1
            ks = [rng.randrange(1 << 127) if trial < 3 else rng.randrange(N) for _ in range(terms)]
            if trial == 2:
                ks[0], ks[-1] = 0, 1
            if trial == 5:
                ks = [0] * terms
            coords = [c for p in pts for c in _proj(p, rng)]
            got = fn(*ks, *coords, A, D, P)
            want = bsn.IDENTITY
            for p, k in zip(pts, ks):
                want = bsn.add(want, coracle.te_mul(p, k))
            assert _norm(got) == want
    # identity operands (Z arbitrary) and the reference's argument check
    ident = (0, 7, 7, 0)
    got = te.scalar_mult_windowed_native_w2_cy(5, 9, *ident, *_proj(bsn.G, rng), A, D, P)
    assert _norm(got) == coracle.te_mul(bsn.G, 9)
    with pytest.raises(ValueError):
        te.scalar_mult_windowed_native_w2_cy(1, 1, *ident, *ident, A, (D + 1) % P, P)
    with pytest.raises(OverflowError):
        te.scalar_mult_windowed_native_w2_cy(1 << 256, 1, *ident, *ident, A, D, P)


@pytest.mark.parametrize("n", [5, 96, 1024, 5122])
def test_msm_pippenger_signed_matches_oracle(ctx, n):
    """bandersnatch_te.pyx:257 with the reference's call shape: affine point objects, scalars centred into (-n/2, n/2]
    (bandersnatch.py:270-284), the window from _pippenger_window_bits (:23-36), both return forms"""
    from dot_ring_amd.shims import bandersnatch_te_hip as te

    rng = random.Random(257 + n)
    base = _points(rng, min(n, 64))
    pts = [base[i % len(base)] for i in range(n)]
    ks = [rng.randrange(N) for _ in range(n)]
    ks[0], ks[1] = 0, N - 1
    centred = [k - N if k > N // 2 else k for k in ks]
    assert any(k < 0 for k in centred)
    window = 2 if n < 8 else 3 if n < 96 else 4 if n < 192 else 5 if n < 384 else 6 if n < 768 else 7 if n < 1024 else 8
    objs = [SimpleNamespace(x=p[0], y=p[1]) for p in pts]
    want = coracle.te_msm(pts, ks)
    assert te.msm_pippenger_signed_native_cy(objs, centred, A, D, P, window_bits=window, affine=True) == want
    assert _norm(te.msm_pippenger_signed_native_cy(objs, centred, A, D, P, window)) == want


def test_msm_pippenger_edge_behaviour(ctx):
    from dot_ring_amd.shims import bandersnatch_te_hip as te

    g = SimpleNamespace(x=bsn.G[0], y=bsn.G[1])
    assert te.msm_pippenger_signed_native_cy([], [], A, D, P) == (0, 1, 1, 0)
    assert te.msm_pippenger_signed_native_cy([], [], A, D, P, affine=True) == (0, 1)
    assert te.msm_pippenger_signed_native_cy([g, g], [0, 0], A, D, P, affine=True) == (0, 1)
    assert te.msm_pippenger_signed_native_cy([g, g], [3, -3], A, D, P, affine=True) == (0, 1)            # P + (-P)
    assert te.msm_pippenger_signed_native_cy([g], [-2], A, D, P, affine=True) == bsn.neg(coracle.te_mul(bsn.G, 2))
    with pytest.raises(ValueError, match="Points and scalars must have same length"):
        te.msm_pippenger_signed_native_cy([g], [1, 2], A, D, P)
    with pytest.raises(ValueError, match="window_bits must be between 2 and 8"):
        te.msm_pippenger_signed_native_cy([g], [1], A, D, P, window_bits=9)


def test_sqrt_and_projective_to_affine(ctx):
    from dot_ring_amd.shims import bandersnatch_te_hip as te

    rng = random.Random(421)
    for _ in range(20):
        v = rng.randrange(P)
        sq = v * v % P
        r = te.sqrt_mod_bls_scalar_cy(sq)
        assert r * r % P == sq and r in (coracle.fr_sqrt(sq), P - coracle.fr_sqrt(sq))
    assert te.sqrt_mod_bls_scalar_cy(0) == 0
    with pytest.raises(ValueError, match="non-square"):
        te.sqrt_mod_bls_scalar_cy(5)                      # 5 is the field's non-residue
    x, y, z, _t = _proj(bsn.G, rng)
    assert te.projective_to_affine_cy(x, y, z, P) == bsn.G
    assert te.projective_to_affine_cy(3, 4, 0, P) == (0, 1)


# ------------------------------------------------------------------ seam C
def _bit_reverse(n):                # polynomial/fft.py:14-27
    bits = n.bit_length() - 1
    return [int(f"{i:0{bits}b}"[::-1], 2) if bits else 0 for i in range(n)]


def _twiddles(n, omega):            # polynomial/fft.py:30-55
    out, m = [], 2
    while m <= n:
        step = pow(omega, n // m, P)
        stage, w = [], 1
        for _ in range(m >> 1):
            stage.append(w)
            w = w * step % P
        out.append(stage)
        m <<= 1
    return out


@pytest.mark.parametrize("log2n", [1, 2, 5, 11, 13])
def test_ntt_plan_matches_oracle(ctx, log2n):
    from dot_ring_amd.shims.ntt_plan import BlsScalarNTTPlan

    n = 1 << log2n
    omega_2048 = 49307615728544765012166121802278658070711169839041683575071795236746050763237
    omega = pow(omega_2048, 2048 // n, P) if n <= 2048 else None
    if omega is None:               # extend the base root by square roots as params.py:108-115 does
        omega = omega_2048
        for _ in range(log2n - 11):
            omega = coracle.fr_sqrt(omega)
    rng = random.Random(log2n)
    vals = [rng.randrange(P) for _ in range(n)]
    plan = BlsScalarNTTPlan(_twiddles(n, omega), _bit_reverse(n))
    a = list(vals)
    plan.transform(a)
    assert a == coracle.ntt(vals, omega)
    scale = pow(n, -1, P)
    b = list(vals)
    plan.transform_scaled(b, scale)
    assert b == coracle.ntt(vals, omega, scale)
    # forward then inverse (inverse plan from omega^-1, scaled by 1/n) is the identity — what fft.py:87-110 does
    inv = BlsScalarNTTPlan(_twiddles(n, pow(omega, -1, P)), _bit_reverse(n))
    inv.transform_scaled(a, scale)
    assert a == vals


def test_ntt_plan_rejects_what_the_reference_rejects(ctx):
    from dot_ring_amd.shims.ntt_plan import BlsScalarNTTPlan

    omega = pow(49307615728544765012166121802278658070711169839041683575071795236746050763237, 256, P)       # n = 8
    tw, rev = _twiddles(8, omega), _bit_reverse(8)
    with pytest.raises(ValueError, match="power of two"):
        BlsScalarNTTPlan([], [0, 1, 2])
    with pytest.raises(ValueError, match="twiddle stages"):
        BlsScalarNTTPlan(tw[:2], rev)
    with pytest.raises(ValueError, match="outside plan size"):
        BlsScalarNTTPlan(tw, rev[:-1] + [8])
    with pytest.raises(ValueError):
        BlsScalarNTTPlan(tw, list(range(8)))             # not the bit-reversal permutation: another transform
    plan = BlsScalarNTTPlan(tw, rev)
    with pytest.raises(ValueError, match="does not match native NTT plan size"):
        plan.transform([1, 2, 3, 4])
    one = [7]
    plan.transform(one)                                  # len <= 1: untouched, as ntt.pyx:106
    assert one == [7]


# ------------------------------------------------------------------ seam B
def test_hip_kzg_is_a_pcs_over_a_reference_style_srs(ctx, srs_bytes):
    """HipKZG bound to an object shaped like the reference's SRS (integer pairs): every protocol member against the oracle"""
    from dot_ring_amd.shims import hip_kzg
    from oracle.pyref import kzg as okzg

    m = 300
    g1_points = [(int.from_bytes(srs_bytes[96 * i : 96 * i + 48], "big"), int.from_bytes(srs_bytes[96 * i + 48 : 96 * i + 96], "big")) for i in range(m)]
    import dot_ring_amd as d

    ref_like = SimpleNamespace(g1_points=g1_points, g2_points=d.KZG._srs().g2_points)
    pcs = hip_kzg.bind_reference_srs(ref_like)
    for member in ("normalize_g1", "compress_g1", "serialize_g1_uncompressed", "decompress_g1", "msm_g1", "commit", "open", "verify",
                   "batch_verify", "batch_verify_linear_preconverted", "ensure_srs_size"):
        assert callable(getattr(pcs, member))
    assert pcs.commitment_size == 48 and pcs.scalar_modulus == coracle.FR_P
    assert pcs.srs.g1_points[0] == g1_points[0] and len(pcs.srs.g2_points) == 2
    rng = random.Random(152)
    coeffs = [rng.randrange(coracle.FR_P) for _ in range(m)]
    coeffs[3] += coracle.FR_P                            # scalars >= r are accepted (ops.py:215-220 produces them)
    com = pcs.commit(coeffs)
    want = coracle.g1_msm(g1_points, [c % coracle.FR_P for c in coeffs])
    assert pcs.normalize_g1(com) == want
    assert pcs.compress_g1(com) == okzg.compress(want)
    assert pcs.serialize_g1_uncompressed(com) == okzg.serialize(want)
    assert pcs.normalize_g1(pcs.decompress_g1(pcs.compress_g1(com))) == want
    assert pcs.commit([0] * 10) is None and pcs.compress_g1(None) == b"\xc0" + bytes(47)
    assert pcs.normalize_g1(pcs.msm_g1([com, com], [2, 3])) == coracle.g1_mul(want, 5)
    x = rng.randrange(coracle.FR_P)
    opening = pcs.open(coeffs, x)
    y = sum(c * pow(x, i, coracle.FR_P) for i, c in enumerate(coeffs)) % coracle.FR_P
    assert opening.y == y
    assert pcs.verify(com, opening.proof, x, y)
    assert not pcs.verify(com, opening.proof, x, (y + 1) % coracle.FR_P)
    second = pcs.open(coeffs, x + 1)
    assert pcs.batch_verify([(com, opening.proof, x, y), (com, second.proof, x + 1, second.y)])
    assert not pcs.batch_verify([(com, opening.proof, x, y), (com, second.proof, x + 1, (second.y + 1) % coracle.FR_P)])
    with pytest.raises(ValueError):                      # beyond every available SRS (kzg.py:155-160 -> srs.py:44-47)
        pcs.commit([1] * 7000)
    with pytest.raises(ValueError):
        pcs.decompress_g1(bytes(47))


def test_ring_proof_through_the_shim_pcs(ctx, golden_dir):
    """RingProofParams(pcs=HipKZG): the reference's ring KAT (784 bytes incl. ten commitments) through the shim class"""
    import json
    import os

    import dot_ring_amd as d
    from dot_ring_amd.shims.hip_kzg import HipKZG

    cv = d.Bandersnatch
    vrf = d.RingVRF[cv]
    v = json.load(open(os.path.join(golden_dir, "ark-vrf", "bandersnatch_sha-512_ell2_ring.json")))[0]
    keys = vrf.parse_keys(bytes.fromhex(v["ring_pks"]))
    params = d.RingProofParams(test_vectors=True, cv=cv, pcs=HipKZG)
    ring = d.Ring(keys, params)
    root = d.RingRoot.from_ring(ring, params)
    assert root.encode().hex() == v["ring_pks_com"]
    sk = bytes.fromhex(v["sk"])
    proof = vrf.prove(bytes.fromhex(v["alpha"]), bytes.fromhex(v["ad"]), sk, cv.public_key_from_secret(sk), ring, root)
    assert proof.encode().hex() == (v["gamma"] + v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"] + v["ring_proof"])
    assert proof.verify(bytes.fromhex(v["alpha"]), bytes.fromhex(v["ad"]), ring, root)
