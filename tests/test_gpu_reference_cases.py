"""GPU: more of the reference's own test cases, run through the API mirror — the assertions of
/root/reference/tests/test_verify_ring_sig.py (ring proofs produced by the Rust implementation, domain 512..2048, verified
under their own verifier key), tests/test_ring_vrf/test_audit_regressions.py and the negative / batch-API cases of
tests/test_ark_vrf.py:133-256.  Fixtures: tests/golden/others/*.json, tests/golden/ark-vrf/*.json (data files of the
reference's tests)."""
import copy
import json
import os
from types import SimpleNamespace

import pytest

pytestmark = pytest.mark.gpu

Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB


def _load(golden_dir, rel):
    with open(os.path.join(golden_dir, rel)) as f:
        return json.load(f)


def _fq2_sqrt(a0, a1):
    """A square root of a0 + a1*u in Fq[u]/(u^2+1) (q = 3 mod 4), or None."""
    if a1 == 0:
        r = pow(a0, (Q + 1) // 4, Q)
        if r * r % Q == a0:
            return r, 0
        r = pow(-a0 % Q, (Q + 1) // 4, Q)          # sqrt(-|a0|) = sqrt(|a0|) * u
        return (0, r) if r * r % Q == -a0 % Q else None
    norm_root = pow((a0 * a0 + a1 * a1) % Q, (Q + 1) // 4, Q)
    if norm_root * norm_root % Q != (a0 * a0 + a1 * a1) % Q:
        return None
    half = pow(2, -1, Q)
    for delta in ((a0 + norm_root) * half % Q, (a0 - norm_root) * half % Q):
        x0 = pow(delta, (Q + 1) // 4, Q)
        if x0 and x0 * x0 % Q == delta:
            x1 = a1 * half % Q * pow(x0, -1, Q) % Q
            if (x0 * x0 - x1 * x1) % Q == a0 and 2 * x0 * x1 % Q == a1:
                return x0, x1
    return None


def _g2_uncompress(data: bytes) -> bytes:
    """zcash-compressed G2 (96 bytes: x.c1 || x.c0, flags in the top three bits) -> x.c1 || x.c0 || y.c1 || y.c0, the byte
    order of the SRS file and of the verifier-key transcript item."""
    assert len(data) == 96 and data[0] & 0x80 and not data[0] & 0x40
    largest = bool(data[0] & 0x20)
    x1 = int.from_bytes(bytes([data[0] & 0x1F]) + data[1:48], "big")
    x0 = int.from_bytes(data[48:], "big")
    # y^2 = x^3 + 4(1 + u)
    s0, s1 = (x0 * x0 - x1 * x1) % Q, 2 * x0 * x1 % Q
    c0, c1 = (s0 * x0 - s1 * x1 + 4) % Q, (s0 * x1 + s1 * x0 + 4) % Q
    y = _fq2_sqrt(c0, c1)
    assert y is not None
    y0, y1 = y
    is_largest = y1 > (Q - 1) // 2 if y1 else y0 > (Q - 1) // 2
    if is_largest != largest:
        y0, y1 = -y0 % Q, -y1 % Q
    return b"".join(v.to_bytes(48, "big") for v in (x1, x0, y1, y0))


OTHERS = ["ring_proof_ring64_domain512.json", "ring_proof_ring128_domain512.json", "ring_proof_ring256_domain1024.json",
          "ring_proof_ring1024_domain2048.json", "ring_proof_rust_generated.json"]


@pytest.mark.parametrize("name", OTHERS)
def test_verify_ring_sig_vectors(ctx, golden_dir, name):
    """tests/test_verify_ring_sig.py:92-180: a ring proof from the Rust implementation, its verifier key (G1[0], the two
    G2 points of ITS trusted setup, the three fixed-column commitments) and the statement; the transcript label is
    b"w3f-ring-proof-test".  The scalar pass is the mirror's, the two folds run on the GPU, the pairing in the library;
    tampering with an evaluation or with the statement must be rejected."""
    import dot_ring_amd as d
    from dot_ring_amd.ring_proof.pcs import KZG, SRS
    from dot_ring_amd.ring_proof.transcript import FiatShamirTranscript
    from dot_ring_amd.ring_proof.verifier import linear_pcs_verifications

    data = _load(golden_dir, "others/" + name)
    par = data["metadata"]["parameters"]
    pr = data["proof"]
    vk = bytes.fromhex(data["verifier_key"]["verification_key"])
    assert len(vk) == 384
    g1_0 = KZG.decompress_g1(vk[:48])
    g2 = [_g2_uncompress(vk[48:144]), _g2_uncompress(vk[144:240])]
    pcs = KZG.with_srs(SRS(g1_0, g2))
    params = d.RingProofParams(domain_size=par["domain_size"], max_ring_size=1, pcs=pcs)
    fixed = [pcs.decompress_g1(vk[240 + 48 * i : 288 + 48 * i]) for i in range(3)]
    le = lambda h: int.from_bytes(bytes.fromhex(h), "little")
    cv = d.Bandersnatch
    seed = cv.point(le(par["seed"]["x"]), le(par["seed"]["y"]))
    result = cv.point(le(par["result"]["x"]), le(par["result"]["y"]))
    cols = bytes.fromhex(pr["column_commitments"])
    evs = bytes.fromhex(pr["columns_at_zeta"])
    col = lambda raw: SimpleNamespace(commitment=pcs.decompress_g1(raw))
    names = ("px_zeta", "py_zeta", "s_zeta", "b_zeta", "accip_zeta", "accx_zeta", "accy_zeta")
    proof = SimpleNamespace(
        c_b=col(cols[0:48]), c_accip=col(cols[48:96]), c_accx=col(cols[96:144]), c_accy=col(cols[144:192]),
        c_q=col(bytes.fromhex(pr["quotient_commitment"])), l_zeta_omega=le(pr["lin_at_zeta_omega"]),
        open_agg_zeta=pcs.decompress_g1(bytes.fromhex(pr["agg_at_zeta_proof"])),
        open_l_zeta_omega=pcs.decompress_g1(bytes.fromhex(pr["lin_at_zeta_omega_proof"])),
        **{nm: int.from_bytes(evs[32 * i : 32 * i + 32], "little") for i, nm in enumerate(names)})

    def prefix():
        t = FiatShamirTranscript(params.prime, b"w3f-ring-proof-test")
        t.absorb_labeled(b"vk", pcs.serialize_g1_uncompressed(g1_0) + g2[0] + g2[1] + b"".join(pcs.serialize_g1_uncompressed(c) for c in fixed))
        return t

    def valid(p, relation):
        claims = linear_pcs_verifications(p, fixed, relation, relation + seed, seed, params, prefix())
        return pcs.batch_verify_linear_preconverted(list(claims))

    assert valid(proof, result), name
    bad = copy.copy(proof)
    bad.accx_zeta = (bad.accx_zeta + 1) % params.prime
    assert not valid(bad, result)
    bad = copy.copy(proof)
    bad.l_zeta_omega = (bad.l_zeta_omega + 1) % params.prime
    assert not valid(bad, result)
    assert not valid(proof, result + seed)


# ------------------------------------------------------------------ tests/test_ring_vrf/test_audit_regressions.py
def _keys(count):
    import dot_ring_amd as d

    return [d.Bandersnatch.public_key_from_secret((i + 1).to_bytes(32, "little")) for i in range(count)]


@pytest.fixture(scope="module")
def audit():
    import dot_ring_amd as d

    sk = bytes.fromhex("01" * 32)
    pk = d.Bandersnatch.public_key_from_secret(sk)
    params = d.RingProofParams(test_vectors=True)
    ring = d.Ring([pk, *_keys(7)[1:]], params)
    root = d.RingRoot.from_ring(ring, params)
    proof = d.RingVRF[d.Bandersnatch].prove(b"audit-input", b"audit-ad", sk, pk, ring, root)
    return SimpleNamespace(sk=sk, pk=pk, ring=ring, root=root, proof=proof)


def test_audit_mismatched_ring_for_same_root(ctx, audit):
    import dot_ring_amd as d

    other = d.Ring([audit.pk, *_keys(7)], audit.ring.params)
    assert audit.proof.verify(b"audit-input", b"audit-ad", audit.ring, audit.root)
    assert not audit.proof.verify(b"audit-input", b"audit-ad", other, audit.root)
    assert audit.root.matches_ring(audit.ring) and audit.root.matches_ring(audit.ring)
    assert not audit.root.matches_ring(other) and not audit.root.matches_ring(other)


def test_audit_ring_keys_pad_decode_failures_in_place(ctx):
    import dot_ring_amd as d

    cv = d.Bandersnatch
    pk1, pk2 = _keys(2)
    params = d.RingProofParams(test_vectors=True)
    ring = d.Ring([pk1, b"", (1).to_bytes(32, "little"), pk2], params)
    pad = cv.curve.params.auxiliary_points.padding_point
    tup = lambda key: (lambda p: (p.x, p.y))(cv.point_type.string_to_point(key))
    assert ring.nm_points[0] == tup(pk1) and ring.nm_points[3] == tup(pk2)
    assert ring.nm_points[1] == pad and ring.nm_points[2] == pad


def test_audit_decode_rejects_trailing_bytes_and_noncanonical_scalars(ctx, audit):
    import dot_ring_amd as d
    from dot_ring_amd.vrf.codec import point_len

    cv = d.Bandersnatch
    with pytest.raises(ValueError, match="Ring VRF proof must be exactly"):
        d.RingVRF[cv].decode(audit.proof.encode() + b"junk")
    with pytest.raises(ValueError, match="ring root must be exactly"):
        d.RingRoot.decode(audit.root.encode() + b"junk")
    raw = audit.root.encode()
    one, two = d.RingRoot.decode(raw, audit.root.params), d.RingRoot.decode(raw, audit.root.params)
    assert one.encode() == raw and two.encode() == raw and one is not two
    ped = bytearray(audit.proof.pedersen_proof.encode())
    off = 4 * point_len(cv)
    s = int.from_bytes(ped[off : off + 32], "little")
    ped[off : off + 32] = (s + cv.curve.params.subgroup_order).to_bytes(32, "little")
    with pytest.raises(ValueError, match="not canonical"):
        d.PedersenVRF[cv].decode(bytes(ped))
    with pytest.raises(ValueError, match="not canonical"):
        d.RingVRF[cv].decode(bytes(ped) + audit.proof.encode()[len(ped) :])


def test_audit_prove_rejects_wrong_producer_key_and_bad_params(ctx):
    import dot_ring_amd as d

    cv = d.Bandersnatch
    sk1 = bytes.fromhex("01" * 32)
    pk1, pk2 = cv.public_key_from_secret(sk1), cv.public_key_from_secret(bytes.fromhex("02" * 32))
    params = d.RingProofParams(test_vectors=True)
    ring = d.Ring([pk1, pk2], params)
    root = d.RingRoot.from_ring(ring, params)
    with pytest.raises(ValueError, match="producer_key does not match secret_key"):
        d.RingVRF[cv].prove(b"audit-input", b"audit-ad", sk1, pk2, ring, root)
    with pytest.raises(ValueError, match="padding_rows must be 4"):
        d.RingProofParams(padding_rows=5, max_ring_size=1)
    big = d.RingProofParams.from_ring_size(2047)
    assert (big.domain_size, big.max_ring_size) == (4096, 3839)
    with pytest.raises(ValueError, match="ring proofs require a primitive"):
        d.RingProofParams(base_root=3)


# ------------------------------------------------------------------ tests/test_ark_vrf.py:133-256
@pytest.mark.parametrize("cvname,prefix", [("Bandersnatch", "bandersnatch_sha-512_ell2"), ("JubJub", "jubjub_sha-512_tai")])
def test_batch_verify_apis_and_negative_cases(ctx, golden_dir, cvname, prefix):
    import dot_ring_amd as d

    cv = getattr(d, cvname)
    order = cv.curve.params.subgroup_order
    load = lambda scheme: _load(golden_dir, f"ark-vrf/{prefix}_{scheme}.json")
    hx = lambda v, *ks: [bytes.fromhex(v[k]) for k in ks]
    # Thin / Pedersen batch_verify with one response off by one
    vs = load("thin")[:2]
    thin = [d.ThinVRF[cv].prove(*hx(v, "alpha", "sk", "ad")) for v in vs]
    ins, ads, pks = ([bytes.fromhex(v[k]) for v in vs] for k in ("alpha", "ad", "pk"))
    assert d.ThinVRF[cv].batch_verify(thin, pks, ins, ads)
    bad = copy.copy(thin[0])
    bad.s = (bad.s + 1) % order
    assert not d.ThinVRF[cv].batch_verify([bad, thin[1]], pks, ins, ads)
    vs = load("pedersen")[:2]
    ped = [d.PedersenVRF[cv].prove(*hx(v, "alpha", "sk", "ad")) for v in vs]
    ins, ads = ([bytes.fromhex(v[k]) for v in vs] for k in ("alpha", "ad"))
    assert d.PedersenVRF[cv].batch_verify(ped, ins, ads)
    bad = d.PedersenVRF[cv](ped[0].output_point, ped[0].blinded_pk, ped[0].result_point, ped[0].ok, (ped[0].s + 1) % order, ped[0].sb)
    assert not d.PedersenVRF[cv].batch_verify([bad, ped[1]], ins, ads)
    # single-proof negatives and length errors
    v = load("tiny")[0]
    alpha, ad, pk, sk = hx(v, "alpha", "ad", "pk", "sk")
    tiny = d.TinyVRF[cv].prove(alpha, sk, ad)
    assert not tiny.verify(pk, alpha, b"wrong-ad") and not tiny.verify(pk, b"wrong-input", ad)
    with pytest.raises(ValueError, match="invalid Tiny VRF proof length"):
        d.TinyVRF[cv].decode(tiny.encode()[:-1])
    with pytest.raises(ValueError, match="invalid Thin VRF proof length"):
        d.ThinVRF[cv].decode(d.ThinVRF[cv].prove(alpha, sk, ad).encode()[:-1])
    p1 = d.PedersenVRF[cv].prove(alpha, sk, ad)
    assert not p1.verify(alpha, b"wrong-ad")
    with pytest.raises(ValueError, match="invalid Pedersen VRF proof length"):
        d.PedersenVRF[cv].decode(p1.encode()[:-1])
    # invalid point encodings (test_ark_vrf.py:88-107)
    junk = b"\xff" * 32
    for scheme, fields in ((d.TinyVRF, ("proof_c", "proof_s")), (d.ThinVRF, ("proof_r", "proof_s"))):
        vec = load("tiny" if scheme is d.TinyVRF else "thin")[0]
        with pytest.raises(ValueError, match="INVALID|Invalid"):
            scheme[cv].decode(junk + b"".join(hx(vec, *fields)))
    # ring: batch API, tampered evaluation, wrong input / ad / root, malformed encodings, wrong prover key
    rv = load("ring")[0]
    alpha, ad, sk, pk = hx(rv, "alpha", "ad", "sk", "pk")
    raw = b"".join(hx(rv, "gamma", "proof_pk_com", "proof_r", "proof_ok", "proof_s", "proof_sb", "ring_proof"))
    keys = d.RingVRF[cv].parse_keys(bytes.fromhex(rv["ring_pks"]))
    params = d.RingProofParams(test_vectors=True, cv=cv)
    ring = d.Ring(keys, params)
    root = d.RingRoot.from_ring(ring, params)
    proof = d.RingVRF[cv].decode(raw)
    second = d.RingVRF[cv].prove(b"ring-batch-second", b"ring-batch-ad", sk, pk, ring, root)
    ins, ads = [alpha, b"ring-batch-second"], [ad, b"ring-batch-ad"]
    assert d.RingVRF[cv].batch_verify([proof, second], ins, ads, ring, root)
    tampered = d.RingVRF[cv].decode(second.encode())
    tampered.l_zeta_omega = (tampered.l_zeta_omega + 1) % params.prime
    assert not d.RingVRF[cv].batch_verify([proof, tampered], ins, ads, ring, root)
    wrong_root = d.RingRoot.from_ring(d.Ring(list(reversed(keys)), params), params)
    assert not proof.verify(alpha, b"wrong-ad", ring, root)
    assert not proof.verify(b"wrong-input", ad, ring, root)
    assert not proof.verify(alpha, ad, ring, wrong_root)
    with pytest.raises(ValueError, match="invalid Ring VRF proof length"):
        d.RingVRF[cv].decode(raw[:-1])
    with pytest.raises(ValueError):
        d.RingVRF[cv].decode(raw[:192] + b"\xff" * 48 + raw[240:])
    with pytest.raises(ValueError, match="invalid ring root length"):
        d.RingRoot.decode(root.encode()[:-1])
    with pytest.raises(ValueError):
        d.RingRoot.decode(b"\xff" * 48 + root.encode()[48:])
    wrong_sk = bytes.fromhex(load("tiny")[1]["sk"])
    with pytest.raises(ValueError, match="producer key is not in ring"):
        d.RingVRF[cv].prove(alpha, ad, wrong_sk, cv.public_key_from_secret(wrong_sk), ring, root)


# ------------------------------------------------------------------ tests/test_coverage/test_kzg.py (the PCS seam by itself)
def test_kzg_commit_open_verify_batch_verify(ctx, monkeypatch):
    from dot_ring_amd.ring_proof.pcs import KZG, Opening
    from oracle.pyref import kzg as okzg

    assert KZG.commit([1, 2, 3]) is not None and KZG.commit([5]) is not None
    assert KZG.commit([0, 0, 0]) is None                                   # the point at infinity
    sparse = [1, 0, 0, 0, 0, 5, 0, 0, 0, 3]
    assert KZG.compress_g1(KZG.commit(sparse)) == okzg.compress(okzg.commit(sparse))
    asked = []
    monkeypatch.setattr(KZG, "ensure_srs_size", classmethod(lambda cls, deg: asked.append(deg)))
    KZG.commit([1, 2, 3])
    assert asked == [2]
    monkeypatch.undo()
    opening = KZG.open([1, 2, 3], 5)
    assert isinstance(opening, Opening) and opening.y == 86 and KZG.open([7, 2, 3], 0).y == 7
    cm = KZG.commit([1, 2, 3])
    assert KZG.verify(cm, opening.proof, 5, opening.y) is True
    assert KZG.verify(cm, opening.proof, 5, (opening.y + 1) % (1 << 256)) is False
    assert KZG.batch_verify([]) is True
    assert KZG.batch_verify([(cm, opening.proof, 5, opening.y)]) is True
    claims = []
    for i in range(3):
        coeffs, x = [1 + i, 2 + i, 3 + i], 5 + i
        op = KZG.open(coeffs, x)
        claims.append((KZG.commit(coeffs), op.proof, x, op.y))
    assert KZG.batch_verify(claims) is True
    op = KZG.open([1, 2, 3], 10)
    assert KZG.batch_verify(claims[:2] + [(cm, op.proof, 10, (op.y + 1000) % (1 << 256))]) is False
