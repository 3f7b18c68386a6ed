"""CPU: the oracle against every golden vector the reference's own tests hold for the hot path."""
import ctypes
import hashlib
import json
import os
import random

import pytest

from oracle import coracle
from oracle.pyref import bandersnatch as bsn
from oracle.pyref import kzg, ring, vrf

SUITES = {"sha512": bsn.SHA512, "shake128": bsn.SHAKE128, "jubjub": bsn.JUBJUB}


def _load(golden_dir, rel):
    with open(os.path.join(golden_dir, rel)) as f:
        return json.load(f)


# ---------------------------------------------------------------- field / curve restatement
def test_fr_matches_python_bigints_and_reference_c():
    rng = random.Random(1)
    p = coracle.FR_P
    ref = coracle.ref_lib()

    class S(ctypes.Structure):
        _fields_ = [("v", ctypes.c_uint64 * 4)]

    def mk(x):
        s = S()
        for i in range(4):
            s.v[i] = (x >> (64 * i)) & (2**64 - 1)
        return s

    for _ in range(200):
        a, b = rng.randrange(p), rng.randrange(p)
        assert coracle.fr_add(a, b) == (a + b) % p
        assert coracle.fr_sub(a, b) == (a - b) % p
        assert coracle.fr_mul(a, b) == a * b % p
        if ref is not None:   # the reference's own bls12_381_scalar.c (oracle/_ref), limb for limb
            o = S()
            ref.bls_scalar_mul_mont(ctypes.byref(o), ctypes.byref(mk(a)), ctypes.byref(mk(b)))
            assert sum(int(o.v[i]) << (64 * i) for i in range(4)) == coracle.fr_mul_mont_raw(a, b)
            ref.bls_scalar_add(ctypes.byref(o), ctypes.byref(mk(a)), ctypes.byref(mk(b)))
            assert sum(int(o.v[i]) << (64 * i) for i in range(4)) == (a + b) % p
    a = rng.randrange(1, p)
    assert coracle.fr_inv(a) == pow(a, -1, p)
    assert coracle.fr_sqrt(a * a % p) in (a, p - a)
    assert coracle.fr_sqrt(5) is None          # 5 is the non-residue of the Tonelli-Shanks setup
    assert coracle.fr_sqrt(0) == 0


def test_te_kernels_match_affine_law():
    rng = random.Random(2)
    for _ in range(4):
        k = rng.randrange(bsn.N)
        want = bsn.mul_py(bsn.G, k)
        assert coracle.te_mul(bsn.G, k) == want
        assert coracle.te_mul(bsn.G, k, glv=True) == want
    p1, p2 = bsn.mul_py(bsn.G, 12345), bsn.mul_py(bsn.G, 99991)
    k1, k2 = rng.randrange(1 << 127), rng.randrange(1 << 127)
    assert coracle.te_mul2(p1, k1, p2, k2) == bsn.add(bsn.mul_py(p1, k1), bsn.mul_py(p2, k2))
    for n in (1, 4, 5, 9, 40):
        pts = [coracle.te_mul(bsn.G, rng.randrange(1, bsn.N)) for _ in range(n)]
        ks = [rng.randrange(bsn.N) for _ in range(n)]
        want = bsn.IDENTITY
        for pt, k in zip(pts, ks):
            want = bsn.add(want, coracle.te_mul(pt, k))
        assert coracle.te_msm(pts, ks) == want
    assert coracle.te_msm([], []) == bsn.IDENTITY


def test_ntt_matches_definition():
    rng = random.Random(3)
    n = 16
    w = pow(ring.ROOT_OF_UNITY_2048, 2048 // n, bsn.P)
    v = [rng.randrange(bsn.P) for _ in range(n)]
    want = [sum(v[j] * pow(w, i * j, bsn.P) for j in range(n)) % bsn.P for i in range(n)]
    assert coracle.ntt(v, w) == want
    assert coracle.ntt(want, pow(w, -1, bsn.P), pow(n, -1, bsn.P)) == v
    with pytest.raises(ValueError):
        coracle.ntt([1, 2, 3], w)


def test_g1_pippenger_matches_naive():
    rng = random.Random(4)
    pts = [coracle.g1_mul(kzg.G1_GEN, rng.randrange(1, coracle.FR_P)) for _ in range(33)]
    ks = [rng.randrange(coracle.FR_P) for _ in range(33)]
    want = coracle.g1_msm_naive(pts, ks)
    for c in (0, 3, 8, 13):
        assert coracle.g1_msm(pts, ks, c) == want
    assert coracle.g1_msm(pts, [0] * 33) is None
    assert coracle.g1_mul(kzg.G1_GEN, coracle.FR_P) is None
    neg = (kzg.G1_GEN[0], coracle.FP_P - kzg.G1_GEN[1])
    assert coracle.g1_add(kzg.G1_GEN, neg) is None
    assert kzg.decompress(kzg.compress(pts[0])) == pts[0]
    assert kzg.decompress(kzg.compress(None)) is None
    with pytest.raises(ValueError):
        kzg.decompress(bytes(48))


# ---------------------------------------------------------------- reference KATs
def test_hash_to_curve_kat():
    # /root/reference/tests/test_h2c_suites/test_e2c_bandersnatch.py:4-34
    u = bsn.hash_to_field(bsn.SHA512, b"foo", 2)
    assert u == [51868557272037678616201174487618104615692125483749830231812383640086259094753,
                 28148112010555661709849764589968816930893551111405645473366527007776586648740]
    assert bsn.from_mont(*bsn.map_to_curve_ell2(u[0])) == (
        8864805491392651408849860969071502422330330403516577719645408615048305804698,
        3141991639291324936022779954882288522159181883763336030616570191472121730763)
    assert bsn.from_mont(*bsn.map_to_curve_ell2(u[1])) == (
        15951435375274270238335190310049552077179069857488797671983601737994539723286,
        7365450909453271239422049886386045513107923934059991464613784473257165293593)
    assert bsn.encode_to_curve(bsn.SHA512, b"foo") == (
        41706851287321768980670436615954402659160947743433584884323702829779219804533,
        45261115535002764022712885934321790255618221679857277845409327068867281988279)


def test_keygen_kat():
    # /root/reference/tests/test_keygen.py:6-24
    pk, sk = vrf.secret_from_seed(bsn.SHA512, (0).to_bytes(32, "little"))
    assert pk.hex() == "dff68d8158281c3ee65e678d75c7f5c007de51d0c3a800675208b7c61d2e6f98"
    assert sk.hex() == "cc1a43aef9a710b8def623da1eae8f35d7992f46302c08242e0a2bb823ccac08"
    pk, sk = vrf.secret_from_seed(bsn.SHA512, (100).to_bytes(32, "little"))
    assert pk.hex() == "84c569f6371c182164b6ca1b94097274c7071d3a005050df39c14275f60b01cf"
    assert sk.hex() == "0d28a81b0a4b8d197c7c10d60472d9ab9c5b7743803c4b68dc1a274d34009104"


TINY = [("sha512", "ark-vrf/bandersnatch_sha-512_ell2_tiny.json"), ("sha512", "ark-vrf/bandersnatch_ed_sha512_ell2_ietf.json"),
        ("sha512", "dot-ring/bandersnatch_sha-512_ell2_tiny.json"), ("shake128", "ark-vrf/bandersnatch_shake128_ell2_tiny.json"),
        ("shake128", "dot-ring/bandersnatch_shake128_ell2_tiny.json"), ("jubjub", "ark-vrf/jubjub_sha-512_tai_tiny.json"),
        ("jubjub", "ark-vrf/jubjub_sha_512_tai_ietf.json"), ("jubjub", "dot-ring/jubjub_sha-512_tai_tiny.json")]
PEDERSEN = [("sha512", "ark-vrf/bandersnatch_sha-512_ell2_pedersen.json"), ("sha512", "ark-vrf/bandersnatch_ed_sha512_ell2_pedersen.json"),
            ("sha512", "dot-ring/bandersnatch_sha-512_ell2_pedersen.json"), ("shake128", "ark-vrf/bandersnatch_shake128_ell2_pedersen.json"),
            ("shake128", "dot-ring/bandersnatch_shake128_ell2_pedersen.json"), ("jubjub", "ark-vrf/jubjub_sha-512_tai_pedersen.json"),
            ("jubjub", "ark-vrf/jubjub_sha512_tai_pedersen.json"), ("jubjub", "dot-ring/jubjub_sha-512_tai_pedersen.json")]
RING = [("sha512", "ark-vrf/bandersnatch_sha-512_ell2_ring.json"), ("sha512", "ark-vrf/bandersnatch_ed_sha512_ell2_ring.json"),
        ("sha512", "dot-ring/bandersnatch_sha-512_ell2_ring.json"), ("shake128", "ark-vrf/bandersnatch_shake128_ell2_ring.json"),
        ("shake128", "dot-ring/bandersnatch_shake128_ell2_ring.json"), ("jubjub", "ark-vrf/jubjub_sha-512_tai_ring.json"),
        ("jubjub", "dot-ring/jubjub_sha-512_tai_ring.json")]


@pytest.mark.parametrize("suite,rel", TINY)
def test_tiny_vrf_kats(golden_dir, suite, rel):
    s = SUITES[suite]
    with bsn.using(s):
        for v in _load(golden_dir, rel):
            sk, al, ad = (bytes.fromhex(v[k]) for k in ("sk", "alpha", "ad"))
            assert bsn.public_key_from_secret(sk).hex() == v["pk"]
            assert bsn.enc_point(bsn.encode_to_curve(s, al)).hex() == v["h"]
            proof = vrf.tiny_prove(s, al, sk, ad)
            assert proof.hex() == v["gamma"] + v["proof_c"] + v["proof_s"]
            assert vrf.tiny_verify(s, proof, bytes.fromhex(v["pk"]), al, ad)
            assert not vrf.tiny_verify(s, proof, bytes.fromhex(v["pk"]), al + b"x", ad)
            assert vrf.point_to_hash(s, bsn.decompress(proof[:32])).hex() == v["beta"][:64]


@pytest.mark.parametrize("suite,rel", PEDERSEN)
def test_pedersen_vrf_kats(golden_dir, suite, rel):
    s = SUITES[suite]
    with bsn.using(s):
        for v in _load(golden_dir, rel):
            sk, al, ad = (bytes.fromhex(v[k]) for k in ("sk", "alpha", "ad"))
            proof, blinding = vrf.pedersen_prove(s, al, sk, ad)
            assert proof.hex() == v["gamma"] + v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"] + v["proof_sb"]
            assert bsn.enc_scalar(blinding).hex() == v["blinding"]
            assert vrf.pedersen_verify(s, proof, al, ad)
            assert not vrf.pedersen_verify(s, proof, al, ad + b"x")


@pytest.mark.parametrize("suite,rel", RING)
def test_ring_vrf_kats_byte_exact(golden_dir, suite, rel):
    """Full 784-byte proofs incl. all ten G1 commitments per vector (the only tests that pin G1 MSM values:
    /root/reference/tests/test_ring_vrf/test_ring_vrf.py:41-45, tests/test_dot_ring_vectors.py:72)."""
    s = SUITES[suite]
    with bsn.using(s):
        for v in _load(golden_dir, rel):
            sk, al, ad = (bytes.fromhex(v[k]) for k in ("sk", "alpha", "ad"))
            raw = bytes.fromhex(v["ring_pks"])
            keys = [raw[i : i + 32] for i in range(0, len(raw), 32)]
            params = ring.Params(test_vectors=True, suite=s)
            rg = ring.Ring(keys, params)
            root = ring.RingRoot(rg)
            assert root.encode().hex() == v["ring_pks_com"]
            proof = ring.ring_vrf_prove(rg, root, al, ad, sk)
            assert proof.hex() == (v["gamma"] + v["proof_pk_com"] + v["proof_r"] + v["proof_ok"] + v["proof_s"]
                                   + v["proof_sb"] + v["ring_proof"])


def test_safrole_selector_commitment_n2048(golden_dir):
    """tests/vectors/full/safrole-ring-root.json (unused by the reference's tests): its third commitment — the
    N=2048 selector column — is reproduced; the px/py thirds of that file do not follow the reference's own
    padding rule (they differ from what Ring()/RingRoot.from_ring produce), so only the selector is pinned."""
    d = _load(golden_dir, "full/safrole-ring-root.json")
    params = ring.Params(domain_size=d["domain_size"], max_ring_size=d["max_ring_size"])
    s_evals = [1 if i < params.max_ring else 0 for i in range(params.N)]
    c_s = kzg.commit(ring.intt(s_evals, params.omega))
    assert kzg.compress(c_s).hex() == d["ring_root_hex"][192:]


def test_params_capacity_table():
    # /root/reference/tests/test_coverage/test_params.py:64-88, test_audit_regressions.py:112-114
    assert ring.Params.from_ring_size(8).N == 512
    assert ring.Params.from_ring_size(255).N == 512
    assert ring.Params.from_ring_size(256).N == 1024
    assert ring.Params.from_ring_size(1024).N == 2048
    assert ring.Params.from_ring_size(3839).N == 4096
    with pytest.raises(ValueError):
        ring.Params.from_ring_size(3840)
    with pytest.raises(ValueError):
        ring.Params.from_ring_size(0)


from oracle.gen_ntt_fixtures import seeded_inputs as _ntt_fixture_inputs      # the input rule of tests/golden/ntt/ntt_cases.json


def test_ntt_matches_the_reference_kernel_fixtures(golden_dir):
    """tests/golden/ntt/ntt_cases.json holds outputs of the REFERENCE'S OWN NTT kernel (ntt.pyx + scalar.pyx over bls12_381_scalar.c,
    built into oracle/_ref by oracle/build_ref_ntt.sh, vectors written by oracle/gen_ntt_fixtures.py): n = 2 ... 16384, forward,
    inverse with scale 1/n, and an arbitrary scale.  The oracle's NTT must reproduce every one of them."""
    with open(os.path.join(golden_dir, "ntt", "ntt_cases.json")) as f:
        fx = json.load(f)
    assert int(fx["modulus"], 16) == bsn.P and len(fx["cases"]) == 42
    for case in fx["cases"]:
        n = 1 << case["log2n"]
        vals = _ntt_fixture_inputs(n, case["input_tag"])
        raw = b"".join(v.to_bytes(32, "little") for v in vals)
        assert hashlib.sha256(raw).hexdigest() == case["input_sha256"]
        if "input" in case:
            assert [int(v, 16) for v in case["input"]] == vals
        scale = None if case["scale"] is None else int(case["scale"], 16)
        out = bytes(coracle.ntt_raw(raw, n, int(case["omega"], 16), scale))
        assert hashlib.sha256(out).hexdigest() == case["output_sha256"], (case["log2n"], case["kind"])
        got = [int.from_bytes(out[32 * i : 32 * i + 32], "little") for i in range(n)]
        assert [hex(v) for v in got[:4]] == case["output_head"] and [hex(v) for v in got[-4:]] == case["output_tail"]
        if "output" in case:
            assert [hex(v) for v in got] == case["output"]


def test_reference_ntt_build_matches_fixtures_when_present(golden_dir):
    """Where oracle/_ref/pyx exists (this container: built from /root/reference by oracle/build_ref_ntt.sh), the reference kernel
    itself is run again on three cases and must equal both the committed fixtures and the oracle."""
    root = os.path.abspath(os.path.join(golden_dir, "..", ".."))
    pyx = os.path.join(root, "oracle", "_ref", "pyx")
    if not os.path.isdir(os.path.join(pyx, "dot_ring")):
        pytest.skip("oracle/_ref/pyx not built (the reference sources are absent here)")
    import subprocess
    import sys

    code = (
        "import sys, json, hashlib; sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])\n"
        "from oracle import gen_ntt_fixtures as g\n"
        "from dot_ring.ring_proof.polynomial.ntt import BlsScalarNTTPlan\n"
        "out = {}\n"
        "for log2n, kind in ((3, 'forward'), (11, 'inverse'), (13, 'scaled')):\n"
        "    n = 1 << log2n; w = g.root_of_unity(n)\n"
        "    omega = w if kind != 'inverse' else pow(w, -1, g.P)\n"
        "    scale = None if kind == 'forward' else pow(n, -1, g.P) if kind == 'inverse' else (0x1234567 + 977 * log2n) * pow(3, 200 + log2n, g.P) % g.P\n"
        "    v = g.seeded_inputs(n, f'{log2n}-{kind}')\n"
        "    plan = BlsScalarNTTPlan(g.stage_twiddles(n, omega), g.bit_reverse(n))\n"
        "    plan.transform(v) if scale is None else plan.transform_scaled(v, scale)\n"
        "    out[f'{log2n}-{kind}'] = g.digest(v)\n"
        "print(json.dumps(out))\n"
    )
    proc = subprocess.run([sys.executable, "-c", code, os.path.abspath(pyx), root], capture_output=True, text=True, check=True)
    got = json.loads(proc.stdout)
    with open(os.path.join(golden_dir, "ntt", "ntt_cases.json")) as f:
        fx = {c["input_tag"]: c["output_sha256"] for c in json.load(f)["cases"]}
    for tag, dg in got.items():
        assert fx[tag] == dg, tag
