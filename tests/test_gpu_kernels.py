"""GPU parity: HIP kernels (through the C ABI) against the CPU oracle on the same seeded inputs. Bit-exact."""
import hashlib
import random

import pytest

from oracle import coracle
from oracle.pyref import bandersnatch as bsn

pytestmark = pytest.mark.gpu

N = bsn.N
P = bsn.P


def _seeded_points(count, tag=b"bsn-pt"):
    pts = []
    for i in range(count):
        k = int.from_bytes(hashlib.sha256(tag + i.to_bytes(8, "little")).digest(), "little") % N
        pts.append(coracle.te_mul(bsn.G, k or 1))
    return pts


def _seeded_scalars(count, tag=b"bsn-k", mod=N):
    return [int.from_bytes(hashlib.sha256(tag + i.to_bytes(8, "little")).digest(), "little") % mod for i in range(count)]


# ------------------------------------------------------------------ seam A
def test_bsn_scalar_mul_batch_matches_oracle(ctx):
    n = 300
    pts, ks = _seeded_points(n), _seeded_scalars(n)
    # edge scalars: 0, 1, n-1, n (== 0), 2^256-1 (reduced inside), small values
    ks[:8] = [0, 1, N - 1, N, (1 << 256) - 1, 2, 8, 16]
    got = ctx.bsn_scalar_mul_batch(coracle.te_pack(pts), b"".join(k.to_bytes(32, "little") for k in ks))
    want = coracle.te_mul_batch_raw(coracle.te_pack(pts), coracle.scalars_pack([k % N for k in ks]), n, glv=False)
    assert got == want


def config2_inputs(n=4096):
    """BASELINE configs[1] / SURVEY 8(d) config 2: P_i = public key of secret_from_seed(sha256("bsn-pt" || LE64(i))) — valid
    prime-order points — and k_i = sha256("bsn-k" || LE64(i)) as a little-endian integer mod n."""
    from oracle.pyref import vrf as ovrf

    pks = [ovrf.secret_from_seed(bsn.SHA512, hashlib.sha256(b"bsn-pt" + i.to_bytes(8, "little")).digest())[0] for i in range(n)]
    return [bsn.dec_point(pk) for pk in pks], _seeded_scalars(n)


def test_bsn_scalar_mul_config2_4096(ctx):
    """BASELINE configs[1] at its stated size: 4096 variable-base scalar multiplications, bit-exact against the oracle's
    restatement of the reference kernel (curve/specs/bandersnatch.py:177-191 -> glv.py:191 -> bandersnatch_te.pyx:480)."""
    n = 4096
    pts, ks = config2_inputs(n)
    raw_p, raw_k = coracle.te_pack(pts), coracle.scalars_pack(ks)
    got = ctx.bsn_scalar_mul_batch(raw_p, raw_k)
    assert got == coracle.te_mul_batch_raw(raw_p, raw_k, n, glv=True)
    assert got == coracle.te_mul_batch_raw(raw_p, raw_k, n, glv=False)
    # device-resident variant (the one bench.py times) gives the same bytes
    d_p, d_k, d_o = ctx.alloc(64 * n).upload(raw_p), ctx.alloc(32 * n).upload(raw_k), ctx.alloc(64 * n)
    ctx.bsn_scalar_mul_batch_dev(d_p, d_k, n, d_o)
    assert d_o.download() == got
    for b in (d_p, d_k, d_o):
        b.free()


def test_bsn_scalar_mul_device_resident_glv_split_on_device(ctx):
    """The device-resident entry point decomposes every scalar on the device (k_bsn_scalar_mul_glv<true>: reduce mod n, the lattice
    rounding of glv.py:128-163 in 32-bit words): edge scalars — 0, 1, n - 1, n, n + 1, 2^128 +- 1, lambda, values >= n up to
    2^256 - 1, powers of two, all-ones halves — the identity as input, ragged sizes, against the oracle's plain and GLV kernels."""
    lam = 0x13B4F3DC4A39A493EDF849562B38C72BCFC49DB970A5056ED13D21408783DF05
    edge = [0, 1, 2, N - 1, N, N + 1, (1 << 128) - 1, 1 << 128, (1 << 128) + 1, lam, lam - 1, N - lam, (1 << 256) - 1, (1 << 255) + 12345,
            (1 << 252), (1 << 253) - 1, 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF << 128, 0x8000000000000000 << 64, 4 * N - 1, 8 * N + 7]
    edge += [1 << b for b in range(0, 256, 17)]
    for n in (1, 2, 63, 64, 65, 700):
        pts = _seeded_points(n, b"dv")
        ks = (edge + _seeded_scalars(max(0, n - len(edge)), b"dvk", 1 << 256))[:n]
        if n >= 3:
            pts[2] = bsn.IDENTITY
        raw_p, raw_k = coracle.te_pack(pts), b"".join(k.to_bytes(32, "little") for k in ks)
        want = coracle.te_mul_batch_raw(raw_p, coracle.scalars_pack([k % N for k in ks]), n, glv=False)
        assert want == coracle.te_mul_batch_raw(raw_p, coracle.scalars_pack([k % N for k in ks]), n, glv=True)
        d_p, d_k, d_o = ctx.alloc(64 * n).upload(raw_p), ctx.alloc(32 * n).upload(raw_k), ctx.alloc(64 * n)
        ctx.bsn_scalar_mul_batch_dev(d_p, d_k, n, d_o)
        assert d_o.download() == want, n
        for b in (d_p, d_k, d_o):
            b.free()


def test_bsn_scalar_mul_identity_and_kat(ctx):
    # identity input, and the reference KAT pk = sk*G (tests/golden/ark-vrf/bandersnatch_sha-512_ell2_tiny.json #1)
    sk = int.from_bytes(bytes.fromhex("c9922b7a9849b9928e15c655dd2f22ceef737cc355024f43d4b04bf4398c270d"), "little")
    out = ctx.bsn_scalar_mul_batch(coracle.te_pack([bsn.IDENTITY, bsn.G]), (5).to_bytes(32, "little") + sk.to_bytes(32, "little"))
    pts = coracle.te_unpack(out)
    assert pts[0] == bsn.IDENTITY
    assert bsn.enc_point(pts[1]).hex() == "5a538209ff1fc7b1c9c8e1da05b3e169acf10a8b1591b3af029fe4eede0bbc71"


def test_bsn_scalar_mul_ragged_sizes(ctx):
    for n in (1, 63, 64, 65, 129):
        pts, ks = _seeded_points(n, b"rag"), _seeded_scalars(n, b"ragk")
        got = ctx.bsn_scalar_mul_batch(coracle.te_pack(pts), coracle.scalars_pack(ks))
        assert got == coracle.te_mul_batch_raw(coracle.te_pack(pts), coracle.scalars_pack(ks), n, glv=True)
    assert ctx.bsn_scalar_mul_batch(b"", b"") == b""


def test_bsn_rejects_non_canonical_coordinate(ctx):
    bad = (P).to_bytes(32, "little") + (1).to_bytes(32, "little")
    with pytest.raises(ValueError):
        ctx.bsn_scalar_mul_batch(bad, (1).to_bytes(32, "little"))


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 17, 64, 65, 200])
def test_bsn_msm_matches_oracle(ctx, n):
    pts, ks = _seeded_points(n, b"msm"), _seeded_scalars(n, b"msmk")
    got = coracle.te_unpack(ctx.bsn_msm(coracle.te_pack(pts), coracle.scalars_pack(ks)))[0]
    assert got == (coracle.te_msm(pts, ks) if n else bsn.IDENTITY)


@pytest.mark.parametrize("n", [256, 257, 1024, 5122, 20482, 65536])
def test_bsn_msm_pippenger_matches_oracle(ctx, n):
    """K4: from 256 terms dr_bsn_msm is a signed-digit bucket Pippenger (the reference's msm_pippenger_signed_native_cy,
    bandersnatch_te.pyx:257-418); sizes incl. PedersenVRF.batch_verify's 5B + 2 at B = 1024 / 4096; duplicates, zero
    scalars, the identity point, k = n - 1"""
    base = _seeded_points(min(n, 512), b"pip")
    pts = [base[i % len(base)] for i in range(n)]
    ks = _seeded_scalars(n, b"pipk")
    ks[0], ks[1], ks[2] = 0, N - 1, 1
    pts[3] = bsn.IDENTITY
    pts[5] = pts[4]
    ks[5] = (N - ks[4]) % N                      # P and -P in the same buckets
    got = coracle.te_unpack(ctx.bsn_msm(coracle.te_pack(pts), coracle.scalars_pack(ks)))[0]
    assert got == coracle.te_msm(pts, ks)


def test_bsn_msm_pippenger_all_zero_and_window_choices(ctx, monkeypatch):
    n = 300
    pts = _seeded_points(n, b"pz")
    assert coracle.te_unpack(ctx.bsn_msm(coracle.te_pack(pts), bytes(32 * n)))[0] == bsn.IDENTITY
    ks = _seeded_scalars(n, b"pzk")
    want = coracle.te_msm(pts, ks)
    assert coracle.te_unpack(ctx.bsn_msm(coracle.te_pack(pts), coracle.scalars_pack(ks)))[0] == want


def test_fixed_base_tables_match_oracle(ctx):
    """dr_te_fixed_base_msm_groups: k*G, x*G + b*B (the Pedersen prover's constant bases, vrf/pedersen/vrf.py:94-111) from the
    window tables against the oracle's variable-base arithmetic; edge scalars 0, 1, n-1, n, 2^256-1, every digit +-8"""
    gen = bsn.G
    bb = bsn.SHA512.blinding_base
    ks = _seeded_scalars(150, b"fixed", mod=1 << 256)
    ks[:8] = [0, 1, N - 1, N, (1 << 256) - 1, int("8" * 63, 16), int("7" * 63, 16), 2]
    raw = b"".join(k.to_bytes(32, "little") for k in ks)
    got = coracle.te_unpack(ctx.te_fixed_base_msm_groups(coracle.te_pack([gen]), raw))
    assert got == [coracle.te_mul(gen, k % N) for k in ks]
    got = coracle.te_unpack(ctx.te_fixed_base_msm_groups(coracle.te_pack([gen, bb]), raw))
    assert got == [bsn.add(coracle.te_mul(gen, ks[2 * i] % N), coracle.te_mul(bb, ks[2 * i + 1] % N)) for i in range(75)]
    got = coracle.te_unpack(ctx.te_fixed_base_msm_groups(coracle.te_pack([gen, bb, gen]), raw))      # 3 bases: groups padded to 4 terms
    assert got == [bsn.add(bsn.add(coracle.te_mul(gen, ks[3 * i] % N), coracle.te_mul(bb, ks[3 * i + 1] % N)), coracle.te_mul(gen, ks[3 * i + 2] % N))
                   for i in range(50)]
    with bsn.using(bsn.JUBJUB):                                      # a = -1 curve, 252-bit order
        jk = [k % (1 << 256) for k in ks[:40]]
        got = coracle.te_unpack(ctx.te_fixed_base_msm_groups(coracle.te_pack([bsn.G]), b"".join(k.to_bytes(32, "little") for k in jk), 1))
        assert got == [bsn.mul(bsn.G, k) for k in jk]
    with pytest.raises(ValueError):
        ctx.te_fixed_base_msm_groups(coracle.te_pack([gen] * 5), raw[: 32 * 5])


def test_bsn_msm_groups(ctx):
    for m in (2, 3, 4):
        groups = 37
        pts, ks = _seeded_points(groups * m, b"grp"), _seeded_scalars(groups * m, b"grpk")
        out = coracle.te_unpack(ctx.bsn_msm_groups(coracle.te_pack(pts), coracle.scalars_pack(ks), m))
        for g in range(groups):
            assert out[g] == coracle.te_msm(pts[g * m : (g + 1) * m], ks[g * m : (g + 1) * m], 2)


# ------------------------------------------------------------------ seam B
def _be_to_le(raw96: bytes) -> bytes:
    return raw96[:48][::-1] + raw96[48:][::-1]


def _le_pack_from_be(blob: bytes, n: int) -> bytes:
    return b"".join(_be_to_le(blob[96 * i : 96 * i + 96]) for i in range(n))


def _oracle_msm_be(blob_be: bytes, ks: bytes, n: int):
    out = coracle.g1_msm_raw(_le_pack_from_be(blob_be, n), ks, n)
    if out == bytes(96):
        return None
    return out[:48][::-1] + out[48:][::-1]


@pytest.mark.parametrize("n", [1, 2, 7, 100, 1000, 2048, 6145])
def test_g1_msm_matches_oracle(ctx, srs_bytes, n):
    srs = ctx.srs_load(srs_bytes)
    rng = random.Random(n)
    ks = b"".join(rng.randrange(coracle.FR_P).to_bytes(32, "little") for _ in range(n))
    assert ctx.g1_msm(srs, ks) == _oracle_msm_be(srs_bytes, ks, n)
    srs.close()


def test_g1_msm_edge_scalars(ctx, srs_bytes):
    srs = ctx.srs_load(srs_bytes[: 96 * 64])
    n = 64
    assert ctx.g1_msm(srs, bytes(32 * n)) is None                      # all-zero -> infinity (kzg.py:167)
    assert ctx.g1_msm(srs, b"") is None
    # 0/1 column (the ring prover's bit column), tiny values, r-1, values >= r (unreduced quotient coefficients)
    r = coracle.FR_P
    vals = [1, 0] * 16 + [3, 5, 255, 256, 65535, 65536] + [r - 1, r, r + 1, (1 << 256) - 1] + [7] * 22
    ks = b"".join(v.to_bytes(32, "little") for v in vals)
    assert ctx.g1_msm(srs, ks) == _oracle_msm_be(srs_bytes, b"".join((v % r).to_bytes(32, "little") for v in vals), n)
    # offset into the SRS
    ks2 = b"".join((v % r).to_bytes(32, "little") for v in vals[:20])
    assert ctx.g1_msm(srs, ks2, offset=11) == _oracle_msm_be(srs_bytes[96 * 11 :], ks2, 20)
    with pytest.raises(ValueError):
        ctx.g1_msm(srs, bytes(32 * 65))
    srs.close()


def test_g1_msm_points_duplicates_and_cancellation(ctx, srs_bytes):
    g = srs_bytes[:96]
    neg_y = (coracle.FP_P - int.from_bytes(g[48:], "big")).to_bytes(48, "big")
    pts = g * 5 + g[:48] + neg_y + bytes(96)                 # 5 x G, -G, infinity
    ks = b"".join(v.to_bytes(32, "little") for v in (1, 1, 2, 3, 9, 16, 12345))
    assert ctx.g1_msm_points(pts, ks) is None                # 16 G - 16 G
    ks = b"".join(v.to_bytes(32, "little") for v in (1, 1, 2, 3, 9, 15, 12345))
    assert ctx.g1_msm_points(pts, ks) == g                   # = 1 * G
    with pytest.raises(ValueError):
        ctx.g1_msm_points(g[:48] + bytes(47) + b"\x01", (1).to_bytes(32, "little"))   # not on the curve


@pytest.mark.parametrize("window", [0])
def test_g1_msm_batch_matches_singles(ctx, srs_bytes, window):
    srs = ctx.srs_load(srs_bytes[: 96 * 512])
    n, batch = 512, 5
    rng = random.Random(99)
    ks = b"".join(rng.randrange(coracle.FR_P).to_bytes(32, "little") for _ in range(n * batch))
    ks = ks[: 32 * n * 4] + bytes(32 * n)                    # last vector all zero -> infinity
    got = ctx.g1_msm_batch(srs, ks, n)
    for b in range(batch):
        assert got[b] == _oracle_msm_be(srs_bytes, ks[32 * n * b : 32 * n * (b + 1)], n)
    srs.close()


@pytest.mark.parametrize("table", [0, 12])
def test_g1_msm_skewed_scalars_take_the_heavy_bucket_kernel(ctx, srs_bytes, table):
    """every scalar equal / a 0-1 column / r - 1 everywhere: thousands of points in ONE bucket per window — walked by a whole
    wave (k_g1_accumulate_heavy) instead of one lane; results against the oracle, alone and inside a batch"""
    import time

    n = 6145
    srs = ctx.srs_load(srs_bytes[: 96 * n])
    if table:
        srs.precompute(table)
    rng = random.Random(255)
    cols = {
        "ones": [1] * n,
        "same": [0x123456789ABCDEF0123456789ABCDEF] * n,
        "bits": [rng.randrange(2) for _ in range(n)],
        "minus one": [coracle.FR_P - 1] * n,
        "three values": [rng.choice([5, 1 << 200, coracle.FR_P - 7]) for _ in range(n)],
    }
    for name, col in cols.items():
        raw = b"".join(k.to_bytes(32, "little") for k in col)
        t0 = time.perf_counter()
        got = ctx.g1_msm(srs, raw)
        assert time.perf_counter() - t0 < 0.5, name          # one lane per bucket took 28 ms and more here; the wave kernel < 2 ms
        assert got == _oracle_msm_be(srs_bytes, raw, n), name
    batch = b"".join(b"".join(k.to_bytes(32, "little") for k in col) for col in cols.values())
    got = ctx.g1_msm_batch(srs, batch, n)
    for g, col in zip(got, cols.values()):
        assert g == _oracle_msm_be(srs_bytes, b"".join(k.to_bytes(32, "little") for k in col), n)
    srs.close()


@pytest.mark.parametrize("log2n,table", [(16, 0), (16, 12), (18, 16), (20, 16)])
def test_g1_msm_synthetic_bases_closed_form(ctx, log2n, table):
    """BASELINE configs[2] sizes (2^16 and 2^20): no SRS of that size exists, so bases are (1+i)*G
    generated on the GPU and the MSM must equal [sum k_i (1+i)]*G — one oracle scalar multiplication; the first 4096
    pairs are also checked against the oracle's Pippenger."""
    import bench

    n = 1 << log2n
    srs = ctx.srs_synthetic(bench.G1_BE, n, first=1)
    if table:
        srs.precompute(table)
    vals, raw = bench.seeded_scalars(n, b"test")
    got = ctx.g1_msm(srs, raw)
    expect = sum(k * (1 + i) for i, k in enumerate(vals)) % coracle.FR_P
    want = coracle.g1_msm_raw(bench.be_to_le_points(bench.G1_BE), expect.to_bytes(32, "little"), 1)
    assert got == bytes(want)[:48][::-1] + bytes(want)[48:][::-1]
    m = 4096
    cpu = coracle.g1_msm_raw(bench.be_to_le_points(srs.download(0, m)), raw[: 32 * m], m)
    gpu = ctx.g1_msm(srs, raw[: 32 * m])
    assert gpu == bytes(cpu)[:48][::-1] + bytes(cpu)[48:][::-1]
    srs.close()


@pytest.mark.parametrize("log2n,table", [(16, 16), (20, 20), (13, 0)])
def test_g1_msm_on_the_surveys_config3_bases(ctx, log2n, table):
    """BASELINE configs[2] on SURVEY 8(d)'s inputs: bases = the real SRS (6145 points of the shipped file) followed by [t^i] G1.  The
    expected value accounts for all 2^log2n pairs: oracle Pippenger over the real prefix + [sum_{i >= 6145} k_i t^i] G1; the tables
    are bench.py's (16-bit windows at 2^16, 20-bit at 2^20), and 2^13 runs over plain bases."""
    import bench

    n = 1 << log2n
    srs, real_be, t = bench.survey_msm_bases(ctx, n)
    assert srs.download(0, 2) == real_be[:192] and srs.download(6144, 1) == real_be[96 * 6144 :]
    one = srs.download(6145, 2)                       # [t^6145] G1, [t^6146] G1: the oracle's scalar multiplications
    for j in range(2):
        w = coracle.g1_msm_raw(bench.be_to_le_points(bench.G1_BE), pow(t, 6145 + j, coracle.FR_P).to_bytes(32, "little"), 1)
        assert one[96 * j : 96 * j + 96] == bytes(w)[:48][::-1] + bytes(w)[48:][::-1]
    if table:
        srs.precompute(table)
    vals, raw = bench.seeded_scalars(n, b"\0\0\0\0")
    assert ctx.g1_msm(srs, raw) == bench.survey_msm_expected(real_be, t, vals, raw)
    srs.close()


def test_fr29_field_arithmetic_against_big_integers(ctx):
    """the unsaturated Fr of the twisted Edwards kernels (csrc/fr29.hip.h: 9 signed limbs of 29 bits, lazy reduction) through
    dr_fr_ops_selftest: products, squarings, sums, differences, inverses (division steps), a product of two lazy operands,
    the -5 a addition chain of the group law, the fused two-product and Tonelli-Shanks, on the edges of the field (0, 1, 2,
    p - 1, p - 2, values around 2^29 k and 2^255 - small, all-ones limb patterns) and 20 000 random pairs"""
    p = coracle.FR_P
    rng = random.Random(29)
    edge = [0, 1, 2, 3, 5, p - 1, p - 2, p - 5, (p - 1) // 2, (p + 1) // 2, 1 << 29, (1 << 29) - 1, (1 << 58) + 1, (1 << 232) - 1, 1 << 232,
            (1 << 254) - 1, 1 << 254, p - (1 << 29), p - (1 << 232), sum(((1 << 29) - 1) << (29 * i) for i in range(8)) % p,
            pow(5, (p - 1) >> 32, p), pow(7, -1, p)]
    pairs = [(x, y) for x in edge for y in edge] + [(rng.randrange(p), rng.randrange(p)) for _ in range(20000)]
    r261 = pow(2, 261, p)
    a = b"".join(x.to_bytes(32, "little") for x, _ in pairs)
    b = b"".join(y.to_bytes(32, "little") for _, y in pairs)
    out, flags = ctx.fr_ops_selftest(a, b)
    for i, (x, y) in enumerate(pairs):
        rec = [int.from_bytes(out[384 * i + 32 * k : 384 * i + 32 * k + 32], "little") for k in range(12)]
        want = [x * y % p, x * x % p, (x + y) % p, (x - y) % p, pow(x, -1, p) if x else 0, (x * x - y * y) % p, (-5 * x) % p, 2 * x * y % p]
        assert rec[:8] == want, (hex(x), hex(y))
        square = x == 0 or pow(x, (p - 1) // 2, p) == 1
        assert flags[i] == (1 if square else 0), hex(x)
        assert (rec[8] * rec[8] - x) % p == 0 if square else rec[8] == 0, hex(x)
        # the lazy-sum helpers of the NTT / polynomial kernels (records in the device's Montgomery form, R = 2^261)
        assert rec[9:] == [(x * y + x) * r261 % p, (x + 27 * y) * r261 % p, (x - 28 * y) * r261 % p], (hex(x), hex(y))


def _skewed_columns(n, rng):
    import bench

    vals, _ = bench.seeded_scalars(n, b"part")
    return {
        "random": vals,
        "same": [0x1234567 << 180 | 0xABCDEF] * n,
        "bits": [rng.randrange(2) for _ in range(n)],
        "three values, zeros": [rng.choice([0, 5, 1 << 200, coracle.FR_P - 7]) for _ in range(n)],
        "minus one": [coracle.FR_P - 1] * n,
    }


def _closed_form_be(col):
    import bench

    expect = sum(k * (1 + i) for i, k in enumerate(col)) % coracle.FR_P
    want = coracle.g1_msm_raw(bench.be_to_le_points(bench.G1_BE), expect.to_bytes(32, "little"), 1)
    return bytes(want)[:48][::-1] + bytes(want)[48:][::-1]


def test_g1_msm_partition_sort_skewed_and_ragged(ctx):
    """the two-pass partition sort of one huge table MSM (k_g1_part_scatter / k_g1_part_sort) away from uniformly random
    scalars: a ragged size (index groups of unequal length, a last tile of a few scalars), every scalar equal (one bucket per
    window gets everything: a stream overfilled on the first try, the second run with exact stream offsets, a partition of many
    stage chunks, entries past the LDS stage), a 0/1 column, three distinct values, zero scalars mixed in — closed form
    [sum k_i (1 + i)] G each time; 16- and 18-bit windows (32 and 256 partitions: per-wave and per-workgroup counters)"""
    import bench

    n = (1 << 18) + 37
    for bits in (16, 18):
        srs = ctx.srs_synthetic(bench.G1_BE, n, first=1)
        srs.precompute(bits)
        for name, col in _skewed_columns(n, random.Random(4242)).items():
            raw = b"".join(k.to_bytes(32, "little") for k in col)
            assert ctx.g1_msm(srs, raw) == _closed_form_be(col), (bits, name)
        srs.close()


@pytest.mark.parametrize("table", [0, 13])
def test_g1_msm_many_heavy_lists_one_entry_over_a_segment(ctx, table):
    """51 distinct scalars, each 1025 times: ~1000 bucket lists of 1025 entries, one over the 1024-entry segment, so every list is
    cut in two and the segment sums number twice the heavy lists (the bound ctx->heavy is reserved for: total / seg + n_heavy);
    closed form [sum k_i (1 + i)] G, then the same call again (the scratch neighbours of `heavy` must have survived)"""
    import bench

    rng = random.Random(1025)
    distinct = [rng.randrange(coracle.FR_P) for _ in range(51)]
    col = [distinct[i % 51] for i in range(51 * 1025)]
    n = len(col)
    srs = ctx.srs_synthetic(bench.G1_BE, n, first=1)
    if table:
        srs.precompute(table)
    raw = b"".join(k.to_bytes(32, "little") for k in col)
    want = _closed_form_be(col)
    assert ctx.g1_msm(srs, raw) == want
    assert ctx.g1_msm(srs, raw) == want
    vals, raw2 = bench.seeded_scalars(n, b"after-heavy")
    assert ctx.g1_msm(srs, raw2) == _closed_form_be(vals)
    srs.close()


def test_g1_msm_degenerate_scalars_at_full_size_within_twice_the_random_time(ctx):
    """BASELINE configs[2] at 2^20 pairs over 20-bit windows (one set of 2^19 buckets, 512 partitions): all-equal scalars, a 0/1
    column, r - 1 everywhere and three distinct values put a million entries into a handful of buckets.  Their lists are cut into
    segments over 2048 waves (k_g1_accumulate_heavy / k_g1_heavy_fold) and the sort runs a second time with exact stream offsets:
    the results still equal the closed form, and no such vector takes more than twice the time of a random one."""
    import time

    import bench

    n = 1 << 20
    srs = ctx.srs_synthetic(bench.G1_BE, n, first=1)
    srs.precompute(20)
    times = {}
    for name, col in _skewed_columns(n, random.Random(77)).items():
        raw = b"".join(k.to_bytes(32, "little") for k in col)
        d = ctx.alloc(32 * n).upload(raw)
        got = ctx.g1_msm_dev(srs, d, n)
        assert got == _closed_form_be(col), name
        for _ in range(2):
            ctx.g1_msm_dev(srs, d, n)
        t0 = time.perf_counter()
        for _ in range(4):
            ctx.g1_msm_dev(srs, d, n)
        times[name] = (time.perf_counter() - t0) / 4
        d.free()
    srs.close()
    for name, t in times.items():
        assert t <= 2.0 * times["random"], (name, times)


@pytest.mark.parametrize("bits,batch", [(12, 2100), (9, 16500)])
def test_g1_msm_many_bucket_sets_level_reduction(ctx, srs_bytes, bits, batch):
    """Thousands of small MSMs over a window table take the level-wise bucket reduction (k_g1_reduce_level/_final:
    two levels at H = 2048, one at H = 256); sampled results against the oracle, plus all-zero and single-term vectors."""
    n = 24
    rng = random.Random(bits)
    tabled = ctx.srs_load(srs_bytes[: 96 * n]).precompute(bits)
    vecs = [b"".join(rng.randrange(coracle.FR_P).to_bytes(32, "little") for _ in range(n)) for _ in range(5)]
    vecs.append(bytes(32 * n))                                                     # -> infinity
    vecs.append((coracle.FR_P - 1).to_bytes(32, "little") + bytes(32 * (n - 1)))   # -G
    vecs.append(bytes(32 * (n - 1)) + (1).to_bytes(32, "little"))                  # last base
    ks = b"".join(vecs[i % len(vecs)] for i in range(batch))
    got = ctx.g1_msm_batch(tabled, ks, n)
    want = [_oracle_msm_be(srs_bytes, v, n) for v in vecs]
    assert want[5] is None
    for i in range(batch):
        assert got[i] == want[i % len(vecs)], i
    tabled.close()


@pytest.mark.parametrize("bits,n,batch", [(9, 700, 40), (12, 300, 70), (10, 64, 4100), (8, 1, 33)])
def test_g1_msm_batched_over_table_matches_plain(ctx, srs_bytes, bits, n, batch):
    """batches of MSMs over a fixed-base table (window rows below 256 vectors, the non-adjacent form over bit rows from there on) —
    identical results to plain bases and the oracle, incl. zero vectors, scalars >= r and 2^255"""
    rng = random.Random(bits * 1000 + n)
    plain = ctx.srs_load(srs_bytes[: 96 * (n + 5)])
    comb = ctx.srs_load(srs_bytes[: 96 * (n + 5)]).precompute(bits)
    vecs = [b"".join(rng.randrange(coracle.FR_P).to_bytes(32, "little") for _ in range(n)) for _ in range(6)]
    vecs.append(bytes(32 * n))
    vecs.append(b"".join(v.to_bytes(32, "little") for v in ([1, coracle.FR_P - 1, (1 << 256) - 1, 2**255, coracle.FR_P] * n)[:n]))
    ks = b"".join(vecs[i % len(vecs)] for i in range(batch))
    got = ctx.g1_msm_batch(comb, ks, n)
    want = [_oracle_msm_be(srs_bytes, v, n) for v in vecs]
    for i in range(batch):
        assert got[i] == want[i % len(vecs)], i
    assert ctx.g1_msm_batch(plain, ks[: 32 * n * 3], n) == got[:3]
    plain.close()
    comb.close()


@pytest.mark.parametrize("bits,n", [(9, 513), (10, 2048), (12, 6145)])
def test_g1_msm_a_few_over_a_table_take_the_workgroup_scan(ctx, srs_bytes, bits, n):
    """1 .. 33 MSMs over a window table in one call — RingVRF.prove of ONE proof commits 4, 1 and 2 polynomials this way; up to 32 take the
    workgroup-scan reduction with a host fold per index group, 33 the chunk kernels: every result equals the oracle's, incl. a zero
    vector, an all-equal vector, scalars >= r and a vector whose scalars are +-1 (most buckets empty)."""
    rng = random.Random(bits * 7 + n)
    srs = ctx.srs_load(srs_bytes[: 96 * n]).precompute(bits)
    vecs = [b"".join(rng.randrange(coracle.FR_P).to_bytes(32, "little") for _ in range(n)) for _ in range(5)]
    vecs.append(bytes(32 * n))
    vecs.append((0x1234567890ABCDEF1234567890ABCDEF).to_bytes(32, "little") * n)
    vecs.append(b"".join(v.to_bytes(32, "little") for v in ([1, coracle.FR_P - 1, (1 << 256) - 1, 2**255, coracle.FR_P] * n)[:n]))
    vecs.append(b"".join((1 if rng.random() < 0.5 else coracle.FR_P - 1).to_bytes(32, "little") for _ in range(n)))
    want = [_oracle_msm_be(srs_bytes, v, n) for v in vecs]
    for batch in (1, 2, 3, 5, 8, 9, 20, 32, 33):
        ks = b"".join(vecs[(batch + i) % len(vecs)] for i in range(batch))
        got = ctx.g1_msm_batch(srs, ks, n)
        assert got == [want[(batch + i) % len(vecs)] for i in range(batch)], batch
    srs.close()


@pytest.mark.parametrize("n", [300, 1000, 3000, 7172, 20000, 50000])
def test_g1_msm_one_plain_call_takes_the_workgroup_scan(ctx, n):
    """one MSM over plain bases (no table): the window width follows the size (256 .. 4096 buckets per window), every window is reduced by
    a few workgroup scans and folded on the host — the closed form [sum k_i (1 + i)] G over synthetic bases, a prefix against the
    oracle's Pippenger, and degenerate vectors (all equal, r - 1 everywhere, zero)."""
    import bench

    srs = ctx.srs_synthetic(bench.G1_BE, n, first=1)
    vals, raw = bench.seeded_scalars(n, b"plain%d" % n)
    le_g = bench.be_to_le_points(bench.G1_BE)

    def closed(values):
        e = sum(k * (1 + i) for i, k in enumerate(values)) % coracle.FR_P
        if e == 0:
            return None
        w = bytes(coracle.g1_msm_raw(le_g, e.to_bytes(32, "little"), 1))
        return w[:48][::-1] + w[48:][::-1]

    assert ctx.g1_msm(srs, raw) == closed(vals)
    m = min(n, 2000)
    cpu = bytes(coracle.g1_msm_raw(bench.be_to_le_points(srs.download(0, m)), raw[: 32 * m], m))
    assert ctx.g1_msm(srs, raw[: 32 * m]) == cpu[:48][::-1] + cpu[48:][::-1]
    for values in ([0x0F0E0D0C0B0A09080706050403020100] * n, [coracle.FR_P - 1] * n, [0] * n):
        assert ctx.g1_msm(srs, b"".join(v.to_bytes(32, "little") for v in values)) == closed(values)
    srs.close()


@pytest.mark.parametrize("bits", [7, 12, 16])
def test_g1_msm_fixed_base_table_matches_plain(ctx, srs_bytes, bits):
    """dr_srs_precompute: one bucket set per MSM over the window table — identical results, single and batched."""
    rng = random.Random(bits)
    plain = ctx.srs_load(srs_bytes[: 96 * 1500])
    tabled = ctx.srs_load(srs_bytes[: 96 * 1500]).precompute(bits)
    for n, offset in ((1, 0), (2, 7), (333, 100), (1500, 0)):
        ks = b"".join(rng.randrange(coracle.FR_P).to_bytes(32, "little") for _ in range(n))
        want = _oracle_msm_be(srs_bytes[96 * offset :], ks, n)
        assert ctx.g1_msm(tabled, ks, offset=offset) == want == ctx.g1_msm(plain, ks, offset=offset)
    n, batch = 256, 4
    ks = b"".join(rng.randrange(coracle.FR_P).to_bytes(32, "little") for _ in range(n * (batch - 1))) + bytes(32 * n)
    assert ctx.g1_msm_batch(tabled, ks, n) == ctx.g1_msm_batch(plain, ks, n)
    vals = [1, 0, coracle.FR_P - 1, (1 << 256) - 1, 2**255, 5]
    ks = b"".join(v.to_bytes(32, "little") for v in vals)
    assert ctx.g1_msm(tabled, ks) == ctx.g1_msm(plain, ks)
    tabled.precompute(0)                                   # drop the table again
    assert ctx.g1_msm(tabled, ks) == ctx.g1_msm(plain, ks)
    plain.close()
    tabled.close()


@pytest.mark.parametrize("bits,n,batch,width", [(9, 96, 8192, 10), (12, 1500, 1024, 13), (12, 5000, 300, 13)])
def test_g1_msm_batched_odd_multiple_buckets(ctx, srs_bytes, bits, n, batch, width):
    """A small SRS gets a table with a row per bit, and a batch of hundreds of MSMs over it recodes its scalars in width-w
    non-adjacent form (odd digits at arbitrary bit positions into odd-multiple buckets, dr_srs_table_info): the very same scalar
    vectors in batches of 64 take the window rows — the results must agree, and a sample of them with the oracle.  Edge vectors:
    powers of two (every row), r - 1, all equal, all zero, 2^c - 1 patterns, runs of ones that carry across every slot, alternating
    bits, values whose top digit lands on the last rows.  (96 points: the plain LDS sort; 1500 and 5000: the staged one; the width is the
    planner's choice for the size — tiling_for: n x digits + 1.5 addition-equivalents per bucket.)"""
    rng = random.Random(bits * 1000 + n)
    tabled = ctx.srs_load(srs_bytes[: 96 * n]).precompute(bits)
    info = tabled.table_info(n, batch)
    assert info["window_bits"] == bits and info["rows"] == 256
    assert info["tiling"] == "non-adjacent form" and info["tiling_bits"] == width
    assert info["batched_windows"] == -(-256 // width) and 0.3 < info["digits_per_scalar"] - 256 / (width + 1) < 0.8
    assert not tabled.table_info(n, 64)["odd_buckets"]                         # few MSMs: the window rows
    r = coracle.FR_P
    vecs = []
    for b in range(batch):
        if b == 0:
            v = [0] * n
        elif b == 1:
            v = [r - 1] * n
        elif b == 2:
            v = [(1 << (i % 255)) % r for i in range(n)]                       # single bits: every row
        elif b == 3:
            v = [((1 << (bits + 1)) - 1) << ((bits + 1) * (i % 16)) for i in range(n)]
        elif b == 4:
            v = [5] * n                                                        # one long list
        elif b == 5:
            v = [(1 << 255) % r if i % 2 else ((1 << (bits)) << (7 * (i % 30))) % r for i in range(n)]
        elif b == 6:
            v = [((1 << (200 + i % 55)) - 1 - (i % 7)) % r for i in range(n)]  # long runs of ones: the carry crosses every slot
        elif b == 7:
            v = [(int("5" * 64, 16) >> (i % 9)) % r for i in range(n)]         # alternating bits
        elif b == 8:
            v = [(int("a" * 64, 16) >> (i % 5)) % r for i in range(n)]
        elif b == 9:
            v = [r - 1 - i if i % 2 else (r >> 1) + i for i in range(n)]       # top digits on the last rows, (r - 1) / 2 and neighbours
        elif b == 10:
            v = [((1 << width) - 1) * (1 << (i % 240)) % r for i in range(n)]  # 2^w - 1 at every offset of every slot
        elif b == 11:
            v = [(((1 << (width - 1)) + (i % 3) - 1) << (i % 241)) % r for i in range(n)]   # around the sign threshold of a digit
        else:
            v = [rng.randrange(r) for _ in range(n)]
        vecs.append(b"".join(x.to_bytes(32, "little") for x in v))
    ctx.prof_reset()
    ctx.prof_enable(True)
    got = ctx.g1_msm_batch(tabled, b"".join(vecs), n)
    ctx.prof_enable(False)
    assert len(got) == batch
    for lo in range(0, batch, 64 * 16):                                        # window rows: 64 vectors per call, every 16th chunk
        assert ctx.g1_msm_batch(tabled, b"".join(vecs[lo : lo + 64]), n) == got[lo : lo + 64], lo
    for b in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, batch - 1):
        assert got[b] == _oracle_msm_be(srs_bytes, vecs[b], n), b
    tabled.close()


# ------------------------------------------------------------------ seam C
@pytest.mark.parametrize("log2n", [1, 2, 5, 9, 10, 11, 13, 14])
def test_ntt_matches_oracle(ctx, log2n):
    from oracle.pyref.ring import ROOT_OF_UNITY_2048, _sqrt_mod_prime

    n = 1 << log2n
    root, size = ROOT_OF_UNITY_2048, 2048
    while size < n:
        root, size = _sqrt_mod_prime(root), size * 2
    omega = pow(root, size // n, P)
    rng = random.Random(log2n)
    batch = 3
    data = b"".join(rng.randrange(P).to_bytes(32, "little") for _ in range(n * batch))
    fwd = ctx.ntt(data, log2n, omega)
    for b in range(batch):
        assert fwd[32 * n * b : 32 * n * (b + 1)] == coracle.ntt_raw(data[32 * n * b : 32 * n * (b + 1)], n, omega)
    back = ctx.ntt(fwd, log2n, pow(omega, -1, P), pow(n, -1, P))
    assert back == data


def test_ntt_matches_the_reference_kernel_fixtures(ctx, golden_dir):
    """dr_ntt against outputs of the REFERENCE'S OWN NTT kernel (tests/golden/ntt/ntt_cases.json, written by oracle/gen_ntt_fixtures.py
    from ntt.pyx + scalar.pyx over bls12_381_scalar.c): n = 2 ... 16384, forward, inverse with scale 1/n, an arbitrary scale; three
    transforms per call so the batched path runs too."""
    import hashlib
    import json

    import os

    from oracle.gen_ntt_fixtures import seeded_inputs as _ntt_fixture_inputs

    with open(os.path.join(golden_dir, "ntt", "ntt_cases.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) == 42
    for case in cases:
        n = 1 << case["log2n"]
        raw = b"".join(v.to_bytes(32, "little") for v in _ntt_fixture_inputs(n, case["input_tag"]))
        assert hashlib.sha256(raw).hexdigest() == case["input_sha256"]
        scale = None if case["scale"] is None else int(case["scale"], 16)
        out = ctx.ntt(raw * 3, case["log2n"], int(case["omega"], 16), scale)
        for b in range(3):
            assert hashlib.sha256(out[32 * n * b : 32 * n * (b + 1)]).hexdigest() == case["output_sha256"], (case["log2n"], case["kind"], b)


def test_bsn_decode_points_matches_oracle(ctx):
    """dr_bsn_decode_points (decompression + subgroup check on the GPU) against the oracle's dec_point on valid keys,
    random strings (about half are not x-coordinates of anything, most of the rest have a torsion component),
    the identity, the order-2 point, non-canonical y and both sign bits."""
    rng = random.Random(2024)
    encs = [bsn.enc_point(coracle.te_mul(bsn.G, rng.randrange(1, bsn.N))) for _ in range(70)]
    encs += [bytes(rng.randrange(256) for _ in range(32)) for _ in range(120)]
    encs += [(1).to_bytes(32, "little"), (bsn.P - 1).to_bytes(32, "little"), (bsn.P + 3).to_bytes(32, "little"),
             (0).to_bytes(32, "little"), bytes(31) + b"\x80"]
    encs += [e[:31] + bytes([e[31] ^ 0x80]) for e in encs[:10]]          # the other root: valid too (it is -P)
    raw, ok = ctx.bsn_decode_points(b"".join(encs))
    n_valid = 0
    for i, e in enumerate(encs):
        try:
            want = bsn.dec_point(e)
        except ValueError:
            want = None
        assert bool(ok[i]) == (want is not None), (i, e.hex())
        if want is not None:
            n_valid += 1
            assert (int.from_bytes(raw[64 * i : 64 * i + 32], "little"), int.from_bytes(raw[64 * i + 32 : 64 * i + 64], "little")) == want
    assert n_valid >= 80
    assert ctx.bsn_decode_points(b"") == (b"", b"")


def test_bsn_scalar_mul_large_batch_two_bit_window_kernel(ctx):
    """From 32768 scalar multiplications per launch the 2-bit-window kernel (k_bsn_scalar_mul_w2, 16 KiB of LDS per wave)
    takes over: same results as the 4-bit kernel on the same inputs and as the oracle on a sample, incl. edge scalars."""
    rng = random.Random(77)
    n = 40000
    base = [coracle.te_mul(bsn.G, rng.randrange(1, bsn.N)) for _ in range(50)]
    ks_small = [rng.randrange(bsn.N) for _ in range(50)]
    edge = [0, 1, 2, 3, bsn.N - 1, bsn.N - 2, bsn.N, (1 << 253) - 1, (1 << 256) - 1, 1 << 252, 0x5555555555555555555555555555555555555555555555555555555555555555 % (1 << 253),
            0xAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA]
    ks_small[: len(edge)] = edge
    pts = coracle.te_pack(base) * (n // 50)
    ks = b"".join(k.to_bytes(32, "little") for k in ks_small) * (n // 50)
    got = ctx.bsn_scalar_mul_batch(pts, ks)
    want = coracle.te_mul_batch_raw(coracle.te_pack(base), b"".join((k % (1 << 256)).to_bytes(32, "little") for k in ks_small), 50, glv=True)
    assert got[: 64 * 50] == bytes(want)
    assert got[64 * 50 * 3 : 64 * 50 * 4] == bytes(want)
    assert got[-64 * 50 :] == bytes(want)
    small = ctx.bsn_scalar_mul_batch(pts[: 64 * 50], ks[: 32 * 50])            # 4-bit kernel
    assert small == got[: 64 * 50]


# ------------------------------------------------------------------ seam A on JubJub (a = -1, cofactor 8)
def _jub_points(rng, count):
    return [bsn.mul_py(bsn.G, rng.randrange(1, bsn.N)) for _ in range(count)]


def test_jubjub_scalar_mul_msm_and_groups_match_oracle(ctx):
    """dr_te_scalar_mul_batch / dr_te_msm / dr_te_msm_groups with DR_CURVE_JUBJUB against the oracle's affine double-and-add
    (bandersnatch.py under using(JUBJUB)), incl. the edge scalars 0, 1, n-1, n, 2^256-1 (17 subtractions of n)."""
    rng = random.Random(4242)
    with bsn.using(bsn.JUBJUB):
        order = bsn.N
        n = 70
        pts = _jub_points(rng, n)
        ks = [rng.randrange(1 << 256) for _ in range(n)]
        ks[:8] = [0, 1, order - 1, order, (1 << 256) - 1, 2, 8, 17 * order - 1]
        got = coracle.te_unpack(ctx.bsn_scalar_mul_batch(coracle.te_pack(pts), b"".join(k.to_bytes(32, "little") for k in ks), 1))
        want = [bsn.mul(p, k) for p, k in zip(pts, ks)]
        assert got == want
        for m in (1, 5, 64, 70):
            msm = coracle.te_unpack(ctx.bsn_msm(coracle.te_pack(pts[:m]), coracle.scalars_pack([k % order for k in ks[:m]]), 1))[0]
            acc = bsn.IDENTITY
            for w in want[:m]:
                acc = bsn._te_add_ref(acc, w)
            assert msm == acc
        groups = coracle.te_unpack(ctx.bsn_msm_groups(coracle.te_pack(pts[:69]), coracle.scalars_pack([k % order for k in ks[:69]]), 3, 1))
        assert groups == [bsn._te_add_ref(bsn._te_add_ref(want[3 * g], want[3 * g + 1]), want[3 * g + 2]) for g in range(23)]
        # 280 terms: the bucket method (K4) on the a = -1 curve, unreduced scalars on input
        big_p = [pts[i % n] for i in range(280)]
        big_k = [rng.randrange(1 << 256) for _ in range(280)]
        msm = coracle.te_unpack(ctx.bsn_msm(coracle.te_pack(big_p), b"".join(k.to_bytes(32, "little") for k in big_k), 1))[0]
        total = {}
        for p_, k_ in zip(big_p, big_k):
            total[p_] = (total.get(p_, 0) + k_) % order
        acc = bsn.IDENTITY
        for p_, k_ in total.items():
            acc = bsn._te_add_ref(acc, bsn.mul(p_, k_))
        assert msm == acc
    # the default curve is untouched by the template parameter
    assert coracle.te_unpack(ctx.bsn_scalar_mul_batch(coracle.te_pack([bsn.G]), (3).to_bytes(32, "little")))[0] == coracle.te_mul(bsn.G, 3)


def test_jubjub_decode_points_and_try_and_increment_match_oracle(ctx):
    """dr_te_decode_points on JubJub (cofactor-8 subgroup check without an endomorphism) and dr_encode_to_curve_batch
    (try-and-increment: several counters per launch) against the oracle, incl. inputs whose first candidates fail."""
    import dot_ring_amd as d

    rng = random.Random(99)
    with bsn.using(bsn.JUBJUB):
        encs = [bsn.enc_point(p) for p in _jub_points(rng, 40)]
        encs += [bytes(rng.randrange(256) for _ in range(32)) for _ in range(120)]
        encs += [(1).to_bytes(32, "little"), (bsn.P - 1).to_bytes(32, "little"), (bsn.P + 3).to_bytes(32, "little"), bytes(32), bytes(31) + b"\x80"]
        encs += [e[:31] + bytes([e[31] ^ 0x80]) for e in encs[:10]]
        raw, ok = ctx.bsn_decode_points(b"".join(encs), 1)
        n_valid = 0
        for i, e in enumerate(encs):
            try:
                want = bsn.dec_point(e)
            except ValueError:
                want = None
            assert bool(ok[i]) == (want is not None), (i, e.hex())
            if want is not None:
                n_valid += 1
                assert coracle.te_unpack(raw[64 * i : 64 * i + 64])[0] == want
        assert n_valid >= 50
        msgs = [b"", b"foo"] + [bytes(rng.randrange(256) for _ in range(rng.randrange(0, 200))) for _ in range(60)]
        salts = [b""] * 32 + [bytes(rng.randrange(256) for _ in range(rng.randrange(0, 40))) for _ in range(30)]
        got = d.JubJub.point_type.encode_to_curve_batch(msgs, salts)
        want = [bsn.encode_to_curve(bsn.JUBJUB, m, s) for m, s in zip(msgs, salts)]
        assert [(p.x, p.y) for p in got] == want
        assert d.JubJub.point_type.encode_to_curve(b"foo") == got[1]
        assert d.JubJub.point_type.encode_to_curve_batch([]) == []
