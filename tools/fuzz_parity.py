#!/usr/bin/env python3
"""One-off large differential run of the Bandersnatch / JubJub kernels against the oracle (beyond the sizes the test suite
uses): python tools/fuzz_parity.py [scale]   (scale 1 = about two minutes of oracle time).  Exit code 1 on any mismatch."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dot_ring_amd as d
from dot_ring_amd import runtime
from oracle import coracle
from oracle.pyref import bandersnatch as bsn

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
rng = random.Random(20260)
ctx = runtime.context()
bad = 0


def report(name, n, wrong, t0):
    global bad
    bad += wrong
    print(f"{name}: {n} cases, {wrong} mismatches, {time.perf_counter() - t0:.1f} s", flush=True)


# 1. Bandersnatch scalar multiplication (both kernels: GLV lane pairs below 16384 terms, 2-bit windows from 32768)
for n in (int(3000 * scale), int(40000 * scale)):
    t0 = time.perf_counter()
    base = [coracle.te_mul(bsn.G, rng.randrange(1, bsn.N)) for _ in range(200)]
    pts = [base[rng.randrange(200)] for _ in range(n)]
    ks = [rng.randrange(1 << 256) for _ in range(n)]
    got = ctx.bsn_scalar_mul_batch(coracle.te_pack(pts), b"".join(k.to_bytes(32, "little") for k in ks))
    want = coracle.te_mul_batch_raw(coracle.te_pack(pts), coracle.scalars_pack([k % bsn.N for k in ks]), n, glv=True)
    report(f"bsn scalar_mul n={n}", n, sum(got[64 * i : 64 * i + 64] != bytes(want[64 * i : 64 * i + 64]) for i in range(n)), t0)

# 2. Bandersnatch point decoding: random strings, valid keys, sign flips
t0 = time.perf_counter()
n = int(60000 * scale)
encs = [bytes(rng.randrange(256) for _ in range(32)) for _ in range(n)]
for i in range(0, n, 7):
    encs[i] = bsn.enc_point(coracle.te_mul(bsn.G, rng.randrange(1, bsn.N)))
raw, ok = ctx.bsn_decode_points(b"".join(encs))
wrong = 0
for i, e in enumerate(encs):
    try:
        want = bsn.dec_point(e)
    except ValueError:
        want = None
    if bool(ok[i]) != (want is not None) or (want is not None and coracle.te_unpack(raw[64 * i : 64 * i + 64])[0] != want):
        wrong += 1
report("bsn decode_points", n, wrong, t0)

# 3. Elligator 2 on both Bandersnatch suites
for cv, suite in ((d.Bandersnatch, bsn.SHA512), (d.Bandersnatch_SHAKE128, bsn.SHAKE128)):
    t0 = time.perf_counter()
    n = int(4000 * scale)
    msgs = [bytes(rng.randrange(256) for _ in range(rng.randrange(0, 90))) for _ in range(n)]
    got = cv.point_type.encode_to_curve_batch(msgs)
    report(f"elligator {cv.name}", n, sum((p.x, p.y) != bsn.encode_to_curve(suite, m) for p, m in zip(got, msgs)), t0)

# 4. JubJub: try-and-increment, decoding, scalar multiplication (the oracle's affine double-and-add is slow: smaller counts)
with bsn.using(bsn.JUBJUB):
    t0 = time.perf_counter()
    n = int(6000 * scale)
    msgs = [bytes(rng.randrange(256) for _ in range(rng.randrange(0, 90))) for _ in range(n)]
    got = d.JubJub.point_type.encode_to_curve_batch(msgs)
    report("jubjub try-and-increment", n, sum((p.x, p.y) != bsn.encode_to_curve(bsn.JUBJUB, m) for p, m in zip(got, msgs)), t0)
    t0 = time.perf_counter()
    n = int(1500 * scale)
    encs = [bytes(rng.randrange(256) for _ in range(32)) for _ in range(n)]
    for i in range(0, n, 5):
        encs[i] = bsn.enc_point(bsn.mul_py(bsn.G, rng.randrange(1, bsn.N)))
    raw, ok = ctx.bsn_decode_points(b"".join(encs), 1)
    wrong = 0
    for i, e in enumerate(encs):
        try:
            want = bsn.dec_point(e)
        except ValueError:
            want = None
        if bool(ok[i]) != (want is not None) or (want is not None and coracle.te_unpack(raw[64 * i : 64 * i + 64])[0] != want):
            wrong += 1
    report("jubjub decode_points", n, wrong, t0)
    t0 = time.perf_counter()
    n = int(1500 * scale)
    base = [bsn.mul_py(bsn.G, rng.randrange(1, bsn.N)) for _ in range(20)]
    pts = [base[rng.randrange(20)] for _ in range(n)]
    ks = [rng.randrange(1 << 256) for _ in range(n)]
    got = coracle.te_unpack(ctx.bsn_scalar_mul_batch(coracle.te_pack(pts), b"".join(k.to_bytes(32, "little") for k in ks), 1))
    report("jubjub scalar_mul", n, sum(g != bsn.mul(p, k) for g, p, k in zip(got, pts, ks)), t0)

# 5. whole Ring-VRF proofs (native batch prover) against the oracle prover: random ring sizes, signers, input lengths
from oracle.pyref import ring as oring

for cv, suite in ((d.Bandersnatch, bsn.SHA512), (d.Bandersnatch_SHAKE128, bsn.SHAKE128), (d.JubJub, bsn.JUBJUB)):
    with bsn.using(suite):
        t0 = time.perf_counter()
        wrong = total = 0
        for _ in range(max(1, int(2 * scale))):
            size = rng.choice([1, 2, 9, 100, 255, 256, 300, 700, 767, 768, 1500] if cv is not d.JubJub else [1, 2, 9, 100, 255, 256, 300, 700])
            sks = [rng.randrange(1, bsn.N).to_bytes(32, "little") for _ in range(size)]
            keys = [cv.public_key_from_secret(sk) for sk in sks]
            params = d.RingProofParams.from_ring_size(size, test_vectors=True, cv=cv)
            ring = d.Ring(keys, params)
            root = d.RingRoot.from_ring(ring, params)
            o_ring = oring.Ring(keys, oring.Params.from_ring_size(size, test_vectors=True, suite=suite))
            o_root = oring.RingRoot(o_ring)
            wrong += root.encode() != o_root.encode()
            who = [rng.randrange(size) for _ in range(4)]
            als = [bytes(rng.randrange(256) for _ in range(rng.randrange(0, 70))) for _ in who]
            ads = [bytes(rng.randrange(256) for _ in range(rng.randrange(0, 70))) for _ in who]
            proofs = d.RingVRF[cv].prove_batch(als, ads, [sks[w] for w in who], [keys[w] for w in who], ring, root)
            for pf, w, al, ad in zip(proofs, who, als, ads):
                wrong += pf.encode() != oring.ring_vrf_prove(o_ring, o_root, al, ad, sks[w])
                total += 1
            wrong += not d.RingVRF[cv].batch_verify(proofs, als, ads, ring, root)
        report(f"ring proofs {cv.name}", total, wrong, t0)

# 6. G1 MSMs of random shape (size, batch, plain bases / window tables of several widths, sparse and dense scalars)
srs_be = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dot_ring_amd", "data", "bls12-381-srs-2-11-uncompressed-zcash.bin"), "rb").read()[8 : 8 + 96 * 6145]
srs_le = b"".join(srs_be[96 * i : 96 * i + 48][::-1] + srs_be[96 * i + 48 : 96 * i + 96][::-1] for i in range(6145))
t0 = time.perf_counter()
wrong = total = 0
for table in (0, 9, 12, 14):
    srs = ctx.srs_load(srs_be)
    if table:
        srs.precompute(table)
    for _ in range(max(1, int(3 * scale))):
        n = rng.choice([1, 2, 3, 17, 64, 65, 300, 2047, 2048, 6145])
        batch = rng.choice([1, 1, 2, 5, 33])
        dense = rng.random() < 0.6
        ks = [[(rng.randrange(coracle.FR_P) if dense or rng.random() < 0.1 else rng.choice([0, 0, 0, 1])) for _ in range(n)] for _ in range(batch)]
        raw = b"".join(k.to_bytes(32, "little") for row in ks for k in row)
        got = ctx.g1_msm_batch(srs, raw, n) if batch > 1 else [ctx.g1_msm(srs, raw)]
        for row, g in zip(ks, got):
            w = coracle.g1_msm_raw(srs_le[: 96 * n], coracle.scalars_pack(row), n)
            w_be = None if w == bytes(96) else w[:48][::-1] + w[48:][::-1]
            wrong += g != w_be
            total += 1
    srs.close()
report("g1 msm shapes", total, wrong, t0)

# 6b. large batches over a bit-row table (non-adjacent form, odd-multiple buckets) against the same vectors in batches of 32
# (window rows) and a sample against the oracle; scalars dense, sparse, short, repeated
t0 = time.perf_counter()
wrong = total = 0
for table, n, batch in ((9, 50, 8192), (10, 333, 4096), (12, 2047, 1024), (13, 700, 1100)):
    if scale < 1 and n > 400:
        continue
    srs = ctx.srs_load(srs_be[: 96 * n]).precompute(table)
    assert srs.table_info(n, batch)["odd_buckets"]
    kinds = [rng.choice(["dense", "sparse", "short", "repeated"]) for _ in range(batch)]
    rows = []
    for kind in kinds:
        if kind == "dense":
            row = [rng.randrange(coracle.FR_P) for _ in range(n)]
        elif kind == "sparse":
            row = [rng.randrange(coracle.FR_P) if rng.random() < 0.1 else rng.choice([0, 0, 1, coracle.FR_P - 1]) for _ in range(n)]
        elif kind == "short":
            row = [rng.randrange(1 << rng.choice([1, 13, 14, 64, 200])) for _ in range(n)]
        else:
            vals = [rng.randrange(coracle.FR_P) for _ in range(2)] + [1 << rng.randrange(255)]
            row = [rng.choice(vals) for _ in range(n)]
        rows.append(b"".join(k.to_bytes(32, "little") for k in row))
    got = ctx.g1_msm_batch(srs, b"".join(rows), n)
    for lo in range(0, batch, 32 * 8):
        wrong += ctx.g1_msm_batch(srs, b"".join(rows[lo : lo + 32]), n) != got[lo : lo + 32]
        total += 1
    for b in rng.sample(range(batch), 6):
        w = coracle.g1_msm_raw(srs_le[: 96 * n], rows[b], n)
        wrong += got[b] != (None if w == bytes(96) else w[:48][::-1] + w[48:][::-1])
        total += 1
    srs.close()
report("g1 msm non-adjacent form over bit rows", total, wrong, t0)

# 7. Bandersnatch bucket Pippenger (K4, from 256 terms): random sizes, uniform / short / repeated scalars (heavy buckets)
t0 = time.perf_counter()
wrong = total = 0
base = [coracle.te_mul(bsn.G, rng.randrange(1, bsn.N)) for _ in range(300)]
for _ in range(max(2, int(4 * scale))):
    n = rng.choice([256, 257, 1000, 4095, 4096, 5122, 16383, 20482])
    kind = rng.choice(["uniform", "short", "repeated", "sparse"])
    pts = [base[rng.randrange(300)] for _ in range(n)]
    if kind == "uniform":
        ks = [rng.randrange(bsn.N) for _ in range(n)]
    elif kind == "short":
        ks = [rng.randrange(1 << rng.choice([1, 9, 64, 128, 129])) for _ in range(n)]
    elif kind == "repeated":
        vals = [rng.randrange(bsn.N) for _ in range(3)]
        ks = [rng.choice(vals) for _ in range(n)]
    else:
        ks = [rng.randrange(bsn.N) if rng.random() < 0.05 else 0 for _ in range(n)]
    got = coracle.te_unpack(ctx.bsn_msm(coracle.te_pack(pts), coracle.scalars_pack(ks)))[0]
    wrong += got != coracle.te_msm(pts, ks)
    total += 1
report("bsn pippenger shapes", total, wrong, t0)

# 8. fixed-base tables (generator, blinding base): full-width scalars
t0 = time.perf_counter()
n = int(2000 * scale)
ks = [rng.randrange(1 << 256) for _ in range(2 * n)]
raw = b"".join(k.to_bytes(32, "little") for k in ks)
got = coracle.te_unpack(ctx.te_fixed_base_msm_groups(coracle.te_pack([bsn.G, bsn.SHA512.blinding_base]), raw))
want = [bsn.add(coracle.te_mul(bsn.G, ks[2 * i] % bsn.N), coracle.te_mul(bsn.SHA512.blinding_base, ks[2 * i + 1] % bsn.N)) for i in range(n)]
report("fixed-base x*G + b*B", n, sum(g != w for g, w in zip(got, want)), t0)

# 9. round 3: the NTT on the unsaturated field — random sizes, batches, scales, forward and inverse roots — against the oracle's radix-2 network
t0 = time.perf_counter()
wrong = total = 0
root_2_32 = pow(7, (coracle.FR_P - 1) >> 32, coracle.FR_P)
for _ in range(max(4, int(12 * scale))):
    k = rng.choice([1, 2, 3, 6, 9, 10, 11, 12, 13, 14, 15])
    n, batch = 1 << k, rng.choice([1, 1, 2, 5])
    omega = pow(root_2_32, 1 << (32 - k), coracle.FR_P)
    if rng.random() < 0.5:
        omega = pow(omega, -1, coracle.FR_P)
    scale_by = rng.choice([None, pow(n, -1, coracle.FR_P), rng.randrange(1, coracle.FR_P)])
    rows = [[rng.choice([0, 1, coracle.FR_P - 1, rng.randrange(coracle.FR_P)]) if rng.random() < 0.2 else rng.randrange(coracle.FR_P) for _ in range(n)] for _ in range(batch)]
    data = b"".join(v.to_bytes(32, "little") for row in rows for v in row)
    got = ctx.ntt(data, k, omega, scale=scale_by)
    for b, row in enumerate(rows):
        want = coracle.ntt_raw(b"".join(v.to_bytes(32, "little") for v in row), n, omega)
        if scale_by is not None:
            want = b"".join((int.from_bytes(want[32 * i : 32 * i + 32], "little") * scale_by % coracle.FR_P).to_bytes(32, "little") for i in range(n))
        wrong += got[32 * n * b : 32 * n * (b + 1)] != bytes(want)
        total += 1
report("ntt shapes", total, wrong, t0)

# 10. round 3: the device-resident scalar multiplication (GLV decomposition in the kernel), full-width scalars
t0 = time.perf_counter()
n = int(3000 * scale)
base = [coracle.te_mul(bsn.G, rng.randrange(1, bsn.N)) for _ in range(200)]
pts = [base[rng.randrange(200)] for _ in range(n)]
ks = [rng.choice([0, 1, bsn.N - 1, bsn.N, (1 << 256) - 1]) if rng.random() < 0.02 else rng.randrange(1 << 256) for _ in range(n)]
raw_p, raw_k = coracle.te_pack(pts), b"".join(k.to_bytes(32, "little") for k in ks)
d_p, d_k, d_o = ctx.alloc(64 * n).upload(raw_p), ctx.alloc(32 * n).upload(raw_k), ctx.alloc(64 * n)
ctx.bsn_scalar_mul_batch_dev(d_p, d_k, n, d_o)
got = d_o.download()
want = coracle.te_mul_batch_raw(raw_p, coracle.scalars_pack([k % bsn.N for k in ks]), n, glv=True)
report("bsn scalar_mul device-resident (GLV split on device)", n, sum(got[64 * i : 64 * i + 64] != bytes(want[64 * i : 64 * i + 64]) for i in range(n)), t0)

print("FUZZ", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
