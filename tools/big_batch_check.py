"""Batches above the per-call limits of the native entry points (4096 ring proofs, 65536 sigma proofs) are split by the Python layer: python tools/big_batch_check.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import dot_ring_amd as d
for cv in (d.Bandersnatch, d.JubJub):
    sks = [(900 + i).to_bytes(32, "little") for i in range(9)]
    keys = [cv.public_key_from_secret(sk) for sk in sks]
    params = d.RingProofParams.from_ring_size(9, cv=cv)
    ring = d.Ring(keys, params); root = d.RingRoot.from_ring(ring, params)
    n = 5003
    al = [b"x%d" % i for i in range(n)]
    t = time.time()
    pr = d.RingVRF[cv].prove_batch(al, al, [sks[i % 9] for i in range(n)], [keys[i % 9] for i in range(n)], ring, root)
    t1 = time.time()
    ok = d.RingVRF[cv].batch_verify(pr, al, al, ring, root)
    bad = d.RingVRF[cv].batch_verify(pr, al[1:] + al[:1], al, ring, root)
    print(cv.name, n, "prove %.2fs verify %.2fs" % (t1 - t, time.time() - t1), ok, bad, len(set(p.encode() for p in pr)))
    # Pedersen / Tiny big batches
    pp = d.PedersenVRF[cv].prove_batch(al, [sks[0]] * n, al)
    print(" pedersen", d.PedersenVRF[cv].batch_verify(pp, al, al), d.PedersenVRF[cv].batch_verify(pp, al, al[::-1]))
    tp = d.TinyVRF[cv].prove_batch(al, [sks[0]] * n, al)
    print(" tiny", all(p.verify(keys[0], a, a) for p, a in list(zip(tp, al))[:20]))
