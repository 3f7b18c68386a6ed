"""Experiment: one batch of 1024 proofs as 2 (or 4) concurrent sub-batches from separate host threads, each with its own
context / stream / prover state, against the single-call baseline."""
import os, sys, threading, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import prove_sweep, dot_ring_amd as d
from dot_ring_amd.curve import scalar_mul_batch
from dot_ring_amd.vrf.primitives import secret_from_seed_scalar

cv = d.Bandersnatch; vrf = d.RingVRF[cv]
pk, sk = cv.secret_from_seed(prove_sweep.seed("signer", 0, 0))
sks = [secret_from_seed_scalar(cv, prove_sweep.seed("ring-member", 0, i)) for i in range(1024)]
keys = [p.point_to_string() for p in scalar_mul_batch([cv.point_type.generator_point()] * 1024, sks)]
keys[3] = pk
ring = d.Ring(keys); root = d.RingRoot.from_ring(ring)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
al = [b"a" + i.to_bytes(8, "little") for i in range(B)]
vrf.prove_batch(al, al, [sk] * B, [pk] * B, ring, root)
t = time.perf_counter(); vrf.prove_batch(al, al, [sk] * B, [pk] * B, ring, root); base = time.perf_counter() - t
print(f"single call: {base*1e3:.1f} ms")
cuts = [B * i // parts for i in range(parts + 1)]
res = [None] * parts
start = threading.Barrier(parts)
rounds = 4
done = threading.Barrier(parts)
times = []
def work(k):
    lo, hi = cuts[k], cuts[k + 1]
    n = hi - lo
    for r in range(rounds):
        start.wait()
        t = time.perf_counter()
        res[k] = vrf.prove_batch(al[lo:hi], al[lo:hi], [sk] * n, [pk] * n, ring, root)
        done.wait()
        if k == 0:
            times.append(time.perf_counter() - t)
ths = [threading.Thread(target=work, args=(k,)) for k in range(1, parts)]
for th in ths: th.start()
work(0)
for th in ths: th.join()
print(f"{parts} concurrent parts (first round = per-thread setup):", " ".join(f"{x*1e3:.1f}" for x in times), "ms")
