"""Single-proof latencies and Tiny / Thin / Pedersen batch throughput (python tools/latency_test.py)."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import prove_sweep, dot_ring_amd as d
from dot_ring_amd.curve import scalar_mul_batch
from dot_ring_amd.vrf.primitives import secret_from_seed_scalar
cv = d.Bandersnatch
pk, sk = cv.secret_from_seed(prove_sweep.seed("signer", 0, 0))
sks = [secret_from_seed_scalar(cv, prove_sweep.seed("ring-member", 0, i)) for i in range(1024)]
keys = [p.point_to_string() for p in scalar_mul_batch([cv.point_type.generator_point()] * 1024, sks)]
keys[3] = pk
ring = d.Ring(keys); root = d.RingRoot.from_ring(ring)
def best(f, reps=5):
    f(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3, r
ms, pr = best(lambda: d.RingVRF[cv].prove(b"alpha", b"ad", sk, pk, ring, root))
print(f"RingVRF.prove  (ring 1024, one proof): {ms:.1f} ms")
ms, ok = best(lambda: pr.verify(b"alpha", b"ad", ring, root)); print(f"RingVRF.verify (one proof): {ms:.1f} ms ok={ok}")
ms, dec = best(lambda: d.RingVRF[cv].decode(pr.encode())); print(f"RingVRF.decode: {ms:.1f} ms")
for name in ("TinyVRF", "ThinVRF", "PedersenVRF"):
    vrf = getattr(d, name)[cv]
    B = 4096
    al = [b"a" + i.to_bytes(4, "little") for i in range(B)]
    ms, proofs = best(lambda: vrf.prove_batch(al, [sk] * B, al), 2)
    line = f"{name}.prove_batch({B}): {ms:.1f} ms = {B / ms * 1e3:.0f} proofs/s"
    if name == "PedersenVRF":
        ms2, ok = best(lambda: vrf.batch_verify(proofs, al, al), 2)
    elif name == "ThinVRF":
        ms2, ok = best(lambda: vrf.batch_verify(proofs, [pk] * B, al, al), 2)
    else:
        print(line); continue
    print(line + f" ; batch_verify: {ms2:.1f} ms = {B / ms2 * 1e3:.0f} proofs/s ok={ok}")
