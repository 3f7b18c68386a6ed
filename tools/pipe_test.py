import os, sys, time
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
al = [b"a" + i.to_bytes(8, "little") for i in range(B)]
for pipe in (1, 2, 3, 4, 8):
    vrf.prove_batch(al, al, [sk] * B, [pk] * B, ring, root, pipeline=pipe)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); pr = vrf.prove_batch(al, al, [sk] * B, [pk] * B, ring, root, pipeline=pipe); ts.append(time.perf_counter() - t)
    print(f"batch={B} pipeline={pipe}: {min(ts)*1e3:.1f} ms -> {B/min(ts):.0f} proofs/s", flush=True)
print("verify", vrf.batch_verify(pr[:16], al[:16], al[:16], ring, root))
