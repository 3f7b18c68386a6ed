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
B = 1024
al = [b"a" + i.to_bytes(8, "little") for i in range(B)]
for interval in (0.005, 0.0005, 0.0001):
    sys.setswitchinterval(interval)
    for pipe in (1, 2, 4):
        vrf.prove_batch(al, al, [sk] * B, [pk] * B, ring, root, pipeline=pipe)
        ts = []
        for _ in range(3):
            t = time.perf_counter(); vrf.prove_batch(al, al, [sk] * B, [pk] * B, ring, root, pipeline=pipe); ts.append(time.perf_counter() - t)
        print(f"switch={interval} pipeline={pipe}: {min(ts)*1e3:.1f} ms -> {B/min(ts):.0f} proofs/s", flush=True)
