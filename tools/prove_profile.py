#!/usr/bin/env python3
"""cProfile of RingVRF.prove_batch (host side) on the GPU box."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prove_sweep, dot_ring_amd as d
from dot_ring_amd.curve import scalar_mul_batch
from dot_ring_amd.vrf.primitives import secret_from_seed_scalar
cv = d.Bandersnatch; vrf = d.RingVRF[cv]
ring_size, batch = 1024, 1024
pk, sk = cv.secret_from_seed(prove_sweep.seed("signer", 0, 0))
sks = [secret_from_seed_scalar(cv, prove_sweep.seed("ring-member", 0, i)) for i in range(ring_size)]
keys = [p.point_to_string() for p in scalar_mul_batch([cv.point_type.generator_point()] * ring_size, sks)]
keys[3] = pk
ring = d.Ring(keys); root = d.RingRoot.from_ring(ring)
al = [b"a" + i.to_bytes(8, "little") for i in range(batch)]
vrf.prove_batch(al[:2], al[:2], [sk] * 2, [pk] * 2, ring, root)
pr = cProfile.Profile(); pr.enable()
vrf.prove_batch(al, al, [sk] * batch, [pk] * batch, ring, root)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
