"""cProfile of the host side of prove_batch / batch_verify (run on the GPU box): python tools/host_profile.py [prove|verify]"""
import cProfile, os, pstats, sys
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
pr = vrf.prove_batch(al, al, [sk] * B, [pk] * B, ring, root, pipeline=1)
assert vrf.batch_verify(pr, al, al, ring, root)
which = sys.argv[1] if len(sys.argv) > 1 else "verify"
prof = cProfile.Profile()
prof.enable()
if which == "prove":
    vrf.prove_batch(al, al, [sk] * B, [pk] * B, ring, root, pipeline=1)
else:
    vrf.batch_verify(pr, al, al, ring, root)
prof.disable()
pstats.Stats(prof).sort_stats("tottime").print_stats(28)
