import os, sys, time, collections
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import prove_sweep, dot_ring_amd as d
from dot_ring_amd import _native
from dot_ring_amd.curve import scalar_mul_batch
from dot_ring_amd.vrf.primitives import secret_from_seed_scalar
acc = collections.defaultdict(float)
def wrap(cls, name):
    f = getattr(cls, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t; return r
    setattr(cls, name, g)
for n in ("witness", "quotient", "evals", "openings"): wrap(_native.RingProver, n)
for n in ("bsn_scalar_mul_batch", "bsn_msm_groups", "bsn_encode_to_curve_batch", "g1_msm_points", "bsn_msm"): wrap(_native.Context, n)
cv = d.Bandersnatch; vrf = d.RingVRF[cv]
pk, sk = cv.secret_from_seed(prove_sweep.seed("signer", 0, 0))
sks = [secret_from_seed_scalar(cv, prove_sweep.seed("ring-member", 0, i)) for i in range(1024)]
keys = [p.point_to_string() for p in scalar_mul_batch([cv.point_type.generator_point()] * 1024, sks)]
keys[3] = pk
ring = d.Ring(keys); root = d.RingRoot.from_ring(ring)
B = 1024
al = [b"a" + i.to_bytes(8, "little") for i in range(B)]
vrf.prove_batch(al, al, [sk] * B, [pk] * B, ring, root, pipeline=1)
acc.clear()
t = time.perf_counter(); pr = vrf.prove_batch(al, al, [sk] * B, [pk] * B, ring, root, pipeline=1); tot = time.perf_counter() - t
print(f"prove total {tot*1e3:.1f} ms; inside GPU calls {sum(acc.values())*1e3:.1f} ms:", {k: round(v*1e3, 1) for k, v in acc.items()})
acc.clear()
t = time.perf_counter(); ok = vrf.batch_verify(pr, al, al, ring, root); tot = time.perf_counter() - t
print(f"verify total {tot*1e3:.1f} ms ok={ok}; inside GPU/native calls {sum(acc.values())*1e3:.1f} ms:", {k: round(v*1e3, 1) for k, v in acc.items()})
