"""Per-kernel breakdown of RingVRF.batch_verify (native path) at batch B: python tools/verify_profile.py [B]"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import prove_sweep, dot_ring_amd as d
from dot_ring_amd import runtime
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
pr = vrf.prove_batch(al, al, [sk] * B, [pk] * B, ring, root)
assert vrf.batch_verify(pr, al, al, ring, root)
ctx = runtime.context()
names = ("k_bsn_decode_points", "k_g1_decompress", "k_bsn_encode_to_curve", "k_bsn_scalar_mul", "k_bsn_msm_groups", "k_g1_digits", "k_scan", "k_g1_scatter",
         "k_g1_sort_sets", "k_size_sort", "k_g1_accumulate", "k_g1_reduce_chunks", "k_g1_reduce_windows", "k_g1_results_affine")
for prof in (False, True):
    ctx.prof_reset(); ctx.prof_enable(prof)
    t = time.perf_counter(); ok = vrf.batch_verify(pr, al, al, ring, root); dt = time.perf_counter() - t
    ctx.prof_enable(False)
    print(f"batch_verify({B}) ok={ok} {dt*1e3:.1f} ms" + ("  kernels: " + " ".join(f"{n[2:]}={ctx.prof_get(n)[0]:.2f}" for n in names if ctx.prof_get(n)[1]) if prof else ""))
