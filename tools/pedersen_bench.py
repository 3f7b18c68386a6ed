"""Pedersen / Thin batch_verify throughput (the (5B+2)-term Bandersnatch MSM is their dominant kernel) and the raw dr_bsn_msm time.
    python tools/pedersen_bench.py [B]            (DOTRING_BSN_PIPPENGER_FROM=0 selects the pre-K4 path: n full scalar multiplications)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dot_ring_amd as d
from dot_ring_amd import runtime

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cv = d.Bandersnatch
ped = d.PedersenVRF[cv]
sk = (1234567).to_bytes(32, "little")
alphas = [b"ped-bench" + i.to_bytes(4, "little") for i in range(B)]
ads = [b"ad"] * B
proofs = ped.prove_batch(alphas, [sk] * B, ads)
assert ped.batch_verify(proofs, alphas, ads)
for _ in range(2):
    ped.batch_verify(proofs, alphas, ads)
t = time.perf_counter()
reps = 5
for _ in range(reps):
    ok = ped.batch_verify(proofs, alphas, ads)
dt = (time.perf_counter() - t) / reps
print(f"PedersenVRF.batch_verify B={B}: {dt * 1e3:.2f} ms = {B / dt:,.0f} proofs/s ok={ok}")
t = time.perf_counter()
for _ in range(reps):
    ped.prove_batch(alphas, [sk] * B, ads)
dt = (time.perf_counter() - t) / reps
print(f"PedersenVRF.prove_batch   B={B}: {dt * 1e3:.2f} ms = {B / dt:,.0f} proofs/s")
# the MSM alone at 5B + 2 terms
import hashlib
from dot_ring_amd.curve import pack_points, scalar_mul_batch
n = 5 * B + 2
base = scalar_mul_batch([cv.point_type.generator_point()] * 256, list(range(2, 258)))
pts = pack_points([base[i % 256] for i in range(n)])
ks = b"".join(hashlib.sha256(i.to_bytes(4, "little")).digest()[:31] + b"\0" for i in range(n))
ctx = runtime.context()
ctx.bsn_msm(pts, ks)
t = time.perf_counter()
for _ in range(reps):
    ctx.bsn_msm(pts, ks)
print(f"dr_bsn_msm n={n}: {(time.perf_counter() - t) / reps * 1e3:.2f} ms")
ctx.prof_reset(); ctx.prof_enable(True)
t = time.perf_counter()
ctx.bsn_msm(pts, ks)
wall = time.perf_counter() - t
ctx.prof_enable(False)
print("one profiled call %.2f ms:" % (wall * 1e3), {k: round(ctx.prof_get(k)[0], 3) for k in ("k_te_msm_prepare", "k_g1_sort_sets", "k_size_sort", "k_te_msm_accumulate", "k_te_msm_reduce", "k_bsn_msm_groups")})
