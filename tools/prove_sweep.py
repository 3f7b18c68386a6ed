#!/usr/bin/env python3
"""Time RingVRF.prove_batch / verify on the GPU for a given ring size and batch; print per-kernel breakdown."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dot_ring_amd as d
from dot_ring_amd import runtime

def seed(*parts):
    h = hashlib.sha256()
    for p in parts:
        h.update(p if isinstance(p, bytes) else (p.to_bytes(8, "little") if isinstance(p, int) else p.encode()))
        h.update(b"\0")
    return h.digest()

def main(ring_size, batch, reps=2, curve="Bandersnatch"):
    cv = getattr(d, curve)
    vrf = d.RingVRF[cv]
    t0 = time.perf_counter()
    signer_pk, signer_sk = cv.secret_from_seed(seed("signer", 0, 0))
    from dot_ring_amd.curve import scalar_mul_batch
    from dot_ring_amd.vrf.primitives import secret_from_seed_scalar
    sks = [secret_from_seed_scalar(cv, seed("ring-member", 0, i)) for i in range(ring_size)]
    pts = scalar_mul_batch([cv.point_type.generator_point()] * ring_size, sks)
    keys = [p.point_to_string() for p in pts]
    keys[min(3, ring_size - 1)] = signer_pk
    t1 = time.perf_counter()
    ring = d.Ring(keys, d.RingProofParams.from_ring_size(ring_size, cv=cv))
    root = d.RingRoot.from_ring(ring)
    t2 = time.perf_counter()
    print(f"ring {ring_size}: domain {ring.params.domain_size}; keygen {t1 - t0:.2f}s, Ring+RingRoot {t2 - t1:.3f}s", flush=True)
    alphas = [b"bench-batch-input" + i.to_bytes(8, "little") for i in range(batch)]
    ads = [b"bench-batch-ad" + i.to_bytes(8, "little") for i in range(batch)]
    ctx = runtime.context()
    proofs = vrf.prove_batch(alphas[:512], ads[:512], [signer_sk] * 512, [signer_pk] * 512, ring, root)   # warm-up (both pipeline workers)
    for r in range(reps):
        ctx.prof_reset(); ctx.prof_enable(True)
        t = time.perf_counter()
        proofs = vrf.prove_batch(alphas, ads, [signer_sk] * batch, [signer_pk] * batch, ring, root)
        dt = time.perf_counter() - t
        ctx.prof_enable(False)
        names = ["k_bsn_scalar_mul", "k_ring_chain", "k_ring_columns", "k_ntt_local", "k_ntt_strided", "k_ring_pad", "k_ring_constraints",
                 "k_ring_quotient", "k_ring_eval", "k_ring_linpoly", "k_ring_aggpoly", "k_syndiv", "k_bsn_encode_to_curve", "k_bsn_msm_groups", "k_g1_sort_sets", "k_size_sort", "k_g1_digits", "k_scan", "k_g1_scatter",
                 "k_g1_accumulate", "k_g1_comb_msm", "k_g1_reduce_chunks", "k_g1_reduce_windows", "k_g1_horner", "k_g1_results_affine"]
        parts = {k: ctx.prof_get(k)[0] for k in names}
        gpu_ms = sum(parts.values())
        print(f"batch {batch}: {dt * 1e3:.1f} ms -> {batch / dt:.1f} proofs/s ; GPU kernels {gpu_ms:.1f} ms | " +
              " ".join(f"{k[2:]}={v:.1f}" + (f"(x{ctx.prof_get(k)[1]})" if k in ("k_g1_comb_msm", "k_g1_accumulate") else "") for k, v in parts.items() if v > 0.05), flush=True)
    t = time.perf_counter()
    ok = proofs[0].verify(alphas[0], ads[0], ring, root)
    print(f"verify: {ok} in {(time.perf_counter() - t) * 1e3:.1f} ms; batch_verify(8): ", end="")
    t = time.perf_counter()
    ok = vrf.batch_verify(proofs[:8], alphas[:8], ads[:8], ring, root)
    print(ok, f"{(time.perf_counter() - t) * 1e3:.1f} ms", flush=True)
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    t = time.perf_counter()
    ok = vrf.batch_verify(proofs, alphas, ads, ring, root)
    dt = time.perf_counter() - t
    pr.disable()
    print(f"batch_verify({batch}): {ok} {dt * 1e3:.1f} ms -> {batch / dt:.1f} proofs/s", flush=True)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), curve=sys.argv[3] if len(sys.argv) > 3 else "Bandersnatch")
