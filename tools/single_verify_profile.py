"""Where a single RingVRF.verify spends its time: python tools/single_verify_profile.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import prove_sweep, dot_ring_amd as d
from dot_ring_amd import runtime
cv = d.Bandersnatch
sks = [(3000 + i).to_bytes(32, "little") for i in range(40)]
keys = [cv.public_key_from_secret(s) for s in sks]
params = d.RingProofParams.from_ring_size(1000)
ring = d.Ring(keys, params); root = d.RingRoot.from_ring(ring, params)
pr = d.RingVRF[cv].prove(b"alpha", b"ad", sks[0], keys[0], ring, root)
assert pr.verify(b"alpha", b"ad", ring, root)
ctx = runtime.context()
ts = []
for _ in range(10):
    t = time.perf_counter(); pr.verify(b"alpha", b"ad", ring, root); ts.append(time.perf_counter() - t)
print(f"verify: min {min(ts)*1e3:.2f} ms, median {sorted(ts)[5]*1e3:.2f} ms")
vrf = d.RingVRF[cv]
vrf.batch_verify([pr], [b"alpha"], [b"ad"], ring, root)
ts = []
for _ in range(10):
    t = time.perf_counter(); ok = vrf.batch_verify([pr], [b"alpha"], [b"ad"], ring, root); ts.append(time.perf_counter() - t)
print(f"batch_verify([one]): min {min(ts)*1e3:.2f} ms, median {sorted(ts)[5]*1e3:.2f} ms ok={ok}")
dec = vrf.decode(pr.encode())
ts = []
for _ in range(10):
    t = time.perf_counter(); ok = vrf.batch_verify([dec], [b"alpha"], [b"ad"], ring, root); ts.append(time.perf_counter() - t)
print(f"batch_verify([decoded one]): min {min(ts)*1e3:.2f} ms ok={ok}")
ctx.prof_reset(); ctx.prof_enable(True)
pr.verify(b"alpha", b"ad", ring, root)
ctx.prof_enable(False)
names = ("k_bsn_decode_points", "k_g1_decompress", "k_bsn_encode_to_curve", "k_bsn_scalar_mul", "k_bsn_msm_groups", "k_g1_digits", "k_scan", "k_g1_scatter",
         "k_g1_sort_sets", "k_size_sort", "k_g1_accumulate", "k_g1_reduce_chunks", "k_g1_reduce_windows", "k_g1_results_affine", "k_g1_horner")
print("kernels:", " ".join(f"{n[2:]}={ctx.prof_get(n)[0]:.3f}(x{ctx.prof_get(n)[1]})" for n in names if ctx.prof_get(n)[1]))
p = cProfile.Profile(); p.enable()
for _ in range(5): pr.verify(b"alpha", b"ad", ring, root)
p.disable(); pstats.Stats(p).sort_stats("tottime").print_stats(8)
