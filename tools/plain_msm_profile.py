"""One plain G1 MSM (no table, batch 1: the batch verifier's folds, a single KZG.commit) at several sizes: wall time and per-kernel times.
   python3 tools/plain_msm_profile.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import bench
from dot_ring_amd import _native

FR = bench.FR
ctx = _native.Context(0)
for n in [int(x) for x in sys.argv[1:]] or [2048, 6145, 7172]:
    srs = ctx.srs_synthetic(bench.G1_BE, n, first=1)          # no precompute: plain bases
    import hashlib
    ks = b"".join((int.from_bytes(hashlib.sha256(b"k%d" % i).digest(), "little") % FR).to_bytes(32, "little") for i in range(n))
    d = ctx.alloc(32 * n).upload(ks)
    for _ in range(3):
        ctx.g1_msm_dev(srs, d, n)
    t = time.perf_counter()
    for _ in range(20):
        ctx.g1_msm_dev(srs, d, n)
    dt = (time.perf_counter() - t) / 20
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(5):
        ctx.g1_msm_dev(srs, d, n)
    ctx.prof_enable(False)
    kern = {k[2:]: round(ctx.prof_get(k)[0] / 5, 3) for k in bench.MSM_KERNELS if ctx.prof_get(k)[1]}
    print(f"n={n}: {dt * 1e3:.3f} ms per MSM; kernels {kern} (sum {sum(kern.values()):.3f})", flush=True)
    d.free(); srs.close()
ctx.close()
