"""Comb-table MSM rate (G mixed additions/s) vs window width / table size: python tools/comb_sweep.py"""
import os, sys, time, random
sys.path.insert(0, os.getcwd())
from dot_ring_amd import _native
ctx = _native.Context(0)
blob = open("dot_ring_amd/data/bls12-381-srs-2-11-uncompressed-zcash.bin", "rb").read()
n, batch = 6145, 1024
srs_be = blob[8 : 8 + 96 * n]
rng = random.Random(5)
ks = rng.randbytes(32 * n * batch)
ks = bytes(b & 0x3F if (i % 32) == 31 else b for i, b in enumerate(ks)) if False else ks
d = ctx.alloc(len(ks)).upload(ks)
for bits, comb in ((12, False), (12, True), (10, True), (9, True), (8, True)):
    srs = ctx.srs_load(srs_be).precompute(bits)
    t0 = time.perf_counter()
    if comb:
        srs.precompute_comb()
    build = time.perf_counter() - t0
    ctx.g1_msm_batch_dev(srs, d, n, batch)
    t = time.perf_counter(); ctx.g1_msm_batch_dev(srs, d, n, batch); dt = time.perf_counter() - t
    W = -(-256 // bits)
    print(f"bits={bits} comb={comb} table={(n * W * (1 << (bits - 1)) * 96 / 2**30) if comb else 0:.1f} GiB build={build:.2f}s  msm batch: {dt*1e3:.1f} ms  {n*W*batch/dt/1e9:.2f} G adds/s", flush=True)
    srs.close()
