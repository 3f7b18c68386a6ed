"""Bandersnatch variable-base scalar multiplication throughput by batch size (python tools/bsn_sweep.py)."""
import os, sys, time, hashlib
sys.path.insert(0, os.getcwd())
from dot_ring_amd import _native
ctx = _native.Context(0)
G = bytes.fromhex("") if False else None
import dot_ring_amd as d
g = d.Bandersnatch.point_type.generator_point()
gxy = g.x.to_bytes(32, "little") + g.y.to_bytes(32, "little")
N = 0x1CFB69D4CA675F520CCE760202687600FF8F87007419047174FD06B52876E7E1
for n in (4096, 16384, 65536, 262144, 1048576):
    ks = b"".join((int.from_bytes(hashlib.sha256(b"k%d" % i).digest(), "little") % N).to_bytes(32, "little") for i in range(min(n, 4096))) * (n // min(n, 4096))
    pts = ctx.bsn_scalar_mul_batch(gxy * n, ks)            # distinct valid points as inputs
    ctx.bsn_scalar_mul_batch(pts, ks)
    ctx.prof_reset(); ctx.prof_enable(True)
    t = time.perf_counter(); ctx.bsn_scalar_mul_batch(pts, ks); dt = time.perf_counter() - t
    ctx.prof_enable(False)
    km = ctx.prof_get("k_bsn_scalar_mul")[0]
    print(f"n={n}: kernel {km:.2f} ms = {n/km/1e3:.2f} M scalar-mults/s ; call {dt*1e3:.2f} ms (host checks + PCIe of {n*160/1e6:.1f} MB)")
