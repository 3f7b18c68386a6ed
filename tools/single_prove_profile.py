"""Phase and kernel breakdown of a single RingVRF.prove: DOTRING_TRACE=1 python tools/single_prove_profile.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import dot_ring_amd as d
from dot_ring_amd import runtime
cv = d.Bandersnatch
sks = [(3000 + i).to_bytes(32, "little") for i in range(40)]
keys = [cv.public_key_from_secret(s) for s in sks]
params = d.RingProofParams.from_ring_size(1000)
ring = d.Ring(keys, params); root = d.RingRoot.from_ring(ring, params)
vrf = d.RingVRF[cv]
vrf.prove(b"a", b"b", sks[0], keys[0], ring, root)
ts = []
for i in range(8):
    t = time.perf_counter(); vrf.prove(b"a%d" % i, b"b", sks[0], keys[0], ring, root); ts.append(time.perf_counter() - t)
print(f"prove: min {min(ts)*1e3:.2f} ms, median {sorted(ts)[4]*1e3:.2f} ms", flush=True)
ctx = runtime.context()
ctx.prof_reset(); ctx.prof_enable(True)
vrf.prove(b"zz", b"b", sks[0], keys[0], ring, root)
ctx.prof_enable(False)
import bench
tot = 0
out = []
for n in bench.MSM_KERNELS + bench.RING_KERNELS:
    ms, cnt = ctx.prof_get(n)
    if cnt:
        out.append(f"{n[2:]}={ms:.3f}(x{cnt})"); tot += ms
print("kernels (sum %.2f ms):" % tot, " ".join(out))
