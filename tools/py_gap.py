"""How much of a bench step is spent outside the two native calls (python3 tools/py_gap.py [steps]): wall time of the step, of
RingProver.ringvrf_prove_batch and Context.ringvrf_verify_batch (wrapped), and of the pieces of the Python layer around them."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import bench
import dot_ring_amd as d
from dot_ring_amd import _native

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
acc = {"prove_native": 0.0, "verify_native": 0.0, "random_expand": 0.0, "from_batch": 0.0, "indices_of": 0.0}

def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[key] += time.perf_counter() - t
    setattr(obj, name, g)

wrap(_native.RingProver, "ringvrf_prove_batch", "prove_native")
wrap(_native.Context, "ringvrf_verify_batch", "verify_native")
wrap(_native, "random_expand", "random_expand")
w = bench.RingWorkload(d, 1024, 1024)
wrap(type(w.ring), "indices_of", "indices_of")
for _ in range(3):
    w.step()
for k in acc:
    acc[k] = 0.0
tp = tv = 0.0
t0 = time.perf_counter()
for _ in range(steps):
    _, ok, a, b = w._span(0, w.batch)
    tp += a; tv += b
total = time.perf_counter() - t0
ms = lambda x: x / steps * 1e3
print(f"step {ms(total):.2f} ms = prove_batch {ms(tp):.2f} (native {ms(acc['prove_native']):.2f}, random_expand {ms(acc['random_expand']):.2f}, "
      f"indices_of {ms(acc['indices_of']):.2f}) + batch_verify {ms(tv):.2f} (native {ms(acc['verify_native']):.2f}); "
      f"outside the native calls: {ms(total - acc['prove_native'] - acc['verify_native']):.2f} ms")
