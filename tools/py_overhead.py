"""Python-side cost of one prove_batch / batch_verify call at batch B (cProfile, top entries): python tools/py_overhead.py [B]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench, dot_ring_amd as d
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = bench.RingWorkload(d, 1024, B)
for _ in range(2):
    w.step()
for name, fn in (("prove_batch", lambda: w.vrf.prove_batch(w.alphas, w.ads, w.sks, w.pks, w.ring, w.root)),):
    t = time.perf_counter(); proofs = fn(); dt = time.perf_counter() - t
    p = cProfile.Profile(); p.enable(); proofs = fn(); p.disable()
    print(f"== {name}: {dt * 1e3:.2f} ms wall"); pstats.Stats(p).sort_stats("tottime").print_stats(12)
t = time.perf_counter(); ok = w.vrf.batch_verify(proofs, w.alphas, w.ads, w.ring, w.root); dt = time.perf_counter() - t
p = cProfile.Profile(); p.enable(); ok = w.vrf.batch_verify(proofs, w.alphas, w.ads, w.ring, w.root); p.disable()
print(f"== batch_verify: {dt * 1e3:.2f} ms wall ok={ok}"); pstats.Stats(p).sort_stats("tottime").print_stats(10)
