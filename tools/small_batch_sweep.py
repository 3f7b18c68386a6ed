"""prove_batch + batch_verify at small batch sizes (the MSM paths change with the number of bucket sets): python3 tools/small_batch_sweep.py [batch ...]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import bench
import dot_ring_amd as d

for b in [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4, 8, 9, 16, 32, 64, 128, 255, 256]:
    w = bench.RingWorkload(d, 1024, b)
    for _ in range(3):
        w.step()
    reps = max(3, min(20, 400 // max(1, b)))
    tp = tv = 0.0
    ok = True
    for _ in range(reps):
        _, o, a, c = w._span(0, b)
        tp += a; tv += c; ok = ok and o
    print(f"batch {b:4d}: prove {tp / reps * 1e3:8.2f} ms ({b / (tp / reps):8.0f}/s)  verify {tv / reps * 1e3:6.2f} ms  all verified {ok}", flush=True)
