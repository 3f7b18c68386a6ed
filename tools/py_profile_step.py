"""Where the interpreter spends its time in one bench step (prove_batch + batch_verify, ring 1024, 1024 proofs): cProfile of the
calling thread over 5 steps, native calls show as ctypes entries.  Run on the GPU box:  python3 tools/py_profile_step.py"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.getcwd())
import bench
import dot_ring_amd as d

w = bench.RingWorkload(d, 1024, 1024)
for _ in range(3):
    w.step()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(5):
    w.step()
pr.disable()
print(f"5 steps under the profiler: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per step")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
