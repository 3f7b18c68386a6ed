#!/usr/bin/env python3
"""Sweep G1 MSM sizes / window widths on the GPU and print the per-kernel breakdown (hipEvent timing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dot_ring_amd import _native

KERNELS = ("k_g1_digits", "k_scan", "k_g1_scatter", "k_g1_part_scatter", "k_g1_part_sort", "k_g1_sort_sets", "k_size_sort", "k_g1_accumulate", "k_g1_reduce_chunks", "k_g1_reduce_windows", "k_g1_horner", "k_g1_results_affine")

def run(log2n, window=0, reps=2, table=0):
    if window:
        os.environ["DOTRING_MSM_WINDOW"] = str(window)
    else:
        os.environ.pop("DOTRING_MSM_WINDOW", None)
    ctx = _native.Context(0)
    n = 1 << log2n
    srs = ctx.srs_synthetic(bench.G1_BE, n)
    if table:
        srs.precompute(table)
    _, raw = bench.seeded_scalars(n, b"sweep")
    d = ctx.alloc(32 * n).upload(raw)
    ctx.g1_msm_dev(srs, d, n)
    ctx.prof_reset(); ctx.prof_enable(True)
    t = time.perf_counter()
    for _ in range(reps):
        ctx.g1_msm_dev(srs, d, n)
    dt = (time.perf_counter() - t) / reps
    ctx.prof_enable(False)
    parts = " ".join(f"{k[2:]}={ctx.prof_get(k)[0] / reps:.3f}" for k in KERNELS if ctx.prof_get(k)[1])
    print(f"log2n={log2n} c={window or 'auto'} table={table} total={dt * 1e3:.2f} ms  {n / dt / 1e6:.2f} Mpairs/s | {parts}", flush=True)
    d.free(); srs.close(); ctx.close()

if __name__ == "__main__":
    for spec in sys.argv[1:]:
        parts = spec.split(":")
        run(int(parts[0]), int(parts[1] or 0) if len(parts) > 1 else 0, table=int(parts[2]) if len(parts) > 2 else 0)
