"""Host-side timeline of one bench step (prove_batch + batch_verify, ring 1024, 1024 proofs): the library's phase trace
(DOTRING_TRACE=1 on stderr) plus wall-clock marks around the two Python calls, to see where the GPU waits for the host.
    DOTRING_TRACE=1 python tools/host_timeline.py 2> gpurun_out/host_timeline.txt"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import bench, dot_ring_amd as d

w = bench.RingWorkload(d, 1024, 1024)
for _ in range(3):
    w.step()
sys.stderr.write("==== measured steps\n")
for _ in range(3):
    t0 = time.perf_counter()
    proofs = w.vrf.prove_batch(w.alphas, w.ads, w.sks, w.pks, w.ring, w.root)
    t1 = time.perf_counter()
    ok = w.vrf.batch_verify(proofs, w.alphas, w.ads, w.ring, w.root)
    t2 = time.perf_counter()
    sys.stderr.write(f"PY prove_batch {1e3 * (t1 - t0):.2f} ms, batch_verify {1e3 * (t2 - t1):.2f} ms, ok={ok}\n")
