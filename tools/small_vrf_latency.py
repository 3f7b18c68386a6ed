"""One-call latencies of the small VRFs (Tiny / Thin / Pedersen prove and verify of ONE proof): python3 tools/small_vrf_latency.py"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import prove_sweep, dot_ring_amd as d
cv = d.Bandersnatch
pk, sk = cv.secret_from_seed(prove_sweep.seed("signer", 0, 0))
def best(f, reps=20):
    f(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t)
    ts.sort()
    return ts[0] * 1e3, ts[len(ts) // 2] * 1e3, r
for name in ("TinyVRF", "ThinVRF", "PedersenVRF"):
    vrf = getattr(d, name)[cv]
    lo, med, pr = best(lambda: vrf.prove(b"alpha", sk, b"ad"))
    print(f"{name}.prove: min {lo:.2f} ms, median {med:.2f} ms", flush=True)
    try:
        if name == "PedersenVRF":
            lo, med, ok = best(lambda: pr.verify(b"alpha", b"ad"))
        else:
            lo, med, ok = best(lambda: pr.verify(pk, b"alpha", b"ad"))
        print(f"{name}.verify: min {lo:.2f} ms, median {med:.2f} ms ok={ok}", flush=True)
    except Exception as e:
        print(f"{name}.verify: {type(e).__name__}: {e}")
