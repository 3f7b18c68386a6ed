"""ORACLE (test infrastructure only) — one worker of bench.py's all-cores CPU baseline: proves a slice of the benchmark's
proofs with the oracle prover and reports how long the proving took.
    python -m oracle.cpu_worker JOB.json FIRST COUNT
JOB.json: {"keys": [hex...], "ring_size": R, "signer_sk": hex, "alpha_prefix": hex, "ad_prefix": hex}."""
import hashlib
import json
import sys
import time


def main() -> int:
    from oracle.pyref import bandersnatch as obsn
    from oracle.pyref import ring as oring

    job = json.load(open(sys.argv[1]))
    first, count = int(sys.argv[2]), int(sys.argv[3])
    keys = [bytes.fromhex(k) for k in job["keys"]]
    params = oring.Params.from_ring_size(job["ring_size"], test_vectors=True, suite=obsn.SHA512)
    ring = oring.Ring(keys, params)
    root = oring.RingRoot(ring)
    sk = bytes.fromhex(job["signer_sk"])
    al, ad = bytes.fromhex(job["alpha_prefix"]), bytes.fromhex(job["ad_prefix"])
    print("READY", flush=True)
    sys.stdin.readline()                      # the parent releases all workers together
    t = time.perf_counter()
    digest = hashlib.sha256()
    for i in range(first, first + count):
        digest.update(oring.ring_vrf_prove(ring, root, al + i.to_bytes(8, "little"), ad + i.to_bytes(8, "little"), sk))
    print("DONE", time.perf_counter() - t, digest.hexdigest(), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
