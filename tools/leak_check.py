"""Repeated prove_batch / batch_verify / Ring / RingRoot / single-proof calls: host RSS and device memory must stop growing
after the first iterations (the reference keeps tests/benchmark/memory_regression.py for the same purpose).
python tools/leak_check.py [iterations]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psutil
import dot_ring_amd as d


def vram_used() -> int:
    out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--csv"], capture_output=True, text=True).stdout
    for line in out.splitlines():
        parts = line.split(",")
        if len(parts) >= 3 and parts[0].startswith("card"):
            return int(parts[2])
    return -1


iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
proc = psutil.Process()
rows = []
for cv in (d.Bandersnatch, d.JubJub):
    sks = [(700 + i).to_bytes(32, "little") for i in range(300)]
    keys = [cv.public_key_from_secret(sk) for sk in sks[:40]] + [bytes(32)] * 260
    for it in range(iters):
        params = d.RingProofParams.from_ring_size(300, cv=cv)
        ring = d.Ring(keys, params)                      # a fresh ring (and device prover) every iteration
        root = d.RingRoot.from_ring(ring, params)
        n = 256
        al = [b"leak%d-%d" % (it, i) for i in range(n)]
        proofs = d.RingVRF[cv].prove_batch(al, al, [sks[i % 40] for i in range(n)], [keys[i % 40] for i in range(n)], ring, root)
        assert d.RingVRF[cv].batch_verify(proofs, al, al, ring, root)
        one = d.RingVRF[cv].prove(b"x", b"y", sks[0], keys[0], ring, root)
        assert one.verify(b"x", b"y", ring, root)
        ped = d.PedersenVRF[cv].prove_batch(al, [sks[0]] * n, al)
        assert d.PedersenVRF[cv].batch_verify(ped, al, al)
        del ring, root, proofs, one, ped
        if it in (2, iters // 2, iters - 1):
            rows.append((cv.name, it, proc.memory_info().rss >> 20, vram_used() >> 20))
            print(f"{cv.name} iteration {it}: RSS {rows[-1][2]} MiB, VRAM in use {rows[-1][3]} MiB", flush=True)
# contexts created and destroyed (a thread pool of callers): scratch, streams, twiddle tables and SRS uploads must go with them
from dot_ring_amd import _native
from oracle import coracle
from oracle.pyref import bandersnatch as obsn
pts = coracle.te_pack([obsn.G] * 256)
ks = b"".join((i + 1).to_bytes(32, "little") for i in range(256))
srs_be = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dot_ring_amd", "data", "bls12-381-srs-2-11-uncompressed-zcash.bin"), "rb").read()[8 : 8 + 96 * 2048]
ctx_rows = []
for it in range(30):
    c = _native.Context(0)
    c.bsn_scalar_mul_batch(pts, ks)
    c.ntt(ks * 4, 10, pow(49307615728544765012166121802278658070711169839041683575071795236746050763237, 2, obsn.P))
    srs = c.srs_load(srs_be)
    srs.precompute(12)
    c.g1_msm(srs, ks * 8)
    srs.close()
    c.close()
    if it in (2, 15, 29):
        ctx_rows.append((proc.memory_info().rss >> 20, vram_used() >> 20))
        print(f"context cycle {it}: RSS {ctx_rows[-1][0]} MiB, VRAM in use {ctx_rows[-1][1]} MiB", flush=True)
rows.append(("ctx", 0, *ctx_rows[0])); rows.append(("ctx", 1, *ctx_rows[1])); rows.append(("ctx", 2, *ctx_rows[2]))
grow_rss = max(rows[i + 2][2] - rows[i + 1][2] for i in (0, 3, 6))
grow_vram = max(rows[i + 2][3] - rows[i + 1][3] for i in (0, 3, 6))
print(f"growth over the second half: RSS {grow_rss} MiB, VRAM {grow_vram} MiB")
sys.exit(1 if grow_rss > 64 or grow_vram > 64 else 0)
