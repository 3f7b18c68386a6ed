#!/usr/bin/env python3
"""bench.py — headline measurement of the Ring-VRF hot path on MI355X.

Headline workload (BASELINE.json configs[3], the configuration the metric "RingVRF proofs/sec" is quoted on that
fits one GPU): RingVRF[Bandersnatch] prove + verify, ring_size = 1024 (PIOP domain N = 2048), batch = 1024
proofs per GPU.  A "step" = prove_batch of 1024 proofs over the ring followed by batch_verify of those proofs;
ring keys follow the reference's bench scheme (tests/benchmark/bench_ring_proof.py:47-77: sha256-seeded key
pairs, signer at index 3; inputs b"bench-batch-input"/b"bench-batch-ad" || LE64(i), :149-150).  The ring, its
RingRoot, the SRS and the per-ring prover tables are resident in HBM before the timed region; inputs of a step
are 2 short byte strings per proof.  Hidden (ZK) rows are drawn at random as in production (test_vectors=False);
a parity subset is proved again with test_vectors=True and compared byte-for-byte with the CPU oracle.

The timed region runs with the library's per-kernel timers OFF; the same K steps are then repeated with the timers on
(HIP events on the streams the kernels are launched on) and that pass supplies `gpu_kernel_ms_per_step` and the
dominant kernel's average launch duration of `roofline`.

Secondary measurements in the same JSON line (rank 0, one GPU):
  bsn_scalar_mul     BASELINE configs[1]: 4096 Bandersnatch variable-base scalar multiplications (SURVEY 8(d) config 2 inputs)
  g1_msm             BASELINE configs[2]: one G1 Pippenger MSM over 2^20 (and 2^16) synthetic bases
  other_ring_sizes   the metric's other ring sizes: 256 (N = 1024) and 3000 (N = 4096, known-tau SRS: BASELINE configs[4]'s shape)
  distinct_signers   the headline workload with 1024 different signing keys instead of one
  single_call_ms     latency of one RingVRF.prove / one RingVRF.verify (the reference's own benchmark shape, docs/BENCHMARK.md:63-73)
                     and of one Tiny / Thin / Pedersen prove and verify (docs/BENCHMARK.md:20-47), the reference's numbers beside them
  host_threads       the headline step with 4 and 2 host worker threads (what a rank gets when N ranks share a host), phase times by name
  pipelined_prove_verify  the headline's work with batch_verify of batch k on a helper thread beside prove_batch of batch k + 1
  batch_sweep        the headline's step at 512 and 2048 proofs per call
With N ranks (one per GPU) every rank proves and verifies its own 1024 proofs — independent units, no collective.  N > 1 adds
  config5            BASELINE configs[4] through the library call: ONE batch of 1024 x N proofs over ring 3839 (the largest ring of
                     domain 4096, known-tau SRS) sharded by parallel.prove_batch_sharded / batch_verify_sharded, the gather of the
                     784-byte proofs inside the timed region and timed on its own; parity: proofs of different shards vs the oracle
  g1_msm_sharded     one MSM over bases sharded across the ranks with an RCCL all-gather of the partial points
                     (dot_ring_amd/parallel.py, dr_comm_*), `rccl_ranks` = ncclCommCount
No PyTorch anywhere: a launcher only has to export RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.

Launch: python bench.py [--gpus N --steps K --warmup W].  For N > 1 either through torch.distributed.run (one rank per GPU) or as
this bare command: with no RANK in the environment the process starts the N rank processes itself (fresh children, before any
GPU call of its own — the shape of the reference's tests/benchmark/bench_ring_proof.py:168-182), relays rank 0's JSON line and
exits with the worst child's code.  A collective that fails or stalls still prints the line, and the exit code is non-zero.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FR = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
G1_BE = bytes.fromhex(
    "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
    "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1"
)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
# VALU ceiling of the bucket walk: one XYZZ mixed addition = 8 products (461 instructions each) + 2 squarings (383) over
# 14 x 28-bit limbs + ~330 add / sub / carry / unpack instructions = ~4780 wave-instructions per 64 additions; 1024 SIMDs
# x 2.4 GHz at ~4.2 cycles per instruction (profiles/r02_ubench_valu.txt: v_mad_i64_i32 4.2-4.4, VOP2 2.6)
MADD_INSTRUCTIONS_LITERAL = 8 * 461 + 2 * 383 + 330


def _madd_instructions():
    """wave-instructions of one pass of k_g1_accumulate's inner loop, counted from the built code object by
    tools/count_kernel_insts.py (run by __graft_entry__.build()); the literal above only when that file is missing"""
    try:
        with open(os.path.join(ROOT, "dot_ring_amd", "kernel_counts.json")) as f:
            rec = json.load(f)["k_g1_accumulate"]
        return int(rec["loop_instructions"]), int(rec.get("loop_valu_instructions", rec["loop_instructions"])), rec["source"]
    except Exception:
        return MADD_INSTRUCTIONS_LITERAL, MADD_INSTRUCTIONS_LITERAL, "literal in bench.py (dot_ring_amd/kernel_counts.json missing)"


# all wave-instructions of the loop, and the VALU ones among them: the issue-rate ceiling is a VALU ceiling and is priced with the latter
MADD_INSTRUCTIONS, MADD_VALU_INSTRUCTIONS, MADD_SOURCE = _madd_instructions()
VALU_PEAK_GADD_S = 1024 * 2.4e9 / 4.2 * 64 / MADD_VALU_INSTRUCTIONS / 1e9
# the same bucket walk in isolation (tools/ubench_limbs.hip: 2048 chained additions per lane over the prover's 13 MB table,
# profiles/r02_ubench_limbs_fused.txt): an empirical ceiling for the kernel's inner loop on this chip
MEASURED_CHAIN_GADD_S = 7.56
ALG_BYTES_PER_PAIR = 128       # 96 B affine base + 32 B scalar per (base, scalar) pair
ALG_BYTES_PER_SCALAR_MUL = 160  # 64 B point + 32 B scalar + 64 B result
MSM_KERNELS = ("wipe", "k_g1_digits", "k_scan", "k_g1_scatter", "k_g1_part_scatter", "k_g1_part_sort", "k_g1_sort_sets", "k_size_sort", "k_g1_accumulate", "k_g1_reduce_chunks",
               "k_g1_reduce_windows", "k_g1_horner", "k_g1_results_affine")
RING_KERNELS = ("k_bsn_scalar_mul", "k_bsn_encode_to_curve", "k_bsn_msm_groups", "k_bsn_fixed_base", "k_te_msm_prepare", "k_te_msm_accumulate",
                "k_te_msm_reduce", "k_ring_chain", "k_ring_columns", "k_ntt_local",
                "k_ntt_strided", "k_ring_pad", "k_ring_constraints", "k_ring_quotient", "k_ring_eval", "k_ring_linpoly",
                "k_ring_aggpoly", "k_syndiv", "k_ring_diff", "k_bsn_decode_points", "k_g1_decompress")


def seeded_scalars(n: int, tag: bytes):
    """n scalars uniform in [0, r): SHAKE256(tag) stream, 48 bytes each reduced mod r (bias < 2^-128)."""
    stream = hashlib.shake_256(b"g1msm" + tag).digest(48 * n)
    vals = [int.from_bytes(stream[48 * i : 48 * i + 48], "little") % FR for i in range(n)]
    return vals, b"".join(v.to_bytes(32, "little") for v in vals)


def be_to_le_points(raw: bytes) -> bytes:
    return b"".join(raw[i : i + 48][::-1] + raw[i + 48 : i + 96][::-1] for i in range(0, len(raw), 96))


def _seed(*parts) -> bytes:
    """tests/benchmark/bench_ring_proof.py:47 — sha256 over the parts, each followed by a zero byte."""
    h = hashlib.sha256()
    for part in parts:
        h.update(part if isinstance(part, bytes) else (part.to_bytes(8, "little") if isinstance(part, int) else part.encode()))
        h.update(b"\0")
    return h.digest()


def bench_ring_keys(cv, ring_size: int, sample_index: int):
    """Signer key pair + ring keys of the reference bench (bench_ring_proof.py:60-77), keys derived on the GPU.
    Also returns every member's secret key (the distinct-signers variant signs with all of them)."""
    from dot_ring_amd.curve import scalar_mul_batch
    from dot_ring_amd.vrf.primitives import secret_from_seed_scalar

    signer_pk, signer_sk = cv.secret_from_seed(_seed("signer", sample_index, 0))
    secrets_ = [secret_from_seed_scalar(cv, _seed("ring-member", sample_index, i)) for i in range(ring_size)]
    keys = [p.point_to_string() for p in scalar_mul_batch([cv.point_type.generator_point()] * ring_size, secrets_)]
    member_sks = [int(s).to_bytes(32, "little") for s in secrets_]
    at = min(3, ring_size - 1)
    keys[at] = signer_pk
    member_sks[at] = signer_sk
    return signer_pk, signer_sk, keys, member_sks


def cpu_baseline_all_cores(keys, ring_size: int, signer_sk: bytes, per_worker: int, workers: int, first_proofs):
    """The oracle prover in `workers` child processes at once (each proves `per_worker` proofs of the benchmark's inputs;
    worker 0's first proofs are compared with the single-process run).  Children never touch the GPU."""
    import subprocess
    import tempfile

    if workers < 0:
        workers = min(16, len(os.sched_getaffinity(0)))
    workers = max(1, workers)
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump({"keys": [k.hex() for k in keys], "ring_size": ring_size, "signer_sk": signer_sk.hex(),
                   "alpha_prefix": b"bench-batch-input".hex(), "ad_prefix": b"bench-batch-ad".hex()}, f)
        job = f.name
    env = {k: v for k, v in os.environ.items()}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    procs = []
    try:
        for w in range(workers):
            procs.append(subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", job, str(w * per_worker), str(per_worker)], cwd=ROOT, env=env,
                                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
        for p in procs:
            if p.stdout.readline().strip() != "READY":
                raise RuntimeError("oracle worker failed to start")
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        outs = [p.stdout.readline().split() for p in procs]
        wall = time.perf_counter() - t0
        for p in procs:
            p.wait(timeout=60)
        if any(len(o) != 3 or o[0] != "DONE" for o in outs):
            raise RuntimeError("oracle worker failed")
        want = hashlib.sha256(b"".join(first_proofs[:per_worker])).hexdigest()
        return {"value": workers * per_worker / wall, "unit": "proofs/s", "cores": workers, "kind": "port", "work": "prove only",
                "sample": f"{workers} oracle processes x {per_worker} proofs each (prove only), released together, {wall:.1f} s wall",
                "matches_single_process": outs[0][2] == want}
    except Exception as exc:          # a baseline that cannot run is reported, not fatal
        return {"value": None, "unit": "proofs/s", "cores": workers, "kind": "port", "sample": f"failed: {exc}"}
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        os.unlink(job)


# ------------------------------------------------------------------------------------------------ secondary legs
def single_msm_table_bits(log2n: int) -> int:
    """window width of the fixed-base table a single MSM of 2^log2n pairs is measured over (13 windows of 20 bits at 2^20: one set
    of 2^19 buckets; fewer, fuller buckets for the small sizes)"""
    return int(os.environ.get("DOTRING_BENCH_MSM_TABLE", "0")) or (20 if log2n >= 20 else 18 if log2n >= 18 else 16)


SRS_FILE = os.path.join(ROOT, "dot_ring_amd", "data", "bls12-381-srs-2-11-uncompressed-zcash.bin")


def survey_msm_bases(ctx, n: int):
    """SURVEY 8(d) config 3's bases: the real SRS (the shipped 2^11 file's 6145 G1 points, /root/reference/dot_ring/ring_proof/pcs/
    srs.py:42-90) followed by [t^i] G1 for i >= 6145, t a fixed public test scalar (dr_srs_powers generates them on the GPU).
    Returns (Srs in HBM, real prefix as BE records, t)."""
    with open(SRS_FILE, "rb") as f:
        blob = f.read()
    m = min(n, int.from_bytes(blob[:8], "little"))
    real = blob[8 : 8 + 96 * m]
    t = int.from_bytes(hashlib.sha256(b"bench-known-tau").digest(), "little") % FR
    powers = ctx.srs_powers(G1_BE, t, n)
    tail = powers.download(m, n - m) if n > m else b""
    powers.close()
    return ctx.srs_load(real + tail), real, t


def survey_msm_expected(real_be: bytes, t: int, vals, raw: bytes):
    """The whole MSM over survey_msm_bases' points from its structure: sum_{i < m} k_i S_i (oracle Pippenger over the m real SRS points)
    + [sum_{i >= m} k_i t^i] G1 — ONE oracle MSM of m + 1 terms that accounts for every one of the n pairs.  BE x||y."""
    from oracle import coracle

    m = len(real_be) // 96
    acc, tp = 0, pow(t, m, FR)
    for k in vals[m:]:
        acc = (acc + k * tp) % FR
        tp = tp * t % FR
    bases = be_to_le_points(real_be + G1_BE)
    want = bytes(coracle.g1_msm_raw(bases, raw[: 32 * m] + acc.to_bytes(32, "little"), m + 1))
    return want[:48][::-1] + want[48:][::-1]


def g1_msm_measurement(ctx, log2n: int, steps: int, cpu_sample_log2: int, do_cpu: bool):
    """BASELINE configs[2] on SURVEY 8(d)'s inputs: G1 MSM over 2^log2n bases = real SRS prefix + [t^i] G1. Returns a dict (rank-local)."""
    n = 1 << log2n
    srs, real_be, t = survey_msm_bases(ctx, n)
    # fixed-base window table in HBM (W * n * 128 B = 1.7 GB at 2^20 with 20-bit windows): one bucket set per MSM
    table_bits = single_msm_table_bits(log2n)
    srs.precompute(table_bits)
    vals, raw = seeded_scalars(n, b"\0\0\0\0")
    d_scalars = ctx.alloc(32 * n).upload(raw)
    for _ in range(3):                     # untimed warm-up calls: scratch allocations, first-touch of the 1.7 GB table, clocks
        ctx.g1_msm_dev(srs, d_scalars, n)
    t0 = time.perf_counter()
    result = None
    for _ in range(steps):
        result = ctx.g1_msm_dev(srs, d_scalars, n)
    elapsed = time.perf_counter() - t0
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(min(steps, 5)):
        ctx.g1_msm_dev(srs, d_scalars, n)
    ctx.prof_enable(False)
    acc_ms, acc_n = ctx.prof_get("k_g1_accumulate")
    kern = {k: round(ctx.prof_get(k)[0] / max(1, min(steps, 5)), 3) for k in MSM_KERNELS if ctx.prof_get(k)[1]}
    total_kernel_ms = sum(kern.values())
    out = {"pairs": n, "table_window_bits": table_bits,
           "inputs": f"SURVEY 8(d) config 3: the shipped SRS's {len(real_be) // 96} G1 points followed by [t^i] G1 (dr_srs_powers, t public); "
                     "scalars uniform in [0, r) from SHAKE256('g1msm' || tag)",
           "scalar_muls_per_s": n * steps / elapsed, "ms_per_msm": elapsed / steps * 1e3,
           "k_g1_accumulate_avg_ms": acc_ms / max(1, acc_n), "kernel_ms_per_msm": kern,
           "non_accumulate_share": 1.0 - kern.get("k_g1_accumulate", 0.0) / (elapsed / steps * 1e3) if elapsed else None,
           "kernel_ms_sum": round(total_kernel_ms, 3),
           "roofline_frac_hbm": (ALG_BYTES_PER_PAIR * n / (acc_ms / max(1, acc_n) / 1e3) / 1e9) / HBM_PEAK_GBS if acc_ms else None}
    if do_cpu:
        from oracle import coracle

        # every pair accounted for: oracle Pippenger over the real prefix + the closed form of the [t^i] G1 tail
        out["parity_closed_form"] = result == survey_msm_expected(real_be, t, vals, raw)
        if cpu_sample_log2 > 0:
            m = min(n, 1 << cpu_sample_log2)
            bases = be_to_le_points(srs.download(0, m))
            t1 = time.perf_counter()
            cpu_out = coracle.g1_msm_raw(bases, raw[: 32 * m], m)
            cpu_s = time.perf_counter() - t1
            gpu_out = ctx.g1_msm_dev(srs, d_scalars, m)
            out["parity_sample"] = bytes(cpu_out) == (bytes(96) if gpu_out is None else gpu_out[:48][::-1] + gpu_out[48:][::-1])
            out["cpu_port_scalar_muls_per_s"] = m / cpu_s
    d_scalars.free()
    srs.close()
    return out


def g1_msm_batched_leg(ctx, pcs, domain: int, batch: int, steps: int, valu_peak_gadd_s: float):
    """The second half of BASELINE's metric at the ring sizes it names: the batched G1 MSMs a batch of ring proofs runs —
    `KZG.commit` of `batch` polynomials of 3N coefficients (/root/reference/dot_ring/ring_proof/pcs/kzg.py:152-175) over the SRS of
    that domain (the shipped file up to N = 2048, the known-tau SRS at N = 4096) as ONE dr_g1_msm_batch_dev call on device-resident
    scalar vectors: base-scalar pairs per second, the bucket walk's share of the HBM and VALU ceilings, 2 vectors against the oracle.
    Scalars: uniform 255-bit values from a seeded generator (reduced mod r on the device)."""
    import numpy as np

    from oracle import coracle

    n = 3 * domain
    srs = pcs._srs().device()
    rng = np.random.default_rng(20261005 + domain)
    raw = bytearray(rng.bytes(32 * n * batch))
    raw[31::32] = bytes(b & 0x7F for b in raw[31::32])                 # < 2^255
    raw = bytes(raw)
    d_scalars = ctx.alloc(32 * n * batch).upload(raw)
    out = ctx.g1_msm_batch_dev(srs, d_scalars, n, batch)               # warm-up + the checked results
    bases_le = be_to_le_points(srs.download(0, n))
    ok = True
    for b in (0, batch - 1):
        want = bytes(coracle.g1_msm_raw(bases_le, raw[32 * n * b : 32 * n * (b + 1)], n))
        ok = ok and out[b] == want[:48][::-1] + want[48:][::-1]
    ctx.g1_msm_batch_dev(srs, d_scalars, n, batch)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.g1_msm_batch_dev(srs, d_scalars, n, batch)
    elapsed = time.perf_counter() - t0
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(steps):
        ctx.g1_msm_batch_dev(srs, d_scalars, n, batch)
    ctx.prof_enable(False)
    kern = {k: round(ctx.prof_get(k)[0] / steps, 3) for k in MSM_KERNELS if ctx.prof_get(k)[1]}
    acc_ms = ctx.prof_get("k_g1_accumulate")[0] / steps
    tinfo = srs.table_info(n, batch)
    pairs = n * batch
    adds = pairs * tinfo["digits_per_scalar"]
    d_scalars.free()
    return {"domain_size": domain, "pairs_per_msm": n, "msms_per_call": batch, "pairs_per_call": pairs, "steps": steps,
            "scalar_muls_per_s": pairs * steps / elapsed, "ms_per_call": elapsed / steps * 1e3, "kernel_ms_per_call": kern,
            "parity_ok": bool(ok), "parity_vectors": 2, "table": tinfo,
            "roofline": {"bound": "hbm", "kernel": "k_g1_accumulate", "avg_kernel_ms": acc_ms,
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_PAIR * pairs,
                         "achieved": ALG_BYTES_PER_PAIR * pairs / (acc_ms / 1e3) / 1e9 if acc_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ALG_BYTES_PER_PAIR * pairs / (acc_ms / 1e3) / 1e9 / HBM_PEAK_GBS if acc_ms else None,
                         "valu": {"achieved_gadd_s": adds / (acc_ms / 1e3) / 1e9 if acc_ms else None, "peak_gadd_s": valu_peak_gadd_s,
                                  "frac": adds / (acc_ms / 1e3) / 1e9 / valu_peak_gadd_s if acc_ms else None,
                                  "additions_per_pair": tinfo["digits_per_scalar"]}}}


def bsn_scalar_mul_measurement(ctx, cv, n: int, steps: int):
    """BASELINE configs[1] / SURVEY 8(d) config 2: n variable-base scalar multiplications, P_i = public key of
    secret_from_seed(sha256("bsn-pt" || LE64(i))), k_i = sha256("bsn-k" || LE64(i)) mod n, all n compared with the oracle."""
    from dot_ring_amd.curve import pack_points, scalar_mul_batch
    from dot_ring_amd.vrf.primitives import secret_from_seed_scalar
    from oracle import coracle
    from oracle.pyref import bandersnatch as obsn

    secrets_ = [secret_from_seed_scalar(cv, hashlib.sha256(b"bsn-pt" + i.to_bytes(8, "little")).digest()) for i in range(n)]
    raw_p = pack_points(scalar_mul_batch([cv.point_type.generator_point()] * n, secrets_))
    ks = [int.from_bytes(hashlib.sha256(b"bsn-k" + i.to_bytes(8, "little")).digest(), "little") % obsn.N for i in range(n)]
    raw_k = b"".join(k.to_bytes(32, "little") for k in ks)
    d_p, d_k, d_o = ctx.alloc(64 * n).upload(raw_p), ctx.alloc(32 * n).upload(raw_k), ctx.alloc(64 * n)
    ctx.bsn_scalar_mul_batch_dev(d_p, d_k, n, d_o)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.bsn_scalar_mul_batch_dev(d_p, d_k, n, d_o)                       # device-resident in and out, one launch + sync
    wall = (time.perf_counter() - t0) / steps
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(steps):
        ctx.bsn_scalar_mul_batch_dev(d_p, d_k, n, d_o)
    ctx.prof_enable(False)
    k_ms, k_n = ctx.prof_get("k_bsn_scalar_mul")
    kernel_ms = k_ms / max(1, k_n)
    got = d_o.download()
    t1 = time.perf_counter()
    want = coracle.te_mul_batch_raw(raw_p, raw_k, n, glv=True)
    cpu_s = time.perf_counter() - t1
    # the host-buffer entry point (GLV lane pairs below 16384 terms: half the dependent chain), PCIe included
    t2 = time.perf_counter()
    for _ in range(steps):
        host_out = ctx.bsn_scalar_mul_batch(raw_p, raw_k)
    host_wall = (time.perf_counter() - t2) / steps
    for b in (d_p, d_k, d_o):
        b.free()
    return {"n": n, "scalar_muls_per_s": n / wall, "ms_per_call": wall * 1e3, "kernel_ms": kernel_ms,
            "scalar_muls_per_s_kernel": n / (kernel_ms / 1e3) if kernel_ms else None,
            "host_buffers_scalar_muls_per_s": n / host_wall, "host_buffers_ms_per_call": host_wall * 1e3,
            "roofline_frac_hbm": ALG_BYTES_PER_SCALAR_MUL * n / (kernel_ms / 1e3) / 1e9 / HBM_PEAK_GBS if kernel_ms else None,
            "algorithmic_bytes_per_scalar_mul": ALG_BYTES_PER_SCALAR_MUL,
            "cpu_port": {"scalar_muls_per_s": n / cpu_s, "cores": 1, "kind": "port",
                         "sample": f"all {n} through oracle/c te_mul_batch (GLV + joint 2-bit windows, bandersnatch_te.pyx:480), {cpu_s:.2f} s"},
            "parity_ok": got == want and host_out == want}


class RingWorkload:
    """ring + root + inputs of one ring size, and the prove_batch + batch_verify step over them"""

    def __init__(self, d, ring_size: int, batch: int, first_index: int = 0):
        from dot_ring_amd.ring_proof.pcs import SRS

        self.d, self.ring_size, self.batch = d, ring_size, batch
        self.cv = d.Bandersnatch
        self.vrf = d.RingVRF[self.cv]
        self.signer_pk, self.signer_sk, self.keys, self.member_sks = bench_ring_keys(self.cv, ring_size, 0)
        # domains above 2048 need more SRS points than the shipped file holds: known-tau SRS, same on every rank
        self.big = d.RingProofParams.from_ring_size(ring_size).domain_size > 2048
        self.tau = int.from_bytes(hashlib.sha256(b"bench-known-tau").digest(), "little") % d.KZG.scalar_modulus
        self.pcs = d.KZG.with_srs(SRS.synthetic(self.tau, 3 * 4096 + 1)) if self.big else d.KZG
        self.params = d.RingProofParams.from_ring_size(ring_size, pcs=self.pcs)
        t = time.perf_counter()
        self.ring = d.Ring(self.keys, self.params)
        self.root = d.RingRoot.from_ring(self.ring, self.params)
        self.ring_root_s = time.perf_counter() - t
        self.alphas = [b"bench-batch-input" + (first_index + i).to_bytes(8, "little") for i in range(batch)]
        self.ads = [b"bench-batch-ad" + (first_index + i).to_bytes(8, "little") for i in range(batch)]
        self.sks, self.pks = [self.signer_sk] * batch, [self.signer_pk] * batch
        self.prove_s = self.verify_s = 0.0

    def distinct_signers(self):
        """every proof signed by another ring member (ring_size >= batch)"""
        self.sks = [self.member_sks[i % self.ring_size] for i in range(self.batch)]
        self.pks = [self.keys[i % self.ring_size] for i in range(self.batch)]

    def _span(self, lo: int, hi: int):
        t = time.perf_counter()
        proofs = self.vrf.prove_batch(self.alphas[lo:hi], self.ads[lo:hi], self.sks[lo:hi], self.pks[lo:hi], self.ring, self.root)
        t1 = time.perf_counter()
        ok = self.vrf.batch_verify(proofs, self.alphas[lo:hi], self.ads[lo:hi], self.ring, self.root)
        return proofs, ok, t1 - t, time.perf_counter() - t1

    def step(self):
        """One step = the whole batch proved and verified: one prove_batch call, then one batch_verify call."""
        proofs, ok, tp, tv = self._span(0, self.batch)
        self.prove_s += tp
        self.verify_s += tv
        return proofs, ok

    def run(self, steps: int, warmup: int, barrier=lambda: None):
        for _ in range(warmup):
            self.step()
        self.prove_s = self.verify_s = 0.0
        barrier()
        t0 = time.perf_counter()
        all_ok = True
        for _ in range(steps):
            _, ok = self.step()
            all_ok = all_ok and ok
        barrier()
        return time.perf_counter() - t0, all_ok

    def deterministic_ring(self):
        """(ring, root) of the same keys in test-vector mode (hidden rows zero: proofs are a function of their inputs)"""
        d = self.d
        tv_params = d.RingProofParams.from_ring_size(self.ring_size, test_vectors=True, pcs=self.pcs)
        tv_ring = d.Ring(self.keys, tv_params)
        return tv_ring, d.RingRoot.from_ring(tv_ring, tv_params)

    def oracle_ring(self):
        """(ring, root) of the CPU oracle for these keys, test-vector mode, over the same SRS"""
        from oracle.pyref import bandersnatch as obsn
        from oracle.pyref import ring as oring

        o_srs = None
        if self.big:
            from oracle.pyref import kzg as okzg

            o_srs = okzg.SRS.from_tau(self.tau, 3 * 4096 + 1)
            o_srs.g2_raw = list(self.pcs.srs.g2_raw)
        o_ring = oring.Ring(self.keys, oring.Params.from_ring_size(self.ring_size, test_vectors=True, suite=obsn.SHA512, srs=o_srs))
        return o_ring, oring.RingRoot(o_ring)

    def parity(self, m: int, time_it: bool = False):
        """m deterministic proofs (test_vectors=True) byte-compared with the CPU oracle; returns (ok, cpu proofs, cpu seconds)"""
        from oracle.pyref import ring as oring

        tv_ring, tv_root = self.deterministic_ring()
        # the WHOLE batch in deterministic mode, so that the m proofs compared below come through the same code path as the timed
        # batches (from a few hundred MSMs on, the KZG commitments recode their scalars in non-adjacent form over the bit-row SRS table)
        nb = len(self.alphas)
        gpu_all = self.vrf.prove_batch(self.alphas, self.ads, self.sks[:nb], self.pks[:nb], tv_ring, tv_root)
        gpu_proofs = gpu_all[:m]
        o_ring, o_root = self.oracle_ring()
        ok = o_root.encode() == tv_root.encode() == self.root.encode()
        t1 = time.perf_counter()
        cpu_proofs = [oring.ring_vrf_prove(o_ring, o_root, self.alphas[i], self.ads[i], self.sks[i]) for i in range(m)]
        cpu_s = time.perf_counter() - t1
        ok = ok and [p.encode() for p in gpu_proofs] == cpu_proofs
        ok = ok and self.vrf.batch_verify(gpu_proofs, self.alphas[:m], self.ads[:m], tv_ring, tv_root)
        return ok, cpu_proofs, cpu_s


def ring_roofline(w: "RingWorkload", steps: int, barrier=lambda: None):
    """`steps` more steps of the workload with the per-kernel timers on (HIP events on the streams the kernels are launched on):
    the bucket walk's average launch time against its algorithmic bytes (128 B per dense pair, 32 B per by-parts scalar) and against
    the VALU issue ceiling.  Returns (roofline dict, per-kernel ms per step)."""
    from dot_ring_amd import runtime

    all_ctx = runtime.contexts()
    for c in all_ctx:
        c.prof_reset()
        c.prof_enable(True)
    elapsed_prof, _ = w.run(steps, 0, barrier)
    for c in all_ctx:
        c.prof_enable(False)

    def prof_sum(name):
        ms = cnt = 0
        for c in all_ctx:
            m_, n_ = c.prof_get(name)
            ms, cnt = ms + m_, cnt + n_
        return ms, cnt

    kernel_ms = {name: prof_sum(name)[0] / max(1, steps) for name in MSM_KERNELS + RING_KERNELS}
    acc_ms, acc_launches = prof_sum("k_g1_accumulate")
    n_dom, batch = w.ring.params.domain_size, w.batch
    # dense (base, scalar) pairs per proof: quotient 3N+1, two opening quotients 3N + (N-1)  (SURVEY 3.3); the four witness columns
    # (4N) are committed by summation by parts: 4N scalars are read, only ~1.1k bases gathered, so they are priced at the 32 B scalar
    pairs_per_proof, scalar_only_per_proof = 7 * n_dom, 4 * n_dom
    # digits per scalar of each dense call as the library plans it: the quotient (batch vectors of 3N + 1 terms) and the two opening
    # quotients (ONE call of 2 * batch vectors of 3N terms; the N - 1 terms of the second are followed by zeros, which have no digits)
    dev_srs = w.pcs._srs().device()
    tinfo, tinfo_open = dev_srs.table_info(3 * n_dom + 1, batch), dev_srs.table_info(3 * n_dom, 2 * batch)
    dense_adds = float(batch) * steps * ((3 * n_dom + 1) * tinfo["digits_per_scalar"] + (4 * n_dom - 1) * tinfo_open["digits_per_scalar"])
    avg_acc_s = (acc_ms / max(1, acc_launches)) / 1e3
    alg_bytes_launch = (ALG_BYTES_PER_PAIR * batch * pairs_per_proof + 32.0 * batch * scalar_only_per_proof) * steps / max(1, acc_launches)
    achieved = alg_bytes_launch / avg_acc_s / 1e9 if avg_acc_s > 0 else 0.0
    gadd = dense_adds / (acc_ms / 1e3) / 1e9 if acc_ms else None
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "kernel": "k_g1_accumulate", "avg_kernel_ms": avg_acc_s * 1e3, "launches_per_step": acc_launches / max(1, steps),
            "algorithmic_bytes_per_launch": alg_bytes_launch,
            "timers": "HIP events per launch, in a second pass of the same steps (the timed region runs without them); "
                      f"that pass took {elapsed_prof / steps * 1e3:.2f} ms per step",
            "valu": {"achieved_gadd_s": gadd, "peak_gadd_s": VALU_PEAK_GADD_S, "frac": gadd / VALU_PEAK_GADD_S if gadd else None,
                     "instructions_per_addition": MADD_INSTRUCTIONS, "valu_instructions_per_addition": MADD_VALU_INSTRUCTIONS,
                     "table": tinfo, "table_openings": tinfo_open, "additions_per_pair": tinfo["digits_per_scalar"],
                     "note": "dense bucket additions only (7N pairs x non-zero digits per proof, per call shape); by-parts and verify-side additions "
                             "not counted; the ceilings are VALU issue rates over the loop's VALU instructions"}}
    return roof, kernel_ms


def single_call_leg(w: "RingWorkload", reps: int = 10):
    """Latency of ONE RingVRF.prove and ONE RingVRF.verify on the headline ring (the reference publishes 534.57 ms / 3.99 ms for
    them, docs/BENCHMARK.md:72-73); fresh inputs per call so nothing is memoised, minimum and median over `reps`."""
    vrf, ring, root = w.vrf, w.ring, w.root
    proofs = []
    t_prove, t_verify = [], []
    for i in range(reps + 1):
        al, ad = b"single-call-input" + i.to_bytes(4, "little"), b"single-call-ad"
        t = time.perf_counter()
        pr = vrf.prove(al, ad, w.signer_sk, w.signer_pk, ring, root)
        t_prove.append(time.perf_counter() - t)
        proofs.append((al, ad, vrf.decode(pr.encode())))
    ok = True
    for al, ad, pr in proofs:
        t = time.perf_counter()
        ok = pr.verify(al, ad, ring, root) and ok
        t_verify.append(time.perf_counter() - t)
    bad = proofs[0][2].verify(proofs[1][0], proofs[0][1], ring, root)               # wrong input must fail
    t_prove, t_verify = sorted(t_prove[1:]), sorted(t_verify[1:])
    return {"prove_ms_min": t_prove[0] * 1e3, "prove_ms_median": t_prove[len(t_prove) // 2] * 1e3,
            "verify_ms_min": t_verify[0] * 1e3, "verify_ms_median": t_verify[len(t_verify) // 2] * 1e3,
            "reference_ms": {"prove": 534.57, "verify": 3.99, "source": "docs/BENCHMARK.md:72-73 (M1 Max, one core)"},
            "verified": bool(ok and not bad), "reps": reps}


def pipelined_leg(w: "RingWorkload", steps: int):
    """The same K x (prove_batch + batch_verify) as the headline, software-pipelined by the application: while the calling thread
    proves batch k + 1, a helper thread verifies batch k on its own context — the verifier's latency-bound kernels and host phases
    (decode, transcripts, pairing: ~6 ms) and the interpreter's work between the calls run beside the prover's bucket walk.  Every
    batch is proved AND verified inside the timed region; the last verification runs alone.  A secondary figure: the headline
    keeps the two calls strictly one after the other."""
    from concurrent.futures import ThreadPoolExecutor

    prove = lambda: w.vrf.prove_batch(w.alphas, w.ads, w.sks, w.pks, w.ring, w.root)
    verify = lambda proofs: w.vrf.batch_verify(proofs, w.alphas, w.ads, w.ring, w.root)
    with ThreadPoolExecutor(max_workers=1) as pool:
        pool.submit(verify, prove()).result()                     # warm-up: the helper thread's context and verifier state
        t0 = time.perf_counter()
        proofs, ok = prove(), True
        for k in range(steps):
            pending = pool.submit(verify, proofs)
            if k + 1 < steps:
                proofs = prove()
            ok = pending.result() and ok
        elapsed = time.perf_counter() - t0
    return {"proofs_per_s": w.batch * steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "steps": steps, "all_verified": bool(ok),
            "note": "batch_verify of batch k on a helper thread while prove_batch of batch k + 1 runs; same work as the headline"}


def ring_size_leg(d, ring_size: int, batch: int, steps: int, parity_proofs: int, rank: int = 0, ctl=None, barrier=lambda: None, with_roofline: bool = False):
    """prove_batch + batch_verify at another ring size.  With a control communicator every rank runs its own `batch` proofs
    between two barriers, the time is the max over ranks and proofs_per_s the whole job's; parity (rank 0) against the oracle."""
    world = 1 if ctl is None else ctl.world
    w = RingWorkload(d, ring_size, batch, first_index=rank * batch)
    elapsed, all_ok = w.run(steps, 1, barrier)
    if ctl is not None:
        stats = [struct.unpack("<dB", b) for b in ctl.all_gather(struct.pack("<dB", elapsed, 1 if all_ok else 0))]
        elapsed, all_ok = max(s_[0] for s_ in stats), all(s_[1] for s_ in stats)
    prove_s, verify_s = w.prove_s, w.verify_s                # (the profiled pass below runs — and times — more steps)
    ok, _, _ = w.parity(parity_proofs) if parity_proofs and rank == 0 else (True, None, 0.0)
    roof = msm_batched = None
    if rank == 0 and with_roofline:
        roof, _ = ring_roofline(w, min(steps, 2))
        from dot_ring_amd import runtime

        msm_batched = g1_msm_batched_leg(runtime.context(), w.pcs, w.ring.params.domain_size, batch, 3, VALU_PEAK_GADD_S)
        ok = ok and msm_batched["parity_ok"]
    out = {"ring_size": ring_size, "domain_size": w.ring.params.domain_size, "max_ring_size": w.ring.params.max_ring_size,
           "batch": batch, "steps": steps, "ranks": world,
           "proofs_per_s": batch * world * steps / elapsed, "ms_per_step": elapsed / steps * 1e3,
           "prove_only_proofs_per_s": batch * steps / prove_s, "verify_only_proofs_per_s": batch * steps / verify_s,
           "parity_ok": bool(ok and all_ok), "parity_proofs": parity_proofs if rank == 0 else 0,
           "ring_root_s": w.ring_root_s, "srs": "known-tau, 12289 points" if w.big else "shipped 2^11 file"}
    if ctl is not None:
        out["ms_per_step_by_rank"] = [s_[0] / steps * 1e3 for s_ in stats]
    if roof is not None:
        out["roofline"] = roof
        out["g1_msm_batched"] = msm_batched
    del w
    return out


def config5_sharded_leg(d, ring_size: int, batch_per_rank: int, steps: int, parity_proofs: int, comm, barrier):
    """BASELINE configs[4] as the library call: ONE batch of batch_per_rank x world proofs over a ring of domain 4096, the same
    arguments on every rank; parallel.prove_batch_sharded gives rank g its slice and gathers the 784-byte proofs to EVERY rank,
    parallel.batch_verify_sharded verifies the slices and ANDs the verdicts.  The gather is inside the timed region; its own time
    (payload phase, after the 9-byte status header has absorbed the wait for the slowest rank) is reported beside it, also for the
    gather-to-rank-0 form.  Parity (rank 0): deterministic proofs from DIFFERENT shards against the CPU oracle."""
    from dot_ring_amd import parallel

    rank, world = comm.rank, comm.world
    total = batch_per_rank * world
    w = RingWorkload(d, ring_size, total)                          # the same whole batch on every rank
    vrf = w.vrf
    args = (w.alphas, w.ads, w.sks, w.pks, w.ring, w.root)

    def step(dst):
        t = time.perf_counter()
        proofs = parallel.prove_batch_sharded(comm, vrf, *args, dst=dst)
        ex = dict(parallel.last_exchange)
        t1 = time.perf_counter()
        ok = parallel.batch_verify_sharded(comm, vrf, proofs, w.alphas, w.ads, w.ring, w.root)
        return ok, t1 - t, time.perf_counter() - t1, ex

    step(None)                                                     # warm-up: prover tables, verifier key, the sockets' buffers
    barrier()
    t0 = time.perf_counter()
    all_ok, prove_s, verify_s, gather_s = True, 0.0, 0.0, 0.0
    for _ in range(steps):
        ok, tp, tv, ex = step(None)
        all_ok, prove_s, verify_s, gather_s = all_ok and ok, prove_s + tp, verify_s + tv, gather_s + ex["payload_s"]
    barrier()
    elapsed = time.perf_counter() - t0
    # the other form of the gather: all proofs to rank 0 only (batch_verify_sharded then starts with a broadcast from there)
    ok0, tp0, tv0, ex0 = step(0)
    stats = [struct.unpack("<dddddB", b) for b in comm.all_gather(struct.pack("<dddddB", elapsed, prove_s, verify_s, gather_s, ex0["payload_s"],
                                                                               1 if all_ok and ok0 else 0))]
    elapsed = max(s_[0] for s_ in stats)
    # parity: deterministic mode through the same sharded call; rank 0 compares proofs of different shards with the oracle
    parity_ok, checked = True, []
    if parity_proofs:
        tv_ring, tv_root = w.deterministic_ring()
        tv = parallel.prove_batch_sharded(comm, vrf, w.alphas, w.ads, w.sks, w.pks, tv_ring, tv_root, dst=0)
        if rank == 0:
            from oracle.pyref import ring as oring

            o_ring, o_root = w.oracle_ring()
            checked = sorted({0, parallel.shard_range(total, world - 1, world)[0], total - 1})      # first shard, first and last proof of the last shard
            parity_ok = o_root.encode() == tv_root.encode() and len(tv) == total
            for i in checked:
                parity_ok = parity_ok and tv[i].encode() == oring.ring_vrf_prove(o_ring, o_root, w.alphas[i], w.ads[i], w.sks[i])
        verdict = parallel.batch_verify_sharded(comm, vrf, tv, w.alphas, w.ads, tv_ring, tv_root)
        parity_ok = parity_ok and verdict
    return {"config": f"BASELINE configs[4]: RingVRF[Bandersnatch] ring {ring_size} (domain {w.ring.params.domain_size}, known-tau SRS), ONE batch of "
                      f"{total} proofs sharded over {world} ranks by parallel.prove_batch_sharded / batch_verify_sharded",
            "ring_size": ring_size, "domain_size": w.ring.params.domain_size, "batch_total": total, "batch_per_rank": batch_per_rank, "ranks": world,
            "steps": steps, "proofs_per_s": total * steps / elapsed, "ms_per_step": elapsed / steps * 1e3,
            "ms_per_step_by_rank": [s_[0] / steps * 1e3 for s_ in stats],
            "prove_sharded_ms": max(s_[1] for s_ in stats) / steps * 1e3, "verify_sharded_ms": max(s_[2] for s_ in stats) / steps * 1e3,
            "gather": {"bytes": 784 * total, "communicator": type(comm).__name__, "inside_timed_region": True,
                       "to_every_rank_ms": max(s_[3] for s_ in stats) / steps * 1e3, "to_rank0_only_ms": stats[0][4] * 1e3},
            "parity_ok": bool(parity_ok and all(s_[5] for s_ in stats)), "parity_proof_indices": checked,
            "srs": "known-tau, 12289 points" if w.big else "shipped 2^11 file"}


def host_threads_sweep(args, counts=(4, 2)):
    """The headline step with fewer host worker threads (DOTRING_HOST_THREADS; the N-rank launcher gives each rank cores // N of them):
    one child process per setting — the pool is sized once per process — running this script's timed region only, with the library's
    phase trace on (DOTRING_TRACE) so that the host phases of the verifier show by name.  Children run one after the other while this
    process is idle; they share its GPU."""
    import re
    import subprocess

    out = {}
    for t in counts:
        env = dict(os.environ)
        env.update({"DOTRING_HOST_THREADS": str(t), "DOTRING_TRACE": "1"})
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "3", "--warmup", "1", "--ring-size", str(args.ring_size),
               "--batch", str(args.batch), "--cpu-proofs", "0", "--msm-log2n", "0", "--extras", "0"]
        try:
            proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            line = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
            phases = {}
            for what in ("prove_batch", "verify_batch"):
                rows = [ln for ln in proc.stderr.splitlines() if ln.startswith(f"[dotring] {what} ")][-3:]   # the timed steps
                acc = {}
                for ln in rows:
                    for name, ms in re.findall(r" ([A-Za-z0-9+*_ ]+?)=([0-9.]+)", ln.split("|", 1)[1] if "|" in ln else ""):
                        acc[name.strip()] = acc.get(name.strip(), 0.0) + float(ms) / len(rows)
                    m_ = re.search(r"total=([0-9.]+)", ln)
                    if m_:
                        acc["total"] = acc.get("total", 0.0) + float(m_.group(1)) / len(rows)
                phases[what] = {k: round(v, 3) for k, v in acc.items()}
            out[str(t)] = {"proofs_per_s": line["value"], "ms_per_step": line["ms_per_step"],
                           "prove_only_proofs_per_s": line["prove_only_proofs_per_s"], "verify_only_proofs_per_s": line["verify_only_proofs_per_s"],
                           "phase_ms": phases}
        except Exception as exc:          # noqa: BLE001 — a sweep point that cannot run is reported, not fatal
            out[str(t)] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def small_vrf_single_call_leg(d, reps: int = 20):
    """Latency of ONE prove and ONE verify of the three small schemes (BASELINE configs[0]'s shape; the reference publishes them in
    docs/BENCHMARK.md:20-21,33-34,46-47 for one M1 Max core): fresh inputs per call, minimum and median over `reps`."""
    cv = d.Bandersnatch
    pk, sk = cv.secret_from_seed(_seed("signer", 0, 0))
    ref = {"TinyVRF": (2.21, 1.97, "docs/BENCHMARK.md:20-21"), "ThinVRF": (2.21, 1.99, "docs/BENCHMARK.md:33-34"),
           "PedersenVRF": (2.40, 1.74, "docs/BENCHMARK.md:46-47")}
    out, all_ok = {}, True
    for name, (ref_p, ref_v, src) in ref.items():
        vrf = getattr(d, name)[cv]
        tp, tv, ok = [], [], True
        for i in range(reps + 2):
            al = b"single-call-input" + i.to_bytes(4, "little")
            t = time.perf_counter()
            pr = vrf.prove(al, sk, b"ad")
            tp.append(time.perf_counter() - t)
            pr = vrf.decode(pr.encode())
            t = time.perf_counter()
            good = pr.verify(al, b"ad") if name == "PedersenVRF" else pr.verify(pk, al, b"ad")
            tv.append(time.perf_counter() - t)
            bad = pr.verify(al + b"x", b"ad") if name == "PedersenVRF" else pr.verify(pk, al + b"x", b"ad")
            ok = ok and good and not bad
        tp, tv = sorted(tp[2:]), sorted(tv[2:])
        out[name] = {"prove_ms_min": tp[0] * 1e3, "prove_ms_median": tp[len(tp) // 2] * 1e3, "verify_ms_min": tv[0] * 1e3,
                     "verify_ms_median": tv[len(tv) // 2] * 1e3, "reference_ms_min": {"prove": ref_p, "verify": ref_v, "source": src + " (M1 Max, one core)"},
                     "verified": bool(ok)}
        all_ok = all_ok and ok
    out["verified"] = bool(all_ok)
    out["reps"] = reps
    return out


def sharded_msm_leg(ctx, comm, log2_total: int, steps: int, scaling: str):
    """One G1 MSM whose 2^log2_total bases are sharded over the ranks (each rank: srs_synthetic(first = lo + 1) for its slice,
    scalars seeded per rank); closed-form check on every rank, time = max over ranks through the communicator."""
    from dot_ring_amd import parallel
    from oracle import coracle

    n = 1 << log2_total
    lo, hi = parallel.shard_range(n, comm.rank, comm.world)
    cnt = hi - lo
    srs = ctx.srs_synthetic(G1_BE, cnt, first=lo + 1)
    srs.precompute(16 if cnt >= (1 << 18) else 12)
    vals, raw = seeded_scalars(cnt, b"shard" + comm.rank.to_bytes(4, "little"))
    d_scalars = ctx.alloc(32 * cnt).upload(raw)
    local_sum = sum(k * (lo + 1 + i) for i, k in enumerate(vals)) % FR
    expect = sum(int.from_bytes(b, "little") for b in comm.all_gather(local_sum.to_bytes(32, "little"))) % FR
    got = parallel.g1_msm_sharded(ctx, comm, srs, d_scalars, cnt)            # warm-up + the checked result
    want = bytes(coracle.g1_msm_raw(be_to_le_points(G1_BE), expect.to_bytes(32, "little"), 1))
    ok = got == want[:48][::-1] + want[48:][::-1]
    ctx.sync()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        parallel.g1_msm_sharded(ctx, comm, srs, d_scalars, cnt)
    ctx.sync()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    stats = [struct.unpack("<dB", b) for b in comm.all_gather(struct.pack("<dB", elapsed, 1 if ok else 0))]
    elapsed = max(s[0] for s in stats)
    d_scalars.free()
    srs.close()
    return {"sharding": "bases", "scaling": scaling, "pairs_total": n, "pairs_per_rank": cnt, "ranks": comm.world,
            "rccl_ranks": comm.rccl_ranks() if hasattr(comm, "rccl_ranks") else None, "collective": type(comm).__name__,
            "rehearsal": None if hasattr(comm, "rccl_ranks") else "TCP star with every rank on ONE GPU (DOTRING_BENCH_SHARE_GPU=1): not an RCCL / multi-GPU figure",
            "scalar_muls_per_s": n * steps / elapsed, "ms_per_msm": elapsed / steps * 1e3,
            "parity_closed_form_all_ranks": all(s[1] for s in stats)}


def launch_ranks(n: int, argv, script: str | None = None) -> int:
    """`python bench.py --gpus N` as a bare command: start the N rank processes — fresh children of this process, which itself
    never touches the GPU (nothing of dot_ring_amd is imported here) and never execs — with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, relay rank 0's stdout (the JSON line) and return the worst child's exit code."""
    import socket
    import subprocess

    port = None
    for _ in range(64):                                 # MASTER_PORT, + 1 (the control star) and + 2 (the status star) all free
        with socket.socket() as a:
            a.bind(("127.0.0.1", 0))
            cand = a.getsockname()[1]
            if cand >= 65533:
                continue
            try:
                with socket.socket() as b, socket.socket() as c:
                    b.bind(("127.0.0.1", cand + 1))
                    c.bind(("127.0.0.1", cand + 2))
            except (OSError, OverflowError):
                continue
        port = cand
        break
    if port is None:
        print("bench.py: no three free ports in a row for the rank processes", file=sys.stderr)
        return 2
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    import threading

    def relay():                                        # rank 0 prints the one JSON line; anything else it says passes through too
        for text in procs[0].stdout:
            sys.stdout.write(text)
            sys.stdout.flush()

    th = threading.Thread(target=relay, daemon=True)
    th.start()
    deadline = None
    try:
        while any(p.poll() is None for p in procs):
            if deadline is None and any(p.poll() not in (None, 0) for p in procs):
                deadline = time.monotonic() + 60.0      # a rank failed: the others get a minute to notice, then are stopped
            if deadline is not None and time.monotonic() > deadline:
                for p in procs:
                    if p.poll() is None:
                        p.kill()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
        th.join(timeout=10)
    codes = [p.returncode for p in procs]
    worst = max((c if c >= 0 else 128 - c) for c in codes)
    if worst:
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
    return worst


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ring-size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=1024, help="proofs per GPU per step")
    ap.add_argument("--cpu-proofs", type=int, default=16, help="proofs in the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-workers", type=int, default=-1, help="processes of the all-cores CPU baseline (-1 = min(16, cores), 0 = skip)")
    ap.add_argument("--msm-log2n", type=int, default=20, help="secondary G1 MSM size (0 = skip)")
    ap.add_argument("--extras", type=int, default=1, help="0 = skip the secondary legs (bsn_scalar_mul, other ring sizes, distinct signers)")
    args = ap.parse_args()

    if "RANK" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: taking the launcher's {world} ranks", file=sys.stderr)
        args.gpus = world

    # rehearsal on a box with fewer GPUs than ranks (never used by the driver): DOTRING_BENCH_SHARE_GPU=1 puts every rank on
    # device 0 and swaps the RCCL all-gather for the TCP one (RCCL refuses two ranks on one GPU); =rccl shares the device but
    # keeps the RCCL communicator (the test of the failure path)
    share_mode = os.environ.get("DOTRING_BENCH_SHARE_GPU", "")
    share = share_mode == "1"
    if share_mode in ("1", "rccl"):
        local_rank = 0
    os.environ["DOTRING_DEVICE"] = str(local_rank)
    if world > 1 and "DOTRING_HOST_THREADS" not in os.environ:
        # the library's worker threads (hashing between the GPU phases) default to min(16, cores) per process: with one
        # process per GPU share the host's cores between the ranks instead of oversubscribing them
        os.environ["DOTRING_HOST_THREADS"] = str(max(2, min(16, len(os.sched_getaffinity(0)) // world)))
    import dot_ring_amd as d
    from dot_ring_amd import parallel, runtime

    ctx = runtime.context()
    # N > 1: the headline shards PROOFS over the ranks and has no data-path collective; what the ranks exchange for it is the
    # launcher's business — a barrier on both sides of the timed region and 9 bytes of timing per rank — and goes over a TCP
    # star on MASTER_ADDR:MASTER_PORT+1.  RCCL serves the leg that has a real exchange step (the base-sharded MSM below) and
    # is brought up there, after the headline is measured, so that nothing about the collective library can cost the headline.
    ctl = parallel.SocketComm(rank, world, timeout=900.0) if world > 1 else None
    # the outcome of the guarded collective leg travels on a star of its own: a thread abandoned inside that leg may still hold `ctl`
    # (the RCCL bootstrap and the TCP rehearsal use it), and a stale frame of its would desynchronise the status records
    status_star = parallel.SocketComm(rank, world, port=parallel._comm_endpoint()[1] + 1, timeout=900.0) if world > 1 else None
    batch = args.batch

    # ---- setup (untimed): ring, ring root, per-ring prover tables in HBM
    t_setup = time.perf_counter()
    w = RingWorkload(d, args.ring_size, batch, first_index=rank * batch)
    w.step()                       # builds the device prover tables (also those of prove_batch's helper thread)
    setup_s = time.perf_counter() - t_setup

    def barrier():
        for c in runtime.contexts():
            c.sync()
        if ctl is not None:
            ctl.barrier()

    # ---- timed region: per-kernel timers off
    elapsed, all_ok = w.run(args.steps, args.warmup, barrier)
    prove_s, verify_s = w.prove_s, w.verify_s
    if ctl is not None:
        stats = [struct.unpack("<dB", b) for b in ctl.all_gather(struct.pack("<dB", elapsed, 1 if all_ok else 0))]
        elapsed, all_ok = max(s[0] for s in stats), all(s[1] for s in stats)

    rank_ms = None
    if ctl is not None:
        rank_ms = [s_[0] / args.steps * 1e3 for s_ in stats]           # every rank's own timed region (straggling shows here)

    # ---- the same steps again with the per-kernel timers on (every context of this process: prove_batch's helper threads too)
    roof, kernel_ms = ring_roofline(w, args.steps, barrier)
    if ctl is not None:
        # every rank measured its own GPU: the line carries rank 0's roofline in full and the others' figures beside it
        rec = struct.pack("<dddd", roof["frac"], roof["achieved"], roof["avg_kernel_ms"], roof["valu"]["frac"] or 0.0)
        roof["by_rank"] = [dict(zip(("frac", "achieved", "avg_kernel_ms", "valu_frac"), struct.unpack("<dddd", b))) for b in ctl.all_gather(rec)]

    rc = 0
    line = None
    if rank == 0:
        n_dom = w.ring.params.domain_size
        # dense (base, scalar) pairs per proof: quotient 3N+1, two opening quotients 3N + (N-1)  (SURVEY 3.3); the four
        # witness columns (4N) are committed by summation by parts: 4N scalars are read, only ~1.1k bases gathered,
        # so they are priced at the 32 B scalar alone
        parity_ok = all_ok
        cpu = cpu_all = None
        if args.cpu_proofs > 0:
            # (N > 1 as well: rank 0 times the CPU port after the timed region while the other ranks wait at the next exchange)
            m = args.cpu_proofs
            ok, cpu_proofs, cpu_s = w.parity(m)
            parity_ok = parity_ok and ok
            cpu = {
                "value": m / cpu_s, "unit": "proofs/s", "cores": 1, "kind": "port", "work": "prove only",
                "sample": f"{m} proofs of the same workload through oracle/ (Python orchestration + oracle/c kernels for NTT and G1 "
                          f"Pippenger), {cpu_s:.1f} s.  PROVE ONLY — the GPU figure is prove + verify; the reference's verify adds 0.7 % "
                          f"to its prove time (3.99 ms vs 534.57 ms, docs/BENCHMARK.md:72-73)"}
            # the same port on all host cores: one oracle process per core, released together
            if not w.big and args.cpu_workers != 0:
                cpu_all = cpu_baseline_all_cores(w.keys, args.ring_size, w.signer_sk, max(2, m // 4), args.cpu_workers, cpu_proofs)

        total = batch * world * args.steps
        value = total / elapsed
        traffic = traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath))
                traffic = rec.get(f"ringvrf_ring{args.ring_size}_batch{batch}", {}).get("k_g1_accumulate_bytes_per_launch")
                traffic_source = rec.get("_source", "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh, "
                                                    "NOT measured in this run)") if traffic else None
            except Exception:
                traffic = None
        roof["traffic"], roof["traffic_source"] = traffic, traffic_source
        # the clock the chip holds under this kernel, from GRBM_GUI_ACTIVE / duration of its launches in the rocprofv3 --pmc pass
        # (profiles/hbm_traffic.json: "_clock_ghz"); the nominal 2.4 GHz prices the ceiling too high
        clock_ghz, clock_source = 2.0, "default (no _clock_ghz in profiles/hbm_traffic.json)"
        try:
            rec = json.load(open(tpath))
            if rec.get("_clock_ghz"):
                clock_ghz, clock_source = float(rec["_clock_ghz"]), rec.get("_clock_source", "profiles/hbm_traffic.json")
        except Exception:
            pass
        v = roof["valu"]
        v.update({"peak_gadd_s_nominal_clock": VALU_PEAK_GADD_S, "nominal_clock_ghz": 2.4,
                  # at the measured clock the ceiling is priced at the architectural issue rate — one wave-instruction per 4 cycles per SIMD
                  # (16 lanes per cycle) — not at the 4.2 cycles of the v_mad_i64_i32 micro-benchmark; the static instruction count holds
                  # ~150 cold instructions (the exact zero test of the exceptional cases) that a round of the loop does not execute, so a
                  # fraction slightly above 1 means "at the issue rate", not faster than it
                  "peak_gadd_s_at_measured_clock": 1024 * clock_ghz * 1e9 / 4.0 * 64 / MADD_VALU_INSTRUCTIONS / 1e9, "measured_clock_ghz": clock_ghz,
                  "measured_clock_source": clock_source,
                  "frac_at_measured_clock": v["achieved_gadd_s"] / (1024 * clock_ghz * 1e9 / 4.0 * 64 / MADD_VALU_INSTRUCTIONS / 1e9) if v["achieved_gadd_s"] else None,
                  "instructions_source": MADD_SOURCE, "isolated_chain_gadd_s": MEASURED_CHAIN_GADD_S,
                  "isolated_chain_source": "profiles/r02_ubench_limbs_fused.txt (tools/ubench_limbs.hip on another box; not re-measured in this run)"})
        # per-proof unique traffic (11N scalars + 14 NTT passes' data + 784 B out), 3.17 KB per
        # domain point = 6.5 MB per proof at N = 2048, over the whole job
        roof["whole_proof"] = {"algorithmic_bytes_per_proof": 3174 * n_dom, "achieved": 3174 * n_dom * value / 1e9, "unit": "GB/s",
                               "frac": 3174 * n_dom * value / 1e9 / (HBM_PEAK_GBS * world)}
        g1 = bsn = others = distinct = single = pipelined = sweep = msm_batched = threads_sweep = None
        if world == 1 and args.extras:
            # the second half of BASELINE's metric at this ring size: the batched G1 MSM of `batch` 3N-term commitments
            msm_batched = g1_msm_batched_leg(ctx, w.pcs, n_dom, batch, 3, VALU_PEAK_GADD_S)
            parity_ok = parity_ok and msm_batched["parity_ok"]
        if world == 1 and args.msm_log2n > 0:
            g1 = g1_msm_measurement(ctx, args.msm_log2n, 10, 17, True)
            parity_ok = parity_ok and g1.get("parity_closed_form", True) and g1.get("parity_sample", True)
            if args.msm_log2n > 16:             # BASELINE configs[2] names 2^16 as well
                small = g1_msm_measurement(ctx, 16, 20, 0, True)
                parity_ok = parity_ok and small.get("parity_closed_form", True)
                g1["at_2p16"] = {k: small[k] for k in ("pairs", "scalar_muls_per_s", "ms_per_msm", "parity_closed_form", "kernel_ms_per_msm")}
        if world == 1 and args.extras:
            bsn = bsn_scalar_mul_measurement(ctx, d.Bandersnatch, 4096, 20)
            parity_ok = parity_ok and bsn["parity_ok"]
            single = single_call_leg(w)
            parity_ok = parity_ok and single["verified"]
            single["small_vrfs"] = small_vrf_single_call_leg(d)
            parity_ok = parity_ok and single["small_vrfs"]["verified"]
            threads_sweep = host_threads_sweep(args)
            pipelined = pipelined_leg(w, max(3, args.steps))
            parity_ok = parity_ok and pipelined["all_verified"]
            if args.ring_size == 1024 and batch == 1024:
                # the same step at other batch sizes per call: the fixed latencies of a call (head, decoding, host phases) against its batch
                sweep = {}
                for b in (512, 2048):
                    wb = RingWorkload(d, args.ring_size, b)
                    el, okb = wb.run(3, 1)
                    sweep[str(b)] = {"proofs_per_s": b * 3 / el, "ms_per_step": el / 3 * 1e3, "all_verified": bool(okb)}
                    parity_ok = parity_ok and okb
                    del wb

            if args.ring_size == 1024 and batch >= 2:
                # 1024 different signing keys (the reference bench — and the headline — sign every proof with one key)
                w.distinct_signers()
                el, ok_d = w.run(2, 1)
                ok_p, _, _ = w.parity(2)
                distinct = {"proofs_per_s": batch * 2 / el, "prove_only_proofs_per_s": batch * 2 / w.prove_s, "signers": min(batch, args.ring_size),
                            "parity_ok": bool(ok_d and ok_p), "parity_proofs": 2}
                parity_ok = parity_ok and distinct["parity_ok"]
                others = {}
                for rs, st in ((256, 5), (3839, 5)):     # 3839 = the largest ring of domain 4096: BASELINE configs[4]'s per-GPU shape
                    leg = ring_size_leg(d, rs, batch, st, 2, with_roofline=True)
                    others[str(rs)] = leg
                    parity_ok = parity_ok and leg["parity_ok"]
        line = {
            "metric": "ringvrf_proofs_per_sec",
            "value": value,
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_by_rank": rank_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"RingVRF[Bandersnatch] prove_batch + batch_verify, ring_size {args.ring_size} (domain {n_dom}), "
                                   f"{batch} proofs per GPU per step, ZK rows random, ring/SRS/prover tables HBM-resident"
                                   + (", known-tau SRS of 12289 points" if w.big else ""),
                       "ring_size": args.ring_size, "domain_size": n_dom, "batch_per_gpu": batch,
                       "sharding": "proofs sharded per rank, no collective" if world > 1 else "single GPU"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_all,
            "parity_ok": parity_ok,
            "prove_only_proofs_per_s": batch * args.steps / prove_s if prove_s else None,
            "verify_only_proofs_per_s": batch * args.steps / verify_s if verify_s else None,
            "gpu_kernel_ms_per_step": {k: round(v, 3) for k, v in kernel_ms.items() if v > 0.0005},
            "bsn_scalar_mul": bsn,
            "g1_msm": g1,
            "g1_msm_batched": msm_batched,
            "g1_msm_sharded": None,
            "other_ring_sizes": others,
            "distinct_signers": distinct,
            "single_call_ms": single,
            "pipelined_prove_verify": pipelined,
            "batch_sweep": sweep,
            "host_threads": {"this_run": int(os.environ.get("DOTRING_HOST_THREADS", "0")) or min(16, len(os.sched_getaffinity(0))),
                             "sweep": threads_sweep} ,
            "ring_root_s": w.ring_root_s,
            "setup_s": setup_s,
        }
        if not parity_ok:
            print("bench.py: PARITY FAILURE — GPU result differs from the oracle", file=sys.stderr)
            rc = 1

    # ---- N > 1: BASELINE configs[4].  (a) its per-GPU shape — ring 3839 (domain 4096, known-tau SRS), `batch` proofs per rank,
    # prove + verify, proofs sharded per rank with no collective, parity subset on rank 0; (b) the base-sharded MSM, the one
    # place with an exchange step: ncclAllGather of one point per rank through dr_comm_* (TCP star when the ranks share a GPU).
    # The collective part is guarded: a collective library that fails or stalls on this node is reported in the line instead
    # of taking the measured headline with it — and the process then exits NON-ZERO (3 = error, 4 = stalled).
    hung = False
    if ctl is not None and args.extras:
        leg5 = config5_sharded_leg(d, 3839, batch, max(1, min(args.steps, 5)), 2 if args.cpu_proofs > 0 else 0, ctl, barrier)
        if line is not None:
            leg5["rccl_ranks"] = None
            leg5["collective"] = ("the gather of 784 B x B proofs over the launcher's TCP star (no exchange while the GPUs work; RCCL serves the "
                                  "base-sharded MSM, g1_msm_sharded)") + (
                "; every rank on ONE GPU (DOTRING_BENCH_SHARE_GPU rehearsal)" if share_mode in ("1", "rccl") else "")
            line["config5"] = leg5
            if not leg5["parity_ok"]:
                print("bench.py: PARITY FAILURE — config5 leg", file=sys.stderr)
                rc = 1
    if ctl is not None and args.msm_log2n > 0:
        box = {"stage": "start"}

        def sharded_legs():
            box["stage"] = "ncclGetUniqueId / ncclCommInitRank"
            comm = ctl if share else parallel.RcclComm(ctx, rank, world, bootstrap=ctl)
            # strong: 2^msm_log2n pairs in total; weak: 2^msm_log2n pairs per rank (rounded up to a power of two of ranks)
            box["stage"] = "ncclAllGather (strong-scaling MSM)"
            legs = [sharded_msm_leg(ctx, comm, args.msm_log2n, 10, "strong")]
            extra = max(0, (world - 1).bit_length())
            if extra:
                box["stage"] = "ncclAllGather (weak-scaling MSM)"
                legs.append(sharded_msm_leg(ctx, comm, args.msm_log2n + extra, 5, "weak"))
            box["stage"] = "ncclCommDestroy"
            if comm is not ctl:
                comm.close()
            box["legs"] = legs

        def guarded():
            try:
                sharded_legs()
            except Exception as exc:          # noqa: BLE001 — reported in the line
                box["error"] = f"{type(exc).__name__}: {exc}"

        import threading
        th = threading.Thread(target=guarded, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("DOTRING_BENCH_SHARDED_TIMEOUT", "300")))
        hung = th.is_alive()
        state = "timeout" if hung else ("error" if "error" in box else "ok")
        if state != "ok":
            rc = max(rc, 4 if hung else 3)
            print(f"bench.py: rank {rank}: sharded MSM leg {state} in {box['stage']}: {box.get('error', 'no answer within the timeout')}", file=sys.stderr)
        # every rank reaches this point (the guard has a timeout), so the outcome per rank travels over the control star; a
        # rank whose peers failed in a SocketComm rehearsal would otherwise be the only witness
        mine = json.dumps({"rank": rank, "state": state, "stage": box["stage"], "error": box.get("error")}).encode()[:480].ljust(480)
        try:
            reports = [json.loads(b.decode().strip()) for b in status_star.all_gather(mine)]
        except Exception as exc:              # noqa: BLE001 — a peer that is gone: still print, still fail
            reports = [{"rank": rank, "state": state, "stage": box["stage"], "error": box.get("error")},
                       {"rank": None, "state": "error", "stage": "status exchange", "error": f"{type(exc).__name__}: {exc}"}]
        failed = [r_ for r_ in reports if r_["state"] != "ok"]
        if failed:
            rc = max(rc, 3)
        if line is not None:
            if failed:
                line["g1_msm_sharded"] = {"error": "; ".join(f"rank {r_['rank']}: {r_['state']} in {r_['stage']}" + (f" ({r_['error']})" if r_["error"] else "")
                                                             for r_ in failed),
                                          "failed_ranks": [r_["rank"] for r_ in failed], "collective": "SocketComm" if share else "RcclComm",
                                          "legs": box.get("legs")}
            else:
                line["g1_msm_sharded"] = box["legs"]
                if not all(leg["parity_closed_form_all_ranks"] for leg in box["legs"]):
                    print("bench.py: PARITY FAILURE — sharded MSM differs from the closed form", file=sys.stderr)
                    rc = max(rc, 1)
    if line is not None:
        print(json.dumps(line), flush=True)
    if hung:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(rc)                           # a stalled collective holds its thread: leave without joining it — NON-ZERO
    if ctl is not None:
        if rc == 0:
            ctl.barrier()
        ctl.close()
        status_star.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
