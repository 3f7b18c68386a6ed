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
With N ranks every rank proves and verifies its own 1024 proofs (independent units, no collective).

Secondary measurement in the same JSON line ("g1_msm"): one G1 Pippenger MSM over 2^20 synthetic bases
(configs[2]), the kernel north_star puts the roofline target on.

Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 through torch.distributed.run (one rank per GPU).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
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
VALU_PEAK_GADD_S = 1024 * 2.4e9 / 4 * 64 / 6800 / 1e9      # ~5.78 G mixed additions/s (see roofline.valu)
ALG_BYTES_PER_PAIR = 128       # 96 B affine base + 32 B scalar per (base, scalar) pair
MSM_KERNELS = ("k_g1_digits", "k_scan", "k_g1_scatter", "k_g1_sort_sets", "k_size_sort", "k_g1_accumulate", "k_g1_reduce_chunks",
               "k_g1_reduce_windows", "k_g1_horner", "k_g1_results_affine")
RING_KERNELS = ("k_bsn_scalar_mul", "k_bsn_encode_to_curve", "k_bsn_msm_groups", "k_ring_chain", "k_ring_columns", "k_ntt_local",
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
    """Signer key pair + ring keys of the reference bench (bench_ring_proof.py:60-77), keys derived on the GPU."""
    from dot_ring_amd.curve import scalar_mul_batch
    from dot_ring_amd.vrf.primitives import secret_from_seed_scalar

    signer_pk, signer_sk = cv.secret_from_seed(_seed("signer", sample_index, 0))
    secrets_ = [secret_from_seed_scalar(cv, _seed("ring-member", sample_index, i)) for i in range(ring_size)]
    keys = [p.point_to_string() for p in scalar_mul_batch([cv.point_type.generator_point()] * ring_size, secrets_)]
    keys[min(3, ring_size - 1)] = signer_pk
    return signer_pk, signer_sk, keys


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
        return {"value": workers * per_worker / wall, "unit": "proofs/s", "cores": workers, "kind": "port",
                "sample": f"{workers} oracle processes x {per_worker} proofs each (prove only), released together, {wall:.1f} s wall",
                "matches_single_process": outs[0][2] == want}
    except Exception as exc:          # a baseline that cannot run is reported, not fatal
        return {"value": None, "unit": "proofs/s", "cores": workers, "kind": "port", "sample": f"failed: {exc}"}
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        os.unlink(job)


def g1_msm_measurement(ctx, log2n: int, steps: int, cpu_sample_log2: int, do_cpu: bool):
    """Secondary: G1 MSM at 2^log2n synthetic bases. Returns a dict (rank-local)."""
    n = 1 << log2n
    srs = ctx.srs_synthetic(G1_BE, n, first=1)
    # fixed-base window table in HBM (W * n * 96 B = 1.6 GB at 2^20 with 16-bit windows): one bucket set per MSM
    srs.precompute(16 if log2n >= 18 else 12)
    vals, raw = seeded_scalars(n, b"\0\0\0\0")
    d_scalars = ctx.alloc(32 * n).upload(raw)
    ctx.g1_msm_dev(srs, d_scalars, n)
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    result = None
    for _ in range(steps):
        result = ctx.g1_msm_dev(srs, d_scalars, n)
    elapsed = time.perf_counter() - t0
    ctx.prof_enable(False)
    acc_ms, acc_n = ctx.prof_get("k_g1_accumulate")
    out = {"pairs": n, "scalar_muls_per_s": n * steps / elapsed, "ms_per_msm": elapsed / steps * 1e3,
           "k_g1_accumulate_avg_ms": acc_ms / max(1, acc_n),
           "roofline_frac_hbm": (ALG_BYTES_PER_PAIR * n / (acc_ms / max(1, acc_n) / 1e3) / 1e9) / HBM_PEAK_GBS if acc_ms else None}
    if do_cpu:
        from oracle import coracle

        expect = sum(k * (1 + i) for i, k in enumerate(vals)) % FR
        want = coracle.g1_unpack1(bytes(coracle.g1_msm_raw(be_to_le_points(G1_BE), expect.to_bytes(32, "little"), 1)))
        got = None if result is None else (int.from_bytes(result[:48], "big"), int.from_bytes(result[48:], "big"))
        out["parity_closed_form"] = got == want
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
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            return 2
        args.gpus = world

    dist = torch = None
    if world > 1:
        import torch
        import torch.distributed as dist

        # rehearsal on a box with fewer GPUs than ranks (never used by the driver): DOTRING_BENCH_SHARE_GPU=1 puts every
        # rank on device 0 and swaps RCCL for gloo (RCCL refuses two ranks on one GPU)
        share = os.environ.get("DOTRING_BENCH_SHARE_GPU") == "1"
        if share:
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        red_device = "cpu" if share else "cuda"

    os.environ["DOTRING_DEVICE"] = str(local_rank)
    if world > 1 and "DOTRING_HOST_THREADS" not in os.environ:
        # the library's worker threads (hashing between the GPU phases) default to min(16, cores) per process: with one
        # process per GPU share the host's cores between the ranks instead of oversubscribing them
        os.environ["DOTRING_HOST_THREADS"] = str(max(2, min(16, len(os.sched_getaffinity(0)) // world)))
    import dot_ring_amd as d
    from dot_ring_amd import runtime
    from dot_ring_amd.ring_proof.pcs import SRS

    ctx = runtime.context()
    cv = d.Bandersnatch
    vrf = d.RingVRF[cv]
    batch = args.batch

    # ---- setup (untimed): ring, ring root, per-ring prover tables in HBM
    t_setup = time.perf_counter()
    signer_pk, signer_sk, keys = bench_ring_keys(cv, args.ring_size, 0)
    # domains above 2048 need more SRS points than the shipped file holds: known-tau SRS, same on every rank
    big = d.RingProofParams.from_ring_size(args.ring_size).domain_size > 2048
    tau = int.from_bytes(hashlib.sha256(b"bench-known-tau").digest(), "little") % d.KZG.scalar_modulus
    pcs = d.KZG.with_srs(SRS.synthetic(tau, 3 * 4096 + 1)) if big else d.KZG
    params = d.RingProofParams.from_ring_size(args.ring_size, pcs=pcs)
    t_ring = time.perf_counter()
    ring = d.Ring(keys, params)
    root = d.RingRoot.from_ring(ring, params)
    ring_root_s = time.perf_counter() - t_ring
    base = rank * batch
    alphas = [b"bench-batch-input" + (base + i).to_bytes(8, "little") for i in range(batch)]
    ads = [b"bench-batch-ad" + (base + i).to_bytes(8, "little") for i in range(batch)]
    sks, pks = [signer_sk] * batch, [signer_pk] * batch
    vrf.prove_batch(alphas, ads, sks, pks, ring, root)          # builds the device prover tables (also those of prove_batch's helper thread)
    setup_s = time.perf_counter() - t_setup

    def barrier():
        ctx.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    prove_s = verify_s = 0.0

    def step():
        nonlocal prove_s, verify_s
        t = time.perf_counter()
        proofs = vrf.prove_batch(alphas, ads, sks, pks, ring, root)
        t1 = time.perf_counter()
        ok = vrf.batch_verify(proofs, alphas, ads, ring, root)
        t2 = time.perf_counter()
        prove_s += t1 - t
        verify_s += t2 - t1
        return proofs, ok

    for _ in range(args.warmup):
        step()
    prove_s = verify_s = 0.0
    # per-kernel timers of every context of this process (prove_batch runs its two halves on two threads, each with its own)
    all_ctx = runtime.contexts()
    for c in all_ctx:
        c.prof_reset()
        c.prof_enable(True)
    barrier()
    t0 = time.perf_counter()
    all_ok = True
    proofs = None
    for _ in range(args.steps):
        proofs, ok = step()
        all_ok = all_ok and ok
    barrier()
    elapsed = time.perf_counter() - t0
    for c in all_ctx:
        c.prof_enable(False)

    def prof_sum(name):
        ms = cnt = 0
        for c in all_ctx:
            m_, n_ = c.prof_get(name)
            ms, cnt = ms + m_, cnt + n_
        return ms, cnt

    if dist is not None:
        t = torch.tensor([elapsed, 0.0 if all_ok else 1.0], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, all_ok = float(t[0].item()), float(t[1].item()) == 0.0

    kernel_ms = {name: prof_sum(name)[0] / max(1, args.steps) for name in MSM_KERNELS + RING_KERNELS}
    acc_ms, acc_launches = prof_sum("k_g1_accumulate")

    if rank == 0:
        n_dom = ring.params.domain_size
        # dense (base, scalar) pairs per proof: quotient 3N+1, two opening quotients 3N + (N-1)  (SURVEY 3.3); the four
        # witness columns (4N) are committed by summation by parts: 4N scalars are read, only ~1.1k bases gathered,
        # so they are priced at the 32 B scalar alone
        pairs_per_proof = 7 * n_dom
        scalar_only_per_proof = 4 * n_dom
        # ---- parity subset: deterministic proofs (test_vectors=True) byte-compared with the CPU oracle
        from oracle.pyref import bandersnatch as obsn
        from oracle.pyref import ring as oring

        parity_ok = all_ok
        cpu = cpu_all = None
        if args.cpu_proofs > 0:
            tv_params = d.RingProofParams.from_ring_size(args.ring_size, test_vectors=True, pcs=pcs)
            tv_ring = d.Ring(keys, tv_params)
            tv_root = d.RingRoot.from_ring(tv_ring, tv_params)
            m = args.cpu_proofs if world == 1 else min(2, args.cpu_proofs)     # N > 1: parity check only, no CPU timing
            gpu_proofs = vrf.prove_batch(alphas[:m], ads[:m], sks[:m], pks[:m], tv_ring, tv_root)
            o_srs = None
            if big:
                from oracle.pyref import kzg as okzg
                o_srs = okzg.SRS.from_tau(tau, 3 * 4096 + 1)
                o_srs.g2_raw = list(pcs.srs.g2_raw)
            o_params = oring.Params.from_ring_size(args.ring_size, test_vectors=True, suite=obsn.SHA512, srs=o_srs)
            o_ring = oring.Ring(keys, o_params)
            o_root = oring.RingRoot(o_ring)
            parity_ok = parity_ok and o_root.encode() == tv_root.encode() == root.encode()
            t1 = time.perf_counter()
            cpu_proofs = [oring.ring_vrf_prove(o_ring, o_root, alphas[i], ads[i], signer_sk) for i in range(m)]
            cpu_s = time.perf_counter() - t1
            parity_ok = parity_ok and [p.encode() for p in gpu_proofs] == cpu_proofs
            cpu = None if world > 1 else {"value": m / cpu_s, "unit": "proofs/s", "cores": 1, "kind": "port",
                   "sample": f"{m} proofs (prove only) of the same workload through oracle/ (Python orchestration + oracle/c "
                             f"kernels for NTT and G1 Pippenger), {cpu_s:.1f} s"}

            # the same port on all host cores: one oracle process per core, released together
            if world == 1 and not big and args.cpu_workers != 0:
                cpu_all = cpu_baseline_all_cores(keys, args.ring_size, signer_sk, max(2, m // 4), args.cpu_workers, cpu_proofs)

        # bucket additions of the dense MSMs in the timed region: pairs x windows of the SRS table (non-zero digit rate ~1)
        table_windows = -(-256 // int(os.environ.get("DOTRING_SRS_WINDOW", "12") or 12))
        dense_adds = float(batch) * pairs_per_proof * table_windows * args.steps
        total = batch * world * args.steps
        value = total / elapsed
        avg_acc_s = (acc_ms / max(1, acc_launches)) / 1e3
        pairs_per_launch = (batch * pairs_per_proof * args.steps + 0.0) / max(1, acc_launches)      # prove-side MSM pairs / launches
        alg_bytes_launch = ALG_BYTES_PER_PAIR * pairs_per_launch + 32.0 * batch * scalar_only_per_proof * args.steps / max(1, acc_launches)
        achieved = alg_bytes_launch / avg_acc_s / 1e9 if avg_acc_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"ringvrf_ring{args.ring_size}_batch{batch}", {}).get("k_g1_accumulate_bytes_per_launch")
            except Exception:
                traffic = None
        g1 = None
        if args.msm_log2n > 0 and world == 1:
            g1 = g1_msm_measurement(ctx, args.msm_log2n, 10, 17, True)
            parity_ok = parity_ok and g1.get("parity_closed_form", True) and g1.get("parity_sample", True)
            if args.msm_log2n > 16:             # BASELINE configs[2] names 2^16 as well
                small = g1_msm_measurement(ctx, 16, 20, 0, True)
                parity_ok = parity_ok and small.get("parity_closed_form", True)
                g1["at_2p16"] = {k: small[k] for k in ("pairs", "scalar_muls_per_s", "ms_per_msm", "parity_closed_form")}
        line = {
            "metric": "ringvrf_proofs_per_sec",
            "value": value,
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"RingVRF[Bandersnatch] prove_batch + batch_verify, ring_size {args.ring_size} (domain {n_dom}), "
                                   f"{batch} proofs per GPU per step, ZK rows random, ring/SRS/prover tables HBM-resident"
                                   + (", known-tau SRS of 12289 points" if big else ""),
                       "ring_size": args.ring_size, "domain_size": n_dom, "batch_per_gpu": batch,
                       "sharding": "proofs sharded per rank, no collective" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "k_g1_accumulate", "avg_kernel_ms": avg_acc_s * 1e3,
                         "launches_per_step": acc_launches / max(1, args.steps), "algorithmic_bytes_per_launch": alg_bytes_launch,
                         # the kernel is integer-VALU bound, so the informative ceiling is the VALU issue rate: mixed
                         # additions/s against 1024 SIMDs x 2.4 GHz / 4 cycles x 64 lanes / ~6800 instructions per
                         # XYZZ mixed addition (10 Montgomery products of 649 instructions + ~310 add/sub/select)
                         "valu": {"achieved_gadd_s": dense_adds / (acc_ms / 1e3) / 1e9 if acc_ms else None, "peak_gadd_s": VALU_PEAK_GADD_S,
                                  "frac": dense_adds / (acc_ms / 1e3) / 1e9 / VALU_PEAK_GADD_S if acc_ms else None,
                                  "note": "dense bucket additions only (7N pairs x windows per proof); by-parts and verify-side additions not counted"},
                         # per-proof unique traffic (11N scalars + 14 NTT passes' data + 784 B out), 3.17 KB per
                         # domain point = 6.5 MB per proof at N = 2048, over the whole job
                         "whole_proof": {"algorithmic_bytes_per_proof": 3174 * n_dom, "achieved": 3174 * n_dom * value / 1e9, "unit": "GB/s",
                                         "frac": 3174 * n_dom * value / 1e9 / (HBM_PEAK_GBS * world)}},
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_all,
            "parity_ok": parity_ok,
            "prove_only_proofs_per_s": batch * args.steps / prove_s if prove_s else None,
            "verify_only_proofs_per_s": batch * args.steps / verify_s if verify_s else None,
            "gpu_kernel_ms_per_step": {k: round(v, 3) for k, v in kernel_ms.items() if v > 0.0005},
            "g1_msm": g1,
            "ring_root_s": ring_root_s,
            "setup_s": setup_s,
        }
        print(json.dumps(line))
        if not parity_ok:
            print("bench.py: PARITY FAILURE — GPU result differs from the oracle", file=sys.stderr)
            return 1
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
