#!/usr/bin/env python3
"""bench.py — headline measurement of the Ring-VRF hot path on MI355X.

Workload (BASELINE.json configs[2], the kernel north_star sets the roofline target on):
    one BLS12-381 G1 Pippenger MSM over 2^20 bases per GPU (the KZG commit of RingRoot / ring proofs at the
    size north_star names), scalars uniform in [0, r), bases and scalars resident in HBM before the timed region.
    Bases are synthetic (no SRS of that size ships): base[i] = (first+i)*G1, generated on the GPU, so
    the result has the closed form [sum k_i (first+i)]*G1, which is checked on the CPU after the timed region.
A "step" = one such MSM per rank.  With N > 1 ranks the global MSM is N*2^20 pairs sharded by bases: every
rank reduces its shard to one point, the points are all-gathered (RCCL, 96 B per rank) and summed on every
rank — the only exchange step the path has.  value = pairs processed by all ranks / wall time.

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
ALG_BYTES_PER_PAIR = 128       # 96 B affine base + 32 B scalar


def seeded_scalars(n: int, tag: bytes):
    """n scalars uniform in [0, r): SHAKE256(tag) stream, 48 bytes each reduced mod r (bias < 2^-128)."""
    stream = hashlib.shake_256(b"g1msm" + tag).digest(48 * n)
    vals = [int.from_bytes(stream[48 * i : 48 * i + 48], "little") % FR for i in range(n)]
    return vals, b"".join(v.to_bytes(32, "little") for v in vals)


def be_to_le_points(raw: bytes) -> bytes:
    return b"".join(raw[i : i + 48][::-1] + raw[i + 48 : i + 96][::-1] for i in range(0, len(raw), 96))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=20, help="bases per GPU = 2^log2n")
    ap.add_argument("--cpu-sample-log2", type=int, default=17, help="pairs in the CPU baseline sample (0 = skip)")
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

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from dot_ring_amd import _native, parallel

    ctx = _native.Context(local_rank)
    n = 1 << args.log2n
    first = 1 + rank * n
    t_setup = time.time()
    srs = ctx.srs_synthetic(G1_BE, n, first=first)
    vals, raw = seeded_scalars(n, rank.to_bytes(4, "little"))
    d_scalars = ctx.alloc(32 * n).upload(raw)
    setup_s = time.time() - t_setup

    def barrier():
        ctx.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        part = ctx.g1_msm_dev(srs, d_scalars, n)
        if dist is None:
            return part
        return parallel.combine_partials(part)

    for _ in range(args.warmup):
        step()
    ctx.prof_reset()
    ctx.prof_enable(True)
    barrier()
    t0 = time.perf_counter()
    result = None
    for _ in range(args.steps):
        result = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.prof_enable(False)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # closed-form check needs every rank's sum_i k_i*(first+i)
        local = sum(k * (first + i) for i, k in enumerate(vals)) % FR
        parts = [None] * world
        dist.all_gather_object(parts, local)
        expect_scalar = sum(parts) % FR
    else:
        expect_scalar = sum(k * (first + i) for i, k in enumerate(vals)) % FR

    acc_ms, acc_launches = ctx.prof_get("k_g1_accumulate")
    kernel_ms = {name: ctx.prof_get(name)[0] / max(1, args.steps)
                 for name in ("k_g1_digits", "k_scan", "k_g1_scatter", "k_g1_accumulate", "k_g1_reduce_chunks", "k_g1_reduce_windows")}

    if rank == 0:
        from oracle import coracle

        # ---- parity of the timed result: closed form [sum k_i (first+i)] * G on the CPU oracle
        g_le = be_to_le_points(G1_BE)
        want = coracle.g1_unpack1(bytes(coracle.g1_msm_raw(g_le, expect_scalar.to_bytes(32, "little"), 1)))
        got = None if result is None else (int.from_bytes(result[:48], "big"), int.from_bytes(result[48:], "big"))
        parity_ok = got == want

        # ---- CPU baseline: the oracle's Pippenger (a port of the reference algorithm, 1 thread) on a bounded sample
        cpu = None
        if args.cpu_sample_log2 > 0:
            m = min(n, 1 << args.cpu_sample_log2)
            sample_bases = be_to_le_points(srs.download(0, m))
            t1 = time.perf_counter()
            cpu_out = coracle.g1_msm_raw(sample_bases, raw[: 32 * m], m)
            cpu_s = time.perf_counter() - t1
            gpu_out = ctx.g1_msm_dev(srs, d_scalars, m)
            gpu_le = bytes(96) if gpu_out is None else gpu_out[:48][::-1] + gpu_out[48:][::-1]
            parity_ok = parity_ok and (bytes(cpu_out) == gpu_le)
            cpu = {"value": m / cpu_s, "unit": "scalar-muls/s", "cores": 1, "kind": "port",
                   "sample": f"first 2^{m.bit_length() - 1} (base, scalar) pairs of the same workload, oracle/c signed-bucket Pippenger, {cpu_s:.1f} s"}

        total_pairs = n * world * args.steps
        value = total_pairs / elapsed
        avg_acc_s = (acc_ms / max(1, acc_launches)) / 1e3
        achieved = ALG_BYTES_PER_PAIR * n / avg_acc_s / 1e9 if avg_acc_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"g1_msm_2^{args.log2n}", {}).get("k_g1_accumulate_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "g1_msm_scalar_muls_per_sec",
            "value": value,
            "unit": "scalar-muls/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"BLS12-381 G1 Pippenger MSM, 2^{args.log2n} bases per GPU (KZG commit), scalars uniform mod r, HBM-resident",
                       "pairs_per_step_per_gpu": n, "sharding": "bases sharded per rank, all-gather of one partial point per rank" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "k_g1_accumulate", "avg_kernel_ms": avg_acc_s * 1e3,
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_PAIR * n},
            "cpu_baseline": cpu,
            "parity_ok": parity_ok,
            "kernel_ms_per_step": kernel_ms,
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
