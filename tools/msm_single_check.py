#!/usr/bin/env python3
"""One G1 MSM over a fixed-base table at several sizes / table widths, random and degenerate scalars (GPU box):
   python3 tools/msm_single_check.py 20:20 20:16 16:16 16:12 ...      (log2n:table_bits)
Per spec: the bench leg (time, kernel table, closed form + oracle prefix) and, for all-equal / 0-1 / r-1 / three-value scalar
vectors, the time of one call and the closed-form check [sum k_i (1 + i)] G."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dot_ring_amd import _native
from oracle import coracle

FR = bench.FR


def closed_form(vals):
    expect = sum(k * (1 + i) for i, k in enumerate(vals)) % FR
    if expect == 0:
        return None
    return coracle.g1_unpack1(bytes(coracle.g1_msm_raw(bench.be_to_le_points(bench.G1_BE), expect.to_bytes(32, "little"), 1)))


def main():
    ctx = _native.Context(0)
    for spec in sys.argv[1:]:
        log2n, bits = (int(x) for x in spec.split(":"))
        os.environ["DOTRING_BENCH_MSM_TABLE"] = str(bits)
        out = bench.g1_msm_measurement(ctx, log2n, 10, 14, True)
        print(f"log2n={log2n} table={bits}: {out['ms_per_msm']:.3f} ms, non-accumulate {out['non_accumulate_share']:.3f}, closed form {out['parity_closed_form']}, "
              f"oracle prefix {out.get('parity_sample')}, kernels {out['kernel_ms_per_msm']}", flush=True)
        n = 1 << log2n
        srs = ctx.srs_synthetic(bench.G1_BE, n, first=1)
        srs.precompute(bits)
        patterns = {"all equal": [0x1234567890ABCDEF1234567890ABCDEF % FR] * n, "0/1 column": [(i * 7919) % 3 % 2 for i in range(n)],
                    "r - 1": [FR - 1] * n, "three values": [(5, FR - 2, 1 << 200)[i % 3] for i in range(n)], "all zero": [0] * n}
        for name, vals in patterns.items():
            raw = b"".join(v.to_bytes(32, "little") for v in vals)
            d = ctx.alloc(32 * n).upload(raw)
            got = ctx.g1_msm_dev(srs, d, n)
            t = time.perf_counter()
            for _ in range(3):
                got = ctx.g1_msm_dev(srs, d, n)
            dt = (time.perf_counter() - t) / 3
            ctx.prof_reset(); ctx.prof_enable(True)
            ctx.g1_msm_dev(srs, d, n)
            ctx.prof_enable(False)
            kern = {k[2:]: round(ctx.prof_get(k)[0], 3) for k in bench.MSM_KERNELS if ctx.prof_get(k)[1]}
            want = closed_form(vals)
            g = None if got is None else (int.from_bytes(got[:48], "big"), int.from_bytes(got[48:], "big"))
            print(f"   {name:13s} {dt * 1e3:8.3f} ms  ({dt * 1e3 / out['ms_per_msm']:.2f} x random)  closed form {g == want}  {kern}", flush=True)
            d.free()
        srs.close()
    ctx.close()


if __name__ == "__main__":
    main()
