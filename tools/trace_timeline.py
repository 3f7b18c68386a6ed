#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV: how busy the GPU was, where it idled, what overlapped.

    python3 tools/trace_timeline.py <kernel_trace.csv> [--last-ms 160] [--gaps 12] [--dump timeline.txt]

Looks at the last `--last-ms` milliseconds of kernel activity (the bench's final steps).  Prints: the busy fraction (union of all
kernel intervals), the time with >= 2 kernels in flight, per-queue busy time, the largest idle gaps with the kernels before and
after them, and per-kernel totals (count, sum, sum while another queue's kernel was also running)."""
import argparse
import csv
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--last-ms", type=float, default=160.0)
    ap.add_argument("--gaps", type=int, default=12)
    ap.add_argument("--dump", default=None)
    args = ap.parse_args()
    rows = []
    with open(args.csv, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    if not rows:
        print("no kernels in the trace")
        return 1
    rows.sort()
    t_end = max(r[1] for r in rows)
    t_lo = t_end - int(args.last_ms * 1e6)
    rows = [r for r in rows if r[1] > t_lo]
    t0 = rows[0][0]
    span = (t_end - t0) / 1e6
    # sweep: busy union, overlap time
    ev = []
    for s, e, _, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, busy, multi = 0, t0, 0, 0
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            multi += t - last
        depth += d
        last = t
    print(f"window {span:.2f} ms, {len(rows)} kernels: busy {busy / 1e6:.2f} ms ({100 * busy / (t_end - t0):.1f} %), "
          f">= 2 kernels in flight {multi / 1e6:.2f} ms, idle {span - busy / 1e6:.2f} ms")
    per_q = defaultdict(int)
    for s, e, _, q in rows:
        per_q[q] += e - s
    print("per queue (sum of kernel durations, ms):", {q: round(v / 1e6, 2) for q, v in sorted(per_q.items())})
    # idle gaps
    gaps = []
    cur_end, cur_name = rows[0][1], rows[0][2]
    for s, e, n, q in rows[1:]:
        if s > cur_end:
            gaps.append((s - cur_end, cur_end, cur_name, n))
        if e > cur_end:
            cur_end, cur_name = e, n
    gaps.sort(reverse=True)
    print(f"largest idle gaps (of {len(gaps)}, total {sum(g[0] for g in gaps) / 1e6:.2f} ms):")
    for g, at, before, after in gaps[: args.gaps]:
        print(f"  {g / 1e3:8.1f} us at +{(at - t0) / 1e6:8.2f} ms   after {before:28s} before {after}")
    # per kernel
    tot = defaultdict(lambda: [0, 0])
    for s, e, n, q in rows:
        tot[n][0] += 1
        tot[n][1] += e - s
    print("per kernel (count, total ms, avg us):")
    for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"  {n:32s} {c:5d} {d / 1e6:9.2f} {d / c / 1e3:9.1f}")
    if args.dump:
        with open(args.dump, "w") as f:
            for s, e, n, q in rows:
                f.write(f"{(s - t0) / 1e6:10.3f} {(e - t0) / 1e6:10.3f} {(e - s) / 1e3:9.1f}us q{q} {n}\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
