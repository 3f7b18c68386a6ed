"""One fresh-process timing of BASELINE configs[2] (bench.g1_msm_measurement at 2^16 and 2^20 on SURVEY 8(d)'s bases), one JSON line:
run it several times on ONE box to separate run-to-run noise from a change.   python3 tools/msm_repeat.py"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
from dot_ring_amd import runtime
ctx = runtime.context()
out = {}
for log2n, steps in ((16, 20), (20, 10)):
    r = bench.g1_msm_measurement(ctx, log2n, steps, 0, True)
    out[str(log2n)] = {"ms_per_msm": round(r["ms_per_msm"], 4), "parity": r["parity_closed_form"], "kernel_ms": r["kernel_ms_per_msm"]}
print(json.dumps(out), flush=True)
