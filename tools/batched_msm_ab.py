#!/usr/bin/env python3
"""The batched G1 MSM of a 1024-proof batch on its own (bench.py's g1_msm_batched leg), for A/B runs of tiling knobs on one box:
   DOTRING_SRS_TILING=rows python3 tools/batched_msm_ab.py 2048 [batch] [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import dot_ring_amd as d
from dot_ring_amd import runtime

domain = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
out = bench.g1_msm_batched_leg(runtime.context(), d.KZG, domain, batch, steps, bench.VALU_PEAK_GADD_S)
print(os.environ.get("DOTRING_SRS_TILING", "default"), "domain", domain, "ms/call %.3f" % out["ms_per_call"], "pairs/s %.3e" % out["scalar_muls_per_s"], "parity", out["parity_ok"],
      out["kernel_ms_per_call"], "valu frac %.3f" % out["roofline"]["valu"]["frac"], out["table"]["tiling"], out["table"]["tiling_bits"])
