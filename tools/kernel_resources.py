#!/usr/bin/env python3
"""Registers / LDS / scratch of every kernel, from the metadata notes of the built gfx950 code objects
(dot_ring_amd/csrc/build/*.gfx950, extracted by tools/count_kernel_insts.py).  `python3 tools/kernel_resources.py [substring]`."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1] if len(sys.argv) > 1 else ""
for co in sorted(glob.glob(os.path.join(ROOT, "dot_ring_amd", "csrc", "build", "*gfx950"))):
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip().split("(")[0]
        if want in name:
            print(f"{name[:64]:64s} vgpr={g('vgpr_count'):>4s} agpr={blk.split()[0]:>3s} sgpr={g('sgpr_count'):>4s} "
                  f"lds={g('group_segment_fixed_size'):>7s} scratch={g('private_segment_fixed_size'):>5s} spill_v={g('vgpr_spill_count'):>3s}")
