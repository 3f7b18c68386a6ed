#!/usr/bin/env python3
"""Instruction counts of the hot loops, taken from the BUILT code objects (llvm-objdump), not from literals.

`python3 tools/count_kernel_insts.py` (run by __graft_entry__.build()) disassembles the gfx950 code object of every translation
unit under dot_ring_amd/csrc/build/, finds the kernels below and, in each, the innermost loop (smallest backward-branch span)
that holds at least `min_mads` v_mad_i64_i32 — the body of one field-operation chain — and writes
dot_ring_amd/kernel_counts.json: total wave-instructions of that loop body in address order, how many of them are VALU, how many
are multiply-adds.  bench.py prices the VALU ceiling of the bucket walk with `k_g1_accumulate.loop_instructions`.
"""
from __future__ import annotations

import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "dot_ring_amd", "csrc", "build")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# kernel (substring of the mangled name) -> minimal number of v_mad_i64_i32 that makes a loop "the" loop
KERNELS = {"15k_g1_accumulateE": 2000, "k_te_msm_accumulateILi1E": 600}
OUT = os.path.join(ROOT, "dot_ring_amd", "kernel_counts.json")

_HEAD = re.compile(r"^([0-9a-f]+) <(\S+)>:")
_INSN = re.compile(r"^\s+(\S+)\s.*//\s*([0-9A-F]+):")
_TARGET = re.compile(r"<\S+\+0x([0-9a-f]+)>")


def device_objects():
    """extract the gfx950 bundle of every .o (llvm-objdump --offloading writes <obj>.0.hipv4-...gfx950 next to it)"""
    outs = []
    for obj in sorted(glob.glob(os.path.join(BUILD, "*.o"))):
        subprocess.run([OBJDUMP, "--offloading", obj], cwd=BUILD, capture_output=True, check=False)
        outs += glob.glob(obj + ".*gfx950")
    return sorted(set(outs))


def loops_of(lines, base):
    """[(head_offset, branch_offset)] of backward branches inside one function's disassembly"""
    loops = []
    for text in lines:
        m = _INSN.match(text)
        if not m or not m.group(1).startswith(("s_cbranch", "s_branch")):
            continue
        t = _TARGET.search(text)
        if not t:
            continue
        here, target = int(m.group(2), 16) - base, int(t.group(1), 16)
        if target < here:
            loops.append((target, here))
    return loops


def count(lines, base, lo, hi):
    total = valu = mads = 0
    for text in lines:
        m = _INSN.match(text)
        if not m:
            continue
        off = int(m.group(2), 16) - base
        if lo <= off <= hi:
            total += 1
            op = m.group(1)
            valu += op.startswith("v_")
            mads += op.startswith("v_mad_i64_i32")
    return total, valu, mads


def main() -> int:
    result = {}
    for co in device_objects():
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout.splitlines()
        heads = [(i, _HEAD.match(t)) for i, t in enumerate(dis)]
        heads = [(i, int(m.group(1), 16), m.group(2)) for i, m in heads if m]
        for k, (i, base, name) in enumerate(heads):
            want = [(key, n) for key, n in KERNELS.items() if key in name]
            if not want:
                continue
            body = dis[i + 1 : heads[k + 1][0] if k + 1 < len(heads) else len(dis)]
            best = None
            for lo, hi in loops_of(body, base):
                total, valu, mads = count(body, base, lo, hi)
                if mads >= want[0][1] and (best is None or hi - lo < best[0]):
                    best = (hi - lo, total, valu, mads)
            if best:
                short = re.sub(r"^_ZN2dr\d+", "", name).split("EPK")[0].split("ILi")[0].split("ILb")[0]
                result[short] = {"loop_instructions": best[1], "loop_valu_instructions": best[2], "loop_v_mad_i64_i32": best[3],
                                 "symbol": name, "source": "llvm-objdump -d of " + os.path.basename(co) + " at build time"}
    with open(OUT, "w") as f:
        json.dump(result, f, indent=1, sort_keys=True)
    print(json.dumps(result))
    return 0 if "k_g1_accumulate" in result else 1


if __name__ == "__main__":
    sys.exit(main())
