#!/bin/bash
# kernel trace of the plain headline step: gpurun_out/<tag>_base.txt (summary) and <tag>_base_timeline.txt
tag=${1:-trace}
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/tr_base
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_base -o run -- python3 "$root/bench.py" --steps 3 --warmup 2 --cpu-proofs 0 --msm-log2n 0 --extras 0 > "$root/gpurun_out/${tag}_base.log" 2>&1
f=$(find /tmp/tr_base -name '*kernel_trace.csv' | head -1)
python3 "$root/tools/trace_timeline.py" "$f" --last-ms 150 --gaps 30 --dump "$root/gpurun_out/${tag}_base_timeline.txt" > "$root/gpurun_out/${tag}_base.txt" 2>&1
head -40 "$root/gpurun_out/${tag}_base.txt"
