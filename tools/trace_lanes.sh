#!/bin/bash
# kernel traces of the headline step in three configurations (one GPU-box call): plain, two lanes, two lanes + chip partition
# usage: bash tools/trace_lanes.sh <tag>   -> gpurun_out/<tag>_{base,lanes2,side16_lanes2}.txt (+ timeline dumps)
tag=${1:-r3_trace}
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
one() {
  label=$1; shift
  rm -rf /tmp/tr_$label
  env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$label -o run -- python3 "$root/bench.py" --steps 3 --warmup 2 --cpu-proofs 0 --msm-log2n 0 --extras 0 > "$root/gpurun_out/${tag}_$label.log" 2>&1
  f=$(find /tmp/tr_$label -name '*kernel_trace.csv' | head -1)
  python3 "$root/tools/trace_timeline.py" "$f" --last-ms 150 --dump "$root/gpurun_out/${tag}_${label}_timeline.txt" > "$root/gpurun_out/${tag}_$label.txt" 2>&1
  tail -n 1 "$root/gpurun_out/${tag}_$label.log" | cut -c1-200
  head -3 "$root/gpurun_out/${tag}_$label.txt"
}
one base DOTRING_SIDE_CUS=0 DOTRING_BENCH_LANES=1 && one lanes2 DOTRING_SIDE_CUS=0 DOTRING_BENCH_LANES=2 && one side16_lanes2 DOTRING_SIDE_CUS=16 DOTRING_BENCH_LANES=2
