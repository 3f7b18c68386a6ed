#!/bin/bash
# rocprofv3 kernel-trace statistics of one bench step under different knob sets, inside ONE GPU-box call:
#   bash tools/trace_knobs.sh "X=1" "DOTRING_SRS_BIT_ROWS_MB=0"
# prints the per-kernel totals (ms per step over 3 steps) of the kernels named in the filter.
root=$(pwd)
export TMPDIR=/tmp
for cfg in "$@"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  out=$root/gpurun_out/trk/$tag
  mkdir -p "$out"
  ( export $cfg; cd /tmp; rocprofv3 --kernel-trace --output-format csv -d "$out" -o run -- python "$root/bench.py" --steps 3 --warmup 1 --cpu-proofs 0 --cpu-workers 0 --msm-log2n 0 --extras 0 > "$out/log.txt" 2>&1 )
  python3 - "$out" "$cfg" <<'PY'
import csv, glob, sys, collections, re
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("dr::", "")
        tot[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; cnt[name] += 1
print(sys.argv[2])
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    print("   %-46s %8.2f ms total  %5d launches" % (k, v, cnt[k]))
PY
done
