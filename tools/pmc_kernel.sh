#!/bin/bash
# PMC counters of one kernel (substring of its name) of a bench step, one rocprofv3 --pmc pass (kernel trace only) per counter group:
#   bash tools/pmc_kernel.sh k_g1_sort_sets_staged "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" ...
# prints per counter the average over the kernel's launches with the largest grid.
root=$(pwd)
export TMPDIR=/tmp
kern=$1; shift
for grp in "$@"; do
  name=$(echo "$grp" | tr ' ' '+')
  out=$root/gpurun_out/pmck/$name
  mkdir -p "$out"
  ( cd /tmp; rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$out" -o run -- python "$root/bench.py" --steps 1 --warmup 1 --cpu-proofs 0 --cpu-workers 0 --msm-log2n 0 --extras 0 > "$out/log.txt" 2>&1 )
done
python3 - "$root/gpurun_out/pmck" "$kern" <<'PY'
import csv, glob, sys, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            vals[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g in sorted(vals):
    print("grid", g, {k: round(sum(v) / len(v)) for k, v in sorted(vals[g].items())}, "launches", max(len(v) for v in vals[g].values()))
PY
