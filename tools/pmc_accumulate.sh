#!/bin/bash
# PMC counters of the bucket walk under different knob sets, inside ONE GPU-box call:
#   bash tools/pmc_accumulate.sh "X=1" "DOTRING_SRS_BIT_ROWS_MB=0"
# One rocprofv3 --pmc pass (kernel trace only) per counter group and knob set; prints, per knob set, the counters summed over the
# launches of k_g1_accumulate* divided by the number of launches.
root=$(pwd)
export TMPDIR=/tmp
groups=("SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY")
for cfg in "$@"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  for grp in "${groups[@]}"; do
    name=$(echo "$grp" | tr ' ' '+')
    out=$root/gpurun_out/pmcacc/$tag/$name
    mkdir -p "$out"
    ( export $cfg; cd /tmp; rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$out" -o run -- python "$root/bench.py" --steps 1 --warmup 1 --cpu-proofs 0 --cpu-workers 0 --msm-log2n 0 --extras 0 > "$out/log.txt" 2>&1 )
  done
  python3 - "$root/gpurun_out/pmcacc/$tag" "$cfg" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_g1_accumulate" in r["Kernel_Name"] and "heavy" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print(sys.argv[2], {k: round(v / max(1, n[k])) for k, v in sorted(acc.items())}, "launches", dict(n))
PY
done
