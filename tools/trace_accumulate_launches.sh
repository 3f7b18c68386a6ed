#!/bin/bash
# Every launch of the bucket walk in one bench step, under different knob sets, inside ONE GPU-box call:
#   bash tools/trace_accumulate_launches.sh "DOTRING_SRS_TILING=rows" "X=1"
# pass 1: kernel trace (duration and grid of each k_g1_accumulate launch of the last step); passes 2-3: --pmc groups, per launch.
root=$(pwd)
export TMPDIR=/tmp
groups=("SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum")
for cfg in "$@"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  out=$root/gpurun_out/tral/$tag
  mkdir -p "$out/trace"
  ( export $cfg; cd /tmp; rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -o run -- python "$root/bench.py" --steps 1 --warmup 1 --cpu-proofs 0 --cpu-workers 0 --msm-log2n 0 --extras 0 > "$out/trace/log.txt" 2>&1 )
  for grp in "${groups[@]}"; do
    name=$(echo "$grp" | tr ' ' '+')
    mkdir -p "$out/$name"
    ( export $cfg; cd /tmp; rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$out/$name" -o run -- python "$root/bench.py" --steps 1 --warmup 1 --cpu-proofs 0 --cpu-workers 0 --msm-log2n 0 --extras 0 > "$out/$name/log.txt" 2>&1 )
  done
  python3 - "$out" "$cfg" <<'PY'
import csv, glob, sys, collections
out, cfg = sys.argv[1], sys.argv[2]
print("==", cfg)
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_g1_accumulate<" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
rows.sort()
big = [r for r in rows if r[2] >= 500000]
print("  launches (grid lanes, ms), last 12 with >= 500 k lanes:", [(r[2], round(r[1], 3)) for r in big[-12:]])
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_g1_accumulate<" in r["Kernel_Name"]:
            g = int(r.get("Grid_Size", 0) or 0)
            if g >= 500000:
                per[r["Counter_Name"]][g].append(float(r["Counter_Value"]))
for name, by in sorted(per.items()):
    print("  %-18s" % name, {g: [round(v) for v in vals[-4:]] for g, vals in sorted(by.items())})
PY
done
