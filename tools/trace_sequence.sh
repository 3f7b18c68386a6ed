#!/bin/bash
# kernel sequence around the dense bucket walks of one bench step (rocprofv3 kernel trace): start offset, duration, gap to the previous kernel
#   bash tools/trace_sequence.sh "DOTRING_SRS_TILING=naf" ...
root=$(pwd)
export TMPDIR=/tmp
for cfg in "$@"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  out=$root/gpurun_out/trseq/$tag
  mkdir -p "$out"
  ( export $cfg; cd /tmp; rocprofv3 --kernel-trace --output-format csv -d "$out" -o run -- python "$root/bench.py" --steps 2 --warmup 1 --cpu-proofs 0 --cpu-workers 0 --msm-log2n 0 --extras 0 > "$out/log.txt" 2>&1 )
  python3 - "$out" "$cfg" <<'PY'
import csv, sys, re, json
out, cfg = sys.argv[1], sys.argv[2]
rows = sorted(csv.DictReader(open(out + "/run_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
t = open(out + "/log.txt").read(); i = t.find('{"metric'); l = json.loads(t[i:t.find("\n", i)])
print("==", cfg, "value=%.0f" % l["value"], "event-timed accumulate %.2f ms/step" % l["gpu_kernel_ms_per_step"]["k_g1_accumulate"])
idx = [i for i, r in enumerate(rows) if "k_g1_accumulate<" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= 2000000]
for i in idx[-3:]:
    t0 = int(rows[i - 4]["Start_Timestamp"])
    seq = []
    for j in range(i - 4, min(len(rows), i + 8)):
        q = rows[j]
        nm = re.sub(r"\(.*", "", q["Kernel_Name"]).replace("void ", "").replace("dr::", "")[:28]
        gap = (int(q["Start_Timestamp"]) - int(rows[j - 1]["End_Timestamp"])) / 1e3
        seq.append("%s[gap %.0f us, %.3f ms, scratch %s]" % (nm, gap, (int(q["End_Timestamp"]) - int(q["Start_Timestamp"])) / 1e6, q["Scratch_Size"]))
    print("   " + "\n   ".join(seq)); print()
PY
done
