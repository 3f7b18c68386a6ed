#!/bin/bash
# knob A/B of the single-MSM leg (2^20 and 2^16 pairs): bash tools/ab_knobs_msm.sh "X=1" "DOTRING_SRS_LINE=0"
for cfg in "$@"; do
  env $cfg python bench.py --extras 0 --msm-log2n 20 --cpu-proofs 0 --cpu-workers 0 --steps 2 > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - "$cfg" <<PY
import json,sys
l=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
g=l["g1_msm"]
print(sys.argv[1], "2^20: %.3f ms  walk %.3f  |  2^16: %.3f ms" % (g["ms_per_msm"], g["k_g1_accumulate_avg_ms"], g["at_2p16"]["ms_per_msm"]), g["kernel_ms_per_msm"])
PY
done
