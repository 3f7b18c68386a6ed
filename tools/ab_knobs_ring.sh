#!/bin/bash
# like ab_knobs.sh for another ring size: bash tools/ab_knobs_ring.sh 3839 "X=1" "DOTRING_SRS_BIT_ROWS_MB=0"
ring=$1; shift
for cfg in "$@"; do
  env $cfg python bench.py --ring-size $ring --extras 0 --msm-log2n 0 --cpu-proofs 0 --cpu-workers 0 --steps 4 > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - "$cfg" <<PY
import json,sys
l=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
k=l["gpu_kernel_ms_per_step"]
print(sys.argv[1], "value=%.0f prove_only=%.0f parity=%s" % (l["value"], l["prove_only_proofs_per_s"], l["parity_ok"]), {n:k.get(n) for n in ("k_g1_accumulate","k_g1_reduce_chunks","k_g1_reduce_windows","k_g1_sort_sets")}, l["roofline"]["valu"].get("table"))
PY
done
