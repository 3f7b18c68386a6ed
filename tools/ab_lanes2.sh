#!/bin/bash
# A/B of the wide-token experiment (DOTRING_WIDE_TOKEN=1) with lanes and chip partition; same output format as ab_lanes.sh
out=${1:-gpurun_out/ab_lanes2.txt}
: > "$out"
run() {
  label=$1; shift
  env "$@" python3 bench.py --steps 10 --warmup 2 --cpu-proofs 2 --msm-log2n 0 --extras 0 2> gpurun_out/ab_lanes.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['gpu_kernel_ms_per_step']
print('$label', round(d['value']), round(d['ms_per_step'],2), round(d['prove_only_proofs_per_s']), round(d['verify_only_proofs_per_s']), d['parity_ok'], 'acc', k.get('k_g1_accumulate'), 'enc', k.get('k_bsn_encode_to_curve'), 'smul', k.get('k_bsn_scalar_mul'), 'sort', k.get('k_g1_sort_sets'))" >> "$out" || echo "$label FAILED" >> "$out"
}
run base                 DOTRING_SIDE_CUS=0  DOTRING_BENCH_LANES=1
run tok_lanes2           DOTRING_SIDE_CUS=0  DOTRING_BENCH_LANES=2 DOTRING_WIDE_TOKEN=1
run tok_side8_lanes2     DOTRING_SIDE_CUS=8  DOTRING_BENCH_LANES=2 DOTRING_WIDE_TOKEN=1
run tok_side16_lanes2    DOTRING_SIDE_CUS=16 DOTRING_BENCH_LANES=2 DOTRING_WIDE_TOKEN=1
run tok_side16_lanes3    DOTRING_SIDE_CUS=16 DOTRING_BENCH_LANES=3 DOTRING_WIDE_TOKEN=1
run tok_side16_lanes4    DOTRING_SIDE_CUS=16 DOTRING_BENCH_LANES=4 DOTRING_WIDE_TOKEN=1
cat "$out"
