#!/bin/bash
# A/B of bench.py under two environments: bash tools/ab_bench.sh VAR a b [repeats]
var=$1; a=$2; b=$3; reps=${4:-2}
for i in $(seq $reps); do for v in $a $b; do
  env $var=$v python bench.py --msm-log2n 0 --cpu-proofs 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$var=$v', round(d['value']), 'prove', round(d['prove_only_proofs_per_s']), 'verify', round(d['verify_only_proofs_per_s']), d['parity_ok'], 'setup', round(d['setup_s'],2))"
done; done
