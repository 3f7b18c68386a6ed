#!/bin/bash
# A/B of two source trees on ONE GPU box: the working tree against a git worktree of an older commit built under ab_old/
# (git worktree add -f ab_old <commit>; make -C ab_old/dot_ring_amd/csrc -j5; make -C ab_old/oracle/c).  Alternates the two
# benches `reps` times: label, proofs/s, ms per step, prove-only, then the NTT / ring kernel times per step.
reps=${1:-3}
out=${2:-gpurun_out/ab_commits.txt}
: > "$out"
one() {
  (cd "$1" && python3 bench.py --steps 10 --warmup 2 --cpu-proofs 2 --cpu-workers 0 --msm-log2n 0 --extras 0 2>/dev/null) | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['gpu_kernel_ms_per_step']
names=('k_ntt_local','k_ntt_strided','k_ring_constraints','k_ring_quotient','k_ring_eval','k_ring_linpoly','k_ring_aggpoly','k_syndiv','k_g1_sort_sets')
print('$2', round(d['value']), round(d['ms_per_step'],2), round(d['prove_only_proofs_per_s']), d['parity_ok'], 'acc', k.get('k_g1_accumulate'), 'k6-k8', round(sum(k.get(n,0) for n in names),2), [k.get(n) for n in names], 'kernel sum', round(sum(k.values()),2))" >> "$out"
}
for i in $(seq $reps); do one ab_old old; one . new; done
cat "$out"
