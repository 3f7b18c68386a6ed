#!/bin/bash
# Run-to-run variance on one box: the headline step K times in fresh processes with DOTRING_TRACE=1; per run the value and the mean of
# every phase of prove_batch / batch_verify over the timed steps.   bash tools/run_variance.sh [runs] [steps]
runs=${1:-4}; steps=${2:-10}
for r in $(seq 1 $runs); do
  DOTRING_TRACE=1 python bench.py --extras 0 --msm-log2n 0 --cpu-proofs 0 --cpu-workers 0 --steps $steps --warmup 1 > gpurun_out/var.json 2> gpurun_out/var.err
  python - $steps <<'PY'
import json, re, sys, collections
steps = int(sys.argv[1])
l = json.loads(open("gpurun_out/var.json").read().strip().splitlines()[-1])
for what in ("prove_batch", "verify_batch"):
    rows = [ln for ln in open("gpurun_out/var.err") if ln.startswith("[dotring] " + what)]
    rows = rows[2:2 + steps]              # setup step, warm-up step, then the timed steps (the profiled pass follows)
    acc = collections.OrderedDict()
    for ln in rows:
        tot = float(re.search(r"total=([0-9.]+)", ln).group(1))
        acc["total"] = acc.get("total", 0) + tot
        for k, v in re.findall(r"([a-zA-Z0-9+* ]+)=([0-9.]+)", ln.split("|", 1)[1]):
            acc[k.strip()] = acc.get(k.strip(), 0) + float(v)
    print(what, " ".join(f"{k}={v / len(rows):.2f}" for k, v in acc.items()))
print("value=%.0f ms=%.2f" % (l["value"], l["ms_per_step"]))
PY
done
