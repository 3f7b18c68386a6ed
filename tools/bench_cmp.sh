run() { python bench.py "$@" > gpurun_out/cmp.json 2> gpurun_out/cmp.err; python - "$*" <<PY
import json,sys
l=json.loads(open("gpurun_out/cmp.json").read().strip().splitlines()[-1])
print(sys.argv[1], "| value=%.0f ms=%.2f prove=%.0f verify=%.0f" % (l["value"], l["ms_per_step"], l["prove_only_proofs_per_s"], l["verify_only_proofs_per_s"]))
PY
}
run
run --extras 0 --msm-log2n 0 --cpu-proofs 2 --cpu-workers 0 --steps 10
run --extras 0 --msm-log2n 0 --cpu-proofs 2 --cpu-workers 0 --steps 5
run --steps 20 --warmup 2
run
