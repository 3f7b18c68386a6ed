#!/bin/bash
# One measurement pass on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02
# 1. bench.py (default flags) -> gpurun_out/<tag>/bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command -> gpurun_out/<tag>/stats/
# 3. one rocprofv3 --pmc pass per counter group (kernel-trace only, as the pool requires) -> gpurun_out/<tag>/pmc_<group>/
# tools/summarize_profiles.py then turns these into the files committed under profiles/.
set -eo pipefail
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python bench.py > "$out/bench.json" 2> "$out/bench.err"
tail -n 1 "$out/bench.json"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python "$root/bench.py" --steps 2 --warmup 1 --cpu-proofs 0 --msm-log2n 0 --extras 0 > "$out/stats.log" 2>&1
echo "stats done"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  name=$(echo "$grp" | tr ' ' '+')
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$out/pmc_$name" -o run -- python "$root/bench.py" --steps 1 --warmup 1 --cpu-proofs 0 --msm-log2n 0 --extras 0 > "$out/pmc_$name.log" 2>&1
  echo "pmc $name done"
done
