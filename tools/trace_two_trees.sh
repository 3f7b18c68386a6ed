#!/bin/bash
# per-launch averages of the bucket-walk kernels of two source trees on ONE box: the working tree and ab_old/ (see ab_commits.sh)
export TMPDIR=/tmp
root=$(pwd)
for t in new old; do
  dir=$root; [ $t = old ] && dir=$root/ab_old
  out=/tmp/trk_$t; rm -rf $out; mkdir -p $out
  ( cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $out -o run -- python $dir/bench.py --steps 3 --warmup 1 --cpu-proofs 0 --cpu-workers 0 --msm-log2n 0 --extras 0 > $out/log.txt 2>&1 )
  python3 - $out $t <<'PY'
import csv,re,sys,glob,collections
tot=collections.defaultdict(float); cnt=collections.defaultdict(int)
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void dr::","").replace("dr::","")
        if ("accumulate" in n or "sort_sets" in n or "reduce" in n) and "te_msm" not in n:
            key=(n, r["Grid_Size_X"]); tot[key]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; cnt[key]+=1
print(sys.argv[2])
for k in sorted(tot, key=lambda k:-tot[k])[:12]: print("   %-44s grid %-8s avg %9.1f us x %d" % (k[0], k[1], tot[k]/cnt[k], cnt[k]))
PY
done
