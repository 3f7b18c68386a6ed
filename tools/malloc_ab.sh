#!/bin/bash
# Do the host-side stalls of a noisy box come from mmap / munmap / first-touch faults of the per-call staging vectors?  Alternates the headline
# step with glibc's default thresholds and with thresholds that keep every staging vector in the retained heap.   bash tools/malloc_ab.sh [pairs]
cat /proc/loadavg
for r in $(seq 1 ${1:-3}); do
  echo "-- default allocator"; bash tools/run_variance.sh 1 10
  echo "-- retained heap"; MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TRIM_THRESHOLD_=4294967296 MALLOC_TOP_PAD_=268435456 MALLOC_ARENA_MAX=1 bash tools/run_variance.sh 1 10
done
cat /proc/loadavg
