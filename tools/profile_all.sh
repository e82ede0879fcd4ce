#!/bin/bash
# GPU box: tools/profile.sh for several workloads, one progress line each.  usage: tools/profile_all.sh <round tag> wl...
tag=$1; shift
for wl in "$@"; do
  echo "== profiling $wl =="
  tools/profile.sh $wl ${tag}_$wl > gpurun_out/prof_${tag}_$wl.out 2>&1 || { echo "profile of $wl failed"; tail -5 gpurun_out/prof_${tag}_$wl.out; exit 1; }
  grep -E "calls=|FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU |TCC_HIT|TCC_MISS" gpurun_out/prof_${tag}_$wl/summary.txt | head -12
done
