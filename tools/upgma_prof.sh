#!/bin/bash
# GPU box: kernel-trace stats of apd_clustering on the synthetic blob matrix.  usage: tools/upgma_prof.sh <tag> <n> [more upgma_scale args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/prof_upgma_$tag
echo "profiling upgma_scale $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/upgma_scale.py "$@" > $out.log 2>&1
grep "us/merge" $out.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-150 $f | head -6
