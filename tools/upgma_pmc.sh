#!/bin/bash
# GPU box: memory-side counters of the UPGMA kernels on tools/upgma_scale.py's matrix (graph replay off: the profiler crashes inside
# hipGraphLaunch on runs of this length).  usage: tools/upgma_pmc.sh <tag> <upgma_scale args>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/pmc_upgma_$tag
mkdir -p $out
export APD_UPGMA_NO_GRAPH=1
for pass in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/$name -- python3 tools/upgma_scale.py "$@" > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; }
  echo "pass $name done"
done
python3 - $out > $out.txt <<'PY'
import csv, glob, re, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        m = re.search(r"upgma_\w+", r["Kernel_Name"])
        k = (m.group(0) if m else "other", r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for (kern, ctr), (v, c) in sorted(acc.items()):
        if "upgma" in kern: print("%-42s %-26s launches %7d  sum %.4e  per launch %.4e" % (kern, ctr, c, v, v / c))
PY
cat $out.txt
[ -s $out.txt ] || { find $out | head -20; tail -n 5 $out/*.log; }
rm -rf $out
