#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + PMC passes for one bench workload.
# usage: tools/profile.sh <workload> <tag>     -> gpurun_out/prof_<tag>/...
set -o pipefail
wl=${1:-cfg3}; tag=${2:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CMD="python3 bench.py --workload $wl --steps 2 --warmup 1 --cpu-seconds 0 --verify 0 --census off --secondary off $APD_PROFILE_ARGS"   # APD_PROFILE_ARGS: e.g. --distance strict
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $CMD > $out/trace.log 2>&1 || { echo trace failed; tail -20 $out/trace.log; exit 1; }
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/pmc_$name -- $CMD > $out/pmc_$name.log 2>&1 || { echo "pmc pass $name failed"; tail -5 $out/pmc_$name.log; }
done
python3 tools/summarize_profile.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
