#!/bin/bash
# GPU box: one bench workload on several build/flags/libapd_hip_<name>.so variants.  usage: tools/variant_bench.sh <workload> <steps> name...
wl=$1; steps=$2; shift 2
mkdir -p gpurun_out
for v in "$@"; do
  APD_LIB=build/flags/libapd_hip_$v.so timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup 1 --cpu-seconds 0 --census off --secondary off > gpurun_out/vb_${wl}_$v.log 2>&1 || { echo "$wl $v FAILED"; tail -3 gpurun_out/vb_${wl}_$v.log; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/vb_${wl}_$v.log").read().strip().splitlines()[-1])
print("$wl [$v]: kernel %.2f ms  frac %.4f  err %.2e" % (d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["max_rel_err_vs_oracle"]))
PY
done
