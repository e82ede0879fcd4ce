#!/bin/bash
# GPU box: strict-mode cfg3 bench for the main library and every build/flags/libapd_hip_<name>.so given.  usage: tools/strict_sweep.sh [steps] name...
steps=${1:-3}; shift
mkdir -p gpurun_out
run() {  # label, lib
  APD_LIB=$2 timeout -k 10 240 python bench.py --workload cfg3 --distance strict --steps $steps --warmup 1 --cpu-seconds 0 --census off --secondary off > gpurun_out/strict_$1.log 2>&1 || { echo "$1 FAILED"; tail -5 gpurun_out/strict_$1.log; return 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/strict_$1.log").read().strip().splitlines()[-1])
print("strict cfg3 [$1]: kernel %.1f ms  frac %.4f  err %.2e" % (d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["max_rel_err_vs_oracle"]))
PY
}
run main "" || exit 1
for v in "$@"; do run $v build/flags/libapd_hip_$v.so || exit 1; done
