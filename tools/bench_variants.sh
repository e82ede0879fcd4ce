#!/bin/bash
# usage: tools/bench_variants.sh <workload> <steps> v1 v2 ...   (run on the GPU box; one JSON summary line per variant)
wl=$1; steps=$2; shift 2
mkdir -p gpurun_out
for v in "$@"; do
  timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup 1 --cpu-seconds 0 --variant $v $APD_BENCH_EXTRA > gpurun_out/bench_${wl}_v$v.log 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "variant $v TIMEOUT"; exit 1; fi
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_${wl}_v$v.log").read().strip().splitlines()[-1])
    print("$wl variant $v: %.3e cells/s  step %.1f ms  kernel %.2f ms  frac %.3f  err %.2e" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["max_rel_err_vs_oracle"]))
except Exception as e:
    print("variant $v failed", e); print(open("gpurun_out/bench_${wl}_v$v.log").read()[-1500:])
PY
done
