#!/bin/bash
# Run ON THE GPU BOX: kernel time of every forced full-matrix geometry on the full-band workloads (calibrates pick_full_key).
for w in short9 full6 ship full8; do
  for v in 20105 20107 20109 20111 20113 20205 20207 20209 20211 20213 20405 20407 20409 20411 20413; do
    python bench.py --workload $w --cpu-seconds 0 --verify 0 --steps 2 --warmup 1 --variant $v > gpurun_out/grid_${w}_$v.log 2>&1
    echo "$w $v $(grep '^{' gpurun_out/grid_${w}_$v.log | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.4f" % j["roofline"]["kernel_ms"], "%.3e" % j["value"])' 2>/dev/null)"
  done
done
