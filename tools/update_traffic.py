#!/usr/bin/env python3
"""Container side: fold the PMC summaries of tools/profile.sh runs into profiles/hbm_traffic.json and copy them under profiles/<round>/.
usage: tools/update_traffic.py <round dir, e.g. r04> <workload>=<gpurun_out/prof_.../summary.txt> ...
Each entry records the sha of the kernel sources it was taken at (bench.py withholds counters whose sha no longer matches)."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_sha)

rnd = sys.argv[1]
path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
data = json.load(open(path))
os.makedirs(os.path.join(ROOT, "profiles", rnd), exist_ok=True)
for arg in sys.argv[2:]:
    wl, src = arg.split("=", 1)
    text = open(src).read()

    def counter(name):
        m = re.search(r"^%s\s+per_dispatch=([0-9.e+]+)" % name, text, re.M)
        return float(m.group(1)) if m else None

    fetch, write, valu = counter("FETCH_SIZE"), counter("WRITE_SIZE"), counter("SQ_INSTS_VALU")
    if fetch is None or write is None:
        raise SystemExit("%s: FETCH_SIZE / WRITE_SIZE missing" % src)
    dst = "profiles/%s/%s_rocprofv3_summary.txt" % (rnd, wl)
    shutil.copy(src, os.path.join(ROOT, dst))
    stats = os.path.join(os.path.dirname(src), "trace")
    for dirpath, _, files in os.walk(stats):
        for f in files:
            if f.endswith("kernel_stats.csv"):
                shutil.copy(os.path.join(dirpath, f), os.path.join(ROOT, "profiles", rnd, "%s_kernel_stats.csv" % wl))
    data[wl] = {"n_gpus": 1, "fetch_size_kb": fetch, "write_size_kb": write,
                "traffic_bytes": int(2 * fetch * 1024 + write * 1024),      # gfx950: FETCH_SIZE counts 16 B/lane reads at half (MI355X_MICROARCH.md)
                "sq_insts_valu": valu, "source": dst, "kernel_source_sha": bench.kernel_source_sha()}
    print(wl, data[wl])
json.dump(data, open(path, "w"), indent=1)
