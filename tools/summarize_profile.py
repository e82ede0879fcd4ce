"""Condenses rocprofv3 CSV output (kernel trace stats + PMC passes) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for path in find("trace/**/*kernel_stats.csv"):
    with open(path) as fp:
        for row in csv.DictReader(fp):
            print("%-70s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row.get("Name", "")[:70], row.get("Calls"),
                  row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
print("== per-dispatch (kernel trace) of the alignment kernel ==")
for path in find("trace/**/*kernel_trace.csv"):
    with open(path) as fp:
        for row in csv.DictReader(fp):
            if "apd::dtw_" in row.get("Kernel_Name", ""):   # every alignment kernel: dtw_fused_systolic / _wide / _generic, dtw_full_matrix
                dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                # rocprofv3's VGPR_Count column is NOT the compiler's register count on gfx950 (unified 512-entry file, allocated in
                # blocks of 8): it prints 120 for the kernel -Rpass-analysis=kernel-resource-usage reports at 236 VGPRs / occupancy 2
                # (tools/check_resources.sh).  Both trace columns are shown as the tool gives them, labelled as such.
                print("dur_ns=%d trace_vgpr_field=%s trace_accum_vgpr_field=%s (compiler figures: tools/check_resources.sh) sgpr=%s lds=%s scratch=%s grid=%s wg=%s %s" % (
                    dur, row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"), row.get("Scratch_Size"),
                    row.get("Grid_Size_X"), row.get("Workgroup_Size_X"), row["Kernel_Name"][:60]))
# the dominant alignment kernel of the run (largest total duration): the toy launch bench.py primes the device with, and the
# idle literal-kernel fallback behind every fast launch, are alignment kernels too and must not dilute the per-dispatch figures
dominant, dominant_ns = None, -1
for path in find("trace/**/*kernel_stats.csv"):
    with open(path) as fp:
        for row in csv.DictReader(fp):
            if "apd::dtw_" in row.get("Name", "") and int(row.get("TotalDurationNs") or 0) > dominant_ns:
                dominant, dominant_ns = row["Name"], int(row["TotalDurationNs"])
print("== PMC of the dominant alignment kernel: %s (per-dispatch = sum / n_dispatch) ==" % (dominant or "?")[:90])
for d in find("pmc_*/"):
    sums, n = defaultdict(float), defaultdict(int)
    for path in glob.glob(os.path.join(d, "**/*counter_collection.csv"), recursive=True):
        with open(path) as fp:
            for row in csv.DictReader(fp):
                if (row.get("Kernel_Name", "") == dominant) if dominant else ("apd::dtw_" in row.get("Kernel_Name", "")):
                    sums[row["Counter_Name"]] += float(row["Counter_Value"])
                    n[row["Counter_Name"]] += 1
    for k in sorted(sums):
        print("%-28s per_dispatch=%.6g (n=%d)" % (k, sums[k] / max(n[k], 1), n[k]))
