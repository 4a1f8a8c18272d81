"""Condense rocprofv3 CSV output (kernel trace / counter collection) into a small text summary
for profiles/.  Usage: python tools/rocprof_summary.py <rocprof output dir> [<label>]
(ROCPROF_SUMMARY_TOP=N: list the N longest kernels instead of 12)"""
import csv
import glob
import os
import sys
from collections import defaultdict

TOP = int(os.environ.get("ROCPROF_SUMMARY_TOP", "12"))


def main():
    d = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else d
    print("# rocprofv3 summary: %s" % label)
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        print("\n## kernel stats (%s)" % os.path.basename(f))
        rows = list(csv.DictReader(open(f)))
        for r in rows[:TOP]:
            print("  %-60s calls=%-6s total_ns=%-14s avg_ns=%-12s pct=%s" % (
                r.get("Name", "")[:60], r.get("Calls"), r.get("TotalDurationNs"),
                r.get("AverageNs"), r.get("Percentage")))
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        agg = defaultdict(lambda: [0, 0.0, None])
        for r in rows:
            dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += dur
            a[2] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"),
                    r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
        print("\n## kernel trace (%s): per-kernel launches, average duration" % os.path.basename(f))
        for k, (n, t, regs) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:TOP]:
            print("  %-60s n=%-5d avg_us=%-10.2f total_ms=%-10.3f vgpr/agpr/sgpr/lds/wg/grid=%s" % (
                k[:60], n, t / n * 1e-3, t * 1e-6, regs))
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for r in rows:
            a = agg[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        print("\n## counters (%s): per-kernel average per dispatch" % os.path.basename(f))
        for k, cs in agg.items():
            for c, (n, v) in cs.items():
                print("  %-50s %-28s n=%-5d avg=%.6g" % (k[:50], c, n, v / n))


if __name__ == "__main__":
    main()
