"""GPU idle time between kernels from a rocprofv3 kernel trace: for the last N ttm_kernel-to-ttm_kernel
periods, the busy time (sum of kernel durations) against the wall span, and the largest gaps with the
kernels on either side.  Usage: python tools/kernel_gaps.py <rocprof output dir>"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    # steady state: the second half of the trace
    rows = rows[len(rows) // 2:]
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    print("steady-state window: %d kernels, span %.3f ms, busy %.3f ms, idle %.3f ms (%.1f %%)" % (
        len(rows), span * 1e-6, busy * 1e-6, (span - busy) * 1e-6, 100.0 * (span - busy) / span))
    gaps = []
    for (s0, e0, k0), (s1, e1, k1) in zip(rows, rows[1:]):
        gaps.append((s1 - e0, k0.split("(")[0][-40:], k1.split("(")[0][-40:]))
    agg = {}
    for g, a, b in gaps:
        t = agg.setdefault((a, b), [0, 0])
        t[0] += 1
        t[1] += g
    print("gaps by (previous kernel -> next kernel): count, total us, mean us")
    for (a, b), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        print("  %-42s -> %-42s n=%-4d total=%9.1f mean=%7.1f" % (a, b, n, t * 1e-3, t * 1e-3 / n))


if __name__ == "__main__":
    main()
