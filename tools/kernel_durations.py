"""Per-dispatch durations (min / median / max / sum) of the small kernels from a rocprofv3 --kernel-trace CSV directory:
usage: python tools/kernel_durations.py <rocprof output dir> [name substrings...]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for name in (sys.argv[2:] or ("ls_kernel", "ls_snapshot", "update_lds", "reduce_partials")):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
    d2 = sorted(d)
    print(name, "n", len(d), "min %.1f med %.1f max %.1f sum %.1f us" % (d2[0], d2[len(d2)//2], d2[-1], sum(d)))
    if name == "ls_kernel":
        print("  ", " ".join("%.0f" % v for v in d[:40]))
