"""Print rocprofv3 kernel_stats.csv rows (name, calls, avg ms, total ms, %) for kernels matching a substring."""
import csv, sys
f, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "kz_")
for r in csv.DictReader(open(f)):
    if pat in r["Name"]:
        print("%-64s calls %4s avg_ms %9.3f total_ms %9.2f %6s%%" % (r["Name"].split("(")[0][:64], r["Calls"], float(r["AverageNs"]) / 1e6,
                                                                    float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
