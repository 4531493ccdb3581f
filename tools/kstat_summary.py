"""Per-kernel summary of a rocprofv3 kernel_stats.csv:  python tools/kstat_summary.py <csv> [n]"""
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 28
for r in rows[1:n + 1]:
    name = r[0].split('(')[0].replace('orbfe::', '').replace('void ', '')
    print("%-30s calls %4s avg %9.1f us  %6s%%  min %8.1f max %8.1f" % (name[:30], r[1], float(r[3]) / 1e3, r[4][:5], float(r[5]) / 1e3, float(r[6]) / 1e3))
