"""per-step kernel time table from a rocprofv3 --kernel-trace --stats kernel_stats.csv
usage: python tools/kstats.py <kernel_stats.csv> <steps incl. warmup and instrumented> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time per step: %.3f ms' % (tot / 1e6 / steps))
for r in rows[:nrows]:
    print('%-64s calls/step %6.1f  ms/step %7.3f  avg %8.1f us  %5.1f%%' % (
        r['Name'][:64], float(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3,
        float(r['Percentage'])))
