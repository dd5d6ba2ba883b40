"""per-step kernel time table from a rocprofv3 --kernel-trace --stats kernel_stats.csv
usage: python tools/kstats.py <kernel_stats.csv> <steps | auto> [rows]
steps = every train step the traced process ran (graph-capture set-up, warm-up, timed and instrumented ones); `auto`
counts them from the calls of a kernel that runs exactly once per step (the loss's partial-sum pass)"""
import csv, sys


def steps_in_trace(rows):
    """train steps in a kernel_stats.csv: calls of the once-per-step loss kernel"""
    for key in ('dice_partial_kernel', 'focal_partial_kernel', 'focal_fwd_kernel'):
        for r in rows:
            if r['Name'].split('(')[0].strip().endswith(key):
                return int(r['Calls'])
    raise SystemExit('no once-per-step kernel found: pass the step count explicitly')


if __name__ != '__main__':
    raise ImportError('tools/kstats.py is a script; import steps_in_trace from a copy of this function instead')
rows = list(csv.DictReader(open(sys.argv[1])))
steps = steps_in_trace(rows) if len(sys.argv) < 3 or sys.argv[2] == 'auto' else int(sys.argv[2])
print('steps in the trace: %d' % steps)
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time per step: %.3f ms' % (tot / 1e6 / steps))
for r in rows[:nrows]:
    print('%-64s calls/step %6.1f  ms/step %7.3f  avg %8.1f us  %5.1f%%' % (
        r['Name'][:64], float(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3,
        float(r['Percentage'])))
