"""where does a train step spend its time?  reads a rocprofv3 --kernel-trace csv (per-dispatch start / end timestamps) of a
bench.py run and, for the last full step, splits the wall time into: only MFMA-bound kernels running, only other kernels
running (exposed HBM-bound time), both at once (overlap), nothing running (gaps).
usage: python tools/timeline.py <kernel_trace.csv> [steps_to_skip_from_the_end=0]"""
import csv, sys

MFMA = ('conv3d_k3_wino2d_kernel', 'conv3d_k3_wgrad_wino2d_kernel', 'conv3d_k3_mfma2', 'conv3d_k3_wgrad', 'conv3d_k3_wino_kernel',
        'conv3d_k3_wgrad_wino_kernel')
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# a step ends with the Adam kernel
ends = [i for i, n in enumerate(names) if 'adam_step' in n]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = ends[-1 - skip]
lo = ends[-2 - skip] + 1
step = rows[lo:hi + 1]
t0 = int(step[0]['Start_Timestamp'])
t1 = max(int(r['End_Timestamp']) for r in step)
ev = []
for r in step:
    m = any(k in r['Kernel_Name'] for k in MFMA) and 'reduce' not in r['Kernel_Name'] and 'finish' not in r['Kernel_Name']
    ev.append((int(r['Start_Timestamp']), 1, m))
    ev.append((int(r['End_Timestamp']), -1, m))
ev.sort()
cm = co = 0
acc = {'mfma only': 0, 'other only': 0, 'both': 0, 'idle': 0}
prev = t0
for t, d, m in ev:
    dt = t - prev
    key = 'both' if (cm and co) else ('mfma only' if cm else ('other only' if co else 'idle'))
    acc[key] += dt
    prev = t
    if m: cm += d
    else: co += d
tot = t1 - t0
print('step wall time %.3f ms (%d dispatches)' % (tot / 1e6, len(step)))
for k, v in acc.items():
    print('  %-10s %7.3f ms  %5.1f%%' % (k, v / 1e6, 100.0 * v / tot))
busy_m = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step if any(k in r['Kernel_Name'] for k in MFMA))
busy_o = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step if not any(k in r['Kernel_Name'] for k in MFMA))
print('  sum of MFMA-kernel durations %.3f ms, of the others %.3f ms' % (busy_m / 1e6, busy_o / 1e6))
# forward / backward split: the loss kernel marks the boundary
b = next((i for i, r in enumerate(step) if 'dice_partial' in r['Kernel_Name'] or 'focal' in r['Kernel_Name']), None)
if b is not None:
    tb = int(step[b]['Start_Timestamp'])
    print('  forward %.3f ms, loss + backward + Adam %.3f ms' % ((tb - t0) / 1e6, (t1 - tb) / 1e6))
