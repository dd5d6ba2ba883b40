"""merge the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass on gfx950) of the bench
command into per-kernel averages, the file bench.py reads for roofline.traffic.
usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name'].split('(')[0].strip()
        acc[name][0] += float(r['Counter_Value'])
        acc[name][1] += 1
    return acc


fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
for name in sorted(set(fetch) | set(write)):
    e = {}
    if name in fetch:
        e['FETCH_SIZE_KB_per_launch'] = fetch[name][0] / fetch[name][1]
        e['launches'] = fetch[name][1]
    if name in write:
        e['WRITE_SIZE_KB_per_launch'] = write[name][0] / write[name][1]
        e.setdefault('launches', write[name][1])
    out[name] = e
json.dump(out, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
print('wrote', sys.argv[3], len(out), 'kernels')
