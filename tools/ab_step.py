"""A/B of module-level switches of segmentation3d._ops on ONE box: runs bench.py's train-step measurement once per setting.
usage: python tools/ab_step.py NAME=VALUE[,NAME=VALUE...] [NAME=VALUE ...] -- [bench.py args]   (each positional = one arm)"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
bench_args = []
if '--' in args:
    i = args.index('--')
    args, bench_args = args[:i], args[i + 1:]
code = ("import sys, json; sys.path.insert(0, {repo!r}); sys.path.insert(0, {repo!r} + '/medical-segmentation3d-toolkit_amd');"
        "sys.argv = ['bench.py', '--no-infer', '--no-cpu-baseline', '--no-roofline'] + {bargs!r};"
        "from segmentation3d import _ops\n{sets}\nimport bench; bench.main()")
for arm in args:
    sets = '\n'.join('_ops.{} = {}'.format(*kv.split('=')) for kv in arm.split(',') if kv and kv != 'default')
    r = subprocess.run([sys.executable, '-c', code.format(repo=REPO, bargs=bench_args, sets=sets)], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if not line:
        print(arm, 'FAILED', r.stderr[-2000:])
        continue
    import json
    d = json.loads(line[-1])
    print('{:50s} {:8.3f} ms/step  {:8.2f} patches/s  loss {}'.format(arm, d['ms_per_step'], d['value'], d['final_loss']), flush=True)
