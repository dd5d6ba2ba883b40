"""count instruction kinds per kernel in a hipcc -S listing (LDS reads / writes, MFMAs, waits, packed math, spill traffic)
usage: python tools/isa_count.py listing.s [substring of the mangled kernel name ...]"""
import re, sys, collections
txt = open(sys.argv[1]).read()
pat = sys.argv[2:]
parts = re.split(r'\n(_Z\w+):[^\n]*\n', txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split('.Lfunc_end')[0]
    if pat and not any(p in name for p in pat): continue
    c = collections.Counter()
    for line in body.split('\n'):
        m = re.match(r'\s+([a-z_0-9]+)', line)
        if m: c[m.group(1)] += 1
    keep = [k for k in sorted(c) if k.startswith(('ds_', 's_waitcnt', 'global_', 'v_pk', 'scratch', 'buffer_')) or 'mfma' in k or 'accvgpr' in k or k in ('s_barrier', 'v_mov_b32', 'v_readlane_b32', 'v_writelane_b32')]
    print(name)
    print('   ', ', '.join('{} {}'.format(k, c[k]) for k in keep), '| total', sum(c.values()))
