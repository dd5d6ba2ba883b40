"""LDS bank-conflict model of the F(2x2,3x3) forward kernels' access patterns (csrc/conv_wino2d.hip), per the rules of
/opt/skills/guides/MI355X_MICROARCH.md (LDS section): an access is served in fixed lane groups, one LDS cycle per group when
conflict-free, one more per extra distinct address on a busy bank.  Prints, per access kind, LDS cycles per wave-instruction
(ideal / modelled) for the round-3 layouts and the round-4 ones.  Pure host arithmetic; no GPU.
usage: python tools/lds_bank_sim.py"""
import itertools

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS = B128_GROUPS + [[l + 32 for l in g] for g in B128_GROUPS]
HALVES = [list(range(32)), list(range(32, 64))]
QUARTERS = [list(range(16 * i, 16 * i + 16)) for i in range(4)]
RULES = {  # kind -> (lane groups, bank modulus in dwords, dwords per lane)
    'read_b32': (HALVES, 32, 1), 'read_b64': (HALVES, 64, 2), 'read_b128': (B128_GROUPS, 64, 4),
    'write_b32': (HALVES, 32, 1), 'write_b64': (QUARTERS, 32, 2),
}


def cycles(kind, addr):
    """addr: lane -> float (dword) address or None (inactive).  Returns (ideal, modelled) LDS-array cycles."""
    groups, mod, width = RULES[kind]
    ideal = total = 0
    for g in groups:
        per_bank = {}
        for l in g:
            a = addr(l)
            if a is None:
                continue
            for d in range(width):
                per_bank.setdefault((a + d) % mod, set()).add(a + d)
        if not per_bank:
            continue
        ideal += 1
        total += max(len(v) for v in per_bank.values())
    return ideal, total


def report(name, kind, addrs):
    """addrs: iterable of lane->address functions (one per instruction instance); prints the average"""
    i = t = n = 0
    for f in addrs:
        a, b = cycles(kind, f)
        i += a; t += b; n += 1
    print('  {:<58s} {:<10s} ideal {:5.2f}  modelled {:5.2f}  (x{:.2f})'.format(name, kind, i / n, t / n, t / i))
    return t / n


def tile_round3():
    print('tile kernel, round 3: weights [t][co 32][4], T [p][z 10][quad 16][4], RAW [(z*10+y)*10+x][4]')
    tot = 0
    tot += 48 * report('A: lane (l16, kq) -> t*128 + (16 hh + l16)*4 + kq', 'read_b32', [lambda l, hh=hh: (16 * hh + (l & 15)) * 4 + (l >> 4) for hh in (0, 1)])
    tot += 96 * report('B: (p*10 + z)*64 + l16*4 + kq', 'read_b32', [lambda l, z=z: z * 64 + (l & 15) * 4 + (l >> 4) for z in range(10)])
    def task(w, l):
        t = w * 40 + min(l, 39); i, h = t >> 1, t & 1
        return i >> 4, (i >> 2) & 3, i & 3, h   # z, qy, qx, h
    def rd(w, r, k):
        def f(l):
            z, qy, qx, h = task(w, l)
            return ((z * 10 + 2 * qy + r) * 10 + 2 * qx + k) * 4 + 2 * h
        return f
    tot += 16 * report('transform read: RAW b64 (r, k)', 'read_b64', [rd(w, r, k) for w in range(8) for r in range(4) for k in range(4)])
    def wr(w):
        def f(l):
            z, qy, qx, h = task(w, l)
            return (z * 16 + qy * 4 + qx) * 4 + 2 * h
        return f
    tot += 16 * report('transform write: T b64', 'write_b64', [wr(w) for w in range(8)])
    print('  LDS-array cycles per wave and K chunk: {:.0f}  (x 8 waves = {:.0f} per CU and chunk)'.format(tot, 8 * tot))


def tile_round4():
    print('tile kernel, round 4: weights [g 12][kq 4][co 32][4 steps], T [p][h 2][z 10][quad 16][2], RAW slot x ^= (y >> 1) & 1')
    tot = 0
    tot += 12 * report('A: ((g*4 + kq)*32 + 16 hh + l16)*4, 4 steps per read', 'read_b128', [lambda l, hh=hh: (((l >> 4) * 32) + 16 * hh + (l & 15)) * 4 for hh in (0, 1)])
    tot += 64 * report('B: p*640 + (kq>>1)*320 + z*32 + l16*2 + (kq&1)', 'read_b32', [lambda l, z=z: ((l >> 4) >> 1) * 320 + z * 32 + (l & 15) * 2 + ((l >> 4) & 1) for z in range(10)])
    def task(w, l):   # lanes 0-31: block w (= plane z) whole; lanes 32-39: an eighth of planes 8, 9; a block = [h 2][quad 16]
        l = min(l, 39)
        t = w * 32 + l if l < 32 else (8 + (w >> 2)) * 32 + 8 * (w & 3) + (l - 32)
        return t >> 5, (t >> 2) & 3, t & 3, (t >> 4) & 1   # z, qy, qx, h
    def rd(w, r, k):
        def f(l):
            z, qy, qx, h = task(w, l)
            y = 2 * qy + r
            return ((z * 10 + y) * 10 + ((2 * qx + k) ^ ((y >> 1) & 1))) * 4 + 2 * h
        return f
    tot += 16 * report('transform read: RAW b64 (r, k), swizzled slots', 'read_b64', [rd(w, r, k) for w in range(8) for r in range(4) for k in range(4)])
    def wr(w):
        def f(l):
            z, qy, qx, h = task(w, l)
            return h * 320 + z * 32 + (qy * 4 + qx) * 2
        return f
    tot += 16 * report('transform write: T b64', 'write_b64', [wr(w) for w in range(8)])
    print('  LDS-array cycles per wave and K chunk: {:.0f}  (x 8 waves = {:.0f} per CU and chunk)'.format(tot, 8 * tot))


def cell_round3(NC=4):
    print('cell kernel (NC = {}), round 3: T [p][cell][z 6][quad 4][4], RAW per cell [(z*6+y)*6+x][4]'.format(NC))
    tot = 0
    tot += 48 * report('A: as the tile kernel', 'read_b32', [lambda l, hh=hh: (16 * hh + (l & 15)) * 4 + (l >> 4) for hh in (0, 1)])
    tot += 48 * report('B: (cell*24 + l16)*4 + kq + kz*16', 'read_b32', [lambda l, kz=kz: (l & 15) * 4 + (l >> 4) + kz * 16 for kz in range(3)])
    def task(w, l):
        t = w * 24 + min(l, 23); cell, rem = divmod(t, 48); i, h = rem >> 1, rem & 1
        return cell, i >> 2, (i >> 1) & 1, i & 1, h
    def rd(w, r, k):
        def f(l):
            c, z, qy, qx, h = task(w, l)
            return c * 1024 + ((z * 6 + 2 * qy + r) * 6 + 2 * qx + k) * 4 + 2 * h
        return f
    tot += 16 * report('transform read: RAW b64', 'read_b64', [rd(w, r, k) for w in range(2 * NC) for r in range(4) for k in range(4)])
    def wr(w):
        def f(l):
            c, z, qy, qx, h = task(w, l)
            return (c * 24 + z * 4 + qy * 2 + qx) * 4 + 2 * h
        return f
    tot += 16 * report('transform write: T b64', 'write_b64', [wr(w) for w in range(2 * NC)])
    print('  LDS-array cycles per wave and K chunk: {:.0f}  (x {} waves = {:.0f} per CU and chunk)'.format(tot, 2 * NC, 2 * NC * tot))


def cell_round4(NC=4):
    print('cell kernel (NC = {}), round 4: weights as the tile kernel, T [p][cell][h 2][z 6][quad 4][2], RAW slot x ^= (y >> 1) & 1'.format(NC))
    tot = 0
    tot += 12 * report('A: b128, 4 steps per read', 'read_b128', [lambda l, hh=hh: (((l >> 4) * 32) + 16 * hh + (l & 15)) * 4 for hh in (0, 1)])
    tot += 48 * report('B: cell*96 + (kq>>1)*48 + (zl + kz)*8 + quad*2 + (kq&1)', 'read_b32', [lambda l, kz=kz: ((l >> 4) >> 1) * 48 + (((l & 15) >> 2) + kz) * 8 + (l & 3) * 2 + ((l >> 4) & 1) for kz in range(3)])
    def task(w, l):
        t = w * 24 + min(l, 23); cell, rem = divmod(t, 48); i, h = rem >> 1, rem & 1
        return cell, i >> 2, (i >> 1) & 1, i & 1, h
    def rd(w, r, k):
        def f(l):
            c, z, qy, qx, h = task(w, l)
            y, x = 2 * qy + r, 2 * qx + k
            return c * 1024 + ((z * 6 + y) * 6 + (x ^ ((y >> 1) & 1))) * 4 + 2 * h
        return f
    tot += 16 * report('transform read: RAW b64, swizzled slots', 'read_b64', [rd(w, r, k) for w in range(2 * NC) for r in range(4) for k in range(4)])
    def wr(w):
        def f(l):
            c, z, qy, qx, h = task(w, l)
            return c * 96 + h * 48 + z * 8 + (qy * 2 + qx) * 2
        return f
    tot += 16 * report('transform write: T b64', 'write_b64', [wr(w) for w in range(2 * NC)])
    print('  LDS-array cycles per wave and K chunk: {:.0f}  (x {} waves = {:.0f} per CU and chunk)'.format(tot, 2 * NC, 2 * NC * tot))


if __name__ == '__main__':
    tile_round3(); tile_round4(); cell_round3(); cell_round4()
