"""Costing tool (no GPU): LDS bank conflicts of k_fs_maps4's operand reads / row stores for a slot swizzle f(column) = g[(c >> 2) & 3] ^ h[c & 3].
Model (measured on gfx950, profiles/NOTES_r04.md): a ds_read_b128 pass serves lanes {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of a
half-wave; a pass is conflict-free when its sixteen 16-byte accesses fall into sixteen different bank quads ((address / 16) mod 16).
usage: python tools/costing/fm_swizzle.py"""
import itertools
G1 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
G2 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]


def read_conflicts(f):
    tot = 0
    for base in (0, 2):              # hi / lo parts
        for dx in range(3):
            for grp in (G1, G2):
                quads = set()
                for lane in grp:
                    n, qb = lane & 15, (lane >> 4) & 1
                    c = (n + dx) & 15
                    quads.add((((c & 3) << 2) | ((base + qb) ^ f[c])) & 15)
                tot += 16 - len(quads)
    return tot


def write_conflicts(f):              # ds_write_b64 of a layer's output row: lane (n, q) -> column n + 1, slot (q >> 1) (+ 2), half q & 1
    tot = 0
    for base in (0, 2):
        for half in (0, 1):
            banks = set()
            for n in range(16):
                for qb in (0, 1):
                    c = (n + 1) & 15
                    addr = c * 64 + ((base + half) ^ f[c]) * 16 + 8 * qb
                    banks.update({(addr // 4) % 64, (addr // 4 + 1) % 64})
            tot += 64 - len(banks)
    return tot


res = []
for g in itertools.product(range(4), repeat=4):
    for h in itertools.product(range(4), repeat=4):
        if h[0]:
            continue
        f = [g[(c >> 2) & 3] ^ h[c & 3] for c in range(16)]
        res.append((read_conflicts(f), write_conflicts(f), g, h))
res.sort()
cur = [(c >> 2) & 3 for c in range(16)]
print("rounds 2-3, f = (c >> 2) & 3:        read conflicts", read_conflicts(cur), " write conflicts", write_conflicts(cur))
new = [((c >> 2) & 1) << 1 for c in range(16)]
print("round 4,   f = 2 ((c >> 2) & 1):     read conflicts", read_conflicts(new), " write conflicts", write_conflicts(new))
print("best by reads:", res[:3])
print("best with conflict-free writes:", min((r for r in res if r[1] == 0), key=lambda r: r[0]))
