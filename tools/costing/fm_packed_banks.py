#!/usr/bin/env python3
"""LDS cycles of the fp32-grade FSRCNN mapping stage's operand reads (k_fs_maps4, packed 12-channel records, round 6): ds_read_b64 is served
in two groups of 32 lanes, 64 banks of 4 bytes, one cycle per group plus one per extra distinct address on a busy bank
(MI355X_MICROARCH.md, LDS).  Lane = (pixel n = lane & 15, quarter q = lane >> 4); K-step ks, half e reads piece P(ks, q, e) of the 27
(kernel row p / 9, piece p % 9 of the three-record window).  Enumerates record layouts x piece assignments; prints cycles per unit
(16 reads: 4 K-steps x 2 halves x hi / lo) - 32 is conflict-free."""
import itertools

FM_RW = 50


def cycles(addr_of, assign):
    total = 0
    for ks in range(4):
        for e in range(2):
            for lo in range(2):
                for grp in range(2):
                    banks = {}
                    for lane in range(32 * grp, 32 * grp + 32):
                        n, q = lane & 15, lane >> 4
                        p = min(assign(ks, q, e), 26)
                        a = addr_of(n, p, lo)
                        for d in range(2):
                            banks.setdefault(((a >> 2) + d) % 64, set()).add(a)
                    total += max(len(v) for v in banks.values())
    return total


def layout_interleaved(rowb):
    return lambda n, p, lo: (p // 9) * rowb + (n + (p % 9) // 3) * 48 + 8 * ((p % 9) % 3) + 24 * lo


def layout_split(rowb_half):   # hi row then lo row, records 24 bytes
    return lambda n, p, lo: (p // 9) * 2 * rowb_half + lo * rowb_half + n * 24 + 8 * (p % 9)


assigns = {
    "8ks+q+4e": lambda ks, q, e: 8 * ks + q + 4 * e,
    "8ks+2q+e": lambda ks, q, e: 8 * ks + 2 * q + e,
    "8ks+q+4e, quarters swapped 1<->2": lambda ks, q, e: 8 * ks + (0, 2, 1, 3)[q] + 4 * e,
}
def table():
    for name, a in assigns.items():
        print(f"{name:36s} interleaved hi|lo records (48 B): {cycles(layout_interleaved(FM_RW * 48), a):3d}   "
              f"split hi / lo rows (24 B): {cycles(layout_split(FM_RW * 24), a):3d}   "
              f"split, rows padded to 1216 B: {cycles(layout_split(1216), a):3d}")


def search(seed=0, iters=20000):
    """Random-restart hill climbing over assignments of the 27 pieces (+ 5 pads, which may alias any real piece) to the 32 (ks, q, e) slots."""
    import random
    rng = random.Random(seed)
    addr = layout_interleaved(FM_RW * 48)
    best, best_c = None, 99
    perm = list(range(32))
    cur = cycles(addr, lambda ks, q, e: perm[8 * ks + q + 4 * e])
    for it in range(iters):
        i, j = rng.randrange(32), rng.randrange(32)
        perm[i], perm[j] = perm[j], perm[i]
        c = cycles(addr, lambda ks, q, e: perm[8 * ks + q + 4 * e])
        if c <= cur:
            cur = c
            if c < best_c:
                best, best_c = list(perm), c
                if c == 32:
                    break
        else:
            perm[i], perm[j] = perm[j], perm[i]
    return best_c, best


if __name__ == "__main__":
    import sys
    table()
    if "--search" in sys.argv:
        c, perm = search()
        print("best", c, "slot (ks, e, q) -> piece:", perm)
