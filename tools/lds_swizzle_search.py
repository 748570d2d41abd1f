#!/usr/bin/env python3
"""Which XOR swizzles of the K image make k_attn_encoder16's fragment reads conflict-free?  Model (MI355X_MICROARCH.md, LDS): a ds_read_b128 is served in four
groups of sixteen lanes, each group one pass over 64 banks of 4 bytes; a group is conflict-free when its sixteen 16-byte slots differ modulo 256 bytes.
The K image: 64 rows x 128 B, chunk c of row r stored at chunk position c ^ sw(r); lane (r16, g) reads row kappa(r16) = 4 (r16 & 3) + (r16 >> 2), chunk 4 half + g.
Prints whether round 3's swizzle passes and the linear-in-row-bits swizzles that do."""
import itertools

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def kappa(r16):
    return 4 * (r16 & 3) + (r16 >> 2)


def conflict_free(sw, row_of=kappa):
    for half in (0, 1):
        for grp in GROUPS:
            seen = set()
            for lane in grp:
                r16, g = lane & 15, lane >> 4
                row, c = row_of(r16), 4 * half + g
                slot = ((row & 1) << 3) | ((c ^ sw(row)) & 7)
                if slot in seen:
                    return False
                seen.add(slot)
    return True


if __name__ == "__main__":
    print("round 3's K swizzle ((row & 1) << 2 | (row >> 2) & 3):", conflict_free(lambda r: ((r & 1) << 2) | ((r >> 2) & 3)))
    print("V^T reads (row = r16, chunk ^ (row & 7)):", conflict_free(lambda r: r & 7, row_of=lambda r16: r16))
    sols = []
    for masks in itertools.product(range(16), repeat=3):
        f = lambda row, m=masks: sum(((bin(row & m[b]).count("1") & 1) << b) for b in range(3))
        if conflict_free(f):
            sols.append(masks)
    print("%d linear swizzles are conflict-free; the one used: masks (0, 4, 8) = (row >> 1) & 6:" % len(sols), (0, 4, 8) in sols, conflict_free(lambda r: (r >> 1) & 6))
