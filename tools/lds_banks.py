#!/usr/bin/env python3
"""Exhaustive LDS bank-conflict check of the swizzled observation tile of csrc/critic_rows.hip (no GPU needed).

Tile rows of `ldx` floats (a multiple of 64); the 16-B chunk c of row r is stored at chunk (c & ~15) | ((c & 15) ^ (r & 15)).
Lane groups and bank rules per instruction: MI355X_MICROARCH.md, section LDS."""
G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
        list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
        list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
        list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def off(r, c, ldx):
    return r * ldx + (((c & ~15) | ((c & 15) ^ (r & 15))) << 2)


def ways(groups, addr, width, banks):
    worst = 0
    for g in groups:
        use = {}
        for lane in g:
            a = addr(lane)
            for b in range(width):
                use.setdefault((a + b) % banks, set()).add(a + b)
        worst = max(worst, max(len(v) for v in use.values()))
    return worst


def main():
    for ldx in (64, 128, 192, 256, 384):
        rd = max(ways(G128, lambda lane, j=j: off(lane & 15, 4 * j + (lane >> 4), ldx), 4, 64) for j in range(ldx // 16))
        cpr = ldx // 4
        wr = max(ways([range(g0, g0 + 8) for g0 in range(0, 64, 8)],
                      lambda lane, base=base: off(((base + lane) // cpr) & 31, (base + lane) % cpr, ldx), 4, 32)
                 for base in range(0, 32 * cpr, 64))
        b32 = max(ways([range(0, 32), range(32, 64)],
                       lambda lane, r0=r0, t=t: off(r0 + (lane >> 4), (16 * t + (lane & 15)) >> 2, ldx) + (lane & 3), 1, 32)
                  for r0 in range(0, 32, 4) for t in range(ldx // 16))
        print(f"ldx {ldx:3d}: ds_read_b128 A operand {rd}-way, ds_write_b128 staging {wr}-way, ds_read_b32 B operand (rows r0 + kq) {b32}-way")
        assert rd == 1 and wr == 1


if __name__ == "__main__":
    main()
