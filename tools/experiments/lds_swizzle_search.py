"""LDS bank-conflict model of the fused C2f kernel's 3x3 stages (64-byte pixel records, ds_read_b128 B fragments) and an
exhaustive search over linear swizzles s = (a*px + b*(px>>1) + c*(px>>2) + d*(px>>3) + e*row) & 3.

Result (round 3): the production swizzle (px >> 1) & 3 costs 882 extra LDS cycles over the 22 / 20 / 18 / 16-wide stages of one
tile (ideal 3348), the best member of the family 822: with four slots per record a straddling m-tile cannot be made conflict-free
by a swizzle; only a padded pitch (csrc/common.h upa_lds_pick_pitch) does it.  The model reproduces the PMC ratio
(LDS_BANK_CONFLICT / LDS_IDX_ACTIVE = 34 % for c2f32_fused_kernel<2, 16>)."""
import itertools

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def cycles(addrs):
    tot = 0
    for grp in GROUPS:
        banks = {}
        for lane in grp:
            banks.setdefault((addrs[lane] // 16) % 16, set()).add(addrs[lane])
        tot += max(len(s) for s in banks.values())
    return tot


def stage(sdh, sd, swz, taps=range(9)):
    ss, npx, tot, ideal = sd + 2, sdh * sd, 0, 0
    for mt in range((npx + 15) // 16):
        for tap in taps:
            addrs = []
            for lane in range(32):
                g, r = lane >> 4, lane & 15
                yy, xx = divmod(min(mt * 16 + r, npx - 1), sd)
                px = yy * ss + xx + (tap // 3) * ss + tap % 3
                addrs.append(px * 64 + ((g ^ swz(px, yy + tap // 3)) << 4))
            tot += cycles(addrs)
            ideal += 2
    return tot, ideal


if __name__ == "__main__":
    res = []
    for a, b, c, d, e in itertools.product(range(4), repeat=5):
        f = lambda px, row: (a * px + b * (px >> 1) + c * (px >> 2) + d * (px >> 3) + e * row) & 3  # noqa: E731
        extra = sum(t - i for t, i in (stage(s, s, f) for s in (22, 20, 18, 16)))
        res.append((extra, (a, b, c, d, e)))
    res.sort()
    print("best:", res[:4])
    print("production (0,1,0,0,0):", [r for r in res if r[1] == (0, 1, 0, 0, 0)])
