#!/usr/bin/env python3
"""Bank-conflict model of the split kernel's LDS accesses (csrc/tcs_split.hip) under the rules of MI355X_MICROARCH.md, LDS section:
ds_read_b64 / ds_read_b64_tr_b16 are served in two 32-lane groups, bank = (byte address / 4) mod 64, a lane covers two banks; each extra distinct
dword on a busy bank costs one more LDS cycle.  Prints the cycles per wave-instruction (2 = conflict-free) of
  * the producers' window reads for every row pitch (64 XJ elements + pad) -- no pad is conflict-free: the runs of a row lie TT / 8 dwords apart;
  * the consumers' transposed A-fragment reads with the round-4 swizzle key (c & 3) * 5 and with the round-5 key that adds bit 3 of the row."""


def cycles_b64(addrs):
    tot = 0
    for grp in (range(0, 32), range(32, 64)):
        banks = {}
        for l in grp:
            for d in (0, 1):
                dw = addrs[l] // 4 + d
                banks.setdefault(dw % 64, set()).add(dw)
        tot += max(len(v) for v in banks.values())
    return tot


def taddr(c, t, rowb, key):
    return c * rowb + (((t >> 3) ^ key(c)) << 4) + ((t & 7) << 1)


if __name__ == "__main__":
    for tt, run in ((96, 12), (192, 24)):
        for xj in (2, 3, 4, 5):
            res = [(cycles_b64([((l >> 2) * (32 * xj + pad) + (l & 3) * run) * 4 for l in range(64)]), pad) for pad in range(0, 34, 2)]
            print(f"window reads, {tt}-frame tiles, XJ = {xj}: shipped pad (2 dwords) {dict((p, c) for c, p in res)[2]} cycles; best over all pads {min(res)[0]}")
    keys = {"(c & 3) * 5": lambda c: (c & 3) * 5, "(c & 3) * 5 ^ ((c >> 3) & 1) << 1": lambda c: ((c & 3) * 5) ^ (((c >> 3) & 1) << 1)}
    for rowb in (256, 512):
        for name, key in keys.items():
            worst = max(cycles_b64([taddr(8 * (l >> 4) + ((l >> 2) & 3) + hi, wm * 96 + 16 * mt + 4 * (l & 3), rowb, key) for l in range(64)])
                        for mt in range(6) for hi in (0, 4) for wm in ((0,) if rowb == 256 else (0, 1)))
            print(f"A-fragment reads, {rowb}-byte rows, key {name}: {worst} cycles")
