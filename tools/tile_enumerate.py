#!/usr/bin/env python3
"""tile_enumerate.py -- VERDICT r05 item 4, as arithmetic: every output tile BM x BN (multiples of 16, <= 160) whose tile
count for a square size p lands on whole rounds of the chip -- [240, 256] tiles (one per CU), [490, 512] (two), [730, 768]
(three) -- priced BEFORE anything is built: price = occupancy (tiles / slots) x fill (p^2 / padded area) x wave balance
(the tile's 16x16 MFMA blocks over the 2 x 2 consumer waves of the K2W kernels: the fullest wave's share).  A candidate must
also be buildable: BM and BN multiples of 32 (a wave tile is 16 WTM x 16 WTN, csrc/sgemm_dma5.hpp Dma5Tile).
Beside each size: what ships, priced the same way, and what it measures.  No GPU needed."""
import math
import sys

CUS = 256
SHIPPED = {1152: ("96x64 plain (216 tiles)", 96, 64, 1, 111.6), 1280: ("64x64 stream-K (400 tiles on 256 workgroups)", 64, 64, 0, 118.1),
           1408: ("64x64 plain (484 tiles, up to three co-resident per CU)", 64, 64, 3, 132.8)}


def price(p, bm, bn, per):
    t = math.ceil(p / bm) * math.ceil(p / bn)
    fill = p * p / (math.ceil(p / bm) * bm * math.ceil(p / bn) * bn)
    bm16, bn16 = bm // 16, bn // 16
    bal = (bm16 * bn16) / (4 * math.ceil(bm16 / 2) * math.ceil(bn16 / 2))
    occ = t / (CUS * per)
    return t, occ, fill, bal, occ * fill * bal


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [1152, 1280, 1408]
    B = list(range(16, 161, 16))
    print("# Tile shapes that land N = 1152 / 1280 / 1408 on whole rounds of 256 CUs (tools/tile_enumerate.py)\n")
    for p in sizes:
        rows = []
        for bm in B:
            for bn in B:
                if bn > bm:
                    continue
                for per, lo, hi in ((1, 240, 256), (2, 490, 512), (3, 730, 768)):
                    t, occ, fill, bal, pr = price(p, bm, bn, per)
                    if lo <= t <= hi:
                        rows.append((pr, bm, bn, t, per, occ, fill, bal, bm % 32 == 0 and bn % 32 == 0))
        name, sbm, sbn, sper, tf = SHIPPED.get(p, ("?", 64, 64, 1, 0.0))
        print(f"## N = {p}: ships {name}, measured {tf} TFLOP/s = {tf / 157.3:.3f} of peak\n")
        if sper == 1:
            t, occ, fill, bal, pr = price(p, sbm, sbn, sper)
            print(f"shipped tile priced the same way: occupancy {occ:.3f} x fill {fill:.3f} x wave balance {bal:.3f} = **{pr:.3f}** "
                  f"(x the tile's one-workgroup-per-CU rate)\n")
        elif sper == 0:
            print("shipped launch is stream-K: every CU busy by construction; what it measures is what its hand-overs leave\n")
        else:
            print("shipped launch co-resides (up to three workgroups share a CU's matrix pipe): not a slots-times-rate price; "
                  "what it measures is the bar\n")
        if not rows:
            print("no BM x BN (multiples of 16, <= 160) puts the tile count in [240, 256], [490, 512] or [730, 768]\n")
            continue
        print("| tile | tiles | per CU | occupancy | fill | wave balance | price | buildable (32 | BM, BN) |")
        print("|---|---|---|---|---|---|---|---|")
        for pr, bm, bn, t, per, occ, fill, bal, ok in sorted(rows, reverse=True):
            print(f"| {bm}x{bn} | {t} | {per} | {occ:.3f} | {fill:.3f} | {bal:.3f} | {pr:.3f} | {'yes' if ok else 'no'} |")
        print()
    print("Reading: 1152 has no candidate at all (72-wide tiles are not multiples of 16).  1280: 112x64 prices at 0.781 but is not "
          "buildable (3.5 blocks per wave row) and 80x80 prices BELOW what stream-K already measures; 1408: 128x64 on one round "
          "(242 tiles) prices at 0.945 x the ONE-workgroup-per-CU rate of that tile (0.88-0.90 of peak at this K: 0.83-0.85), level "
          "with the 0.844 the shipped two-per-CU 64x64 launch measures.  Nothing to build: the item is closed.")


if __name__ == "__main__":
    main()
