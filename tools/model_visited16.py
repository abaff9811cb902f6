#!/usr/bin/env python
"""Would an exact visited table of 16-bit entries hold a max_search-200 walk in the 16 KB the front table has today?

Today (wave_prims.h): 4096 slots x 32 bits, double hashing, frozen at 5/8 load; everything after that goes to an overflow
table in global memory (one or more atomicCAS round trips per expansion, queued behind the row loads): 47 % of an
expansion at max_search 200 with four waves per SIMD (profiles/r2b_phase_i8_ef200_nq4096.txt).
Candidate: 8192 slots x 16 bits, linear probing; entry = (remainder of a bijective hash, displacement from the home slot),
home slot = high bits of the hash, so (slot, entry) identifies the id exactly as long as id_bits <= 13 + remainder bits.
This script replays walks' insert streams (expansions of ~27 new ids, as the walker sees them) and reports, per expansion,
what the wave would wait for: the slowest lane's number of 8-slot reads (one ds_read_b128 covers 8 consecutive 16-bit
slots) and the largest displacement (must fit the entry's displacement bits).   Design-space model, not product code."""
import random
import statistics
import sys


def replay(n_ids, slots, per_exp, rnd, id_bits=24):
    tab = [None] * slots
    reads_hist, disp_max, fails = [], 0, 0
    inserted = 0
    while inserted < n_ids:
        ids = [rnd.getrandbits(id_bits) for _ in range(per_exp)]
        worst = 0
        for i in ids:
            h = (i * 0x9E3779B1) & 0xFFFFFFFF
            home = (h * slots) >> 32
            d = 0
            while tab[(home + d) % slots] is not None and tab[(home + d) % slots] != i:
                d += 1
            tab[(home + d) % slots] = i
            disp_max = max(disp_max, d)
            first, last = home // 8, (home + d) // 8  # 8-slot groups touched by the probe sequence
            worst = max(worst, last - first + 1)
        inserted += per_exp
        reads_hist.append((inserted / slots, worst))
    return reads_hist, disp_max


def main():
    rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    for n_ids, slots in ((2100, 4096), (2100, 8192), (6500, 8192), (6500, 12288), (8000, 8192)):
        worst_all, disp_all = [], []
        late = []
        for _ in range(200):
            hist, disp = replay(n_ids, slots, 27, rnd)
            worst_all += [w for _, w in hist]
            late += [w for load, w in hist if load > 0.9 * n_ids / slots]
            disp_all.append(disp)
        print("%5d ids into %5d 16-bit slots (%4.1f KB, final load %.2f): 8-slot reads of the slowest lane per expansion "
              "mean %.2f, at the end of the walk %.2f, max %d | largest displacement: median %d, max %d"
              % (n_ids, slots, slots * 2 / 1024, n_ids / slots, statistics.mean(worst_all), statistics.mean(late),
                 max(worst_all), statistics.median(disp_all), max(disp_all)))


if __name__ == "__main__":
    main()
