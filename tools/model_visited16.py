#!/usr/bin/env python
"""How many ids does the two-choice bucket table of the register walkers (VisitedSetB, granne_amd/csrc/wave_prims.h) hold
before the first id finds BOTH of its buckets full and goes to the global overflow table?

Placement as in the kernel: an id's two buckets are independent of each other, it goes to the emptier one (ties: the
first), buckets hold 8 entries (16-bit entries) or 6 (20-bit entries) and never lose one.  Printed per table size: the
number of ids inserted when the first one spilled (min / 1 % / 10 % / median over the trials) and the load that is.
A max_search-50 walk on 10M uniform points visits 1,990 +- 240 ids, a max_search-200 walk ~7,000.

Round 2's model of a 16-bit table with linear probing (home slot + remainder + displacement) is in the git history:
its displacements outgrow four bits at the loads a walk reaches, which is why the table has buckets and two choices.
Design-space model, not product code."""
import sys

import numpy as np


def first_spill(nb, per_bucket, trials, rng):
    out = []
    n = nb * per_bucket
    for _ in range(trials):
        cnt = np.zeros(nb, np.int32)
        b1 = rng.integers(0, nb, n)
        b2 = b1 ^ rng.integers(1, nb, n)  # never the same bucket (the kernel xors an odd function of the tag)
        f = n
        for i in range(n):
            a, b = b1[i], b2[i]
            c = a if cnt[a] <= cnt[b] else b
            if cnt[c] >= per_bucket:
                f = i
                break
            cnt[c] += 1
        out.append(f)
    return np.array(out)


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    for bits, per in ((16, 8), (20, 6)):
        for lg in (6, 9, 10, 11):
            nb = 1 << lg
            fa = first_spill(nb, per, 100 if lg >= 10 else 300, rng)
            print("%d-bit entries, %4d buckets x %d (%4.1f KB): first spill after min %5d, 1 %% %5d, 10 %% %5d, median %5d ids "
                  "(load %.2f)" % (bits, nb, per, nb * 16 / 1024, fa.min(), np.percentile(fa, 1), np.percentile(fa, 10),
                                  np.median(fa), np.median(fa) / (nb * per)))


if __name__ == "__main__":
    main()
