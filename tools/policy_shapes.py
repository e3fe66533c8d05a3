#!/usr/bin/env python3
"""policy_shapes.py -- the two shape lists MMH_KERNEL_AUTO's table is fitted on and judged on (no GPU).

    python tools/policy_shapes.py          # writes tools/policy_shapes_fit.txt and tools/policy_shapes_heldout.txt

fit:      the reference's square sweep (cuda/parameters.h:5-7: 1024 .. 4096 step 128), every size of it +- 1, the
          steps 1000 .. 4100 by 100, twelve non-square shapes (the M / N / K macros of armv7/parameters.h:15-17),
          a few whole-round calibration shapes per tile family and K, and 120 random shapes (seed 1);
held-out: 500 random shapes (seed 2), m, n, k drawn log-uniformly from [256, 8192] and kept when they are multiples
          of nothing in particular (at least one of m, n, k odd or not a multiple of 64) and m n k <= 2^37.
One `m,n,k` per line."""
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
SWEEP = list(range(1024, 4097, 128))
NONSQUARE = [(8192, 1024, 4096), (1024, 8192, 512), (16384, 128, 4096), (300, 5000, 7000), (4096, 4096, 4100),
             (2049, 2049, 2049), (6000, 3000, 1000), (1000, 6000, 3000), (128, 16384, 4096), (5000, 5000, 5000),
             (3000, 4000, 8192), (12288, 512, 2048)]


def rand_shapes(seed, count, ragged_only):
    rng = random.Random(seed)
    out = []
    while len(out) < count:
        m, n, k = (int(round(2 ** rng.uniform(8, 13))) for _ in range(3))
        if m * n * k > 2 ** 37 or min(m, n) < 256:
            continue
        if ragged_only and all(x % 64 == 0 for x in (m, n, k)):
            continue
        out.append((m, n, k))
    return out


def fit_shapes():
    out = [(n, n, n) for n in SWEEP]
    out += [(n + d, n + d, n + d) for n in SWEEP for d in (-1, 1)]
    out += [(n, n, n) for n in range(1000, 4101, 100) if n not in SWEEP]
    out += NONSQUARE
    # whole rounds of each tile family at two depths: c tiles per CU on a 16 x 16c grid of tiles
    for (bm, bn) in ((64, 64), (128, 64), (128, 128), (96, 96), (256, 256)):
        for c in (1, 2, 3, 4, 6):
            for k in (1024, 4096):
                out.append((16 * bm, 16 * c * bn, k))
    out += [(6144, 6144, 6144), (8192, 8192, 8192), (4096, 4096, 16384), (2048, 16384, 16384)]
    out += rand_shapes(1, 120, False)
    seen, uniq = set(), []
    for s in out:
        if s not in seen:
            seen.add(s)
            uniq.append(s)
    return uniq


def main():
    for name, shapes in (("fit", fit_shapes()), ("heldout", rand_shapes(2, 500, True))):
        path = os.path.join(HERE, f"policy_shapes_{name}.txt")
        with open(path, "w") as f:
            for (m, n, k) in shapes:
                f.write(f"{m},{n},{k}\n")
        print(path, len(shapes), "shapes,", round(sum(2.0 * m * n * k for m, n, k in shapes) / 1e12, 1), "TFLOP per pass")


if __name__ == "__main__":
    main()
