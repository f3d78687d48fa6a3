#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  Regenerates tests/golden/sc_accept.txt: 360 random frozen sets with the verdict of the
# unmodified reference's FastSscAvxFloat constructor (1 = builds, 0 = throws std::invalid_argument).
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
python3 - "$T/in.txt" <<'PY'
import sys, numpy as np
rng = np.random.default_rng(20261004)
with open(sys.argv[1], "w") as f:
    for N in (16, 32, 64, 128, 256, 1024):
        for t in range(60):
            K = int(rng.integers(1, N))
            F = np.sort(rng.choice(N, N - K, replace=False))
            f.write(" ".join([str(N)] + [str(int(x)) for x in F]) + "\n")
PY
oracle/_ref/ref_harness sccheck $T/in.txt tests/golden/sc_accept.txt
rm -rf $T
awk '{print $1}' tests/golden/sc_accept.txt | sort | uniq -c
