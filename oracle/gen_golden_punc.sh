#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  Regenerates tests/golden/punc.bin: golden vectors of the reference PUNCTURER
# (PolarCode::Puncturer, puncturer.cpp:51-89 / puncturer.h:66-97) produced by the unmodified reference library
# through oracle/_ref/ref_harness (build it with `make -C oracle` first).  Concatenated 'PFP1' records.
set -e
cd "$(dirname "$0")/.."
H=oracle/_ref/ref_harness
G=tests/golden
T=$(mktemp -d)
seed=9000
: > $G/punc.bin
one() {  # M N K ctor count
  seed=$(( seed + 1 ))
  $H punc $1 $2 $3 $4 0.0 $5 $seed $T/x.bin
  cat $T/x.bin >> $G/punc.bin
}
# qa_pypolar_puncturer.py:29-52: N = 32 ... 1024, K = N/2, punctured lengths N - N/4 and N - N/8
for n in 5 6 7 8 9 10; do
  N=$(( 1 << n )); K=$(( N / 2 ))
  one $(( N - N / 4 )) $N $K BB 3
  one $(( N - N / 8 )) $N $K BB 3
done
one 1024 1024 512 BB 2          # nothing punctured
one 864 1024 512 BB 4           # the bench workload: 160 of the 512 frozen positions punctured
one 520 1024 512 BE 2           # almost every frozen position punctured
one 3000 4096 1024 BB 2
one 1000 1024 176 5G 2          # punctured length that is no multiple of 64 (5G-table set)
rm -rf $T
ls -la $G/punc.bin
