#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  Regenerates tests/golden/randsets.bin: decode vectors of the unmodified reference for
# arbitrary (random) frozen sets -- tree shapes no construction produces.  SC rows only for the sets the reference's
# SC builder accepts (tests/golden/sc_accept.txt); list decoding (L = 4, 8) for every set.
set -e
cd "$(dirname "$0")/.."
H=oracle/_ref/ref_harness
G=tests/golden
T=$(mktemp -d)
seed=11000
: > $G/randsets.bin
n=0
while read -r acc N rest; do
  [ "$N" -gt 256 ] && continue
  n=$(( n + 1 ))
  # every accepted set, every 6th rejected one
  if [ "$acc" = 0 ] && [ $(( n % 6 )) -ne 0 ]; then continue; fi
  printf "%s\n" $rest > $T/f.txt
  for L in 1 4 8; do
    [ $L = 1 ] && [ "$acc" = 0 ] && continue
    for mode in noise ties; do
      seed=$(( seed + 1 ))
      $H gen $N 0 $L @$T/f.txt 0.0 0 crc 2.0 5 $seed $mode float $T/x.bin 2>/dev/null
      cat $T/x.bin >> $G/randsets.bin
    done
  done
done < $G/sc_accept.txt
rm -rf $T
ls -la $G/randsets.bin
