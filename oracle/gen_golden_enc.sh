#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  Regenerates tests/golden/enc.bin: golden vectors of the reference ENCODER
# (ButterflyFipPacked::encode_vector, butterfly_fip_packed.cpp:45-59) produced by the unmodified reference
# library through oracle/_ref/ref_harness (build it with `make -C oracle` first).  Concatenated 'PFE1' records.
set -e
cd "$(dirname "$0")/.."
H=oracle/_ref/ref_harness
G=tests/golden
T=$(mktemp -d)
seed=7000
: > $G/enc.bin
one() {  # N K ctor crcsize crctype systematic count
  seed=$(( seed + 1 ))
  $H enc $1 $2 $3 0.0 $4 $5 $6 $7 $seed $T/x.bin
  cat $T/x.bin >> $G/enc.bin
}
# the five configs (cfg4: K = 176, 5G table; cfg5: BB, CRC-32C), systematic and not
for sys in 1 0; do
  one 64 40 BB 0 crc $sys 16
  one 1024 512 BB 0 crc $sys 8
  one 1024 512 BB 24 crcnr $sys 16
  one 1024 176 5G 11 crcnr $sys 16
  one 4096 2048 BB 32 crc $sys 6
done
# block lengths around the reference's 256-bit vector size, CRC-8, odd rates
for N in 8 16 32 64 128 256 512 2048 8192; do
  for R in 1 2 3; do
    K=$(( N * R / 4 )); [ $N -ge 32 ] && K=$(( (K / 8) * 8 ))
    crc=8; { [ $K -le 8 ] || [ $(( K % 8 )) -ne 0 ]; } && crc=0
    cnt=6; [ $N -ge 2048 ] && cnt=3
    for sys in 1 0; do one $N $K BB $crc crc $sys $cnt; done
  done
  one $N $(( N / 2 )) BE 0 crc 1 4
done
# K not a multiple of 8 (dummy detector), CRC-32 with a byte count that is not a multiple of 4
one 128 59 BB 0 crc 1 6
one 128 59 BB 0 crc 0 6
one 256 104 BB 32 crc 1 6
one 256 104 BB 8 crc 0 6
rm -rf $T
ls -la $G/enc.bin
