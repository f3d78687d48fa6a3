#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  Regenerates tests/golden/*.bin with the UNMODIFIED reference library
# (oracle/_ref, built by `make -C oracle ref` from /root/reference).  Runs in the build container only;
# the fixtures it writes are committed data (inputs + the reference's outputs), format: ref_harness.cpp.
#   args of `gen`: N K L ctor dsnr crcsize crctype ebn0_db count seed mode dectype out
set -euo pipefail
cd "$(dirname "$0")"
H=./_ref/ref_harness
G=../tests/golden
mkdir -p $G
# --- BASELINE.json configs ---------------------------------------------------------
$H gen   64   40  1 BB 1.0  0 crc   3.0  64 101 awgn  float $G/cfg1_n64_k40_sc_awgn.bin
$H gen   64   40  1 BB 1.0  0 crc   3.0  32 102 ties  float $G/cfg1_n64_k40_sc_ties.bin
$H gen 1024  512  1 BB 0.0  0 crc   2.5  96 201 awgn  float $G/cfg2_n1024_k512_sc_awgn.bin
$H gen 1024  512  1 BB 0.0  0 crc   2.5  32 202 ties  float $G/cfg2_n1024_k512_sc_ties.bin
$H gen 1024  512  1 BB 0.0  0 crc   2.5  32 203 quant float $G/cfg2_n1024_k512_sc_quant.bin
$H gen 1024  512  8 BB 0.0 24 crcnr 2.0  96 301 awgn  float $G/cfg3_n1024_k512_l8_crc24_awgn.bin
$H gen 1024  512  8 BB 0.0 24 crcnr 2.0  24 302 noise float $G/cfg3_n1024_k512_l8_crc24_noise.bin
$H gen 1024  512  8 BB 0.0 24 crcnr 2.0  24 303 ties  float $G/cfg3_n1024_k512_l8_crc24_ties.bin
$H gen 1024  512  8 BB 0.0 24 crcnr 2.0  24 304 quant float $G/cfg3_n1024_k512_l8_crc24_quant.bin
$H gen 1024  512  8 BB 0.0 24 crcnr 0.5  32 305 awgn  float $G/cfg3_n1024_k512_l8_crc24_lowsnr.bin
$H gen 1024  176  8 5G 0.0 11 crcnr 1.0  64 401 awgn  float $G/cfg4_n1024_k176_l8_crc11_awgn.bin
$H gen 1024  176  8 5G 0.0 11 crcnr 1.0  16 402 quant float $G/cfg4_n1024_k176_l8_crc11_quant.bin
$H gen 4096 2048 32 BB 0.0 32 crc   1.5  24 501 awgn  float $G/cfg5_n4096_k2048_l32_crc32_awgn.bin
$H gen 4096 2048 32 BB 0.0 32 crc   1.5   6 502 quant float $G/cfg5_n4096_k2048_l32_crc32_quant.bin
# --- mixed (AdaptiveFloat) ----------------------------------------------------------------------
$H gen 1024  512  8 BB 0.0 24 crcnr 1.0  48 601 awgn  mixed $G/mix_n1024_k512_l8_crc24_awgn.bin
$H gen  256  128  4 BB 0.0  8 crc   1.0  48 602 awgn  mixed $G/mix_n256_k128_l4_crc8_awgn.bin
# --- sweep: sizes / rates / list sizes / constructions / detectors (few rows each); the per-case
# fixtures are concatenated into one file per block length (each record keeps its own header).
T=$(mktemp -d)
seed=1000
for N in 8 16 32 64 128 256 512 2048; do
  : > $G/sweep_n${N}.bin
  for R in 1 2 3; do            # K = N*R/4 rounded down to whole bytes
    K=$(( N * R / 4 )); [ $N -ge 32 ] && K=$(( (K / 8) * 8 )); [ $K -ge $N ] && continue
    for L in 1 2 4 8 16 32; do
      [ $N -ge 2048 ] && [ $L -gt 8 ] && continue
      for ctor in BB BE; do
        [ $ctor = BE ] && [ $L -ne 1 ] && [ $L -ne 8 ] && continue
        for mode in awgn ties quant; do
          [ $mode = quant ] && [ $N -ne 128 ] && [ $N -ne 32 ] && continue
          seed=$(( seed + 1 ))
          if [ $ctor = BB ]; then crc=8; else crc=0; fi
          { [ $K -le 8 ] || [ $(( K % 8 )) -ne 0 ]; } && crc=0
          cnt=6; [ $N -ge 512 ] && cnt=3
          $H gen $N $K $L $ctor 0.0 $crc crc 2.0 $cnt $seed $mode float $T/x.bin 2>/dev/null
          cat $T/x.bin >> $G/sweep_n${N}.bin
        done
      done
    done
  done
done
# --- hand-made frozen sets (reach SC nodes no construction produces: ZEROSPC (Q8), DSPC n > 8, TREP, plain n = 8) ---
: > $G/handmade.bin
hm() {  # N  "frozen indices"
  printf "%s\n" $2 > $T/f.txt
  for L in 1 4 8; do for mode in noise ties; do
    seed=$(( seed + 1 ))
    $H gen $1 0 $L @$T/f.txt 0.0 0 crc 2.0 12 $seed $mode float $T/x.bin 2>/dev/null
    cat $T/x.bin >> $G/handmade.bin
  done; done
}
hm 16 "0 1 2 3 4 5 6 7 8"
hm 32 "0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16"
hm 32 "0 1"
hm 16 "0 1 2 3 4 5 6 7 8 9 10 11 12"
hm 8  "0 1 4"
hm 16 "0 1 2 4 8"
hm 64 "0 1 2 3 4 5 6 7 8 9 10 12 16 17 18 20 24 32 33 34 36 40 48"
rm -rf $T
ls -la $G; du -sh --apparent-size $G
