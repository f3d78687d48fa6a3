#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  Round-2 fixtures, produced by the UNMODIFIED reference library (oracle/_ref):
#   nsys_*.bin : setSystematic(false) on encoder and decoder (fastssc_avx_float.cpp:944-948, scl_avx_float.cpp:731-746,
#                PackedContainer::insertLlr with its mFakeSize offset for short blocks, bitcontainer.cpp:975-1000)
#   i8_*.bin   : the row goes through decode_vector(const char*) (decoder.cpp:169-181, bitcontainer.cpp:202-207)
# Same record format as gen_golden.sh (flags in the header's dectype word, see ref_harness.cpp).
#   args of `gen`: N K L ctor dsnr crcsize crctype ebn0_db count seed mode dectype out [systematic] [input]
set -euo pipefail
cd "$(dirname "$0")"
H=./_ref/ref_harness
G=../tests/golden
T=$(mktemp -d)
# --- the BASELINE shapes --------------------------------------------------------------------------------------
# (no "mixed" row with int8 input: AdaptiveFloat does not override setSignal(const char*), so the base class dereferences its
#  null LLR container and the reference crashes, adaptive_float.cpp:59-63 / decoder.cpp:140)
$H gen 1024  512  1 BB 0.0  0 crc   2.5  32 2201 awgn  float $G/nsys_cfg2_n1024_k512_sc.bin 0
$H gen 1024  512  8 BB 0.0 24 crcnr 2.0  32 2301 awgn  float $G/nsys_cfg3_n1024_k512_l8_crc24.bin 0
$H gen 1024  512  8 BB 0.0 24 crcnr 0.5  16 2302 awgn  float $T/a.bin 0; cat $T/a.bin >> $G/nsys_cfg3_n1024_k512_l8_crc24.bin
$H gen 1024  176  8 5G 0.0 11 crcnr 1.0  24 2401 awgn  float $G/nsys_cfg4_n1024_k176_l8_crc11.bin 0
$H gen 1024  512  8 BB 0.0 24 crcnr 1.0  24 2601 awgn  mixed $G/nsys_mix_n1024_k512_l8_crc24.bin 0
$H gen 1024  512  1 BB 0.0  0 crc   2.5  24 2202 quant float $G/i8_cfg2_n1024_k512_sc.bin 1 i8
$H gen 1024  512  8 BB 0.0 24 crcnr 2.0  24 2303 quant float $G/i8_cfg3_n1024_k512_l8_crc24.bin 1 i8
$H gen 1024  176  8 5G 0.0 11 crcnr 1.0  16 2402 quant float $G/i8_cfg4_n1024_k176_l8_crc11.bin 1 i8
$H gen 1024  512  8 BB 0.0 24 crcnr 2.0  12 2304 quant float $G/i8_nsys_cfg3_n1024_k512_l8_crc24.bin 0 i8
# --- sweep over block lengths (N < 256: the packed containers pad), rates, list sizes ---------------------------------
: > $G/nsys_sweep.bin
: > $G/i8_sweep.bin
seed=7000
for N in 8 16 32 64 128 256 512 2048; do
  for R in 1 2 3; do
    K=$(( N * R / 4 )); [ $N -ge 32 ] && K=$(( (K / 8) * 8 )); [ $K -ge $N ] && continue
    for L in 1 4 8; do
      for ctor in BB BE; do
        [ $ctor = BE ] && [ $L -eq 4 ] && continue
        if [ $ctor = BB ]; then crc=8; else crc=0; fi
        { [ $K -le 8 ] || [ $(( K % 8 )) -ne 0 ]; } && crc=0
        cnt=5; [ $N -ge 512 ] && cnt=3
        for mode in awgn ties; do
          seed=$(( seed + 1 ))
          $H gen $N $K $L $ctor 0.0 $crc crc 2.0 $cnt $seed $mode float $T/x.bin 0 2>/dev/null
          cat $T/x.bin >> $G/nsys_sweep.bin
        done
        seed=$(( seed + 1 ))
        $H gen $N $K $L $ctor 0.0 $crc crc 2.0 $cnt $seed quant float $T/x.bin 1 i8 2>/dev/null
        cat $T/x.bin >> $G/i8_sweep.bin
      done
    done
  done
done
rm -rf $T
ls -la $G/nsys_* $G/i8_*; du -sh --apparent-size $G
# --- int8 ("char") decoders: FastSscFipChar (L = 1) / SclFipChar, rows through decode_vector(const char*) --------------
#     (quant: round(8 y) clipped to +-127; ties: +-1; noise: round(N(0,1)) -- mostly -1, 0, 1, i.e. zeros and ties)
T=$(mktemp -d)
$H gen 1024  512  1 BB 0.0  0 crc   2.5  48 3201 quant char $G/char_cfg2_n1024_k512_sc.bin
$H gen 1024  512  1 BB 0.0  0 crc   2.5  16 3202 ties  char $T/a.bin; cat $T/a.bin >> $G/char_cfg2_n1024_k512_sc.bin
$H gen 1024  512  8 BB 0.0 24 crcnr 2.0  48 3301 quant char $G/char_cfg3_n1024_k512_l8_crc24.bin
$H gen 1024  512  8 BB 0.0 24 crcnr 0.0  16 3302 quant char $T/a.bin; cat $T/a.bin >> $G/char_cfg3_n1024_k512_l8_crc24.bin
$H gen 1024  512  8 BB 0.0 24 crcnr 2.0  12 3303 ties  char $T/a.bin; cat $T/a.bin >> $G/char_cfg3_n1024_k512_l8_crc24.bin
$H gen 1024  512  8 BB 0.0 24 crcnr 2.0  12 3304 noise char $T/a.bin; cat $T/a.bin >> $G/char_cfg3_n1024_k512_l8_crc24.bin
$H gen 1024  176  8 5G 0.0 11 crcnr 1.0  32 3401 quant char $G/char_cfg4_n1024_k176_l8_crc11.bin
$H gen 1024  512  8 BB 0.0 24 crcnr 2.0  16 3307 sat   char $T/a.bin; cat $T/a.bin >> $G/char_cfg3_n1024_k512_l8_crc24.bin
$H gen 1024  512  1 BB 0.0  0 crc   2.5  16 3203 sat   char $T/a.bin; cat $T/a.bin >> $G/char_cfg2_n1024_k512_sc.bin
$H gen 4096 3072  4 BB 0.0 32 crc   4.0   8 3501 quant char $G/char_n4096_k3072_l4_crc32.bin
$H gen 4096 3072  1 BB 0.0 32 crc   4.0   8 3502 quant char $T/a.bin; cat $T/a.bin >> $G/char_n4096_k3072_l4_crc32.bin
$H gen 1024  512  8 BB 0.0 24 crcnr 2.0  12 3305 quant char $G/char_nsys_n1024_k512.bin 0
$H gen 1024  512  1 BB 0.0  0 crc   2.5  12 3306 quant char $T/a.bin 0; cat $T/a.bin >> $G/char_nsys_n1024_k512.bin
: > $G/char_sweep.bin
seed=9000
for N in 8 16 32 64 128 256 512 2048; do
  for R in 1 2 3; do
    K=$(( N * R / 4 )); [ $N -ge 32 ] && K=$(( (K / 8) * 8 )); [ $K -ge $N ] && continue
    for L in 1 2 4 8 16 32; do
      [ $N -ge 2048 ] && [ $L -gt 8 ] && continue
      for ctor in BB BE; do
        [ $ctor = BE ] && [ $L -ne 1 ] && [ $L -ne 8 ] && continue
        if [ $ctor = BB ]; then crc=8; else crc=0; fi
        { [ $K -le 8 ] || [ $(( K % 8 )) -ne 0 ]; } && crc=0
        cnt=5; [ $N -ge 512 ] && cnt=3
        for mode in quant ties noise sat; do
          seed=$(( seed + 1 ))
          $H gen $N $K $L $ctor 0.0 $crc crc 2.0 $cnt $seed $mode char $T/x.bin 2>/dev/null
          cat $T/x.bin >> $G/char_sweep.bin
        done
      done
    done
  done
done
rm -rf $T
ls -la $G/char_*
