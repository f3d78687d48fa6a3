// Kernel parameter block and launchers shared by the device code (sc_kernel.hip, scl_kernel.h) and capi.cpp (host).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "program.h"

namespace pamd {

struct DecodeParams {
    const void* llr;         // rows x N  float32 (or int8) channel LLRs, row-major, device memory
    const uint32_t* rowidx;  // nullable: decode rows rowidx[0..rows) of the batch ("mixed" second pass)
    const uint32_t* rows_dev;  // nullable (list kernels): the row count lives in device memory (rows = upper bound, sizes the grid)
    uint32_t rows;
    uint32_t N, logN, K, nb, L;
    const Ins* prog;         // decode program (device)
    const uint16_t* lut;     // K code-bit positions of the information bits, ascending
    const uint32_t* syn;     // K detector syndrome words
    const uint32_t* ext;     // lane kernel: DepositWord per 32 code bits (mask of the information positions, move masks, offset)
    const uint32_t* syn_bytes;  // lane kernel: nb x 256 syndrome words, one per (output byte, byte value)
    uint8_t* out;            // rows x nb
    uint8_t* ok;             // rows (nullable)
    float* metrics;          // rows x L (nullable)
    uint32_t chr;            // 1 = int8 ("char") decoders: int8 rows, saturating int8 arithmetic (FastSscFipChar / SclFipChar)
    uint32_t det_on;         // 0 = dummy detector (always passes)
    uint32_t systematic;
    uint32_t path_stride;    // SCL: floats per path LLR buffer (N + bank padding)
    uint32_t wppL, wppR;     // SCL: bit words per path in the left-child / right-child output areas
    float* arena;            // SCL: global scratch for LLR stages >= s0 (one slice per resident wave)
    uint32_t arena_wave_words, arena_path_stride;  // words of one wave's slice (LLR stages, then the high-stage bits)
    uint32_t arena_bits_off;  // SCL: word offset of the bit blocks inside the wave's slice
    uint32_t cwl;             // SCL: log2 of the codewords one wave decodes side by side
    uint32_t sb;              // SCL: first stage whose bit blocks live in the arena (logN + 1 = none)
    uint32_t gbit_left_words, gbit_path_words;  // SCL: arena bit words per path (left/root outputs; both areas)
    uint32_t s0;             // SCL: first LLR stage kept in the arena (logN = none)
    uint32_t sc_virtual_top; // SC: stage logN-1 is recomputed from the channel row instead of being stored in LDS
    uint32_t prog_root_g0r;  // SC: the root is a ZeroR node (its right child input is a + b)
    uint32_t wave_lds_words; // SCL: LDS words of one wave (all its codewords); SC: of one codeword (64/LC codewords per wave)
    float* trace;            // diagnostics (nullable): row 0 only, (1 + L) floats after every SCL leaf
    uint32_t trace_cap;
    uint32_t prog_len;
    uint32_t ablate;         // perf experiments only (PCD_ABLATE): skips parts of the work, results are wrong
    uint32_t* cycles;        // diagnostics (nullable): row 0 only, s_memtime cycles spent per program instruction
    uint32_t i8rows;         // lane kernels (checked by their launchers): llr points at int8 rows, 1 byte per LLR (at the end: the
                             // kernels of the BASELINE configurations keep their argument offsets, and with them their code)
};

// occupancy != nullptr: do not launch, return the resident workgroups per CU (occupancy API) in *occupancy
hipError_t launch_sc(const DecodeParams& P, int lcl, bool i8, unsigned grid, unsigned waves_per_block, size_t lds,
                     hipStream_t st, int* occupancy = nullptr);
hipError_t launch_scl(const DecodeParams& P, int lpl, bool i8, unsigned grid, unsigned waves_per_block,
                      size_t lds, hipStream_t st, int* occupancy = nullptr);
// lane-per-path list kernel (scl_lane_kernel.hip): N = 1024, L <= 8, fp32 rows; P.prog = the LN_* program, P.arena = grid
// slices of scl_lane_arena_words() 32-bit words
hipError_t launch_scl_lane(const DecodeParams& P, bool diag, unsigned grid, hipStream_t st);
// the same kernel with the RateOne / SPC leaves of 256 LLRs (scl_lane_big.hip): for the codes lower_lane reports needs_big for
hipError_t launch_scl_lane_big(const DecodeParams& P, bool diag, unsigned grid, hipStream_t st);
// the int8 decoder's instantiation (scl_lane_char.hip); no leaves of 256 LLRs
hipError_t launch_scl_lane_char(const DecodeParams& P, bool diag, unsigned grid, hipStream_t st);
hipError_t launch_scl_lane_char_big(const DecodeParams& P, bool diag, unsigned grid, hipStream_t st);  // (scl_lane_char_big.hip)
// the float decoder on int8 rows (decode_vector(const char*)): scl_lane_i8.hip / scl_lane_i8_big.hip
hipError_t launch_scl_lane_i8(const DecodeParams& P, bool diag, unsigned grid, hipStream_t st);
hipError_t launch_scl_lane_i8_big(const DecodeParams& P, bool diag, unsigned grid, hipStream_t st);
unsigned scl_lane_arena_words();
// two-lanes-per-codeword SC kernel (sc_pair_kernel.hip): N = 1024, fp32 or int8 rows (P.chr: the int8 decoder, int8 rows only); P.prog = the PR_* program, P.ext / P.syn_bytes as for
// the lane kernel; single-wave workgroups with 40 KB of dynamic LDS
// big: the code has a leaf above 128 positions (lower_pair reports it): the instantiation that carries those leaves
hipError_t launch_sc_pair(const DecodeParams& P, bool i8rows, bool big, unsigned grid, hipStream_t st);
// ... N = 512 / 256 / 128 (sc_pair_n9 / n8 / n7.hip)
hipError_t launch_sc_pair_n9(const DecodeParams& P, bool i8rows, bool big, unsigned grid, hipStream_t st);
hipError_t launch_sc_pair_n8(const DecodeParams& P, bool i8rows, bool big, unsigned grid, hipStream_t st);
hipError_t launch_sc_pair_n7(const DecodeParams& P, bool i8rows, bool big, unsigned grid, hipStream_t st);
hipError_t launch_sc_pair_n6(const DecodeParams& P, bool i8rows, bool big, unsigned grid, hipStream_t st);
// CharContainer::insertLlr(const float*) (bitcontainer.cpp:449-516, N >= 8): round to nearest even, saturate to int8
hipError_t launch_f32_to_i8(const float* in, int8_t* out, size_t n, hipStream_t st);
hipError_t launch_collect_failed(const uint8_t* ok, uint32_t rows, uint32_t* list, uint32_t* count,
                                 hipStream_t st);

}  // namespace pamd
