// Parameter block and launcher of the batched polar ENCODER kernel (encoder.hip), shared with capi_encoder.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "program.h"

namespace pamd {

struct EncodeParams {
    uint8_t* info;            // rows x nb packed information bytes (MSB first); the checksum field is written in place
    uint8_t* code;            // rows x N/8 packed code words
    uint32_t rows;
    uint32_t N, logN, K, nb;
    const DepositWord* dep;   // max(1, N/32) entries
    const uint32_t* gen4;     // detector generator by nibble: [info dword k][nibble j of the little-endian dword][16 values]
                              // -> XOR of the generator words of the set bits (nullptr = dummy detector)
    uint32_t field_off;       // byte offset of the checksum field in the information word
    uint32_t field_bytes;
    uint32_t systematic;
    uint32_t iw_words;        // LDS words one code word owns (information bits, LSB first)
    uint32_t info_aligned;    // info base and nb are multiples of 4: rows can be read as dwords
};

hipError_t launch_encode(const EncodeParams& P, unsigned grid, unsigned waves_per_block, hipStream_t st);
// code words one wave encodes at a time (launch geometry / LDS sizing)
unsigned encode_codewords_per_wave(unsigned N);

}  // namespace pamd
