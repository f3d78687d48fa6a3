// Launchers of the batched puncture / depuncture kernels (puncturer.hip), shared with capi_puncturer.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pamd {

// out[r, c] = map[c] >= 0 ? in[r, map[c]] : 0   for rows x out_cols elements of elem_bytes (1, 4 or 8) bytes
hipError_t launch_gather_rows(const void* in, void* out, const int32_t* map, uint32_t rows, uint32_t in_cols,
                              uint32_t out_cols, unsigned elem_bytes, unsigned grid, hipStream_t st);
// packed bits, MSB first: output bit i = input bit pos[i]; out_bytes * 8 positions
hipError_t launch_puncture_packed(const uint8_t* in, uint8_t* out, const int32_t* pos, uint32_t rows, uint32_t in_bytes,
                                  uint32_t out_bytes, unsigned grid, hipStream_t st);

}  // namespace pamd
