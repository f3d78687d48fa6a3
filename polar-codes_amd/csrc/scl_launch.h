// Launch entry points of the list-decoder kernel instantiations, one translation unit per
// (log2 lanes per path, log2 codewords per wave) pair (scl_inst_<lpl>_<cwl>.hip).
#pragma once
#include "kernels.h"

namespace pamd {

hipError_t launch_scl_inst_1_0(const DecodeParams& P, bool i8, bool diag, dim3 grid, dim3 block, size_t lds, hipStream_t st, int* occ);
hipError_t launch_scl_inst_2_0(const DecodeParams& P, bool i8, bool diag, dim3 grid, dim3 block, size_t lds, hipStream_t st, int* occ);
hipError_t launch_scl_inst_3_0(const DecodeParams& P, bool i8, bool diag, dim3 grid, dim3 block, size_t lds, hipStream_t st, int* occ);
hipError_t launch_scl_inst_4_0(const DecodeParams& P, bool i8, bool diag, dim3 grid, dim3 block, size_t lds, hipStream_t st, int* occ);
hipError_t launch_scl_inst_5_0(const DecodeParams& P, bool i8, bool diag, dim3 grid, dim3 block, size_t lds, hipStream_t st, int* occ);
hipError_t launch_scl_inst_2_1(const DecodeParams& P, bool i8, bool diag, dim3 grid, dim3 block, size_t lds, hipStream_t st, int* occ);
hipError_t launch_scl_inst_3_1(const DecodeParams& P, bool i8, bool diag, dim3 grid, dim3 block, size_t lds, hipStream_t st, int* occ);
hipError_t launch_scl_inst_3_2(const DecodeParams& P, bool i8, bool diag, dim3 grid, dim3 block, size_t lds, hipStream_t st, int* occ);
hipError_t launch_scl_inst_2_2(const DecodeParams& P, bool i8, bool diag, dim3 grid, dim3 block, size_t lds, hipStream_t st, int* occ);
hipError_t launch_scl_inst_2_3(const DecodeParams& P, bool i8, bool diag, dim3 grid, dim3 block, size_t lds, hipStream_t st, int* occ);

}  // namespace pamd
