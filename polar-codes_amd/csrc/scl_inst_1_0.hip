// List-decoder kernels with 2 lanes per path and 1 codeword(s) per wave (scl_kernel.h).
#include "scl_kernel.h"
#include "scl_launch.h"

namespace pamd {

hipError_t launch_scl_inst_1_0(const DecodeParams& P, bool i8, bool diag, dim3 grid, dim3 block, size_t lds, hipStream_t st, int* occ)
{
    return launch_scl_t<1, 0>(P, i8, diag, grid, block, lds, st, occ);
}

}  // namespace pamd
