// List-decoder kernels with 4 lanes per path and 4 codeword(s) per wave (scl_kernel.h).
#include "scl_kernel.h"
#include "scl_launch.h"

namespace pamd {

hipError_t launch_scl_inst_2_2(const DecodeParams& P, bool i8, bool diag, dim3 grid, dim3 block, size_t lds, hipStream_t st, int* occ)
{
    return launch_scl_t<2, 2>(P, i8, diag, grid, block, lds, st, occ);
}

}  // namespace pamd
