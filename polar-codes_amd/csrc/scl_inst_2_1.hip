// List-decoder kernels with 4 lanes per path and 2 codeword(s) per wave (scl_kernel.h).
#define PAMD_POLAR_F_MINF 1  // (device_common.h: the 3-instruction f pays off for this kernel)
#include "scl_kernel.h"
#include "scl_launch.h"

namespace pamd {

hipError_t launch_scl_inst_2_1(const DecodeParams& P, bool i8, bool diag, dim3 grid, dim3 block, size_t lds, hipStream_t st, int* occ)
{
    return launch_scl_t<2, 1>(P, i8, diag, grid, block, lds, st, occ);
}

}  // namespace pamd
