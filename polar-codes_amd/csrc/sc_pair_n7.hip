// two-lanes-per-codeword SC kernel, N = 128: sc_pair_kernel.hip with the channel row as local stage 6 (float and int8 decoder, fp32 and int8 rows)
#define PR_LOGN 7
#define PR_KERNEL_NAME sc_pair_n7_kernel
#define PR_LAUNCH_NAME launch_sc_pair_n7
#include "sc_pair_kernel.hip"
