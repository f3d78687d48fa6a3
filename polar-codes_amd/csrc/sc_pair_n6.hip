// two-lanes-per-codeword SC kernel, N = 64: sc_pair_kernel.hip with the channel row as local stage 5 (float and int8 decoder, fp32 and int8 rows)
#define PR_LOGN 6
#define PR_KERNEL_NAME sc_pair_n6_kernel
#define PR_LAUNCH_NAME launch_sc_pair_n6
#include "sc_pair_kernel.hip"
