// lane-per-path list kernel, N = 4096, up to 4 paths (16 codewords per wave): scl_lane_kernel.hip under another geometry
#define LN_LOGN 12
#define LN_LCL 2
#ifndef LN_BIG
#define LN_BIG 1
#endif
#define LN_KERNEL_NAME scl_lane_n12_l2_kernel
#define LN_LAUNCH_NAME launch_scl_lane_n12_l2
#define LN_ARENA_NAME scl_lane_n12_l2_arena_words
#include "scl_lane_kernel.hip"
