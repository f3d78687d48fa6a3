// lane-per-path list kernel, N = 1024, up to 2 paths (32 codewords per wave): scl_lane_kernel.hip under another geometry
#define LN_LOGN 10
#define LN_LCL 1
#ifndef LN_BIG
#define LN_BIG 1
#endif
#define LN_KERNEL_NAME scl_lane_n10_l1_kernel
#define LN_LAUNCH_NAME launch_scl_lane_n10_l1
#define LN_ARENA_NAME scl_lane_n10_l1_arena_words
#include "scl_lane_kernel.hip"
