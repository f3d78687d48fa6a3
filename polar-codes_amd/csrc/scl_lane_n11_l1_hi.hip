// lane-per-path list kernel, N = 2048, up to 2 paths (32 codewords per wave), with the all-frozen leaves of 512 / 1024 LLRs (low rates): see scl_lane_geo.h
#define LN_LOGN 11
#define LN_LCL 1
#ifndef LN_BIG
#define LN_BIG 1
#endif
#define LN_HILEAF 1
#define LN_KERNEL_NAME scl_lane_n11_l1_hi_kernel
#define LN_LAUNCH_NAME launch_scl_lane_n11_l1_hi
#define LN_ARENA_NAME scl_lane_n11_l1_hi_arena_words
#include "scl_lane_kernel.hip"
