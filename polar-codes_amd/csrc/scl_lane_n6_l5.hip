// lane-per-path list kernel, N = 64, up to 32 paths (2 codewords per wave): the channel row IS stage 6 (a register block, copied into every lane)
#define LN_LOGN 6
#define LN_LCL 5
#ifndef LN_OPAQUE_LANE
#define LN_OPAQUE_LANE 0
#endif
#ifndef LN_BIG
#define LN_BIG 0
#endif
#define LN_KERNEL_NAME scl_lane_n6_l5_kernel
#define LN_LAUNCH_NAME launch_scl_lane_n6_l5
#define LN_ARENA_NAME scl_lane_n6_l5_arena_words
#include "scl_lane_kernel.hip"
