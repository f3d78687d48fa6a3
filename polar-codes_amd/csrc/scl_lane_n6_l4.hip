// lane-per-path list kernel, N = 64, up to 16 paths (4 codewords per wave): the channel row IS stage 6 (a register block, copied into every lane)
#define LN_LOGN 6
#define LN_LCL 4
#ifndef LN_OPAQUE_LANE
#define LN_OPAQUE_LANE 0
#endif
#ifndef LN_BIG
#define LN_BIG 0
#endif
#define LN_KERNEL_NAME scl_lane_n6_l4_kernel
#define LN_LAUNCH_NAME launch_scl_lane_n6_l4
#define LN_ARENA_NAME scl_lane_n6_l4_arena_words
#include "scl_lane_kernel.hip"
