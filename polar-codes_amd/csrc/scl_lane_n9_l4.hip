// lane-per-path list kernel, N = 512, up to 16 paths (4 codewords per wave): scl_lane_kernel.hip under another geometry
#define LN_LOGN 9
#define LN_LCL 4
#ifndef LN_OPAQUE_LANE
#define LN_OPAQUE_LANE 0  // (N = 512: no scratch without the opaque lane number, and 6 % faster -- A/B on one box)
#endif
#define LN_BIG 1
#define LN_KERNEL_NAME scl_lane_n9_l4_kernel
#define LN_LAUNCH_NAME launch_scl_lane_n9_l4
#define LN_ARENA_NAME scl_lane_n9_l4_arena_words
#include "scl_lane_kernel.hip"
