// lane-per-path list kernel of the int8 decoder (SclFipChar), N = 256, up to 2 paths (32 codewords per wave): the channel row IS stage 8 (no top-of-tree pass)
#define LN_LOGN 8
#define LN_LCL 1
#ifndef LN_BIG
#define LN_BIG 0  // (a leaf of 256 LLRs would be the whole code: such trees keep the path groups)
#endif
#define LN_CHAR 1
#define LN_KERNEL_NAME scl_lane_n8_l1_char_kernel
#define LN_LAUNCH_NAME launch_scl_lane_n8_l1_char
#define LN_ARENA_NAME scl_lane_n8_l1_char_arena_words
#include "scl_lane_kernel.hip"
