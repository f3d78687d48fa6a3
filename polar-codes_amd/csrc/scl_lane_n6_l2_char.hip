// lane-per-path list kernel of the int8 decoder (SclFipChar), N = 64, up to 4 paths (16 codewords per wave): the channel row IS stage 6 (a register block, copied into every lane)
#define LN_LOGN 6
#define LN_LCL 2
#ifndef LN_BIG
#define LN_BIG 0
#endif
#define LN_CHAR 1
#define LN_KERNEL_NAME scl_lane_n6_l2_char_kernel
#define LN_LAUNCH_NAME launch_scl_lane_n6_l2_char
#define LN_ARENA_NAME scl_lane_n6_l2_char_arena_words
#include "scl_lane_kernel.hip"
