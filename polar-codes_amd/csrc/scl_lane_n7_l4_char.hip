// lane-per-path list kernel of the int8 decoder (SclFipChar), N = 128, up to 16 paths (4 codewords per wave): the channel row IS stage 7 (the LDS column of the first path)
#define LN_LOGN 7
#define LN_LCL 4
#ifndef LN_BIG
#define LN_BIG 0
#endif
#define LN_CHAR 1
#define LN_KERNEL_NAME scl_lane_n7_l4_char_kernel
#define LN_LAUNCH_NAME launch_scl_lane_n7_l4_char
#define LN_ARENA_NAME scl_lane_n7_l4_char_arena_words
#include "scl_lane_kernel.hip"
