// lane-per-path list kernel of the int8 decoder (SclFipChar), N = 4096, up to 4 paths: scl_lane_kernel.hip under another geometry
#define LN_LOGN 12
#define LN_LCL 2
#define LN_BIG 1
#define LN_CHAR 1
#define LN_KERNEL_NAME scl_lane_n12_l2_char_kernel
#define LN_LAUNCH_NAME launch_scl_lane_n12_l2_char
#define LN_ARENA_NAME scl_lane_n12_l2_char_arena_words
#include "scl_lane_kernel.hip"
