// ... with the RateOne / SPC leaves of 256 LLRs (see scl_lane_big.hip)
#define LN_I8ROWS 1
#define LN_BIG 1
#define LN_KERNEL_NAME scl_lane_i8_big_kernel
#define LN_LAUNCH_NAME launch_scl_lane_i8_big
#define LN_ARENA_NAME scl_lane_i8_big_arena_words
#include "scl_lane_kernel.hip"
