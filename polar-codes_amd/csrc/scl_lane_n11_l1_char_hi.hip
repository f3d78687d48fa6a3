// ... of the int8 decoder with the all-frozen leaves of 512 / 1024 LLRs (low rates), N = 2048, up to 2 paths
#define LN_LOGN 11
#define LN_LCL 1
#define LN_BIG 1
#define LN_CHAR 1
#define LN_HILEAF 1
#define LN_KERNEL_NAME scl_lane_n11_l1_char_hi_kernel
#define LN_LAUNCH_NAME launch_scl_lane_n11_l1_char_hi
#define LN_ARENA_NAME scl_lane_n11_l1_char_hi_arena_words
#include "scl_lane_kernel.hip"
