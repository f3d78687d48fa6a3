// lane-per-path list kernel of the float decoder on INT8 rows (Decoder::decode_vector(const char*): the rows are widened exactly,
// FloatContainer::insertLlr(const char*) bitcontainer.cpp:202-207, the arithmetic is the float decoder's): scl_lane_kernel.hip
// with the channel rows read as bytes and widened in the top-of-tree pass
#define LN_I8ROWS 1
#define LN_KERNEL_NAME scl_lane_i8_kernel
#define LN_LAUNCH_NAME launch_scl_lane_i8
#define LN_ARENA_NAME scl_lane_i8_arena_words
#include "scl_lane_kernel.hip"
