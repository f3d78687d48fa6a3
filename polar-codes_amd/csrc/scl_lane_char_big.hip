// The lane-per-path list kernel for the int8 decoder with the RateOne / SPC leaves of 256 LLRs (high-rate codes): see the head
// of scl_lane_kernel.hip.
#define LN_CHAR 1
#define LN_BIG 1
#include "scl_lane_kernel.hip"
