// The lane-per-path list kernel with the RateOne / SPC leaves of 256 LLRs (high-rate codes), as its own translation unit: see the
// head of scl_lane_kernel.hip.
#define LN_BIG 1
#include "scl_lane_kernel.hip"
