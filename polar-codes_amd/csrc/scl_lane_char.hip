// The lane-per-path list kernel for the int8 decoder (SclFipChar), as its own translation unit: see the head of
// scl_lane_kernel.hip.
#define LN_CHAR 1
#include "scl_lane_kernel.hip"
