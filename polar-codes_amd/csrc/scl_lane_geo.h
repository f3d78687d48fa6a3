// The lane-per-path list kernel under its other geometries (scl_lane_kernel.hip compiled once per translation unit
// scl_lane_n<logN>_l<lcl>[_char].hip): block length N = 2^logN, LC = 2^lcl lanes -- list positions -- per codeword (2 .. 32: list
// sizes 2, 3 .. 4, 5 .. 8, 9 .. 16, 17 .. 32), 64 / LC codewords per wave.  The default geometry (N = 1024, 8 lanes: launch_scl_lane and its three siblings in kernels.h) is not
// part of this list; N = 4096 with two lanes (32 codewords of 16 KB per wave, 7.9 against 12.9 M cw/s on the path groups) is not built.
// X(logN, lcl): the float decoder's kernels; Y(logN, lcl): the int8 decoder's.
#pragma once
#include "kernels.h"

#define PAMD_LANE_GEOS(X) X(12, 5) X(10, 4) X(10, 5) X(11, 3) X(9, 3) X(9, 4) X(9, 5) X(11, 4) X(11, 5) X(12, 3) X(12, 4) \
    X(9, 1) X(9, 2) X(10, 1) X(10, 2) X(11, 1) X(11, 2) X(12, 2) X(8, 1) X(8, 2) X(8, 3) X(8, 4) X(8, 5) \
    X(7, 1) X(7, 2) X(7, 3) X(7, 4) X(7, 5) X(6, 1) X(6, 2) X(6, 3) X(6, 4) X(6, 5)
#define PAMD_LANE_GEOS_CHAR(Y) Y(9, 3) Y(9, 4) Y(9, 5) Y(10, 4) Y(10, 5) Y(11, 3) Y(11, 4) Y(11, 5) Y(12, 3) Y(12, 4) Y(12, 5) \
    Y(9, 1) Y(9, 2) Y(10, 1) Y(10, 2) Y(11, 1) Y(11, 2) Y(12, 2) Y(8, 1) Y(8, 2) Y(8, 3) Y(8, 4) Y(8, 5) \
    Y(7, 1) Y(7, 2) Y(7, 3) Y(7, 4) Y(7, 5) Y(6, 1) Y(6, 2) Y(6, 3) Y(6, 4) Y(6, 5)
// N >= 2048 with an all-frozen leaf of 512 / 1024 LLRs in the tree (low rates): the same kernels with that leaf, as translation
// units of their own (scl_lane_n<logN>_l<lcl>_hi.hip) -- the code it adds moved cfg 5 by 2.5 % when it sat in the one kernel
#define PAMD_LANE_GEOS_HI(Z) Z(11, 3) Z(11, 4) Z(11, 5) Z(12, 3) Z(12, 4) Z(12, 5) Z(11, 1) Z(11, 2) Z(12, 2)
#define PAMD_LANE_GEOS_CHAR_HI(W) W(11, 3) W(11, 4) W(11, 5) W(12, 3) W(12, 4) W(12, 5) W(11, 1) W(11, 2) W(12, 2)

namespace pamd {

#define PAMD_LANE_DECL(LOGN, LCL)                                                                                    \
    hipError_t launch_scl_lane_n##LOGN##_l##LCL(const DecodeParams& P, bool diag, unsigned grid, hipStream_t st);     \
    unsigned scl_lane_n##LOGN##_l##LCL##_arena_words();
#define PAMD_LANE_DECL_CHAR(LOGN, LCL)                                                                                    \
    hipError_t launch_scl_lane_n##LOGN##_l##LCL##_char(const DecodeParams& P, bool diag, unsigned grid, hipStream_t st);   \
    unsigned scl_lane_n##LOGN##_l##LCL##_char_arena_words();
#define PAMD_LANE_DECL_HI(LOGN, LCL)                                                                                    \
    hipError_t launch_scl_lane_n##LOGN##_l##LCL##_hi(const DecodeParams& P, bool diag, unsigned grid, hipStream_t st);     \
    unsigned scl_lane_n##LOGN##_l##LCL##_hi_arena_words();
PAMD_LANE_GEOS(PAMD_LANE_DECL)
PAMD_LANE_GEOS_CHAR(PAMD_LANE_DECL_CHAR)
#define PAMD_LANE_DECL_CHAR_HI(LOGN, LCL)                                                                                    \
    hipError_t launch_scl_lane_n##LOGN##_l##LCL##_char_hi(const DecodeParams& P, bool diag, unsigned grid, hipStream_t st);   \
    unsigned scl_lane_n##LOGN##_l##LCL##_char_hi_arena_words();
PAMD_LANE_GEOS_HI(PAMD_LANE_DECL_HI)
PAMD_LANE_GEOS_CHAR_HI(PAMD_LANE_DECL_CHAR_HI)
#undef PAMD_LANE_DECL_CHAR_HI
#undef PAMD_LANE_DECL_HI
#undef PAMD_LANE_DECL
#undef PAMD_LANE_DECL_CHAR

struct LaneKernel {
    unsigned logN, lcl;
    bool chr;
    bool hi;  // carries the leaves above stage 8
    hipError_t (*launch)(const DecodeParams&, bool, unsigned, hipStream_t);
    unsigned (*arena_words)();
    const char* name;
};
// the kernel of (logN, list size, decoder type, leaves above stage 8 in the tree), or nullptr: the smallest geometry whose lanes hold the list
inline const LaneKernel* find_lane_kernel(unsigned logN, unsigned L, bool chr, bool hi)
{
    static const LaneKernel table[] = {
#define PAMD_LANE_ROW(LOGN, LCL) { LOGN, LCL, false, false, launch_scl_lane_n##LOGN##_l##LCL, scl_lane_n##LOGN##_l##LCL##_arena_words, "scl_lane_n" #LOGN "_l" #LCL "_kernel" },
#define PAMD_LANE_ROW_HI(LOGN, LCL) { LOGN, LCL, false, true, launch_scl_lane_n##LOGN##_l##LCL##_hi, scl_lane_n##LOGN##_l##LCL##_hi_arena_words, "scl_lane_n" #LOGN "_l" #LCL "_hi_kernel" },
#define PAMD_LANE_ROW_CHAR(LOGN, LCL) { LOGN, LCL, true, false, launch_scl_lane_n##LOGN##_l##LCL##_char, scl_lane_n##LOGN##_l##LCL##_char_arena_words, "scl_lane_n" #LOGN "_l" #LCL "_char_kernel" },
#define PAMD_LANE_ROW_CHAR_HI(LOGN, LCL) { LOGN, LCL, true, true, launch_scl_lane_n##LOGN##_l##LCL##_char_hi, scl_lane_n##LOGN##_l##LCL##_char_hi_arena_words, "scl_lane_n" #LOGN "_l" #LCL "_char_hi_kernel" },
        PAMD_LANE_GEOS(PAMD_LANE_ROW) PAMD_LANE_GEOS_CHAR(PAMD_LANE_ROW_CHAR) PAMD_LANE_GEOS_HI(PAMD_LANE_ROW_HI) PAMD_LANE_GEOS_CHAR_HI(PAMD_LANE_ROW_CHAR_HI)
#undef PAMD_LANE_ROW_CHAR_HI
#undef PAMD_LANE_ROW_HI
#undef PAMD_LANE_ROW
#undef PAMD_LANE_ROW_CHAR
        { 0, 0, false, false, nullptr, nullptr, nullptr }
    };
    const LaneKernel* best = nullptr;
    for (const LaneKernel* k = table; k->launch; ++k)
        if (k->logN == logN && k->chr == chr && k->hi == hi && L <= (1u << k->lcl) && L > (1u << k->lcl) / 2 && (!best || k->lcl < best->lcl)) best = k;
    return best;
}

}  // namespace pamd
