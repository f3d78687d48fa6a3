// Fast-SSCL list decoder, lane-per-path form: N = 1024, list size <= 8.
//
// Reference: SclAvxFloat  src/polarcode/decoding/scl_avx_float.cpp (PathList :19-171, nodes :229-621, extractBestPath
// :711-750).  The path-group kernels of scl_kernel.h give every path 4 lanes and keep the tree in LDS / a global arena;
// here ONE LANE IS ONE PATH: a wave decodes 8 codewords x 8 paths, and the decoding tree of a path lives in the
// lane's own registers:
//
//   stage t <= 6 (block 2^t LLRs)   VGPRs v[128..253], hand-allocated: the compiler is confined to v0..v103
//                                   (amdgpu_num_vgpr) and never sees them; the ops below are asm blocks on fixed
//                                   register numbers (`v[%[d]+k]`, %[d] an immediate)
//   stage 7                         LDS, [16-byte chunk][lane]  (32 KB per wave)
//   stage 8                         per-wave arena in global memory, [16-byte chunk][lane]  (64 KB per wave)
//   stage 9                         never stored: recomputed from the channel row (LN_VT)
//   partial sums                    bit-packed: stages <= 7 in VGPRs v[104..127], stages 8-9 in the arena
//
// One wave per SIMD (4 single-wave workgroups per CU: 40 KB of LDS each), i.e. 32 codewords in flight per CU.
// With one lane per path there is no cross-lane work inside f / g / the leaves at all; what crosses lanes is the
// list management (candidate selection among the 8 lanes of a codeword: a merge network on packed keys with DPP
// partners) and the lazy copy: the reference's DataPool is replaced by a slot word per path -- field t names the lane
// whose registers / LDS column / arena column hold LLR[t] until the G of that stage has read them (ds_bpermute for the
// register stages), and the holder of the left child's bits afterwards.  A fork moves the slot word, nothing else.
//
// Bit-exactness: f exact; g is one fp32 addition (v_fma_f32 with a factor of +-1.0: one rounding, same result); the
// metric sums follow the reference's 8-accumulator order (avxconvenience.h:256-272); findWeakLlrs and
// simplePartialSortDescending (arrayfuncs.h:161-231) are reproduced including their swap-order tie behaviour (the merge
// selection falls back to a literal emulation whenever two keys that matter agree in their upper 26 bits).
// LN_BIG (scl_lane_big.hip includes this file with it set): the same kernel with the RateOne / SPC leaves of 256 LLRs (high rates),
// under another name.  A property of the code -- the host picks the kernel per decoder.  As a SECOND translation unit on purpose:
// with those leaves in the one kernel cfg 3 lost 3.8 %, as a template flag of the one source still 2 % (the lean kernel came out
// with another register allocation); guarded by the preprocessor the lean kernel's token stream is what it was.
// LN_CHAR (scl_lane_char.hip): the int8 decoder's instantiation (CH = true), a third translation unit for the same reason.
// LN_LOGN / LN_LCL (scl_lane_geo.h and the scl_lane_n*_l*.hip stubs, round 4): the same source for other block lengths
// (N = 2^LN_LOGN = 64 .. 4096) and list sizes (LC = 2^LN_LCL = 2 .. 32 lanes per codeword, i.e. 32 .. 2 codewords per
// wave).  Whatever the block length, the tree from stage 8 down is the one described above; the stages between the channel
// row and stage 8 are recomputed from the row by the top-of-tree pass (N = 1024: two levels, LN_VT; otherwise LOGN - 8
// levels, op_vtg), their partial sums live in the arena.  N = 256 / 128 / 64: the row IS stage 8 / 7 / 6 -- copied once into the
// column / registers of the one live path and found from there through the slot word every path inherits (op_vtg_first with zero
// levels, op_copy7, op_copy6).  With the defaults (10, 3) the token stream is what it was.
#ifndef LN_BIG
#define LN_BIG 0
#endif
#ifndef LN_CHAR
#define LN_CHAR 0
#endif
#ifndef LN_I8ROWS
#define LN_I8ROWS LN_CHAR  // the channel rows are int8 (1 byte per LLR): always for the int8 decoder; for the float decoder in scl_lane_i8rows*.hip
#endif
#ifndef LN_HILEAF
#define LN_HILEAF 0  // N >= 2048: the all-frozen leaves of 512 / 1024 LLRs (low rates) -- translation units of their own (*_hi.hip), like LN_BIG
#endif
#ifndef LN_LOGN
#define LN_LOGN 10
#endif
#ifndef LN_LCL
#define LN_LCL 3
#endif
#ifdef LN_FORCE_GENERIC  // (experiments: the default geometry compiled through the code paths of the other geometries)
#define LN_GEO_DEFAULT 0
#else
#define LN_GEO_DEFAULT (LN_LOGN == 10 && LN_LCL == 3)
#endif
// LN_PKFMA: the g's four v_fma_f32 as two v_pk_fma_f32 (and a + b as v_pk_add_f32).  At one wave per SIMD an instruction of any
// kind takes one 4-cycle issue slot, a packed one included (measured on cfg 5: ten instead of twelve instructions per g block of the
// top-of-tree pass = +1.7 %), and the packed fma is the same fused operation per component: results unchanged bit for bit.
// (A/B on one box with the packed forms in every g and a + b: cfg 3 36.82 -> 37.27, cfg 4 47.7 -> 48.7, cfg 5 1.378 -> 1.415,
//  cfg3_i8 27.85 -> 28.07 M cw/s.  -DLN_PKFMA=0 restores the scalar forms.)
// LN_LINKEY: linear selection keys (select_fork).  On where it was measured to pay (A/B on one box, rate 1/2, AWGN at 2 dB):
// 32 lanes at N = 2048 / 4096 + 4 ... 5 % (cfg 5: 1.43 -> 1.50 M cw/s).  16 lanes, N <= 1024 and the 8-lane kernels at
// N = 2048 / 4096: the two reductions in front of the merge network cost a lone wave more than the rarer exact-metric fix-ups /
// tie emulations save (- 0.5 ... - 5 %).  (A variant that carried the scale from fork to fork instead of reducing it in front of
// every network was exact and changed none of these numbers: the reductions are not what the map costs at N = 1024.)
// LN_LINKEY_NOHI: ... measured from 0 instead of from the best metric of the codeword (one reduction less in front of the network, at most one
// bit of resolution lost: metrics and span are of the same order once the list is full): cfg 5 1.489 -> 1.505 M cw/s, A/B on one box
#ifndef LN_LINKEY_NOHI
#define LN_LINKEY_NOHI 1
#endif
#ifndef LN_LINKEY
#define LN_LINKEY (!LN_CHAR && LN_LCL == 5 && LN_LOGN >= 11)
#endif
#ifndef LN_PKFMA
#define LN_PKFMA 1
#endif
#ifdef LN_KERNEL_NAME
// (the stubs of the other geometries -- and of the float decoder on int8 rows -- name their kernel and launcher themselves:
//  LN_KERNEL_NAME, LN_LAUNCH_NAME, LN_ARENA_NAME)
#define scl_lane_kernel LN_KERNEL_NAME
#elif LN_BIG && LN_CHAR
#define scl_lane_kernel scl_lane_char_big_kernel
#elif LN_BIG
#define scl_lane_kernel scl_lane_big_kernel
#elif LN_CHAR
#define scl_lane_kernel scl_lane_char_kernel
#endif
#define PAMD_POLAR_F_MINF 1  // (device_common.h: the 3-instruction f)
#include "device_common.h"
#include "lane_runs.inc"

#include <utility>

namespace pamd {
namespace lane {

#if LN_I8ROWS
using RowT = int8_t;  // element type of the channel rows in HBM
#else
using RowT = float;
#endif
// ---- register map (per lane) ---------------------------------------------------------------------------------------
constexpr int KV = 104;      // the compiler's VGPR budget: v0 .. v103
constexpr int RBL0 = 104;    // left-child / root output bits of stage t <= 5: v[104 + t]
constexpr int RBR0 = 110;    // right-child output bits of stage t <= 5: v[110 + t]
constexpr int RBL6 = 116, RBR6 = 118;  // stage 6: 2 words each
constexpr int RBL7 = 120, RBR7 = 124;  // stage 7: 4 words each
__host__ __device__ constexpr int RS(int t) { return t == 1 ? 128 : t == 2 ? 130 : t == 3 ? 134 : t == 4 ? 142 : t == 5 ? 158 : 190; }
__host__ __device__ constexpr int RB(int side, int t) { return t <= 5 ? (side ? RBR0 : RBL0) + t : t == 6 ? (side ? RBR6 : RBL6) : (side ? RBR7 : RBL7); }
// geometry (see the head of the file): block length, lanes per codeword, codewords per wave, bits of a slot field
constexpr int LOGN = LN_LOGN, LCL = LN_LCL, LC = 1 << LCL, CPW = 64 / LC;
constexpr int TOPD = LOGN - 8;                     // levels between the channel row and stage 8
constexpr int PB = LCL + 3;                        // payload bits of a selection key: (lane of the codeword, candidate)
constexpr uint32_t PBM = (1u << PB) - 1u;
constexpr uint32_t PL_DEAD = LC == 8 ? 15u : 99u;  // list position of a lane that hosts no path
// arena of one wave (32-bit words): stage-8 LLRs [64 chunks][64 lanes][4], then bit words [w][lane], then a dump area
constexpr uint32_t AR_S8 = 0;
#if LN_LOGN == 10
constexpr uint32_t AR_BL9 = 64 * 64 * 4;           // 16 words: output of the root's left child
constexpr uint32_t AR_BR9 = AR_BL9 + 16 * 64;      // 16 words: output of the root's right child
constexpr uint32_t AR_BL8 = AR_BR9 + 16 * 64;      // 8 words
constexpr uint32_t AR_BR8 = AR_BL8 + 8 * 64;       // 8 words
#else
// bits of the blocks of stage t = 8 .. LOGN - 1 (2^(t-5) words per lane and side), the highest stage first
__host__ __device__ constexpr uint32_t ar_bitwords(int t) { return 1u << (t - 5); }
__host__ __device__ constexpr uint32_t AR_BL(int t)
{
    uint32_t o = 64 * 64 * 4;
    for (int u = LOGN - 1; u > t; --u) o += 2 * ar_bitwords(u) * 64;
    return o;
}
__host__ __device__ constexpr uint32_t AR_BR(int t) { return AR_BL(t) + ar_bitwords(t) * 64; }
constexpr uint32_t AR_BL8 = AR_BL(8), AR_BR8 = AR_BR(8);
#endif
constexpr uint32_t AR_DUMP = AR_BR8 + 8 * 64;      // 16 chunks x 64 lanes x 4: stage 5 / 6 leaf input, read back chunk-wise
#if LN_LOGN == 10
constexpr uint32_t AR_WORDS = AR_DUMP + 16 * 64 * 4;
#else
// "virtual rows": the leftmost block of every stage 9 .. LOGN - 1 (f of the channel row, f of that, ...) does not depend on
// the path -- computed once per codeword by the first pass, every later top-of-tree pass whose upper levels are f's starts
// from the highest one it can use instead of the channel row.  Per codeword: stage LOGN - 1 first, then LOGN - 2, ...
constexpr uint32_t AR_VROW = AR_DUMP + 16 * 64 * 4;
constexpr uint32_t VROW_FLOATS = LOGN >= 9 ? (1u << LOGN) - 512u : 0u;  // (N = 256: stage 8 is the channel row, nothing above it)
__host__ __device__ constexpr uint32_t vrow_off(int t) { return (1u << LOGN) - (2u << t); }  // (floats in front of stage t's row)
constexpr uint32_t AR_WORDS = AR_VROW + (uint32_t)CPW * VROW_FLOATS;
#endif
// LDS of one wave (32-bit words)
constexpr uint32_t LDS_S7 = 0;                     // [32 chunks][64 lanes][4 words]
constexpr uint32_t LDS_SCR = 8192;                 // scratch: 2048 words
constexpr uint32_t LDS_WORDS = 10240;
constexpr uint32_t SCR_CM = LDS_SCR;               // candidate metrics [lane][8] floats
constexpr uint32_t SCR_FORK = LDS_SCR + 512;       // fork record [lane][8] words: sign words 0-3, weak indices, parity
constexpr uint32_t SCR_SEL = LDS_SCR + 1024;       // literal selection: ranks [codeword][8]

template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// ---- accessors for the hand-allocated registers (volatile: the compiler does not know their contents change) -------
template <int R>
__device__ __forceinline__ uint32_t rdu()
{
    uint32_t x;
    asm volatile("v_mov_b32 %0, v[%1]" : "=v"(x) : "n"(R));
    return x;
}
template <int R>
__device__ __forceinline__ float rdf()
{
    return bitsf(rdu<R>());
}
template <int R>
__device__ __forceinline__ void wru(uint32_t x)
{
    asm volatile("v_mov_b32 v[%0], %1" ::"n"(R), "v"(x));
}
// register R of lane addr / 4
template <int R>
__device__ __forceinline__ uint32_t rd_lane(uint32_t addr)
{
    uint32_t x;
    asm volatile("ds_bpermute_b32 %0, %1, v[%2]\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x) : "v"(addr), "n"(R));
    return x;
}

// ---- f / g blocks on registers ----------------------------------------------------------------------------------------
// text of f over 4 consecutive elements: v[D+k] = f(v[A+k], v[B+k]), D may equal A (the xor is taken first)
#if LN_PKFMA
// (packed: the sign of a product IS the xor of the signs -- zeros, denormals and infinities included --, so one v_pk_mul_f32
//  replaces two v_xor_b32: ten instructions per block instead of twelve.  LN_FPAIR: a free even register pair for the products --
//  v[254:255] while a walk runs (the stage registers end at v253), v[126:127] in the top-of-tree passes, whose register sets reach
//  v255 while the bit registers v[104 .. 127] are dead.)
#define LN_FPAIR "254"
#define LN_F4T(D_, A_, B_)                                                                               \
    "v_pk_mul_f32 v[" LN_FPAIR ":" LN_FPAIR "+1], v[" A_ ":" A_ "+1], v[" B_ ":" B_ "+1]\n\t"          \
    "v_min_f32_e64 v[" D_ "], |v[" A_ "]|, |v[" B_ "]|\n\t"                                             \
    "v_min_f32_e64 v[" D_ "+1], |v[" A_ "+1]|, |v[" B_ "+1]|\n\t"                                       \
    "v_bfi_b32 v[" D_ "], %[am], v[" D_ "], v[" LN_FPAIR "]\n\t"                                        \
    "v_bfi_b32 v[" D_ "+1], %[am], v[" D_ "+1], v[" LN_FPAIR "+1]\n\t"                                  \
    "v_pk_mul_f32 v[" LN_FPAIR ":" LN_FPAIR "+1], v[" A_ "+2:" A_ "+3], v[" B_ "+2:" B_ "+3]\n\t"      \
    "v_min_f32_e64 v[" D_ "+2], |v[" A_ "+2]|, |v[" B_ "+2]|\n\t"                                       \
    "v_min_f32_e64 v[" D_ "+3], |v[" A_ "+3]|, |v[" B_ "+3]|\n\t"                                       \
    "v_bfi_b32 v[" D_ "+2], %[am], v[" D_ "+2], v[" LN_FPAIR "]\n\t"                                    \
    "v_bfi_b32 v[" D_ "+3], %[am], v[" D_ "+3], v[" LN_FPAIR "+1]\n\t"
#else
#define LN_F4T(D_, A_, B_)                                       \
    "v_xor_b32 %[t0], v[" A_ "], v[" B_ "]\n\t"                  \
    "v_xor_b32 %[t1], v[" A_ "+1], v[" B_ "+1]\n\t"              \
    "v_xor_b32 %[t2], v[" A_ "+2], v[" B_ "+2]\n\t"              \
    "v_xor_b32 %[t3], v[" A_ "+3], v[" B_ "+3]\n\t"              \
    "v_min_f32_e64 v[" D_ "], |v[" A_ "]|, |v[" B_ "]|\n\t"      \
    "v_min_f32_e64 v[" D_ "+1], |v[" A_ "+1]|, |v[" B_ "+1]|\n\t" \
    "v_min_f32_e64 v[" D_ "+2], |v[" A_ "+2]|, |v[" B_ "+2]|\n\t" \
    "v_min_f32_e64 v[" D_ "+3], |v[" A_ "+3]|, |v[" B_ "+3]|\n\t" \
    "v_bfi_b32 v[" D_ "], %[am], v[" D_ "], %[t0]\n\t"           \
    "v_bfi_b32 v[" D_ "+1], %[am], v[" D_ "+1], %[t1]\n\t"       \
    "v_bfi_b32 v[" D_ "+2], %[am], v[" D_ "+2], %[t2]\n\t"       \
    "v_bfi_b32 v[" D_ "+3], %[am], v[" D_ "+3], %[t3]\n\t"
#endif
// text of g over 4 consecutive elements whose inputs are in place: v[D+k] = g(v[D+k], v[B+k], bit E+k of %[w])
#if LN_PKFMA
// (packed: the +-1.0 factors pair up in v[254:255], free while a walk runs -- the stage registers end at v253)
#define LN_G4T(D_, B_, S0, S1, S2, S3)                            \
    "v_lshlrev_b32 %[t0], " #S0 ", %[w]\n\t"                     \
    "v_lshlrev_b32 %[t1], " #S1 ", %[w]\n\t"                     \
    "v_and_or_b32 v254, %[t0], %[sm], 1.0\n\t"                   \
    "v_and_or_b32 v255, %[t1], %[sm], 1.0\n\t"                   \
    "v_lshlrev_b32 %[t2], " #S2 ", %[w]\n\t"                     \
    "v_lshlrev_b32 %[t3], " #S3 ", %[w]\n\t"                     \
    "v_pk_fma_f32 v[" D_ ":" D_ "+1], v[" D_ ":" D_ "+1], v[254:255], v[" B_ ":" B_ "+1]\n\t" \
    "v_and_or_b32 v254, %[t2], %[sm], 1.0\n\t"                   \
    "v_and_or_b32 v255, %[t3], %[sm], 1.0\n\t"                   \
    "v_pk_fma_f32 v[" D_ "+2:" D_ "+3], v[" D_ "+2:" D_ "+3], v[254:255], v[" B_ "+2:" B_ "+3]\n\t"
#else
#define LN_G4T(D_, B_, S0, S1, S2, S3)                            \
    "v_lshlrev_b32 %[t0], " #S0 ", %[w]\n\t"                     \
    "v_lshlrev_b32 %[t1], " #S1 ", %[w]\n\t"                     \
    "v_lshlrev_b32 %[t2], " #S2 ", %[w]\n\t"                     \
    "v_lshlrev_b32 %[t3], " #S3 ", %[w]\n\t"                     \
    "v_and_or_b32 %[t0], %[t0], %[sm], 1.0\n\t"                  \
    "v_and_or_b32 %[t1], %[t1], %[sm], 1.0\n\t"                  \
    "v_and_or_b32 %[t2], %[t2], %[sm], 1.0\n\t"                  \
    "v_and_or_b32 %[t3], %[t3], %[sm], 1.0\n\t"                  \
    "v_fma_f32 v[" D_ "], v[" D_ "], %[t0], v[" B_ "]\n\t"       \
    "v_fma_f32 v[" D_ "+1], v[" D_ "+1], %[t1], v[" B_ "+1]\n\t" \
    "v_fma_f32 v[" D_ "+2], v[" D_ "+2], %[t2], v[" B_ "+2]\n\t" \
    "v_fma_f32 v[" D_ "+3], v[" D_ "+3], %[t3], v[" B_ "+3]\n\t"
#endif

template <int G, int D, int A, int B>
__device__ __forceinline__ void f_block(uint32_t absmask)
{
    static_assert(G == 2 || G == 4, "block size");
    uint32_t t0, t1, t2, t3;
    if constexpr (G == 4) {
        asm volatile(LN_F4T("%[d]", "%[a]", "%[b]")
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [d] "n"(D), [a] "n"(A), [b] "n"(B), [am] "s"(absmask));
    } else {
        asm volatile(
#if LN_PKFMA
            "v_pk_mul_f32 v[254:255], v[%[a]:%[a]+1], v[%[b]:%[b]+1]\n\t"
            "v_min_f32_e64 v[%[d]], |v[%[a]]|, |v[%[b]]|\n\t"
            "v_min_f32_e64 v[%[d]+1], |v[%[a]+1]|, |v[%[b]+1]|\n\t"
            "v_bfi_b32 v[%[d]], %[am], v[%[d]], v254\n\t"
            "v_bfi_b32 v[%[d]+1], %[am], v[%[d]+1], v255"
#else
            "v_xor_b32 %[t0], v[%[a]], v[%[b]]\n\t"
            "v_xor_b32 %[t1], v[%[a]+1], v[%[b]+1]\n\t"
            "v_min_f32_e64 v[%[d]], |v[%[a]]|, |v[%[b]]|\n\t"
            "v_min_f32_e64 v[%[d]+1], |v[%[a]+1]|, |v[%[b]+1]|\n\t"
            "v_bfi_b32 v[%[d]], %[am], v[%[d]], %[t0]\n\t"
            "v_bfi_b32 v[%[d]+1], %[am], v[%[d]+1], %[t1]"
#endif
            : [t0] "=&v"(t0), [t1] "=&v"(t1)
            : [d] "n"(D), [a] "n"(A), [b] "n"(B), [am] "s"(absmask));
    }
}
// llr[T-1] = f(llr[T]) on registers, T = 2 .. 6
template <int T>
__device__ __forceinline__ void f_reg(uint32_t absmask)
{
    constexpr int h = 1 << (T - 1), A = RS(T), D = RS(T - 1);
    constexpr int G = h < 4 ? h : 4;
    static_for<h / G>([&](auto J) {
        constexpr int j = decltype(J)::value * G;
        f_block<G, D + j, A + j, A + h + j>(absmask);
    });
}
// llr[T-1] = a + b (all-frozen left child), T = 2 .. 6
template <int T>
__device__ __forceinline__ void g0_reg()
{
    constexpr int h = 1 << (T - 1), A = RS(T), D = RS(T - 1);
#if LN_PKFMA
    static_for<h / 2>([&](auto I) {  // (every stage's registers start at an even number, h >= 2)
        constexpr int i = 2 * decltype(I)::value;
        asm volatile("v_pk_add_f32 v[%0:%0+1], v[%1:%1+1], v[%2:%2+1]" ::"n"(D + i), "n"(A + i), "n"(A + h + i));
    });
#else
    static_for<h>([&](auto I) {
        constexpr int i = decltype(I)::value;
        asm volatile("v_add_f32 v[%0], v[%1], v[%2]" ::"n"(D + i), "n"(A + i), "n"(A + h + i));
    });
#endif
}
// g over G consecutive elements, inputs fetched from lane addr / 4 (ds_bpermute): a lands in v[D+k], b in a temporary.
// u: the left child's bits, bit k = element k of this block
template <int G, int D, int A, int B>
__device__ __forceinline__ void g_block_lazy(uint32_t addr, uint32_t u, uint32_t signmask)
{
    static_assert(G == 2 || G == 4 || G == 8, "block size");
    uint32_t b0, b1, b2, b3, b4, b5, b6, b7, s0, s1, s2, s3, s4, s5, s6, s7;
#define LN_BPA(k) "ds_bpermute_b32 v[%[d]+" #k "], %[ad], v[%[a]+" #k "]\n\t"
#define LN_BPB(k) "ds_bpermute_b32 %[b" #k "], %[ad], v[%[bb]+" #k "]\n\t"
#define LN_SH(k, s) "v_lshlrev_b32 %[s" #k "], " #s ", %[u]\n\t"
#define LN_AO(k) "v_and_or_b32 %[s" #k "], %[s" #k "], %[sm], 1.0\n\t"
#define LN_FM(k) "v_fma_f32 v[%[d]+" #k "], v[%[d]+" #k "], %[s" #k "], %[b" #k "]\n\t"
    if constexpr (G == 8) {
        asm volatile(LN_BPA(0) LN_BPB(0) LN_BPA(1) LN_BPB(1) LN_BPA(2) LN_BPB(2) LN_BPA(3) LN_BPB(3)
                     LN_BPA(4) LN_BPB(4) LN_BPA(5) LN_BPB(5) LN_BPA(6) LN_BPB(6) LN_BPA(7) LN_BPB(7)
                     LN_SH(0, 31) LN_SH(1, 30) LN_SH(2, 29) LN_SH(3, 28) LN_SH(4, 27) LN_SH(5, 26) LN_SH(6, 25) LN_SH(7, 24)
                     LN_AO(0) LN_AO(1) LN_AO(2) LN_AO(3) LN_AO(4) LN_AO(5) LN_AO(6) LN_AO(7)
                     "s_waitcnt lgkmcnt(8)\n\t" LN_FM(0) LN_FM(1) LN_FM(2) LN_FM(3)
                     "s_waitcnt lgkmcnt(0)\n\t" LN_FM(4) LN_FM(5) LN_FM(6) LN_FM(7)
                     : [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3), [b4] "=&v"(b4), [b5] "=&v"(b5),
                       [b6] "=&v"(b6), [b7] "=&v"(b7), [s0] "=&v"(s0), [s1] "=&v"(s1), [s2] "=&v"(s2), [s3] "=&v"(s3),
                       [s4] "=&v"(s4), [s5] "=&v"(s5), [s6] "=&v"(s6), [s7] "=&v"(s7)
                     : [d] "n"(D), [a] "n"(A), [bb] "n"(B), [ad] "v"(addr), [u] "v"(u), [sm] "s"(signmask)
                     : "memory");
    } else if constexpr (G == 4) {
        asm volatile(LN_BPA(0) LN_BPB(0) LN_BPA(1) LN_BPB(1) LN_BPA(2) LN_BPB(2) LN_BPA(3) LN_BPB(3)
                     LN_SH(0, 31) LN_SH(1, 30) LN_SH(2, 29) LN_SH(3, 28) LN_AO(0) LN_AO(1) LN_AO(2) LN_AO(3)
                     "s_waitcnt lgkmcnt(0)\n\t" LN_FM(0) LN_FM(1) LN_FM(2) LN_FM(3)
                     : [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3), [s0] "=&v"(s0), [s1] "=&v"(s1),
                       [s2] "=&v"(s2), [s3] "=&v"(s3)
                     : [d] "n"(D), [a] "n"(A), [bb] "n"(B), [ad] "v"(addr), [u] "v"(u), [sm] "s"(signmask)
                     : "memory");
    } else {
        asm volatile(LN_BPA(0) LN_BPB(0) LN_BPA(1) LN_BPB(1) LN_SH(0, 31) LN_SH(1, 30) LN_AO(0) LN_AO(1)
                     "s_waitcnt lgkmcnt(0)\n\t" LN_FM(0) LN_FM(1)
                     : [b0] "=&v"(b0), [b1] "=&v"(b1), [s0] "=&v"(s0), [s1] "=&v"(s1)
                     : [d] "n"(D), [a] "n"(A), [bb] "n"(B), [ad] "v"(addr), [u] "v"(u), [sm] "s"(signmask)
                     : "memory");
    }
#undef LN_BPA
#undef LN_BPB
#undef LN_SH
#undef LN_AO
#undef LN_FM
}
// llr[T-1] = g(llr[T] of lane addr / 4, own left bits), T = 2 .. 6
template <int T>
__device__ __forceinline__ void g_reg(uint32_t addr, uint32_t signmask)
{
    constexpr int h = 1 << (T - 1), A = RS(T), D = RS(T - 1);
    constexpr int G = h < 8 ? h : 8;
    const uint32_t u = rdu<RBL0 + (T - 1)>();
    static_for<h / G>([&](auto J) {
        constexpr int j = decltype(J)::value * G;
        g_block_lazy<G, D + j, A + j, A + h + j>(addr, u >> j, signmask);
    });
}

// ---- int8 list decoder (SclFipChar scl_fip_char.cpp, primitives fip_char.h:45-74): an int8 LLR is the float of the same value, ----
// saturation is a clamp: f = the float f with its magnitude clamped to [1, 127] (sign kept, also of a -0), g / a + b clamped
// to [-128, 127].  c127 travels as a scalar operand, -128 in a VGPR (a VALU instruction of gfx9 takes one scalar operand).
template <int R, int CNT>
__device__ __forceinline__ void clamp_f(float c127, uint32_t absmask)
{
    static_for<CNT>([&](auto I) {
        constexpr int i = decltype(I)::value;
        uint32_t t;
        const float cc = c127;
        const uint32_t am = absmask;
        asm volatile("v_med3_f32 %[t], |v[%[r]]|, 1.0, %[c]\n\tv_bfi_b32 v[%[r]], %[am], %[t], v[%[r]]" : [t] "=&v"(t) : [r] "n"(R + i), [c] "s"(cc), [am] "s"(am));
    });
}
template <int R, int CNT>
__device__ __forceinline__ void clamp_g(float c127, float m128)
{
    static_for<CNT>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const float cc = c127, mm = m128;
        asm volatile("v_med3_f32 v[%[r]], v[%[r]], %[m], %[c]" ::[r] "n"(R + i), [c] "s"(cc), [m] "v"(mm));
    });
}

// ---- stage 7 (LDS column, 16-byte chunks 1 KB apart) -> stage 6 (registers) ----------------------------------------------
// Round r loads the chunks 8r .. 8r+7 into v[D .. D+31] and their partners 16+8r .. into v[T .. T+31] (the registers of
// stage 5, dead while stage 6 is being produced), then works on them in place as they arrive.
#define LN_RD2(c) "ds_read_b128 v[%[d]+4*" #c ":%[d]+4*" #c "+3], %[ad] offset:%[oa]+1024*" #c "\n\t" \
                  "ds_read_b128 v[%[t]+4*" #c ":%[t]+4*" #c "+3], %[ad] offset:%[ob]+1024*" #c "\n\t"
template <int R>
__device__ __forceinline__ void f7_round(uint32_t addr, uint32_t absmask)
{
    uint32_t t0, t1, t2, t3;
    asm volatile(LN_RD2(0) LN_RD2(1) LN_RD2(2) LN_RD2(3) LN_RD2(4) LN_RD2(5) LN_RD2(6) LN_RD2(7)
                 "s_waitcnt lgkmcnt(14)\n\t" LN_F4T("%[d]", "%[d]", "%[t]")
                 "s_waitcnt lgkmcnt(12)\n\t" LN_F4T("%[d]+4", "%[d]+4", "%[t]+4")
                 "s_waitcnt lgkmcnt(10)\n\t" LN_F4T("%[d]+8", "%[d]+8", "%[t]+8")
                 "s_waitcnt lgkmcnt(8)\n\t" LN_F4T("%[d]+12", "%[d]+12", "%[t]+12")
                 "s_waitcnt lgkmcnt(6)\n\t" LN_F4T("%[d]+16", "%[d]+16", "%[t]+16")
                 "s_waitcnt lgkmcnt(4)\n\t" LN_F4T("%[d]+20", "%[d]+20", "%[t]+20")
                 "s_waitcnt lgkmcnt(2)\n\t" LN_F4T("%[d]+24", "%[d]+24", "%[t]+24")
                 "s_waitcnt lgkmcnt(0)\n\t" LN_F4T("%[d]+28", "%[d]+28", "%[t]+28")
                 : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                 : [d] "n"(RS(6) + 32 * R), [t] "n"(RS(5)), [ad] "v"(addr), [oa] "n"(4 * LDS_S7 + 8192 * R),
                   [ob] "n"(4 * LDS_S7 + 16384 + 8192 * R), [am] "s"(absmask)
                 : "memory");
}
// the same with g: w = the 32 left-child bits of the round's elements; ZERO: all-frozen left child (a + b)
template <int R, bool ZERO>
__device__ __forceinline__ void g7_round(uint32_t addr, uint32_t w, uint32_t signmask)
{
    uint32_t t0, t1, t2, t3;
    if constexpr (ZERO) {
#if LN_PKFMA
#define LN_ADD4(o) "v_pk_add_f32 v[%[d]+" #o ":%[d]+" #o "+1], v[%[d]+" #o ":%[d]+" #o "+1], v[%[t]+" #o ":%[t]+" #o "+1]\n\t" \
                   "v_pk_add_f32 v[%[d]+" #o "+2:%[d]+" #o "+3], v[%[d]+" #o "+2:%[d]+" #o "+3], v[%[t]+" #o "+2:%[t]+" #o "+3]\n\t"
#else
#define LN_ADD4(o) "v_add_f32 v[%[d]+" #o "], v[%[d]+" #o "], v[%[t]+" #o "]\n\t" "v_add_f32 v[%[d]+" #o "+1], v[%[d]+" #o "+1], v[%[t]+" #o "+1]\n\t" \
                   "v_add_f32 v[%[d]+" #o "+2], v[%[d]+" #o "+2], v[%[t]+" #o "+2]\n\t" "v_add_f32 v[%[d]+" #o "+3], v[%[d]+" #o "+3], v[%[t]+" #o "+3]\n\t"
#endif
        asm volatile(LN_RD2(0) LN_RD2(1) LN_RD2(2) LN_RD2(3) LN_RD2(4) LN_RD2(5) LN_RD2(6) LN_RD2(7)
                     "s_waitcnt lgkmcnt(14)\n\t" LN_ADD4(0) "s_waitcnt lgkmcnt(12)\n\t" LN_ADD4(4)
                     "s_waitcnt lgkmcnt(10)\n\t" LN_ADD4(8) "s_waitcnt lgkmcnt(8)\n\t" LN_ADD4(12)
                     "s_waitcnt lgkmcnt(6)\n\t" LN_ADD4(16) "s_waitcnt lgkmcnt(4)\n\t" LN_ADD4(20)
                     "s_waitcnt lgkmcnt(2)\n\t" LN_ADD4(24) "s_waitcnt lgkmcnt(0)\n\t" LN_ADD4(28)
                     :
                     : [d] "n"(RS(6) + 32 * R), [t] "n"(RS(5)), [ad] "v"(addr), [oa] "n"(4 * LDS_S7 + 8192 * R),
                       [ob] "n"(4 * LDS_S7 + 16384 + 8192 * R)
                     : "memory");
#undef LN_ADD4
    } else {
        asm volatile(LN_RD2(0) LN_RD2(1) LN_RD2(2) LN_RD2(3) LN_RD2(4) LN_RD2(5) LN_RD2(6) LN_RD2(7)
                     "s_waitcnt lgkmcnt(14)\n\t" LN_G4T("%[d]", "%[t]", 31, 30, 29, 28)
                     "s_waitcnt lgkmcnt(12)\n\t" LN_G4T("%[d]+4", "%[t]+4", 27, 26, 25, 24)
                     "s_waitcnt lgkmcnt(10)\n\t" LN_G4T("%[d]+8", "%[t]+8", 23, 22, 21, 20)
                     "s_waitcnt lgkmcnt(8)\n\t" LN_G4T("%[d]+12", "%[t]+12", 19, 18, 17, 16)
                     "s_waitcnt lgkmcnt(6)\n\t" LN_G4T("%[d]+16", "%[t]+16", 15, 14, 13, 12)
                     "s_waitcnt lgkmcnt(4)\n\t" LN_G4T("%[d]+20", "%[t]+20", 11, 10, 9, 8)
                     "s_waitcnt lgkmcnt(2)\n\t" LN_G4T("%[d]+24", "%[t]+24", 7, 6, 5, 4)
                     "s_waitcnt lgkmcnt(0)\n\t" LN_G4T("%[d]+28", "%[t]+28", 3, 2, 1, 0)
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [d] "n"(RS(6) + 32 * R), [t] "n"(RS(5)), [ad] "v"(addr), [oa] "n"(4 * LDS_S7 + 8192 * R),
                       [ob] "n"(4 * LDS_S7 + 16384 + 8192 * R), [w] "v"(w), [sm] "s"(signmask)
                     : "memory");
    }
}
#undef LN_RD2

// registers v[R .. R+4*NCH) -> the dump area of the arena (own column), 16-byte chunks
template <int R, int NCH>
__device__ __forceinline__ void dump_regs(uint32_t* col)
{
    static_for<NCH>([&](auto C) {
        constexpr int c = decltype(C)::value;
        uint32_t* const dst = col + c * 256;
        asm volatile("global_store_dwordx4 %0, v[%1:%1+3], off" ::"v"(dst), "n"(R + 4 * c) : "memory");
    });
}

// ---- small helpers ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 f4(float4 a, float4 b)
{
    return make_float4(polar_f(a.x, b.x), polar_f(a.y, b.y), polar_f(a.z, b.z), polar_f(a.w, b.w));
}
__device__ __forceinline__ float4 g4(float4 a, float4 b, uint32_t u)
{
    return make_float4(polar_g(a.x, b.x, u & 1u), polar_g(a.y, b.y, (u >> 1) & 1u), polar_g(a.z, b.z, (u >> 2) & 1u),
                       polar_g(a.w, b.w, (u >> 3) & 1u));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
// all-reduce over the 8 lanes of a codeword (DPP: lane ^ 1, lane ^ 2, then the mirrored lane of the other quad); 4 / 2 lanes: the
// first two / the first step
#if LN_LCL <= 3
__device__ __forceinline__ uint32_t cw_min(uint32_t v)
{
    v = umin(v, dpp_u32<0xB1>(v));
    if constexpr (LC >= 4) v = umin(v, dpp_u32<0x4E>(v));
    if constexpr (LC >= 8) v = umin(v, dpp_u32<0x141>(v));
    return v;
}
__device__ __forceinline__ uint32_t cw_max(uint32_t v)
{
    v = umax(v, dpp_u32<0xB1>(v));
    if constexpr (LC >= 4) v = umax(v, dpp_u32<0x4E>(v));
    if constexpr (LC >= 8) v = umax(v, dpp_u32<0x141>(v));
    return v;
}
__device__ __forceinline__ uint32_t cw_add(uint32_t v)
{
    v += dpp_u32<0xB1>(v);
    if constexpr (LC >= 4) v += dpp_u32<0x4E>(v);
    if constexpr (LC >= 8) v += dpp_u32<0x141>(v);
    return v;
}
#else
// (16 lanes: + the mirrored lane of the row; 32 lanes: + the other row of the pair -- v_permlane16_swap with both operands the
//  value hands every lane "the even row's" and "the odd row's", whose minimum / sum needs no lane predicate)
__device__ __forceinline__ uint32_t cw_min(uint32_t v)
{
    v = umin(v, dpp_u32<0xB1>(v));
    v = umin(v, dpp_u32<0x4E>(v));
    v = umin(v, dpp_u32<0x141>(v));
    v = umin(v, dpp_u32<0x140>(v));
    if constexpr (LC > 16) {
        const auto s = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        v = umin(s[0], s[1]);
    }
    return v;
}
__device__ __forceinline__ uint32_t cw_max(uint32_t v)
{
    v = umax(v, dpp_u32<0xB1>(v));
    v = umax(v, dpp_u32<0x4E>(v));
    v = umax(v, dpp_u32<0x141>(v));
    v = umax(v, dpp_u32<0x140>(v));
    if constexpr (LC > 16) {
        const auto s = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        v = umax(s[0], s[1]);
    }
    return v;
}
__device__ __forceinline__ uint32_t cw_add(uint32_t v)
{
    v += dpp_u32<0xB1>(v);
    v += dpp_u32<0x4E>(v);
    v += dpp_u32<0x141>(v);
    v += dpp_u32<0x140>(v);
    if constexpr (LC > 16) {
        const auto s = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        v = s[0] + s[1];
    }
    return v;
}
#endif

// diagnostics: cycle stamp k (0 .. 3) of the instruction being profiled (sub: nullptr in the production kernel)
struct Stamps {
    uint32_t* sub;
    uint64_t t0;
    uint32_t ab = 0;  // diagnostics: parts to run twice (PCD_ABLATE bits 16 .. 19)
    __device__ __forceinline__ void mark(int k) const
    {
#ifdef LN_STAMP_SELECT  // (diagnostics: the four stamps go to the parts of select_fork instead -- alt())
        return;
#endif
        if (sub) {
            const uint64_t t = __builtin_amdgcn_s_memtime();
            if (threadIdx.x == 0) sub[k] = (uint32_t)(t - t0);
        }
    }
    __device__ __forceinline__ void alt(int k) const
    {
#ifndef LN_STAMP_SELECT
        return;
#endif
        if (sub) {
            const uint64_t t = __builtin_amdgcn_s_memtime();
            if (threadIdx.x == 0) sub[k] = (uint32_t)(t - t0);
        }
    }
};

// per-lane state the compiler keeps (everything else lives in the hand-allocated registers)
#if LN_GEO_DEFAULT
struct Path {
    float metric;
    uint32_t slot;   // 3-bit fields, stage t at bit 3 * (t - 2), t = 2 .. 10
    uint32_t pl;     // position in the reference's path list; >= P: not a live path
    uint32_t P;      // live paths per codeword (wave uniform)
    uint32_t L;
    int lane, q, g0;
    uint32_t self4;  // lane * 4 (ds_bpermute address of the own lane)
    uint32_t* lds;   // this wave's LDS
    float c127, m128;  // int8 decoder: 127.0 (used as a scalar operand) and -128.0 (kept in a VGPR)
    uint32_t* arena; // this wave's arena slice
    __device__ __forceinline__ uint32_t owner(uint32_t t) const { return (uint32_t)g0 | ((slot >> (3u * (t - 2u))) & 7u); }
    __device__ __forceinline__ void set_field(uint32_t t)
    {
        if (t < 2u) return;  // (stage 1 is never an inner node: nothing is ever looked up for it)
        const uint32_t sh = 3u * (t - 2u);
        slot = (slot & ~(7u << sh)) | ((uint32_t)q << sh);
    }
    __device__ __forceinline__ void reset_slots() { slot = (uint32_t)q * 0x01249249u; }  // every field = own lane
    __device__ __forceinline__ void take_slots(uint32_t src) { slot = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)slot); }
};
#else
// Other geometries: fields of LCL bits; the stages 2 .. 7 in `slot`, the stages 8 .. LOGN in `slot2` (up to 5 x 5 bits each way)
struct Path {
#ifdef LN_STAMP_VT
    uint32_t* vt_sub = nullptr;  // (diagnostics: where the steps of the top-of-tree pass spend their cycles)
#endif
    float metric;
    uint32_t slot, slot2;
    uint32_t pl;     // position in the reference's path list; >= P: not a live path
    uint32_t P;      // live paths per codeword (wave uniform)
    uint32_t L;
    int lane, q, g0;
    uint32_t self4;  // lane * 4 (ds_bpermute address of the own lane)
    uint32_t* lds;   // this wave's LDS
    float c127, m128;  // int8 decoder: 127.0 (used as a scalar operand) and -128.0 (kept in a VGPR)
    uint32_t* arena; // this wave's arena slice
    static constexpr uint32_t FM = (uint32_t)LC - 1u;
    __device__ __forceinline__ uint32_t owner(uint32_t t) const
    {
        return (uint32_t)g0 | (((t >= 8u ? slot2 : slot) >> ((uint32_t)LCL * (t >= 8u ? t - 8u : t - 2u))) & FM);
    }
    __device__ __forceinline__ void set_field(uint32_t t)
    {
        if (t < 2u) return;  // (stage 1 is never an inner node: nothing is ever looked up for it)
        if (t >= 8u) {
            const uint32_t sh = (uint32_t)LCL * (t - 8u);
            slot2 = (slot2 & ~(FM << sh)) | ((uint32_t)q << sh);
        } else {
            const uint32_t sh = (uint32_t)LCL * (t - 2u);
            slot = (slot & ~(FM << sh)) | ((uint32_t)q << sh);
        }
    }
    static constexpr uint32_t ones(int n)
    {
        uint32_t o = 0;
        for (int i = 0; i < n; ++i) o |= 1u << (LCL * i);
        return o;
    }
    __device__ __forceinline__ void reset_slots()
    {
        slot = (uint32_t)q * ones(6);
        slot2 = (uint32_t)q * ones(LOGN - 7);
    }
    __device__ __forceinline__ void take_slots(uint32_t src)
    {
        slot = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)slot);
        slot2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)slot2);
    }
};
#endif

// ---- findWeakLlrs (arrayfuncs.h:209-231) for one path --------------------------------------------------------------
// The reference runs a swap-mutating selection: iteration i takes the first minimum among positions >= i and swaps it
// with position i.  Only the elements that start at positions 0 .. cnt-1 can ever be moved (to the position of a
// winner); everything else stays where it is until it is selected.  The first four elements ("heads") are kept with
// their current position, of the rest ("tail") only the four smallest (value, position) can matter.  One form serves
// cnt = 2 (RateOne) and cnt = 4 (SPC): tracking two heads that never move costs nothing.
struct WeakScan {
    static constexpr uint32_t INF = 0xFFFFFFFFu;
    uint32_t hv[4], tv[4], tp[4];
    uint32_t sg[4];  // hard decisions of the block, 32 per word
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hv[e] = tv[e] = tp[e] = INF;
            sg[e] = 0;
        }
    }
    // tail elements arrive with ascending positions: strict '<' keeps the earlier one first
    __device__ __forceinline__ void tail(uint32_t a, uint32_t ap)
    {
        bool placed = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool lt = placed || (a < tv[k]);
            const uint32_t sv = lt ? tv[k] : a, sp = lt ? tp[k] : ap;
            tv[k] = lt ? a : tv[k];
            tp[k] = lt ? ap : tp[k];
            a = sv;
            ap = sp;
            placed = lt;
        }
    }
    // The same result without the swap emulation, valid while the smallest values are pairwise different (then the
    // selection is a plain sort): heads sorted with their indices, merged with the (sorted) tail minima.  Returns false
    // when two of the values that matter are equal -- the caller then runs finish().  T = log2 of the block length.
    __device__ __forceinline__ bool finish_fast(uint32_t (&wv)[4], uint32_t (&wi)[4], uint32_t T) const
    {
        uint32_t a[4] = { hv[0], hv[1], hv[2], hv[3] }, ai[4] = { 0u, 1u, 2u, 3u };
        auto ce = [](uint32_t& x, uint32_t& xi, uint32_t& y, uint32_t& yi) {
            const bool sw = y < x;
            const uint32_t lo = sw ? y : x, li = sw ? yi : xi, hi = sw ? x : y, hj = sw ? xi : yi;
            x = lo; xi = li; y = hi; yi = hj;
        };
        if (T <= 1u) {  // two elements
            ce(a[0], ai[0], a[1], ai[1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wv[i] = a[i];
                wi[i] = ai[i];
            }
            return a[0] != a[1];
        }
        ce(a[0], ai[0], a[1], ai[1]);
        ce(a[2], ai[2], a[3], ai[3]);
        ce(a[0], ai[0], a[2], ai[2]);
        ce(a[1], ai[1], a[3], ai[3]);
        ce(a[1], ai[1], a[2], ai[2]);
        if (T == 2u) {  // four elements: no tail
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wv[i] = a[i];
                wi[i] = ai[i];
            }
            return umin(umin(a[0] ^ a[1], a[1] ^ a[2]), a[2] ^ a[3]) != 0u;
        }
        // the four smallest of {a[0..3]} U {tv[0..3]}: min(a[i], tv[3 - i]) is bitonic; the discarded maxima hold the fifth
        uint32_t fifth = INF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool sw = tv[3 - i] < a[i];
            fifth = umin(fifth, sw ? a[i] : tv[3 - i]);
            wv[i] = sw ? tv[3 - i] : a[i];
            wi[i] = sw ? tp[3 - i] : ai[i];
        }
        ce(wv[0], wi[0], wv[2], wi[2]);
        ce(wv[1], wi[1], wv[3], wi[3]);
        ce(wv[0], wi[0], wv[1], wi[1]);
        ce(wv[2], wi[2], wv[3], wi[3]);
        // (blocks of 8 and more: the five smallest are all real values)
        return umin(umin(wv[0] ^ wv[1], wv[1] ^ wv[2]), umin(wv[2] ^ wv[3], wv[3] ^ fifth)) != 0u;
    }
    // n = block length, cnt = 2 or 4 (wave uniform); wv / wi: values and original indices at positions 0 .. cnt-1 afterwards
    __device__ __forceinline__ void finish(uint32_t n, uint32_t cnt, uint32_t (&wv)[4], uint32_t (&wi)[4])
    {
        const uint32_t lim = (n - 1 < cnt) ? n - 1 : cnt;
        uint32_t hp[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) hp[e] = ((uint32_t)e < n) ? (uint32_t)e : INF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wv[i] = hv[i];
            wi[i] = (uint32_t)i;
            if ((uint32_t)i < lim) {
                const uint32_t tvmin = tv[0], tpmin = tp[0];
                uint32_t bv = hv[0], bp = hp[0], be = 0;
#pragma unroll
                for (int e = 1; e < 4; ++e) {
                    const bool lt = (hv[e] < bv) || (hv[e] == bv && hp[e] < bp);
                    bv = lt ? hv[e] : bv;
                    bp = lt ? hp[e] : bp;
                    be = lt ? (uint32_t)e : be;
                }
                const bool tail_wins = (tvmin < bv) || (tvmin == bv && tpmin < bp);
                const uint32_t wp = tail_wins ? tpmin : bp;
                wv[i] = tail_wins ? tvmin : bv;
                wi[i] = tail_wins ? wp : be;  // tail elements never move: original index = position
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool is_w = !tail_wins && be == (uint32_t)e;
                    const bool at_i = hp[e] == (uint32_t)i;
                    hp[e] = is_w ? INF : (at_i ? wp : hp[e]);
                    hv[e] = is_w ? INF : hv[e];
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    tv[k] = tail_wins ? tv[k + 1] : tv[k];
                    tp[k] = tail_wins ? tp[k + 1] : tp[k];
                }
                tv[3] = tail_wins ? INF : tv[3];
                tp[3] = tail_wins ? INF : tp[3];
            }
        }
        // positions >= lim keep whatever sits there: the head whose current position is i
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if ((uint32_t)i >= lim && (uint32_t)i < n) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (hp[e] == (uint32_t)i) {
                        wv[i] = hv[e];
                        wi[i] = (uint32_t)e;
                    }
            }
        }
    }
};

// ---- candidate selection: simplePartialSortDescending (arrayfuncs.h:161-183) over the P * C candidates of a codeword ----
// Keys: the raw bits of a candidate metric read as unsigned order the candidates best-first (metrics are never
// positive); the low 6 bits are replaced by (lane, candidate), so that min / max move the payload along and no two
// keys are equal.  The order is exact unless two keys that matter agree in the upper 26 bits: those cases (and real
// ties, whose order depends on the reference's swaps) go to the literal emulation below.
// (One wave per SIMD issues one instruction of any kind every four cycles: the code below avoids lane-mask logic --
//  v_cmp + s_and / s_or + v_cndmask -- wherever min / max / add-with-carry do the same.)
__device__ __forceinline__ void cex(uint32_t& a, uint32_t& b)
{
    const uint32_t lo = umin(a, b), hi = umax(a, b);
    a = lo;
    b = hi;
}
template <int W>
__device__ __forceinline__ void bitonic_clean(uint32_t (&k)[8])
{
#pragma unroll
    for (int d = W / 2; d >= 1; d >>= 1) {
#pragma unroll
        for (int i = 0; i < W; ++i)
            if (!(i & d)) cex(k[i], k[i + d]);
    }
}
// own list k[0 .. M) and the partner lane's (same length), both ascending -> the min(2M, 8) smallest of both, ascending
template <int M, int CTRL>
__device__ __forceinline__ void merge_level(uint32_t (&k)[8])
{
    if constexpr (M == 8) {
        uint32_t b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = dpp_u32<CTRL>(k[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) k[i] = umin(k[i], b[7 - i]);  // (the partner's list reversed: the minima are bitonic)
        bitonic_clean<8>(k);
    } else {
#pragma unroll
        for (int i = 0; i < M; ++i) k[2 * M - 1 - i] = dpp_u32<CTRL>(k[i]);
        bitonic_clean<2 * M>(k);
    }
}
// Literal emulation of the selection sort on the exact metrics: every candidate carries its CURRENT position in the
// reference's array (rank * C + c at the start); round i takes the largest value at positions >= i (first position on
// ties), the candidate sitting at position i moves to the winner's old position.  Two reductions over the codeword per
// round (best value, then first position holding it); a lane notes the round in which a candidate of its own wins and
// posts (lane, candidate) under that rank in the scratch.  Returns the entry of this lane's rank.
template <int C>
__device__ __forceinline__ uint32_t select_literal(const Path& p, const float (&cm)[C], uint32_t np)
{
    constexpr uint32_t INF = 0xFFFFFFFFu;
    const bool live = p.pl < p.P;
    const uint32_t cwbase = SCR_SEL + (uint32_t)(p.lane >> LCL) * (uint32_t)LC;  // ranks [codeword][LC]
    const uint32_t dummy = SCR_SEL + 64u + (uint32_t)p.lane;          // where the losers post
    uint32_t pos[C], key[C], rec[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        pos[c] = live ? (uint32_t)p.q * (uint32_t)C + (uint32_t)c : INF;
        key[c] = live ? fbits(cm[c]) : INF;
        rec[c] = dummy;
    }
    const uint32_t rounds = (uint32_t)__builtin_amdgcn_readfirstlane((int)np);  // (np is the same in every lane)
#pragma unroll 1
    for (uint32_t i = 0; i < rounds; ++i) {
        {
            uint32_t bk = key[0];
#pragma unroll
            for (int c = 1; c < C; ++c) bk = umin(bk, key[c]);
            const uint32_t gk = cw_min(bk);
            uint32_t bp = INF;
#pragma unroll
            for (int c = 0; c < C; ++c) bp = umin(bp, key[c] == gk ? pos[c] : INF);
            const uint32_t gp = cw_min(bp);
            const uint32_t here = cwbase + (uint32_t)i;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const bool win = pos[c] == gp;
                const bool at_i = pos[c] == (uint32_t)i;
                rec[c] = win ? here : rec[c];
                key[c] = win ? INF : key[c];
                pos[c] = win ? INF : (at_i ? gp : pos[c]);
            }
        }
    }
    const uint32_t tag = (uint32_t)p.q << 3;
#pragma unroll
    for (int c = 0; c < C; ++c) p.lds[rec[c]] = tag | (uint32_t)c;
    return p.lds[cwbase + (uint32_t)p.q];
}

// position in the reference's candidate array (rank * C + candidate) of the candidate a key's payload (lane << 3 | candidate) names
template <int C>
__device__ __forceinline__ uint32_t array_pos(uint32_t key)
{
    if constexpr (C == 8) return key & PBM;
    else if constexpr (C == 4) return ((key >> 1) & (PBM >> 1 & ~3u)) | (key & 3u);
    else return ((key >> 2) & (PBM >> 2 & ~1u)) | (key & 1u);
}

// The same emulation over the candidates that can win a round, at most two per lane: the one the merge ranked q (lane q
// holds it: `key`, at array position `at`) and one that was left out AT the cut (equal to the last chosen one: `zkey`,
// INF if the lane has none).  Candidates below the cut never win, so this is exact when the keys are the whole metric
// and no lane has two left out at the cut (the caller checks both).
template <int C>
__device__ __forceinline__ uint32_t select_chosen(const Path& p, uint32_t key, uint32_t at, bool seated, uint32_t np, uint32_t zkey)
{
    constexpr uint32_t INF = 0xFFFFFFFFu;
    const uint32_t cwbase = SCR_SEL + (uint32_t)(p.lane >> LCL) * (uint32_t)LC;
    uint32_t rec = SCR_SEL + 64u + (uint32_t)p.lane, zrec = rec;
    uint32_t comb = seated ? ((key & ~PBM) | at) : INF;
    uint32_t zcomb = zkey != INF ? ((zkey & ~PBM) | array_pos<C>(zkey)) : INF;
    const uint32_t rounds = (uint32_t)__builtin_amdgcn_readfirstlane((int)np);  // (np is the same in every lane)
#pragma unroll 1
    for (uint32_t i = 0; i < rounds; ++i) {
        {
            const uint32_t g = cw_min(umin(comb, zcomb));
            const uint32_t gp = g & PBM;
            const bool win = comb == g, zwin = zcomb == g;
            const bool at_i = (comb & PBM) == (uint32_t)i, zat_i = (zcomb & PBM) == (uint32_t)i;
            rec = win ? cwbase + (uint32_t)i : rec;
            zrec = zwin ? cwbase + (uint32_t)i : zrec;
            comb = win ? INF : (at_i ? ((comb & ~PBM) | gp) : comb);
            zcomb = zwin ? INF : (zat_i ? ((zcomb & ~PBM) | gp) : zcomb);
        }
    }
    p.lds[zrec] = zkey & PBM;  // (a loser posts at the lane's spare word -- first, so that a winner of the same lane overwrites nothing)
    p.lds[rec] = key & PBM;
    return p.lds[cwbase + (uint32_t)p.q];
}

// k[q] for q = 0 .. 7
__device__ __forceinline__ uint32_t pick8(const uint32_t (&k)[8], uint32_t q)
{
    const bool b0 = (q & 1u) != 0, b1 = (q & 2u) != 0, b2 = (q & 4u) != 0;
    const uint32_t a0 = b0 ? k[1] : k[0], a1 = b0 ? k[3] : k[2], a2 = b0 ? k[5] : k[4], a3 = b0 ? k[7] : k[6];
    const uint32_t c0 = b1 ? a1 : a0, c1 = b1 ? a3 : a2;
    return b2 ? c1 : c0;
}

#if LN_LCL <= 3
// (2 / 4 lanes per codeword -- list sizes 2 and 3 .. 4 --: the same with one / two merge levels)
// Chooses the np = min(L, P * C) best candidates of every codeword and re-seats the paths BY RANK: lane q of the codeword
// hosts the path at position q of the reference's list afterwards (p.pl == q for the np live lanes; what a path owns is
// found through its slot word, so a path changes lanes at the price of one ds_bpermute), taken from lane `src`
// (absolute), candidate `cand`; p.metric is the candidate's.  With lane == list position the payload (lane, candidate) of
// the keys orders equal values the way their positions in the reference's candidate array do.
// cm[]: this lane's candidate metrics, in an order whose keys ascend except where `fix` says otherwise (fix: SPC's
// {c3, c4} pair and the two repetition candidates are put in order here).
template <int C>
__device__ __forceinline__ void select_fork(Path& p, const float (&cm)[C], uint32_t& src, uint32_t& cand, const Stamps& sm)
{
    constexpr uint32_t INF = 0xFFFFFFFFu;
    sm.alt(0);
    const bool live = p.pl < p.P;
    const uint32_t np = (p.P * (uint32_t)C < p.L) ? p.P * (uint32_t)C : p.L;
    // the candidates' exact metrics go to the scratch: the new host of a path reads the one it was given
    {
        float* scm = reinterpret_cast<float*>(p.lds + SCR_CM + p.lane * 8);
        if constexpr (C == 8) {
            reinterpret_cast<float4*>(scm)[0] = make_float4(cm[0], cm[1], cm[2], cm[3]);
            reinterpret_cast<float4*>(scm)[1] = make_float4(cm[4], cm[5], cm[6], cm[7]);
        } else if constexpr (C == 4) {
            reinterpret_cast<float4*>(scm)[0] = make_float4(cm[0], cm[1], cm[2], cm[3]);
        } else {
            reinterpret_cast<float2*>(scm)[0] = make_float2(cm[0], cm[1]);
        }
    }
    const uint32_t qc = live ? ((uint32_t)p.q << 3) : INF;  // (dead lanes: every key becomes INF)
    uint32_t k[8];
#if LN_LINKEY == 1
    // linear keys (see the 16 / 32-lane form below): the distance to the best metric of the codeword, 25 bits over the span of the
    // candidates that can be chosen.  At N >= 2048 a sixth of the forks otherwise holds two chosen candidates whose metrics agree
    // in the 26 leading bits (a path and its weakest LLR of hundreds flipped) and runs the full tie emulation, 3 - 5 k cycles each.
    const bool full = p.P == p.L;  // (wave uniform)
    uint32_t bb = fbits(cm[0]), bw = fbits(cm[C - 1]);
    if constexpr (C == 2) {
        bb = umin(fbits(cm[0]), fbits(cm[1]));
        bw = umax(fbits(cm[0]), fbits(cm[1]));
    }
    const uint32_t hiu = cw_min(live ? bb : INF), lou = cw_max(live ? (full ? bb : bw) : 0u);
    const float hi = bitsf(hiu);
    const uint32_t er = umax(fbits(hi - bitsf(lou)) >> 23, 24u);
    const float kscale = bitsf((278u - er) << 23);  // 2^(24 - exponent): span * kscale in [2^24, 2^25)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (c < C) {
            const float t = fminf((hi - cm[c < C ? c : 0]) * kscale, 67108860.0f);
            k[c] = ((uint32_t)t << 6) | qc | (uint32_t)c;
        } else k[c] = INF;
    }
#else
#pragma unroll
    for (int c = 0; c < 8; ++c) k[c] = c < C ? ((fbits(cm[c < C ? c : 0]) & ~63u) | qc | (uint32_t)c) : INF;
#endif
    if constexpr (C == 2) cex(k[0], k[1]);
    if constexpr (C == 8) cex(k[3], k[4]);
    uint32_t own[C];
#pragma unroll
    for (int c = 0; c < C; ++c) own[c] = k[c];
    if (sm.ab & 131072u) {  // (diagnostics: the merge network twice)
        merge_level<C, 0xB1>(k);
        if constexpr (LC >= 4) merge_level<(2 * C < 8 ? 2 * C : 8), 0x4E>(k);
        if constexpr (LC >= 8) merge_level<(4 * C < 8 ? 4 * C : 8), 0x141>(k);
        asm volatile("" : "+v"(k[0]), "+v"(k[7]));
#pragma unroll
        for (int c = 0; c < 8; ++c) k[c] = c < C ? own[c] : INF;
        asm volatile("" : "+v"(k[0]), "+v"(k[1]));
    }
    merge_level<C, 0xB1>(k);
    if constexpr (LC >= 4) merge_level<(2 * C < 8 ? 2 * C : 8), 0x4E>(k);
    if constexpr (LC >= 8) merge_level<(4 * C < 8 ? 4 * C : 8), 0x141>(k);
    sm.alt(1);
    // exactness: neighbours among the np chosen differ above the payload bits, and nobody outside them is at or below
    // the last one (nme = number of this lane's candidates at or below it).
    uint32_t dmin = INF;
#pragma unroll
    for (int i = 0; i < 7; ++i) dmin = umin(dmin, ((uint32_t)(i + 1) < np) ? (k[i] ^ k[i + 1]) : INF);
    uint32_t kcut = k[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) kcut = ((uint32_t)i < np) ? k[i] : kcut;
    const uint32_t kc63 = kcut | 63u;  // (own <= kc63 <=> own >> 6 <= kcut >> 6; INF stays above every valid key)
    uint32_t nme = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) nme += (own[c] <= kc63) ? 1u : 0u;
    const bool cut_tie = cw_add(nme) != np;
    uint32_t mykey = pick8(k, (uint32_t)p.q);
    const bool seated = (uint32_t)p.q < np;
    sm.alt(2);
    if (ballot64((dmin < 64u) || cut_tie) != 0) {  // (wave uniform; rare on real-valued input)
        // Equal metrics.  The merge order is the stable one (value, then position in the candidate array); the reference's
        // selection sort leaves it only where a swap carries a candidate past its equals: a chosen candidate that has an
        // equal (among the chosen, or left out at the cut) and sits at a position below its rank -- round `position` moves
        // it away before its own turn.  Nobody like that in the wave, and keys that are the whole metric (low bits
        // clear, as with quantised LLRs): the merge result stands.  Otherwise the literal emulation decides -- over the
        // chosen candidates alone where that is exact (select_chosen), over all of them where not.
        uint32_t low = 0;
#if LN_LINKEY == 1
#pragma unroll
        for (int c = 0; c < C; ++c) {  // (the linear keys are the whole metric when no distance lost a fraction or was clamped)
            const float t = (hi - cm[c]) * kscale;
            low |= fbits(__builtin_amdgcn_fractf(t)) | (t >= 67108860.0f ? 1u : 0u);
        }
        low = low != 0u ? 1u : 0u;
#else
#pragma unroll
        for (int c = 0; c < C; ++c) low |= fbits(cm[c]);
#endif
        const uint32_t q = (uint32_t)p.q;
        const uint32_t prev = dpp_u32<0x111>(mykey), next = dpp_u32<0x101>(mykey);  // ranks q - 1 and q + 1 (row_shr / row_shl)
        const bool equal = (q > 0u && (mykey ^ prev) < 64u) || (q + 1u < np && (mykey ^ next) < 64u) || (cut_tie && (mykey | 63u) == kc63);
        const uint32_t at = array_pos<C>(mykey);
        const bool moved = seated && equal && at < q;
        bool all = live && (low & 63u) != 0u;  // (keys that are not the whole metric: the emulation over all candidates)
        if (ballot64(moved || all) != 0) {
            uint32_t zkey = INF;
            if (ballot64(cut_tie) != 0) {  // candidates left out at the cut
                uint32_t nz = 0;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const bool z = (own[c] | 63u) == kc63 && own[c] > kcut;
                    zkey = umin(zkey, z ? own[c] : INF);
                    nz += z ? 1u : 0u;
                }
                all = all || nz > 1u;
            }
            // (diagnostics, results wrong: PCD_ABLATE=262144 neither emulation, 524288 the cheap one in place of the full one)
            if (sm.ab & 262144u) {
            } else if (ballot64(all) != 0 && !(sm.ab & 524288u)) mykey = select_literal<C>(p, cm, np);
            else mykey = select_chosen<C>(p, mykey, at, seated, np, zkey);
        }
    }
    sm.alt(3);
    sm.mark(2);
    src = seated ? ((uint32_t)p.g0 | ((mykey >> 3) & 7u)) : (uint32_t)p.lane;
    cand = mykey & 7u;
    // the path's state: metric from the scratch, slot word from the source lane
    const float nm = reinterpret_cast<const float*>(p.lds + SCR_CM)[src * 8 + cand];
#if LN_GEO_DEFAULT
    const uint32_t nslot = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)p.slot);
    p.metric = seated ? nm : p.metric;
    p.slot = nslot;
#else
    p.take_slots(src);  // (both slot words)
    p.metric = seated ? nm : p.metric;
#endif
    p.pl = seated ? (uint32_t)p.q : 15u;
    p.P = np;
}
#else
// ---- the same for 16 / 32 lanes per codeword (list sizes up to 16 / 32) ---------------------------------------------------
// The np <= LC best candidates no longer fit one lane's eight registers: W = LC / 8 neighbouring lanes hold the sorted list
// together, lane j of such a group the ranks 8 j .. 8 j + 7.  (A) the lanes of a group sort their 8 W candidates among
// themselves (bitonic merges: partner's list reversed, the lower lane keeps the minima, the upper one the maxima); (B) groups
// merge pairwise, both keeping the LC smallest -- min(X[m], Y[LC - 1 - m]) is a bitonic sequence that holds exactly those --,
// then clean it up among their lanes; after log2(LC / W) such levels every group of the codeword holds the same sorted list.
// Partners: a group's mirror image inside 4 / 8 / 16 lanes (quad_perm [3,2,1,0], row_half_mirror, row_mirror) is "the other
// group, lanes reversed", which is what the reversed list needs; across the two rows of a 32-lane codeword ds_swizzle
// (lane ^ 0x13).  The sorted list then goes to the LDS (ranks [codeword][LC]), lane q reads entry q: seats by rank as above.
template <int CTRL>
__device__ __forceinline__ void cross_rev_minmax(uint32_t (&k)[8], bool low)
{
    uint32_t b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = dpp_u32<CTRL>(k[7 - i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = low ? umin(k[i], b[i]) : umax(k[i], b[i]);
}
template <int CTRL>
__device__ __forceinline__ void cross_minmax(uint32_t (&k)[8], bool low)
{
    uint32_t b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = dpp_u32<CTRL>(k[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = low ? umin(k[i], b[i]) : umax(k[i], b[i]);
}
// both groups keep the LC smallest of their two sorted lists (as a bitonic sequence)
template <int CTRL>
__device__ __forceinline__ void cross_rev_min(uint32_t (&k)[8])
{
    uint32_t b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if constexpr (CTRL == 0) b[i] = (uint32_t)__builtin_amdgcn_ds_swizzle((int)k[7 - i], 0x4C1F);  // lane ^ 0x13 (bit mode: and 0x1F, xor 0x13)
        else b[i] = dpp_u32<CTRL>(k[7 - i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = umin(k[i], b[i]);
}
// a bitonic sequence of 8 W keys over the W lanes of a group -> ascending
__device__ __forceinline__ void group_clean(uint32_t (&k)[8], int lane)
{
    if constexpr (LC == 32) cross_minmax<0x4E>(k, (lane & 2) == 0);
    cross_minmax<0xB1>(k, (lane & 1) == 0);
    bitonic_clean<8>(k);
}
template <int C>
__device__ __forceinline__ void select_fork(Path& p, const float (&cm)[C], uint32_t& src, uint32_t& cand, const Stamps& sm)
{
    constexpr uint32_t INF = 0xFFFFFFFFu;
    sm.alt(0);
    const bool live = p.pl < p.P;
    const uint32_t np = (p.P * (uint32_t)C < p.L) ? p.P * (uint32_t)C : p.L;
    // the candidates' exact metrics go to the scratch: the new host of a path reads the one it was given
    {
        float* scm = reinterpret_cast<float*>(p.lds + SCR_CM + p.lane * 8);
        if constexpr (C == 8) {
            reinterpret_cast<float4*>(scm)[0] = make_float4(cm[0], cm[1], cm[2], cm[3]);
            reinterpret_cast<float4*>(scm)[1] = make_float4(cm[4], cm[5], cm[6], cm[7]);
        } else if constexpr (C == 4) {
            reinterpret_cast<float4*>(scm)[0] = make_float4(cm[0], cm[1], cm[2], cm[3]);
        } else {
            reinterpret_cast<float2*>(scm)[0] = make_float2(cm[0], cm[1]);
        }
    }
    const uint32_t qc = live ? ((uint32_t)p.q << 3) : INF;  // (dead lanes: every key becomes INF)
    uint32_t k[8];
#if LN_LINKEY == 1
    // Keys.  The 24 - LCL leading bits of the metric resolve |metric| * 2^-15: at 16 / 32 paths most forks then hold two chosen
    // candidates with the same key -- a path's own metric and the one with its weakest LLR flipped, whose magnitude in a block
    // of hundreds of LLRs is tiny -- and take the exact-metric fix-up below (88 % of the forks of cfg 5, a tenth of them the tie
    // emulations).  Any monotone map of the metric is a valid key, so the keys are linear instead: the distance to the best metric
    // of the codeword, scaled so that the span of the candidates that can be chosen fills 23 bits (span * 2^-23: two to three
    // decimal orders finer).  That span: with a full list (P = L) the paths' best candidates -- they are L candidates, so the
    // L-th best of all is among or above them --, before that all candidates.  Metrics are <= 0 and never -0 (sums that start at
    // +0), larger is better, so float bits ascend with the distance as the distance itself does; fl(hi - x) is monotone and
    // injective in x.  No branch on the list state: a taken branch costs a lone wave its instruction buffer.
    const bool full = p.P == p.L;  // (wave uniform)
    uint32_t bb = fbits(cm[0]), bw = fbits(cm[C - 1]);  // best / worst candidate of the lane (C = 2: the repetition pair is not ordered)
    if constexpr (C == 2) {
        bb = umin(fbits(cm[0]), fbits(cm[1]));
        bw = umax(fbits(cm[0]), fbits(cm[1]));
    }
#if LN_LINKEY_NOHI
    // (variant: distances from 0 instead of from the best metric -- one reduction less, at most one bit of resolution lost)
    const uint32_t lou = cw_max(live ? (full ? bb : bw) : 0u);
    const float hi = 0.0f;
    const uint32_t er = umax((lou & 0x7FFFFFFFu) >> 23, 22u);
    const float kscale = bitsf((276u - er) << 23);
#else
    const uint32_t hiu = cw_min(live ? bb : INF), lou = cw_max(live ? (full ? bb : bw) : 0u);
    const float hi = bitsf(hiu);
    const uint32_t er = umax(fbits(hi - bitsf(lou)) >> 23, 22u);  // biased exponent of the span (clamped: a span of 0 still gives a finite scale)
    const float kscale = bitsf((276u - er) << 23);                 // 2^(22 - exponent): span * kscale in [2^22, 2^23)
#endif
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (c < C) {
            const float t = fminf((hi - cm[c < C ? c : 0]) * kscale, 16777214.0f);  // (candidates far outside the span share the last key; INF stays above it)
            k[c] = ((uint32_t)t << PB) | qc | (uint32_t)c;
        } else k[c] = INF;
    }
#else
#pragma unroll
    for (int c = 0; c < 8; ++c) k[c] = c < C ? ((fbits(cm[c < C ? c : 0]) & ~PBM) | qc | (uint32_t)c) : INF;
#endif
    if constexpr (C == 2) cex(k[0], k[1]);
    if constexpr (C == 8) cex(k[3], k[4]);
    uint32_t own[C];
#pragma unroll
    for (int c = 0; c < C; ++c) own[c] = k[c];
    // (A) the W lanes of a group sort their candidates
    cross_rev_minmax<0xB1>(k, (p.lane & 1) == 0);
    bitonic_clean<8>(k);
    if constexpr (LC == 32) {
        cross_rev_minmax<0x1B>(k, (p.lane & 2) == 0);
        cross_minmax<0xB1>(k, (p.lane & 1) == 0);
        bitonic_clean<8>(k);
    }
    // (B) groups merge pairwise
    if constexpr (LC == 16) {
        cross_rev_min<0x1B>(k);
        group_clean(k, p.lane);
        cross_rev_min<0x141>(k);
        group_clean(k, p.lane);
        cross_rev_min<0x140>(k);
        group_clean(k, p.lane);
    } else {
        cross_rev_min<0x141>(k);
        group_clean(k, p.lane);
        cross_rev_min<0x140>(k);
        group_clean(k, p.lane);
        cross_rev_min<0>(k);
        group_clean(k, p.lane);
    }
    sm.alt(1);
    // the sorted list of the codeword: ranks [codeword][LC] in the scratch, written by the first group
    const uint32_t cwbase = SCR_SEL + (uint32_t)(p.lane >> LCL) * (uint32_t)LC;
    if ((uint32_t)p.q < (uint32_t)(LC / 8)) {
        uint4* t = reinterpret_cast<uint4*>(p.lds + cwbase + (uint32_t)p.q * 8u);
        t[0] = make_uint4(k[0], k[1], k[2], k[3]);
        t[1] = make_uint4(k[4], k[5], k[6], k[7]);
    }
    const uint32_t q = (uint32_t)p.q;
    uint32_t mykey = p.lds[cwbase + q];
    const uint32_t kcut = p.lds[cwbase + np - 1u];
    const uint32_t next = p.lds[cwbase + (q + 1u < (uint32_t)LC ? q + 1u : q)], prev = p.lds[cwbase + (q > 0u ? q - 1u : 0u)];
    // exactness: neighbours among the np chosen differ above the payload bits, and nobody outside them is at or below
    // the last one (nme = number of this lane's candidates at or below it).
    const bool near = (q + 1u < np) && ((mykey ^ next) <= PBM);
    const uint32_t kcm = kcut | PBM;  // (own <= kcm <=> own >> PB <= kcut >> PB; INF stays above every valid key)
    uint32_t nme = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) nme += (own[c] <= kcm) ? 1u : 0u;
    const bool cut_tie = cw_add(nme) != np;
    const bool seated = q < np;
    sm.alt(2);
    bool resolved = false;
    if (ballot64(near) != 0 && ballot64(cut_tie) == 0) {
        // Keys that agree above the payload bits while the cut is clean.  With 16 / 32 paths this happens in a good part of the
        // forks on real-valued input (a key keeps 24 - LCL bits of the metric), so it gets a cheap exact answer first: the
        // chosen SET is right, and a run of equal keys only has to be put in the order of the exact metrics -- odd-even
        // transposition between neighbouring ranks on (exact metric bits, key), until two rounds in a row change nothing.
        // If two neighbours then still carry the same exact metric, the order depends on the reference's swaps: the tiers below.
        const uint32_t* scm = p.lds + SCR_CM;
        uint32_t fk = mykey;
        uint32_t tm = seated ? scm[((uint32_t)p.g0 | ((fk >> 3) & (uint32_t)(LC - 1))) * 8u + (fk & 7u)] : INF;
        uint32_t quiet = 0;
#pragma unroll 1
        for (uint32_t round = 0; round < 2u * (uint32_t)LC && quiet < 2u; ++round) {
            const uint32_t pr = (round & 1u) ? (((q + 1u) ^ 1u) - 1u) : (q ^ 1u);  // partner rank (odd rounds: (1,2) (3,4) ...; rank 0 and the last one rest)
            const bool valid = pr < (uint32_t)LC;
            const uint32_t pa = ((uint32_t)p.g0 | (valid ? pr : q)) << 2;
            const uint32_t pk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)pa, (int)fk);
            const uint32_t pt = (uint32_t)__builtin_amdgcn_ds_bpermute((int)pa, (int)tm);
            const bool run = valid && ((fk ^ pk) <= PBM);  // (both seated: an unseated rank holds a key above every run of chosen ones or INF)
            const bool take = run && (q < pr ? (pt < tm) : (tm < pt));
            fk = take ? pk : fk;
            tm = take ? pt : tm;
            quiet = ballot64(take) != 0 ? 0u : quiet + 1u;
        }
        const uint32_t nt = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)p.g0 | (q + 1u < (uint32_t)LC ? q + 1u : q)) << 2), (int)tm);
        const bool exact_tie = (q + 1u < np) && nt == tm;
        if (ballot64(exact_tie) == 0) {
            mykey = fk;
            resolved = true;
        }
    }
    if (!resolved && ballot64(near || cut_tie) != 0) {  // (wave uniform)
        // Equal metrics: see the 8-lane form above.
        uint32_t low = 0;
#if LN_LINKEY
        // (the linear keys are the whole metric when no distance lost a fraction or was clamped)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float t = (hi - cm[c]) * kscale;
            low |= fbits(__builtin_amdgcn_fractf(t)) | (t >= 16777214.0f ? 1u : 0u);
        }
        low = low != 0u ? 1u : 0u;
#else
#pragma unroll
        for (int c = 0; c < C; ++c) low |= fbits(cm[c]);
#endif
        const bool equal = (q > 0u && (mykey ^ prev) <= PBM) || (q + 1u < np && (mykey ^ next) <= PBM) || (cut_tie && (mykey | PBM) == kcm);
        const uint32_t at = array_pos<C>(mykey);
        const bool moved = seated && equal && at < q;
        bool all = live && (low & PBM) != 0u;  // (keys that are not the whole metric: the emulation over all candidates)
        if (ballot64(moved || all) != 0) {
            uint32_t zkey = INF;
            if (ballot64(cut_tie) != 0) {  // candidates left out at the cut
                uint32_t nz = 0;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const bool z = (own[c] | PBM) == kcm && own[c] > kcut;
                    zkey = umin(zkey, z ? own[c] : INF);
                    nz += z ? 1u : 0u;
                }
                all = all || nz > 1u;
            }
            if (ballot64(all) != 0) mykey = select_literal<C>(p, cm, np);
            else mykey = select_chosen<C>(p, mykey, at, seated, np, zkey);
        }
    }
    sm.alt(3);
    sm.mark(2);
    src = seated ? ((uint32_t)p.g0 | ((mykey >> 3) & (uint32_t)(LC - 1))) : (uint32_t)p.lane;
    cand = mykey & 7u;
    // the path's state: metric from the scratch, slot words from the source lane
    const float nm = reinterpret_cast<const float*>(p.lds + SCR_CM)[src * 8 + cand];
    p.take_slots(src);
    p.metric = seated ? nm : p.metric;
    p.pl = seated ? (uint32_t)p.q : PL_DEAD;
    p.P = np;
}
#endif

// ---- leaves ---------------------------------------------------------------------------------------------------------------
// sum of v over the block in the reference's order: 8 strided accumulators (acc[j] += x[8k + j], k ascending), then
// ((((((a0 + a1) + a2) + a3) + a4) + a5) + a6) + a7   (avxconvenience.h:256-272; blocks of fewer than 8: the missing
// lanes hold +0, which never changes a sum whose terms are not -0)
template <int N>
__device__ __forceinline__ float lanesum_regs(const float (&z)[N])
{
    if constexpr (N <= 8) {
        float s = z[0];
#pragma unroll
        for (int i = 1; i < N; ++i) s = s + z[i];
        return s;
    } else {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = z[j];
#pragma unroll
        for (int i = 8; i < N; ++i) acc[i & 7] = acc[i & 7] + z[i];
        return ((((((acc[0] + acc[1]) + acc[2]) + acc[3]) + acc[4]) + acc[5]) + acc[6]) + acc[7];
    }
}

// the output bits of a leaf: stage T, side 0 = left child (or root), 1 = right child; NW words
template <int T>
__device__ __forceinline__ void put_bits(const Path& p, uint32_t side, const uint32_t (&w)[T <= 5 ? 1 : (1 << (T - 5))])
{
    constexpr int NW = T <= 5 ? 1 : (1 << (T - 5));
    if constexpr (T <= 7) {
        if (side) {
            static_for<NW>([&](auto I) { wru<RB(1, T) + decltype(I)::value>(w[decltype(I)::value]); });
        } else {
            static_for<NW>([&](auto I) { wru<RB(0, T) + decltype(I)::value>(w[decltype(I)::value]); });
        }
    } else {
        uint32_t* o = p.arena + (side ? AR_BR8 : AR_BL8) + p.lane;
#pragma unroll
        for (int i = 0; i < NW; ++i) o[i * 64] = w[i];
    }
}

// Input of a leaf at stage T as 16-byte chunks (T >= 5: rolled loops): the registers of stages 5 / 6 are dumped to the
// arena first, stage 7 is read from the LDS column, stage 8 from the arena column.
template <int T>
struct ChunkSrc {
    const float4* base;  // chunk c at base[c * 64]
    __device__ __forceinline__ ChunkSrc(const Path& p)
    {
        if constexpr (T == 7) base = reinterpret_cast<const float4*>(p.lds + LDS_S7) + p.lane;
        else if constexpr (T == 8) base = reinterpret_cast<const float4*>(p.arena + AR_S8) + p.lane;
        else base = reinterpret_cast<const float4*>(p.arena + AR_DUMP) + p.lane;
    }
    __device__ __forceinline__ float4 chunk(uint32_t c) const { return base[c * 64]; }
};
template <int T>
__device__ __forceinline__ void stage_to_dump(const Path& p)
{
    if constexpr (T == 5 || T == 6) {
        dump_regs<RS(T), (1 << T) / 4>(p.arena + AR_DUMP + p.lane * 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// RateZeroDecoder::decode scl_avx_float.cpp:316-337: metric += sum of min(llr, 0); bits = 0
// (v_min_f32 may return -0 for min(-0, +0) where the reference's _mm256_min_ps returns +0: the partial sums start from a
//  value that is not -0 in every case that reaches the metric -- a sum of such terms added to a metric that is never -0 --
//  so the results are the same bit for bit)
template <int R>
__device__ __forceinline__ float min0_reg()
{
    float z;
    asm volatile("v_min_f32 %0, 0, v[%1]" : "=v"(z) : "n"(R));
    return z;
}
template <int T>
__device__ __forceinline__ void leaf_r0(Path& p, uint32_t side)
{
    constexpr int n = 1 << T;
    float sum;
    if constexpr (T <= 3) {  // (one asm statement: chained single-instruction statements cost an s_nop each)
        float t0, t1;
        if constexpr (T == 3) asm volatile(LN_RUN_MIN0SUM_8 : [s] "=&v"(sum), [t0] "=&v"(t0), [t1] "=&v"(t1) : [r] "n"(RS(T)));
        else if constexpr (T == 2) asm volatile(LN_RUN_MIN0SUM_4 : [s] "=&v"(sum), [t0] "=&v"(t0), [t1] "=&v"(t1) : [r] "n"(RS(T)));
        else asm volatile(LN_RUN_MIN0SUM_2 : [s] "=&v"(sum), [t0] "=&v"(t0), [t1] "=&v"(t1) : [r] "n"(RS(T)));
    } else if constexpr (T <= 6) {
        float acc[8], t0, t1, t2, t3;
#define LN_ACC_OPS [a0] "=&v"(acc[0]), [a1] "=&v"(acc[1]), [a2] "=&v"(acc[2]), [a3] "=&v"(acc[3]), [a4] "=&v"(acc[4]), [a5] "=&v"(acc[5]), \
                   [a6] "=&v"(acc[6]), [a7] "=&v"(acc[7]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
        if constexpr (T == 6) asm volatile(LN_RUN_MIN0ACC_64 : LN_ACC_OPS : [r] "n"(RS(T)));
        else if constexpr (T == 5) asm volatile(LN_RUN_MIN0ACC_32 : LN_ACC_OPS : [r] "n"(RS(T)));
        else asm volatile(LN_RUN_MIN0ACC_16 : LN_ACC_OPS : [r] "n"(RS(T)));
#undef LN_ACC_OPS
        sum = ((((((acc[0] + acc[1]) + acc[2]) + acc[3]) + acc[4]) + acc[5]) + acc[6]) + acc[7];
    } else {
        const ChunkSrc<T> x(p);
        float acc[8];
        {
            const float4 a = x.chunk(0), b = x.chunk(1);
            acc[0] = min0(a.x); acc[1] = min0(a.y); acc[2] = min0(a.z); acc[3] = min0(a.w);
            acc[4] = min0(b.x); acc[5] = min0(b.y); acc[6] = min0(b.z); acc[7] = min0(b.w);
        }
#pragma unroll 2
        for (uint32_t c = 2; c < (uint32_t)n / 4; c += 2) {
            const float4 a = x.chunk(c), b = x.chunk(c + 1);
            acc[0] += min0(a.x); acc[1] += min0(a.y); acc[2] += min0(a.z); acc[3] += min0(a.w);
            acc[4] += min0(b.x); acc[5] += min0(b.y); acc[6] += min0(b.z); acc[7] += min0(b.w);
        }
        sum = ((((((acc[0] + acc[1]) + acc[2]) + acc[3]) + acc[4]) + acc[5]) + acc[6]) + acc[7];
    }
    p.metric = p.metric + sum;
    uint32_t w[T <= 5 ? 1 : (1 << (T - 5))];
#pragma unroll
    for (int i = 0; i < (T <= 5 ? 1 : (1 << (T - 5))); ++i) w[i] = 0;
    put_bits<T>(p, side, w);
}

// RepetitionDecoder::decode scl_avx_float.cpp:428-481 (blocks of 2 and 4)
template <int T>
__device__ __forceinline__ void leaf_rep(Path& p, uint32_t side, const Stamps& sm)
{
    constexpr int n = 1 << T;
    float z[n], o[n];
    static_for<n>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const float x = rdf<RS(T) + i>();
        z[i] = min0(x);
        o[i] = max0(x);
    });
    float cm[2];
    cm[0] = p.metric + lanesum_regs<n>(z);
    cm[1] = p.metric - lanesum_regs<n>(o);
    sm.mark(1);
    uint32_t src, cand;
    select_fork<2>(p, cm, src, cand, sm);
    sm.mark(3);
    const uint32_t w[1] = { cand ? ((1u << n) - 1u) : 0u };
    put_bits<T>(p, side, w);
}

// RepetitionDecoder of the int8 decoder (scl_fip_char.cpp:497-547): at EVERY block length, exact sums (int8-valued floats: any order
// gives the same result) -- candidate 0 = all bits 0 at metric + sum of min(x, 0), candidate 1 = all bits 1 at metric - sum of max(x, 0)
template <int T>
__device__ __forceinline__ void leaf_rep_any(Path& p, uint32_t side, const Stamps& sm)
{
    constexpr int n = 1 << T;
    float zs = 0.0f, xs = 0.0f;  // sum of min(x, 0), sum of x (sum of max(x, 0) = xs - zs)
    if constexpr (T <= 6) {
        float z4[4] = { 0.0f, 0.0f, 0.0f, 0.0f }, x4[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        static_for<n>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const float x = rdf<RS(T) + i>();
            z4[i & 3] += min0(x);
            x4[i & 3] += x;
        });
        zs = (z4[0] + z4[1]) + (z4[2] + z4[3]);
        xs = (x4[0] + x4[1]) + (x4[2] + x4[3]);
    } else {
        const ChunkSrc<T> x(p);
#pragma unroll 2
        for (uint32_t c = 0; c < (uint32_t)n / 4; ++c) {
            const float4 a = x.chunk(c);
            zs += (min0(a.x) + min0(a.y)) + (min0(a.z) + min0(a.w));
            xs += (a.x + a.y) + (a.z + a.w);
        }
    }
    float cm[2];
    cm[0] = p.metric + zs;
    cm[1] = p.metric - (xs - zs);
    sm.mark(1);
    uint32_t src, cand;
    select_fork<2>(p, cm, src, cand, sm);
    sm.mark(3);
    uint32_t w[T <= 5 ? 1 : (1 << (T - 5))];
#pragma unroll
    for (int i = 0; i < (T <= 5 ? 1 : (1 << (T - 5))); ++i) w[i] = cand ? (n >= 32 ? 0xFFFFFFFFu : ((1u << (n & 31)) - 1u)) : 0u;
    put_bits<T>(p, side, w);
}

// RateOneDecoder::decode :353-413 and SpcDecoder::decode :498-621, first half: hard decisions and the weak-LLR scan of
// the block (what follows -- findWeakLlrs' iterations, candidates, selection, bits -- is shared: fork_tail)
// CH (int8 decoder): magnitudes are abs(max(x, -127)) (scl_fip_char.cpp:413-417, 587-590)
#ifndef LN_SCAN_REGS
#define LN_SCAN_REGS 5
#endif
template <int T, bool CH = false>
__device__ __forceinline__ void leaf_scan(const Path& p, WeakScan& ws)
{
    auto mag = [](uint32_t b) {
        const uint32_t a = b & 0x7FFFFFFFu;
        if constexpr (CH) return a < 0x42FE0000u ? a : 0x42FE0000u;  // 127.0f
        else return a;
    };
    constexpr int n = 1 << T;
    ws.init();
    if constexpr (T <= LN_SCAN_REGS) {  // straight from the registers (stage 6: through a dump in the arena -- 64 unrolled insertions are 12 KB of code that runs once per pass)
        static_for<(n + 31) / 32>([&](auto Wd) {
            constexpr int wd = decltype(Wd)::value;
            uint32_t s = 0;
            static_for<(n < 32 ? n : 32)>([&](auto I) {
                constexpr int i = 32 * wd + decltype(I)::value;
                const uint32_t b = rdu<RS(T) + i>();
                s |= (b >> 31) << (i & 31);
                if constexpr (i < 4) ws.hv[i] = mag(b);
                else ws.tail(mag(b), (uint32_t)i);
            });
            ws.sg[wd] = s;
        });
    } else {
        stage_to_dump<T>(p);
        const ChunkSrc<T> x(p);
        {
            const float4 a = x.chunk(0);
            const uint32_t e[4] = { fbits(a.x), fbits(a.y), fbits(a.z), fbits(a.w) };
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ws.sg[0] |= (e[i] >> 31) << i;
                ws.hv[i] = mag(e[i]);
            }
        }
        static_for<(n + 31) / 32>([&](auto Wd) {
            constexpr int wd = decltype(Wd)::value;
#if LN_BIG
            uint32_t s = wd < 4 ? ws.sg[wd < 4 ? wd : 0] : 0u;  // (a block of 256 LLRs: the sign words 4 .. 7 go to the LDS)
#else
            uint32_t s = ws.sg[wd];
#endif
#pragma unroll 1
            for (uint32_t c = (wd == 0 ? 1u : 0u); c < 8; ++c) {
                const uint32_t ch = (uint32_t)wd * 8u + c;
                const float4 a = x.chunk(ch);
                const uint32_t e[4] = { fbits(a.x), fbits(a.y), fbits(a.z), fbits(a.w) };
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s |= (e[i] >> 31) << (4u * c + (uint32_t)i);
                    ws.tail(mag(e[i]), 4u * ch + (uint32_t)i);
                }
            }
#if LN_BIG
            if constexpr (wd < 4) ws.sg[wd] = s;
            else p.lds[SCR_SEL + 128 + p.lane * 4 + (wd - 4)] = s;  // ([lane][4] behind the selection scratch)
#else
            ws.sg[wd] = s;
#endif
        });
    }
}

// Second half of the forking leaves.  SPC: SpcDecoder (8 candidates), else RateOne (4); T = log2 of the block length;
// ws: the scan of this lane's path.
template <bool SPC>
__device__ __forceinline__ void fork_tail(Path& p, uint32_t T, uint32_t side, WeakScan& ws, const Stamps& sm)
{
    const uint32_t n = 1u << T;
    uint32_t wv[4], wi[4];
    if (sm.ab & 65536u) {  // (diagnostics: the weak-LLR order twice)
        bool r = ws.finish_fast(wv, wi, T);
        asm volatile("" : "+v"(wv[0]), "+v"(wv[1]), "+v"(wi[0]), "+v"(ws.hv[0]), "+v"(ws.tv[0]), "+v"(ws.tp[0]));
        (void)r;
    }
    if (ballot64(!ws.finish_fast(wv, wi, T)) != 0) ws.finish(n, SPC ? 4u : 2u, wv, wi);  // (equal magnitudes somewhere in the wave)
    const float m = p.metric;
    constexpr int C = SPC ? 8 : 4;
    float cm[C];
    uint32_t par = 0;
    if constexpr (SPC) {
        par = (uint32_t)__builtin_popcount(ws.sg[0] ^ ws.sg[1] ^ ws.sg[2] ^ ws.sg[3]) & 1u;
#if LN_BIG
        if (T == 8u) {  // (a block of 256 LLRs: the upper four sign words sit in the LDS)
            const uint4 sx = reinterpret_cast<const uint4*>(p.lds + SCR_SEL + 128)[p.lane];
            par ^= (uint32_t)__builtin_popcount(sx.x ^ sx.y ^ sx.z ^ sx.w) & 1u;
        }
#endif
        const float v0 = bitsf(wv[0]), v1 = bitsf(wv[1]), v2 = bitsf(wv[2]), v3 = bitsf(wv[3]);
        const float t = m - v0;        // m' - pinv * v0 (par: m' = m - v0, pinv = 0; else m' = m, pinv = 1)
        const float mp = par ? t : m;  // m'
        cm[0] = mp;
        cm[1] = t - v1;
        cm[2] = t - v2;
        cm[3] = t - v3;
        cm[4] = (mp - v1) - v2;
        cm[5] = (mp - v1) - v3;
        cm[6] = (mp - v2) - v3;
        cm[7] = ((t - v1) - v2) - v3;
    } else {
        const float v0 = bitsf(wv[0]), v1 = bitsf(wv[1]);
        cm[0] = m;
        cm[1] = m - v0;
        cm[2] = m - v1;
        cm[3] = (m - v0) - v1;
    }
    // fork record of this lane's path: the new hosts of its children read it
    uint4* rec = reinterpret_cast<uint4*>(p.lds + SCR_FORK + p.lane * 8);
    const uint32_t packed = (wi[0] & 0xFFu) | ((wi[1] & 0xFFu) << 8) | ((wi[2] & 0xFFu) << 16) | ((wi[3] & 0xFFu) << 24);
#if LN_LOGN >= 11 && LN_HILEAF
    if (T >= 9u)  // (a block of 512 / 1024 LLRs: 16-bit weak indices; the sign words are in the arena already, leaf_scan_high)
        rec[0] = make_uint4(0u, (wi[0] & 0xFFFFu) | (wi[1] << 16), par, (wi[2] & 0xFFFFu) | (wi[3] << 16));
    else
#endif
    rec[0] = make_uint4(ws.sg[0], packed, par, ws.sg[1]);
#if LN_BIG
    if (T >= 7u) rec[1] = make_uint4(ws.sg[2], ws.sg[3], 0u, 0u);
#else
    if (T == 7u) rec[1] = make_uint4(ws.sg[2], ws.sg[3], 0u, 0u);
#endif
    sm.mark(1);
    uint32_t src, cand;
    select_fork<C>(p, cm, src, cand, sm);
    sm.mark(3);
    const uint4* sr = reinterpret_cast<const uint4*>(p.lds + SCR_FORK + src * 8);
    const uint4 s0 = sr[0];
    const uint32_t spk = s0.y, spar = s0.z;
    // which of the weak positions the candidate flips (RateOne: 1 -> ix0, 2 -> ix1, 3 -> both; SPC :541-613)
    uint32_t fl[4];
    if constexpr (SPC) {
        fl[0] = ((spar ? 0x71u : 0x8Eu) >> cand) & 1u;
        fl[1] = (0xB2u >> cand) & 1u;
        fl[2] = (0xD4u >> cand) & 1u;
        fl[3] = (0xE8u >> cand) & 1u;
    } else {
        fl[0] = cand & 1u;
        fl[1] = (cand >> 1) & 1u;
        fl[2] = fl[3] = 0;
    }
    constexpr int NF = SPC ? 4 : 2;
#if LN_LOGN >= 11 && LN_HILEAF
    if (T >= 9u) {
        // the source path's sign words (its column of this leaf's output area, where its scan left them) with the candidate's flips,
        // eight words at a time: every lane reads a batch before any lane writes it, so the rewrite in place is safe
        const uint32_t ix[4] = { s0.y & 0xFFFFu, s0.y >> 16, s0.w & 0xFFFFu, s0.w >> 16 };
        uint32_t ln = (uint32_t)p.lane;
        asm volatile("" : "+v"(ln));
        const uint32_t area = T == 9u ? (side ? AR_BR(9) : AR_BL(9)) : (side ? AR_BR(10) : AR_BL(10));
        const uint32_t* sp = p.arena + area + src;
        uint32_t* dp = p.arena + area + ln;
        const uint32_t nw = 1u << (T - 5u);
#pragma unroll 1
        for (uint32_t i0 = 0; i0 < nw; i0 += 8) {
            uint32_t wd[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) wd[i] = sp[(i0 + (uint32_t)i) * 64u];
#pragma unroll
            for (int k = 0; k < NF; ++k) {
#pragma unroll
                for (int i = 0; i < 8; ++i) wd[i] ^= ((ix[k] >> 5) == i0 + (uint32_t)i) ? (fl[k] << (ix[k] & 31u)) : 0u;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (all loads of the batch have landed in every lane)
#pragma unroll
            for (int i = 0; i < 8; ++i) dp[(i0 + (uint32_t)i) * 64u] = wd[i];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
#endif
    uint32_t w[4] = { s0.x, s0.w, 0u, 0u };
#if LN_BIG
    if (T == 8u) {  // a block of 256 LLRs: eight words, to the arena like the other bits of that stage
        const uint4 s1 = sr[1];
        const uint4 s2 = reinterpret_cast<const uint4*>(p.lds + SCR_SEL + 128)[src];
        uint32_t w8[8] = { s0.x, s0.w, s1.x, s1.y, s2.x, s2.y, s2.z, s2.w };
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const uint32_t ix = (spk >> (8 * k)) & 0xFFu;
#pragma unroll
            for (int i = 0; i < 8; ++i) w8[i] ^= ((ix >> 5) == (uint32_t)i) ? (fl[k] << (ix & 31u)) : 0u;
        }
        put_bits<8>(p, side, w8);
        return;
    }
#endif
    if (T <= 5u) {  // one word
#pragma unroll
        for (int k = 0; k < NF; ++k) w[0] ^= fl[k] << ((spk >> (8 * k)) & 31u);
    } else {
        if (T == 7u) {
            const uint4 s1 = sr[1];
            w[2] = s1.x;
            w[3] = s1.y;
        }
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const uint32_t ix = (spk >> (8 * k)) & 0xFFu;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] ^= ((ix >> 5) == (uint32_t)i) ? (fl[k] << (ix & 31u)) : 0u;
        }
    }
    // (one flat switch over (stage, side): a switch over the stage around put_bits' branch on the side made the compiler thread
    //  flag registers through the structured control flow -- see the G step of the walk)
    switch ((T << 1) | side) {
    case 2: wru<RB(0, 1)>(w[0]); break;
    case 3: wru<RB(1, 1)>(w[0]); break;
    case 4: wru<RB(0, 2)>(w[0]); break;
    case 5: wru<RB(1, 2)>(w[0]); break;
    case 6: wru<RB(0, 3)>(w[0]); break;
    case 7: wru<RB(1, 3)>(w[0]); break;
    case 8: wru<RB(0, 4)>(w[0]); break;
    case 9: wru<RB(1, 4)>(w[0]); break;
    case 10: wru<RB(0, 5)>(w[0]); break;
    case 11: wru<RB(1, 5)>(w[0]); break;
    case 12: wru<RBL6>(w[0]); wru<RBL6 + 1>(w[1]); break;
    case 13: wru<RBR6>(w[0]); wru<RBR6 + 1>(w[1]); break;
    case 14: wru<RBL7>(w[0]); wru<RBL7 + 1>(w[1]); wru<RBL7 + 2>(w[2]); wru<RBL7 + 3>(w[3]); break;
    case 15: wru<RBR7>(w[0]); wru<RBR7 + 1>(w[1]); wru<RBR7 + 2>(w[2]); wru<RBR7 + 3>(w[3]); break;
    default: break;
    }
}

// ---- COMBINE: bits[T] = [left ^ right | right], left = the slot holder's, right = own -------------------------------------
template <int T>
__device__ __forceinline__ void op_combine(Path& p, uint32_t side)
{
    const uint32_t ho = p.owner(T) << 2;  // holder of the left child's bits
    if constexpr (T <= 5) {
        constexpr int h = 1 << (T - 1);
        const uint32_t l = rd_lane<RBL0 + (T - 1)>(ho), r = rdu<RBR0 + (T - 1)>();
        const uint32_t w[1] = { ((l ^ r) & ((1u << h) - 1u)) | (r << h) };
        put_bits<T>(p, side, w);
    } else if constexpr (T == 6) {
        const uint32_t l = rd_lane<RBL0 + 5>(ho), r = rdu<RBR0 + 5>();
        const uint32_t w[2] = { l ^ r, r };
        put_bits<6>(p, side, w);
    } else if constexpr (T == 7) {
        const uint32_t l0 = rd_lane<RBL6>(ho), l1 = rd_lane<RBL6 + 1>(ho), r0 = rdu<RBR6>(), r1 = rdu<RBR6 + 1>();
        const uint32_t w[4] = { l0 ^ r0, l1 ^ r1, r0, r1 };
        put_bits<7>(p, side, w);
    } else if constexpr (T == 8) {
        const uint32_t l0 = rd_lane<RBL7>(ho), l1 = rd_lane<RBL7 + 1>(ho), l2 = rd_lane<RBL7 + 2>(ho), l3 = rd_lane<RBL7 + 3>(ho);
        const uint32_t r0 = rdu<RBR7>(), r1 = rdu<RBR7 + 1>(), r2 = rdu<RBR7 + 2>(), r3 = rdu<RBR7 + 3>();
        const uint32_t w[8] = { l0 ^ r0, l1 ^ r1, l2 ^ r2, l3 ^ r3, r0, r1, r2, r3 };
        put_bits<8>(p, side, w);
    } else {
#if LN_LOGN == 10
        // T == 9: 8 + 8 words from the arena -> 16 words to the arena
        const uint32_t* lp = p.arena + AR_BL8 + p.owner(9);
        const uint32_t* rp = p.arena + AR_BR8 + p.lane;
        uint32_t* o = p.arena + (side ? AR_BR9 : AR_BL9) + p.lane;
        uint32_t l[8], r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            l[i] = lp[i * 64];
            r[i] = rp[i * 64];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            o[i * 64] = l[i] ^ r[i];
            o[(8 + i) * 64] = r[i];
        }
#else
        // T = 9 .. LOGN - 1: 2^(T-6) + 2^(T-6) words from the arena -> 2^(T-5) words to the arena, eight at a time
        constexpr uint32_t NW = ar_bitwords(T - 1);
        // (opaque: otherwise the compiler precomputes one 64-bit address per word and block at the head of the kernel and
        //  keeps -- spills -- them all)
        uint32_t ln = (uint32_t)p.lane;
        asm volatile("" : "+v"(ln));
        const uint32_t* lp = p.arena + AR_BL(T - 1) + p.owner(T);
        const uint32_t* rp = p.arena + AR_BR(T - 1) + ln;
        uint32_t* o = p.arena + (side ? AR_BR(T) : AR_BL(T)) + ln;
#pragma unroll 1
        for (uint32_t i0 = 0; i0 < NW; i0 += 8) {
            uint32_t l[8], r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                l[i] = lp[(i0 + i) * 64];
                r[i] = rp[(i0 + i) * 64];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                o[(i0 + i) * 64] = l[i] ^ r[i];
                o[(NW + i0 + i) * 64] = r[i];
            }
        }
#endif
    }
}

#if LN_PKFMA
#undef LN_FPAIR
#define LN_FPAIR "126"  // (the top-of-tree passes: see LN_F4T)
#endif
// int8 channel rows: a dword of four LLRs in v[R] -> four floats in v[R .. R + 3] (v_cvt_f32_i32 on a sign-extended byte)
#define LN_WIDEN4(R_)                                                                                                   \
    "v_cvt_f32_i32_sdwa v[" R_ "+3], sext(v[" R_ "]) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n\t"           \
    "v_cvt_f32_i32_sdwa v[" R_ "+2], sext(v[" R_ "]) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n\t"           \
    "v_cvt_f32_i32_sdwa v[" R_ "+1], sext(v[" R_ "]) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n\t"           \
    "v_cvt_f32_i32_sdwa v[" R_ "], sext(v[" R_ "]) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n\t"
#if LN_LOGN == 10
// ---- the top of the tree -------------------------------------------------------------------------------------------------
// LN_VT: the stage-8 block of this path from the channel row (stage 9 = f / g of the row is never stored), optionally
// with the F of stage 8 on the values in registers.  Element e of stage 8 needs the row at e, e + 256, e + 512, e + 768.
// The registers of stages 1 .. 6 are dead while the top of the tree is being worked on: two sets of eight 16-byte
// row chunks live there (v[128..159], v[160..191]); one step = chunks j (-> stage-8 chunk j) and j + 32 (-> chunk j + 32),
// the loads of step j + 1 are issued before step j is computed.
constexpr int VT_SET0 = 128, VT_SET1 = 160, VT_SET2 = 192, VT_SET3 = 224;  // (v254 / v255 are free as well)
// set layout: +0 row[e], +4 row[e+256], +8 row[e+512], +12 row[e+768], +16 row[e+128], +20 row[e+384], +24 row[e+640], +28 row[e+896]
// (int8 rows: a chunk of four LLRs is one dword, fetched into the first register of its group of four and widened in place by
//  vt_widen -- v_cvt_f32_i32 with a sign-extended byte select, one instruction per LLR)
template <int SET>
__device__ __forceinline__ void vt_load(const int8_t* rowp)
{
    asm volatile(
        "global_load_dword v[%[r]], %[a], off\n\t"
        "global_load_dword v[%[r]+4], %[a], off offset:256\n\t"
        "global_load_dword v[%[r]+8], %[a], off offset:512\n\t"
        "global_load_dword v[%[r]+12], %[a], off offset:768\n\t"
        "global_load_dword v[%[r]+16], %[a], off offset:128\n\t"
        "global_load_dword v[%[r]+20], %[a], off offset:384\n\t"
        "global_load_dword v[%[r]+24], %[a], off offset:640\n\t"
        "global_load_dword v[%[r]+28], %[a], off offset:896"
        :
        : [r] "n"(SET), [a] "v"(rowp)
        : "memory");
}
template <int SET>
__device__ __forceinline__ void vt_widen()
{
    asm volatile("s_waitcnt vmcnt(24)\n\t" LN_WIDEN4("%[r]") LN_WIDEN4("%[r]+4") LN_WIDEN4("%[r]+8") LN_WIDEN4("%[r]+12") LN_WIDEN4("%[r]+16")
                 LN_WIDEN4("%[r]+20") LN_WIDEN4("%[r]+24") LN_WIDEN4("%[r]+28") "s_nop 0" ::[r] "n"(SET)
                 : "memory");
}
template <int SET>
__device__ __forceinline__ void vt_load(const float* rowp)
{
    asm volatile(
        "global_load_dwordx4 v[%[r]:%[r]+3], %[a], off\n\t"
        "global_load_dwordx4 v[%[r]+4:%[r]+7], %[a], off offset:1024\n\t"
        "global_load_dwordx4 v[%[r]+8:%[r]+11], %[a], off offset:2048\n\t"
        "global_load_dwordx4 v[%[r]+12:%[r]+15], %[a], off offset:3072\n\t"
        "global_load_dwordx4 v[%[r]+16:%[r]+19], %[a], off offset:512\n\t"
        "global_load_dwordx4 v[%[r]+20:%[r]+23], %[a], off offset:1536\n\t"
        "global_load_dwordx4 v[%[r]+24:%[r]+27], %[a], off offset:2560\n\t"
        "global_load_dwordx4 v[%[r]+28:%[r]+31], %[a], off offset:3584"
        :
        : [r] "n"(SET), [a] "v"(rowp)
        : "memory");
}
// one step on set SET.  wa..wd: the root's left-child bits for the four stage-9 chunks, ue / uf: the stage-9 node's
// left-child bits for the two stage-8 chunks -- each word shifted so that the chunk's four bits are bits 0 .. 3.
template <int SET, bool R9, bool ISG, bool F8, bool CH = false>
__device__ __forceinline__ void vt_step(uint32_t wa, uint32_t wb, uint32_t wc, uint32_t wd, uint32_t ue, uint32_t uf, uint32_t absmask,
                                        uint32_t signmask, uint32_t s8off, const uint32_t* s8base, uint32_t s7addr, float c127 = 0.0f,
                                        float m128 = 0.0f)
{
    uint32_t t0, t1, t2, t3;
#if LN_PKFMA
// (packed: the pair of +-1.0 factors in v[126:127] -- the bit registers of the stages <= 7 are dead during this pass)
#define LN_VG(D_, B_, W_)                                          \
    "v_lshlrev_b32 %[t0], 31, " W_ "\n\t"                         \
    "v_lshlrev_b32 %[t1], 30, " W_ "\n\t"                         \
    "v_and_or_b32 v126, %[t0], %[sm], 1.0\n\t"                    \
    "v_and_or_b32 v127, %[t1], %[sm], 1.0\n\t"                    \
    "v_lshlrev_b32 %[t2], 29, " W_ "\n\t"                         \
    "v_lshlrev_b32 %[t3], 28, " W_ "\n\t"                         \
    "v_pk_fma_f32 v[" D_ ":" D_ "+1], v[" D_ ":" D_ "+1], v[126:127], v[" B_ ":" B_ "+1]\n\t" \
    "v_and_or_b32 v126, %[t2], %[sm], 1.0\n\t"                    \
    "v_and_or_b32 v127, %[t3], %[sm], 1.0\n\t"                    \
    "v_pk_fma_f32 v[" D_ "+2:" D_ "+3], v[" D_ "+2:" D_ "+3], v[126:127], v[" B_ "+2:" B_ "+3]\n\t"
#else
#define LN_VG(D_, B_, W_)                                          \
    "v_lshlrev_b32 %[t0], 31, " W_ "\n\t"                         \
    "v_lshlrev_b32 %[t1], 30, " W_ "\n\t"                         \
    "v_lshlrev_b32 %[t2], 29, " W_ "\n\t"                         \
    "v_lshlrev_b32 %[t3], 28, " W_ "\n\t"                         \
    "v_and_or_b32 %[t0], %[t0], %[sm], 1.0\n\t"                   \
    "v_and_or_b32 %[t1], %[t1], %[sm], 1.0\n\t"                   \
    "v_and_or_b32 %[t2], %[t2], %[sm], 1.0\n\t"                   \
    "v_and_or_b32 %[t3], %[t3], %[sm], 1.0\n\t"                   \
    "v_fma_f32 v[" D_ "], v[" D_ "], %[t0], v[" B_ "]\n\t"        \
    "v_fma_f32 v[" D_ "+1], v[" D_ "+1], %[t1], v[" B_ "+1]\n\t"  \
    "v_fma_f32 v[" D_ "+2], v[" D_ "+2], %[t2], v[" B_ "+2]\n\t"  \
    "v_fma_f32 v[" D_ "+3], v[" D_ "+3], %[t3], v[" B_ "+3]\n\t"
#endif
#if LN_I8ROWS
    vt_widen<SET>();
#endif
    // stage 9 (in place): a -> +0, b -> +4, c -> +16, d -> +20
    if constexpr (R9) {
        asm volatile("s_waitcnt vmcnt(24)\n\t" LN_VG("%[r]", "%[r]+8", "%[wa]") LN_VG("%[r]+4", "%[r]+12", "%[wb]")
                     LN_VG("%[r]+16", "%[r]+24", "%[wc]") LN_VG("%[r]+20", "%[r]+28", "%[wd]")
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [r] "n"(SET), [wa] "v"(wa), [wb] "v"(wb), [wc] "v"(wc), [wd] "v"(wd), [sm] "s"(signmask)
                     : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(24)\n\t" LN_F4T("%[r]", "%[r]", "%[r]+8") LN_F4T("%[r]+4", "%[r]+4", "%[r]+12")
                     LN_F4T("%[r]+16", "%[r]+16", "%[r]+24") LN_F4T("%[r]+20", "%[r]+20", "%[r]+28")
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [r] "n"(SET), [am] "s"(absmask)
                     : "memory");
    }
    if constexpr (CH) {  // (int8 decoder: the values of stage 9 are clamped like stored ones)
        if constexpr (R9) {
            clamp_g<SET, 8>(c127, m128);
            clamp_g<SET + 16, 8>(c127, m128);
        } else {
            clamp_f<SET, 8>(c127, absmask);
            clamp_f<SET + 16, 8>(c127, absmask);
        }
    }
    // stage 8 (in place): x = op(a, b) -> +0, y = op(c, d) -> +16; stage 7: f(x, y) -> +4
    if constexpr (ISG) {
        asm volatile(LN_VG("%[r]", "%[r]+4", "%[ue]") LN_VG("%[r]+16", "%[r]+20", "%[uf]")
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [r] "n"(SET), [ue] "v"(ue), [uf] "v"(uf), [sm] "s"(signmask));
    } else {
        asm volatile(LN_F4T("%[r]", "%[r]", "%[r]+4") LN_F4T("%[r]+16", "%[r]+16", "%[r]+20")
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [r] "n"(SET), [am] "s"(absmask));
    }
    if constexpr (CH) {
        if constexpr (ISG) {
            clamp_g<SET, 4>(c127, m128);
            clamp_g<SET + 16, 4>(c127, m128);
        } else {
            clamp_f<SET, 4>(c127, absmask);
            clamp_f<SET + 16, 4>(c127, absmask);
        }
    }
    if constexpr (F8 && CH) {  // (the f of stage 8 is clamped before it goes to the LDS)
        asm volatile(LN_F4T("%[r]+4", "%[r]", "%[r]+16")
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [r] "n"(SET), [am] "s"(absmask));
        clamp_f<SET + 4, 4>(c127, absmask);
        asm volatile("global_store_dwordx4 %[o], v[%[r]:%[r]+3], %[sb]\n\t"
                     "global_store_dwordx4 %[o2], v[%[r]+16:%[r]+19], %[sb]\n\t"
                     "ds_write_b128 %[la], v[%[r]+4:%[r]+7]"
                     :
                     : [r] "n"(SET), [o] "v"(s8off), [o2] "v"(s8off + 32768u), [sb] "s"(s8base), [la] "v"(s7addr)
                     : "memory");
    } else if constexpr (F8) {
        asm volatile(LN_F4T("%[r]+4", "%[r]", "%[r]+16")
                     "global_store_dwordx4 %[o], v[%[r]:%[r]+3], %[sb]\n\t"
                     "global_store_dwordx4 %[o2], v[%[r]+16:%[r]+19], %[sb]\n\t"
                     "ds_write_b128 %[la], v[%[r]+4:%[r]+7]"
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [r] "n"(SET), [am] "s"(absmask), [o] "v"(s8off), [o2] "v"(s8off + 32768u), [sb] "s"(s8base), [la] "v"(s7addr)
                     : "memory");
    } else {
        asm volatile("s_nop 0\n\t"
                     "global_store_dwordx4 %[o], v[%[r]:%[r]+3], %[sb]\n\t"
                     "global_store_dwordx4 %[o2], v[%[r]+16:%[r]+19], %[sb]"
                     :
                     : [r] "n"(SET), [o] "v"(s8off), [o2] "v"(s8off + 32768u), [sb] "s"(s8base)
                     : "memory");
    }
#undef LN_VG
}
// six words of partial sums for the eight steps of one group (jo): asm loads, so that the counted waits of the steps cover them
template <bool R9, bool ISG>
__device__ __forceinline__ void vt_words(const uint32_t* bl9, const uint32_t* bl8, uint32_t jo, uint32_t& wa, uint32_t& wb, uint32_t& wc,
                                         uint32_t& wd, uint32_t& ue, uint32_t& uf)
{
    wa = wb = wc = wd = ue = uf = 0;
    if constexpr (R9) {
        const uint32_t* pa = bl9 + jo * 64;
        asm volatile("global_load_dword %0, %4, off\n\t"
                     "global_load_dword %1, %4, off offset:2048\n\t"
                     "global_load_dword %2, %4, off offset:1024\n\t"
                     "global_load_dword %3, %4, off offset:3072"
                     : "=&v"(wa), "=&v"(wb), "=&v"(wc), "=&v"(wd)
                     : "v"(pa)
                     : "memory");
    }
    if constexpr (ISG) {
        const uint32_t* pe = bl8 + jo * 64;
        asm volatile("global_load_dword %0, %2, off\n\t"
                     "global_load_dword %1, %2, off offset:1024"
                     : "=&v"(ue), "=&v"(uf)
                     : "v"(pe)
                     : "memory");
    }
}
template <bool R9, bool ISG, bool F8, bool CH>
__device__ __forceinline__ void op_vt_t(Path& p, const RowT* chan, uint32_t absmask, uint32_t signmask)
{
    if (R9 && !ISG) p.set_field(10);  // the G of the root directly follows its left child: the bits are this lane's own
    const uint32_t* bl9 = p.arena + AR_BL9 + p.owner(10);  // the root's left-child bits: with the slot holder
    const uint32_t* bl8 = p.arena + AR_BL8 + p.lane;       // own (the G follows its left child directly)
    const uint32_t* s8base = p.arena + AR_S8;
    uint32_t s8off = (uint32_t)p.lane * 16u;               // byte offset of chunk j of this lane's column (+ 32 KB: chunk j + 32)
    uint32_t s7addr = (uint32_t)(uintptr_t)p.lds + (uint32_t)p.lane * 16u;
    const RowT* rowp = chan;
    const RowT* const rowend = chan + 4 * 31;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (nothing older in flight: the counted waits below see only this op's accesses)
    uint32_t wa, wb, wc, wd, ue, uf;
    vt_words<R9, ISG>(bl9, bl8, 0, wa, wb, wc, wd, ue, uf);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(wa), "+v"(wb), "+v"(wc), "+v"(wd), "+v"(ue), "+v"(uf)::"memory");
    vt_load<VT_SET0>(rowp);
    vt_load<VT_SET1>(rowp + 4);
    vt_load<VT_SET2>(rowp + 8);
    rowp += 12;
#pragma unroll 1
    for (uint32_t jo = 0; jo < 4; ++jo) {
        // the words of the next group are requested now and are long there when it starts (every step waits for
        // everything older than its own row chunks)
        uint32_t na, nb, nc, nd, ne, nf;
        vt_words<R9, ISG>(bl9, bl8, jo < 3u ? jo + 1u : 3u, na, nb, nc, nd, ne, nf);
#pragma unroll 1
        for (uint32_t ji = 0; ji < 8; ji += 4) {
            vt_load<VT_SET3>(rowp < rowend ? rowp : rowend);
            vt_step<VT_SET0, R9, ISG, F8, CH>(wa, wb, wc, wd, ue, uf, absmask, signmask, s8off, s8base, s7addr, p.c127, p.m128);
            vt_load<VT_SET0>(rowp + 4 < rowend ? rowp + 4 : rowend);
            vt_step<VT_SET1, R9, ISG, F8, CH>(wa >> 4, wb >> 4, wc >> 4, wd >> 4, ue >> 4, uf >> 4, absmask, signmask, s8off + 1024u, s8base,
                                          s7addr + 1024u, p.c127, p.m128);
            vt_load<VT_SET1>(rowp + 8 < rowend ? rowp + 8 : rowend);
            vt_step<VT_SET2, R9, ISG, F8, CH>(wa >> 8, wb >> 8, wc >> 8, wd >> 8, ue >> 8, uf >> 8, absmask, signmask, s8off + 2048u, s8base,
                                          s7addr + 2048u, p.c127, p.m128);
            vt_load<VT_SET2>(rowp + 12 < rowend ? rowp + 12 : rowend);
            vt_step<VT_SET3, R9, ISG, F8, CH>(wa >> 12, wb >> 12, wc >> 12, wd >> 12, ue >> 12, uf >> 12, absmask, signmask, s8off + 3072u,
                                          s8base, s7addr + 3072u, p.c127, p.m128);
            wa >>= 16; wb >>= 16; wc >>= 16; wd >>= 16; ue >>= 16; uf >>= 16;
            rowp += 16;
            s8off += 4096u;
            s7addr += 4096u;
        }
        asm volatile("" : "+v"(na), "+v"(nb), "+v"(nc), "+v"(nd), "+v"(ne), "+v"(nf));  // (nothing derived from them may be computed earlier)
        wa = na; wb = nb; wc = nc; wd = nd; ue = ne; uf = nf;
    }
    if (ISG) p.set_field(9);
    p.set_field(8);
    if (F8) p.set_field(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#if LN_LCL == 3
// The first pass of a codeword (F of the root, F of its left child): one live path, whose lane would do 32 steps while
// the other seven of the codeword idle.  The steps do not depend on the path, so lane q does the steps j = q, q + 8,
// q + 16, q + 24 and stores into the column of the codeword's first lane, where the path lives.
template <bool CH>
__device__ __forceinline__ void op_vt_first(Path& p, const RowT* chan, uint32_t absmask, uint32_t signmask)
{
    const uint32_t* s8base = p.arena + AR_S8;
    const uint32_t s8off = (uint32_t)p.g0 * 16u + (uint32_t)p.q * 1024u;
    const uint32_t s7addr = (uint32_t)(uintptr_t)p.lds + (uint32_t)p.g0 * 16u + (uint32_t)p.q * 1024u;
    const RowT* rowp = chan + 4 * p.q;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    vt_load<VT_SET0>(rowp);
    vt_load<VT_SET1>(rowp + 32);
    vt_load<VT_SET2>(rowp + 64);
    vt_load<VT_SET3>(rowp + 96);
    // (vt_step's counted wait assumes the next set's loads are issued in front of every step; here nothing follows the
    //  four sets, so everything is waited for first)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    vt_step<VT_SET0, false, false, true, CH>(0, 0, 0, 0, 0, 0, absmask, signmask, s8off, s8base, s7addr, p.c127, p.m128);
    vt_step<VT_SET1, false, false, true, CH>(0, 0, 0, 0, 0, 0, absmask, signmask, s8off + 8192u, s8base, s7addr + 8192u, p.c127, p.m128);
    vt_step<VT_SET2, false, false, true, CH>(0, 0, 0, 0, 0, 0, absmask, signmask, s8off + 16384u, s8base, s7addr + 16384u, p.c127, p.m128);
    vt_step<VT_SET3, false, false, true, CH>(0, 0, 0, 0, 0, 0, absmask, signmask, s8off + 24576u, s8base, s7addr + 24576u, p.c127, p.m128);
    p.set_field(8);
    p.set_field(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#elif LN_LCL < 3
// (4 / 2 lanes per codeword: lane q does the steps j = q + LC i, 8 / 16 of them, four sets at a time)
template <bool CH>
__device__ __forceinline__ void op_vt_first(Path& p, const RowT* chan, uint32_t absmask, uint32_t signmask)
{
    const uint32_t* s8base = p.arena + AR_S8;
    uint32_t s8off = (uint32_t)p.g0 * 16u + (uint32_t)p.q * 1024u;
    uint32_t s7addr = (uint32_t)(uintptr_t)p.lds + (uint32_t)p.g0 * 16u + (uint32_t)p.q * 1024u;
    const RowT* rowp = chan + 4 * p.q;
#pragma unroll 1
    for (uint32_t b = 0; b < 8u / (uint32_t)LC; ++b) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        vt_load<VT_SET0>(rowp);
        vt_load<VT_SET1>(rowp + 4 * LC);
        vt_load<VT_SET2>(rowp + 8 * LC);
        vt_load<VT_SET3>(rowp + 12 * LC);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        vt_step<VT_SET0, false, false, true, CH>(0, 0, 0, 0, 0, 0, absmask, signmask, s8off, s8base, s7addr, p.c127, p.m128);
        vt_step<VT_SET1, false, false, true, CH>(0, 0, 0, 0, 0, 0, absmask, signmask, s8off + 1024u * LC, s8base, s7addr + 1024u * LC, p.c127, p.m128);
        vt_step<VT_SET2, false, false, true, CH>(0, 0, 0, 0, 0, 0, absmask, signmask, s8off + 2048u * LC, s8base, s7addr + 2048u * LC, p.c127, p.m128);
        vt_step<VT_SET3, false, false, true, CH>(0, 0, 0, 0, 0, 0, absmask, signmask, s8off + 3072u * LC, s8base, s7addr + 3072u * LC, p.c127, p.m128);
        rowp += 16 * LC;
        s8off += 4096u * (uint32_t)LC;
        s7addr += 4096u * (uint32_t)LC;
    }
    p.set_field(8);
    p.set_field(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#else
// (16 / 32 lanes per codeword: lane q does the steps j = q, q + LC, ... < 32)
template <bool CH>
__device__ __forceinline__ void op_vt_first(Path& p, const RowT* chan, uint32_t absmask, uint32_t signmask)
{
    const uint32_t* s8base = p.arena + AR_S8;
    const uint32_t s8off = (uint32_t)p.g0 * 16u + (uint32_t)p.q * 1024u;
    const uint32_t s7addr = (uint32_t)(uintptr_t)p.lds + (uint32_t)p.g0 * 16u + (uint32_t)p.q * 1024u;
    const RowT* rowp = chan + 4 * p.q;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    vt_load<VT_SET0>(rowp);
    if constexpr (LC == 16) vt_load<VT_SET1>(rowp + 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    vt_step<VT_SET0, false, false, true, CH>(0, 0, 0, 0, 0, 0, absmask, signmask, s8off, s8base, s7addr, p.c127, p.m128);
    if constexpr (LC == 16)
        vt_step<VT_SET1, false, false, true, CH>(0, 0, 0, 0, 0, 0, absmask, signmask, s8off + 16384u, s8base, s7addr + 16384u, p.c127, p.m128);
    p.set_field(8);
    p.set_field(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#endif
template <bool CH>
__device__ __forceinline__ void op_vt(Path& p, uint32_t off, const RowT* chan, uint32_t absmask, uint32_t signmask)
{
    // (bit 2 of off -- "the F of stage 8 follows" -- needs no variant of its own: stage 7 is dead whenever it is clear, so the
    //  F is computed and stored regardless)
    switch (off & 3u) {
    case 0:
        if (p.P == 1u) op_vt_first<CH>(p, chan, absmask, signmask);  // (always: the program starts with this op)
        else op_vt_t<false, false, true, CH>(p, chan, absmask, signmask);
        break;
    case 1: op_vt_t<true, false, true, CH>(p, chan, absmask, signmask); break;
    case 2: op_vt_t<false, true, true, CH>(p, chan, absmask, signmask); break;
    default: op_vt_t<true, true, true, CH>(p, chan, absmask, signmask); break;
    }
}

#endif  // LN_LOGN == 10
#if LN_LOGN != 10
// ---- the top of the tree at other block lengths: LV = LOGN - 8 levels between the channel row and stage 8 --------------------
// One step = ONE 16-byte chunk c of the path's stage-8 block (elements 4 c .. 4 c + 3): the 2^LV row chunks it depends on
// (row[4 c + 256 k ..], k < 2^LV) are loaded into one register set and folded in place, level by level -- level l (0 = the
// channel row's own f / g) pairs chunk k with chunk k + 2^(LV-1-l) --, f or g per level as the instruction's mask says (wave
// uniform), the g's steered by the left-child bits of that level's node (arena, column of the slot owner; one word per block
// and eight steps, bit 4 (c & 7) + i of it for element i: the shift amounts travel as scalar operands).  The result goes to
// the arena; the F of stage 8 pairs chunk c with chunk c + 32: the chunks c < 32 are parked in the LDS slot of stage-7 chunk
// c (the same 16 bytes per lane), the step of chunk c + 32 reads the slot back (into v[104 .. 107]: the bit registers of the
// stages <= 7 are dead during this op), and leaves f there.  NS register sets of 4 * 2^LV registers from v128 on: the loads of
// step c + NS - 1 are issued in front of step c.
#if LN_PKFMA
// (the four fma as two v_pk_fma_f32, the +-1.0 factors built in the free registers v[124 .. 127])
#define LN_VGS(D_, B_, W_)                                        \
    "v_lshlrev_b32 v124, %[s0], " W_ "\n\t"                      \
    "v_lshlrev_b32 v125, %[s1], " W_ "\n\t"                      \
    "v_lshlrev_b32 v126, %[s2], " W_ "\n\t"                      \
    "v_lshlrev_b32 v127, %[s3], " W_ "\n\t"                      \
    "v_and_or_b32 v124, v124, %[sm], 1.0\n\t"                    \
    "v_and_or_b32 v125, v125, %[sm], 1.0\n\t"                    \
    "v_and_or_b32 v126, v126, %[sm], 1.0\n\t"                    \
    "v_and_or_b32 v127, v127, %[sm], 1.0\n\t"                    \
    "v_pk_fma_f32 v[" D_ ":" D_ "+1], v[" D_ ":" D_ "+1], v[124:125], v[" B_ ":" B_ "+1]\n\t" \
    "v_pk_fma_f32 v[" D_ "+2:" D_ "+3], v[" D_ "+2:" D_ "+3], v[126:127], v[" B_ "+2:" B_ "+3]"
#else
#define LN_VGS(D_, B_, W_)                                        \
    "v_lshlrev_b32 %[t0], %[s0], " W_ "\n\t"                     \
    "v_lshlrev_b32 %[t1], %[s1], " W_ "\n\t"                     \
    "v_lshlrev_b32 %[t2], %[s2], " W_ "\n\t"                     \
    "v_lshlrev_b32 %[t3], %[s3], " W_ "\n\t"                     \
    "v_and_or_b32 %[t0], %[t0], %[sm], 1.0\n\t"                  \
    "v_and_or_b32 %[t1], %[t1], %[sm], 1.0\n\t"                  \
    "v_and_or_b32 %[t2], %[t2], %[sm], 1.0\n\t"                  \
    "v_and_or_b32 %[t3], %[t3], %[sm], 1.0\n\t"                  \
    "v_fma_f32 v[" D_ "], v[" D_ "], %[t0], v[" B_ "]\n\t"       \
    "v_fma_f32 v[" D_ "+1], v[" D_ "+1], %[t1], v[" B_ "+1]\n\t" \
    "v_fma_f32 v[" D_ "+2], v[" D_ "+2], %[t2], v[" B_ "+2]\n\t" \
    "v_fma_f32 v[" D_ "+3], v[" D_ "+3], %[t3], v[" B_ "+3]"
#endif
constexpr int VTG_H = 104;  // v[104 .. 107]: the parked stage-8 chunk of the second half's steps
constexpr int VTG_W = 108;  // v[108 .. 122]: the bit words of the current group of eight steps (word i of the op: v[108 + i])
constexpr int VTG_A = 240;  // a[240 .. 254]: the words of the NEXT group land here (the compiler's spills stay below a104)
template <int LV>
struct VtgGeo {
    static constexpr int NR = 1 << LV;                         // row chunks per step (LV = 0: the source row is the block itself)
    static constexpr int NS = LV >= 4 ? 2 : (LV == 3 ? 4 : 8);  // register sets
    static constexpr int SW = 4 * NR;                          // registers per set
    static constexpr int NWD = NR - 1;                         // bit words per group of eight steps (all levels)
    __host__ __device__ static constexpr int wbase(int l) { return NR - (NR >> l); }  // first word of level l
};
// the 2^LV row chunks of one step: base = the row at element 4 c
// (int8 rows: one dword per chunk into the first register of its group, 2^TOUT bytes apart; widened in place by vtg_widen)
template <int S, int LV, int TOUT = 8>
__device__ __forceinline__ void vtg_load(const int8_t* base)
{
    static_for<(1 << LV)>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const int8_t* pm = base;
        asm volatile("global_load_dword v[%0], %1, off offset:%2" ::"n"(S + 4 * k), "v"(pm), "n"(k << TOUT) : "memory");
    });
}
template <int S, int LV, typename RT>
__device__ __forceinline__ void vtg_widen()
{
    if constexpr (sizeof(RT) == 1) {
        static_for<(1 << LV)>([&](auto K) {
            constexpr int k = decltype(K)::value;
            asm volatile(LN_WIDEN4("%[r]") "s_nop 0" ::[r] "n"(S + 4 * k));
        });
    }
}
// (TOUT: the stage of the block the step produces -- chunk k of its input is 2^TOUT elements further on; 8 for the top-of-tree pass)
template <int S, int LV, int TOUT = 8>
__device__ __forceinline__ void vtg_load(const float* base)
{
    static_for<(1 << LV)>([&](auto K) {
        constexpr int k = decltype(K)::value;
        constexpr int bytes = (4 * k) << TOUT;
        const float* pm = base + (bytes & ~4095) / 4;
        asm volatile("global_load_dwordx4 v[%0:%0+3], %1, off offset:%2" ::"n"(S + 4 * k), "v"(pm), "n"(bytes & 4095) : "memory");
    });
}
// the levels of one step on set S, in place: the stage-8 chunk ends up in v[S .. S + 3].  gm bit l: level l is a g (its
// words: v[VTG_W + wbase(l) + k]), sh = 31 - 4 (c & 7)
template <int S, int LV, bool CH>
__device__ __forceinline__ void vtg_levels(uint32_t gm, uint32_t sh, uint32_t absmask, uint32_t signmask, float c127, float m128)
{
    using G = VtgGeo<LV>;
    uint32_t s0 = sh, s1 = sh - 1u, s2 = sh - 2u, s3 = sh - 3u;  // (not const: a const local is not captured by the lambdas' asm operands)
    uint32_t am = absmask, sm = signmask;
    float cc = c127, mm = m128;
    static_for<LV>([&](auto Lc) {
        constexpr int l = decltype(Lc)::value;
        constexpr int NB = 1 << (LV - 1 - l);
        if (gm & (1u << l)) {
            static_for<NB>([&](auto K) {
                constexpr int k = decltype(K)::value;
                uint32_t t0, t1, t2, t3;
                const uint32_t q0 = s0, q1 = s1, q2 = s2, q3 = s3, qm = sm;  // (copies: the asm operands of a nested lambda do not capture)
                asm volatile(LN_VGS("%[d]", "%[b]", "v[%[w]]")
                             : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                             : [d] "n"(S + 4 * k), [b] "n"(S + 4 * (k + NB)), [w] "n"(VTG_W + G::wbase(l) + k), [s0] "s"(q0), [s1] "s"(q1), [s2] "s"(q2),
                               [s3] "s"(q3), [sm] "s"(qm));
            });
            if constexpr (CH) clamp_g<S, 4 * NB>(cc, mm);
        } else {
            static_for<NB>([&](auto K) {
                constexpr int k = decltype(K)::value;
                uint32_t t0, t1, t2, t3;
                const uint32_t qa = am;
                asm volatile(LN_F4T("%[d]", "%[d]", "%[b]")
                             : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                             : [d] "n"(S + 4 * k), [b] "n"(S + 4 * (k + NB)), [am] "s"(qa));
            });
            if constexpr (CH) clamp_f<S, 4 * NB>(cc, am);
        }
    });
}
// what a step does with its chunk.  First half (chunks c < 32): to the arena and into the LDS slot of stage-7 chunk c.
template <int S>
__device__ __forceinline__ void vtg_tail_first(uint32_t s8off, const uint32_t* s8base, uint32_t s7addr)
{
    asm volatile("s_nop 0\n\t"
                 "global_store_dwordx4 %[o], v[%[r]:%[r]+3], %[sb]\n\t"
                 "ds_write_b128 %[la], v[%[r]:%[r]+3]"
                 :
                 : [r] "n"(S), [o] "v"(s8off), [sb] "s"(s8base), [la] "v"(s7addr)
                 : "memory");
}
// Second half (chunk c + 32, its partner c parked in v[104 .. 107] by vtg_fetch_parked): to the arena, f of the two to the LDS slot
template <int S, bool CH>
__device__ __forceinline__ void vtg_tail_second(uint32_t s8off, const uint32_t* s8base, uint32_t s7addr, uint32_t absmask, float c127)
{
    uint32_t t0, t1, t2, t3;
    asm volatile("s_nop 0\n\t"
                 "global_store_dwordx4 %[o], v[%[r]:%[r]+3], %[sb]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t" LN_F4T("%[h]", "%[h]", "%[r]")
                 : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                 : [r] "n"(S), [h] "n"(VTG_H), [o] "v"(s8off), [sb] "s"(s8base), [am] "s"(absmask)
                 : "memory");
    if constexpr (CH) clamp_f<VTG_H, 4>(c127, absmask);
    asm volatile("s_nop 0\n\tds_write_b128 %[la], v[%[h]:%[h]+3]" ::[h] "n"(VTG_H), [la] "v"(s7addr) : "memory");
}
__device__ __forceinline__ void vtg_fetch_parked(uint32_t s7addr)
{
    asm volatile("ds_read_b128 v[%[h]:%[h]+3], %[la]" ::[h] "n"(VTG_H), [la] "v"(s7addr) : "memory");
}
// The bit words of one group of eight steps (g = 0 .. 7) are requested into a[VTG_A ..]: for every g level l the words g + 8 k
// of its node's left-child block (bo[l]: byte offset of that block in the column of the slot owner, relative to the wave's
// arena).  They are older than every row chunk of the group's later steps, so the steps' counted waits cover them.
template <int LV, int TOUT = 8>
__device__ __forceinline__ void vtg_words_issue(uint32_t gm, const uint32_t (&bo)[LV > 0 ? LV : 1], uint32_t g, const uint32_t* abase)
{
    using G = VtgGeo<LV>;
    static_for<LV>([&](auto Lc) {
        constexpr int l = decltype(Lc)::value;
        constexpr int NB = 1 << (LV - 1 - l);
        if (gm & (1u << l)) {
            const uint32_t o = bo[l] + g * 256u;
            const uint32_t* ab = abase;
            static_for<NB>([&](auto K) {
                constexpr int k = decltype(K)::value;
                // word g + 2^(TOUT-5) k of the block: 2^(TOUT+3) k bytes further on in the lane's column
                constexpr uint32_t bytes = (uint32_t)k << (TOUT + 3);
                const uint32_t ok = o + (bytes & ~4095u);
                const uint32_t* qb = ab;
                asm volatile("global_load_dword a[%0], %1, %2 offset:%3" ::"n"(VTG_A + G::wbase(l) + k), "v"(ok), "s"(qb), "n"(bytes & 4095u) : "memory");
            });
        }
    });
}
// ... and moved to v[VTG_W ..] when their group starts (all of them: a word of an f level is never looked at)
template <int LV>
__device__ __forceinline__ void vtg_words_take()
{
    static_for<(1 << LV) - 1>([&](auto I) {
        constexpr int i = decltype(I)::value;
        asm volatile("v_accvgpr_read_b32 v[%0], a[%1]" ::"n"(VTG_W + i), "n"(VTG_A + i));
    });
}
// What a step does with its chunk (the sinks of vtg_core).  The top-of-tree pass: arena + the F of stage 8 through the LDS slots.
template <bool CH>
struct SinkStage8 {
    const uint32_t* s8base;
    uint32_t s8col, s7col, absmask;
    float c127;
    __device__ __forceinline__ void pre(uint32_t c) const
    {
        if (c >= 32u) vtg_fetch_parked(s7col + 1024u * (c - 32u));
    }
    template <int S, int PAR>
    __device__ __forceinline__ void take(uint32_t c)
    {
        if (c >= 32u) vtg_tail_second<S, CH>(s8col + 1024u * c, s8base, s7col + 1024u * (c - 32u), absmask, c127);
        else vtg_tail_first<S>(s8col + 1024u * c, s8base, s7col + 1024u * c);
    }
};
// An all-frozen leaf above stage 8 (RateZeroDecoder::decode scl_avx_float.cpp:316-337 on a block that is never stored): the sum of
// min(x, 0) in the reference's order -- eight strided accumulators, acc[j] += x[8 k + j]: chunk c feeds acc[4 (c & 1) ..].
struct SinkMin0 {
    float acc[8];
    __device__ __forceinline__ void pre(uint32_t) const {}
    template <int S, int PAR>
    __device__ __forceinline__ void take(uint32_t)
    {
        float t0, t1, t2, t3;
        asm volatile("v_min_f32 %[t0], 0, v[%[r]]\n\t"
                     "v_min_f32 %[t1], 0, v[%[r]+1]\n\t"
                     "v_min_f32 %[t2], 0, v[%[r]+2]\n\t"
                     "v_min_f32 %[t3], 0, v[%[r]+3]\n\t"
                     "v_add_f32 %[a0], %[a0], %[t0]\n\t"
                     "v_add_f32 %[a1], %[a1], %[t1]\n\t"
                     "v_add_f32 %[a2], %[a2], %[t2]\n\t"
                     "v_add_f32 %[a3], %[a3], %[t3]"
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [a0] "+v"(acc[4 * PAR]), [a1] "+v"(acc[4 * PAR + 1]),
                       [a2] "+v"(acc[4 * PAR + 2]), [a3] "+v"(acc[4 * PAR + 3])
                     : [r] "n"(S));
    }
};
// The pass itself.  TOP: the stage of the source row (LOGN: the channel row; lower: a virtual row), TOUT: the stage of the block
// it produces chunk by chunk (8: the top-of-tree pass), LV = TOP - TOUT levels; gm bit l: the op at stage TOP - l is a g.
// 2^TOUT / 4 steps in groups of eight (one bit word per block and group), NS register sets.
// (NSO: another number of register sets than VtgGeo's -- the leaves above stage 8 run with two: their sinks are large, and every
//  set is another copy of them)
template <int TOP, int TOUT, bool CH, typename RT, typename SINK, int NSO = 0>
__device__ __forceinline__ void vtg_core(Path& p, uint32_t gm, const RT* chan, uint32_t absmask, uint32_t signmask, SINK& sink)
{
    constexpr int LV = TOP - TOUT;
    using G = VtgGeo<LV>;
    constexpr int NS = NSO ? NSO : G::NS, SW = G::SW;
    constexpr uint32_t NG = (1u << TOUT) / 32u;  // groups of eight steps
    // the g that has just been executed is the one at the lowest stage: its left child's bits are this lane's own
    if (gm) p.set_field((uint32_t)TOP - (31u - (uint32_t)__builtin_clz(gm)));
    uint32_t bo[LV > 0 ? LV : 1];
    static_for<LV>([&](auto Lc) {
        constexpr int l = decltype(Lc)::value;
        bo[l] = 4u * (AR_BL(TOP - l - 1) + p.owner((uint32_t)(TOP - l)));
    });
    const uint32_t* abase = p.arena;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (nothing older in flight: the counted waits below see only this op's accesses)
    vtg_words_issue<LV, TOUT>(gm, bo, 0, abase);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LN_STAMP_VT
    uint32_t vt_acc[4] = { 0u, 0u, 0u, 0u };
#endif
    const RT* const rowlast = chan + 4 * (8 * NG - 1);
    static_for<NS - 1>([&](auto J) { vtg_load<128 + SW * decltype(J)::value, LV, TOUT>(chan + 4 * decltype(J)::value); });
    const RT* rowp = chan + 4 * (NS - 1);  // the row at the chunk of the next step to be loaded
#pragma unroll 1
    for (uint32_t g = 0; g < NG; ++g) {
        vtg_words_take<LV>();
        // the words of the next group are requested now and are long there when it starts
        vtg_words_issue<LV, TOUT>(gm, bo, g + 1u < NG ? g + 1u : NG - 1u, abase);
#pragma unroll 1
        for (uint32_t i = 0; i < 8; i += NS) {
            static_for<NS>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const uint32_t c = 8u * g + i + (uint32_t)j;
#ifdef LN_STAMP_VT
                const uint64_t ta = __builtin_amdgcn_s_memtime();
#endif
                sink.pre(c);
                vtg_load<128 + SW * ((j + NS - 1) % NS), LV, TOUT>(rowp < rowlast ? rowp : rowlast);
                rowp += 4;
#ifdef LN_STAMP_VT
                const uint64_t tb = __builtin_amdgcn_s_memtime();
#endif
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * G::NR) : "memory");
#ifdef LN_STAMP_VT
                const uint64_t tc = __builtin_amdgcn_s_memtime();
#endif
                vtg_widen<128 + SW * j, LV, RT>();
                vtg_levels<128 + SW * j, LV, CH>(gm, 31u - 4u * ((i + (uint32_t)j) & 7u), absmask, signmask, p.c127, p.m128);
#ifdef LN_STAMP_VT
                const uint64_t td = __builtin_amdgcn_s_memtime();
#endif
                sink.template take<128 + SW * j, (j & 1)>(c);
#ifdef LN_STAMP_VT
                const uint64_t te = __builtin_amdgcn_s_memtime();
                vt_acc[0] += (uint32_t)(tb - ta);
                vt_acc[1] += (uint32_t)(tc - tb);
                vt_acc[2] += (uint32_t)(td - tc);
                vt_acc[3] += (uint32_t)(te - td);
#endif
            });
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef LN_STAMP_VT
    if (p.vt_sub && threadIdx.x == 0)
        for (int i = 0; i < 4; ++i) p.vt_sub[i] = vt_acc[i];
#endif
}
template <int TOP, bool CH, typename RT>
__device__ __forceinline__ void op_vtg_t(Path& p, uint32_t gm, const RT* chan, uint32_t absmask, uint32_t signmask)
{
    SinkStage8<CH> sink{ p.arena + AR_S8, (uint32_t)p.lane * 16u, (uint32_t)(uintptr_t)p.lds + (uint32_t)p.lane * 16u, absmask, p.c127 };
    vtg_core<TOP, 8, CH, RT>(p, gm, chan, absmask, signmask, sink);
    p.set_field(8);
    p.set_field(7);
}
// every level an f, and what the levels above the last one produce goes to the codeword's virtual rows: after level l the set
// holds the chunks c + 64 k, k < 2^(LV-1-l), of the leftmost block of stage LOGN - 1 - l
template <int S, int LV, bool CH>
__device__ __forceinline__ void vtg_levels_first(float* vrow, uint32_t c, uint32_t absmask, float c127)
{
    uint32_t am = absmask;
    float cc = c127;
    static_for<LV>([&](auto Lc) {
        constexpr int l = decltype(Lc)::value;
        constexpr int NB = 1 << (LV - 1 - l);
        static_for<NB>([&](auto K) {
            constexpr int k = decltype(K)::value;
            uint32_t t0, t1, t2, t3;
            const uint32_t qa = am;
            asm volatile(LN_F4T("%[d]", "%[d]", "%[b]")
                         : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                         : [d] "n"(S + 4 * k), [b] "n"(S + 4 * (k + NB)), [am] "s"(qa));
        });
        if constexpr (CH) clamp_f<S, 4 * NB>(cc, am);
        if constexpr (l + 1 < LV) {
            float* const dst = vrow + vrow_off(LOGN - 1 - l) + 4u * c;
            static_for<NB>([&](auto K) {
                constexpr int k = decltype(K)::value;
                float* const dk = dst + 256 * (k & ~3);  // (chunk c + 64 k: 1024 k bytes on; the immediate offset takes k & 3)
                asm volatile("s_nop 0\n\tglobal_store_dwordx4 %0, v[%1:%1+3], off offset:%2" ::"v"(dk), "n"(S + 4 * k), "n"((k & 3) * 1024) : "memory");
            });
        }
    });
}
// TAILS = false (LN_VROWS): the virtual rows only -- the program of a low-rate code starts with an all-frozen leaf above stage 8,
// not with this pass
template <int LV, bool CH, bool TAILS = true>
__device__ __forceinline__ void op_vtg_first(Path& p, const RowT* chan, uint32_t absmask, uint32_t signmask)
{
    const uint32_t* s8base = p.arena + AR_S8;
    const uint32_t s8col = (uint32_t)p.g0 * 16u, s7col = (uint32_t)(uintptr_t)p.lds + (uint32_t)p.g0 * 16u;
    float* const vrow = reinterpret_cast<float*>(p.arena + AR_VROW) + ((uint32_t)p.lane >> LCL) * VROW_FLOATS;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
    for (uint32_t c = (uint32_t)p.q; c < 32u; c += (uint32_t)LC) {
        vtg_load<128, LV>(chan + 4 * c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        vtg_widen<128, LV, RowT>();
        vtg_levels_first<128, LV, CH>(vrow, c, absmask, p.c127);
        if constexpr (TAILS) {
            vtg_tail_first<128>(s8col + 1024u * c, s8base, s7col + 1024u * c);
            vtg_fetch_parked(s7col + 1024u * c);
        }
        vtg_load<128, LV>(chan + 4 * (c + 32u));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        vtg_widen<128, LV, RowT>();
        vtg_levels_first<128, LV, CH>(vrow, c + 32u, absmask, p.c127);
        if constexpr (TAILS) vtg_tail_second<128, CH>(s8col + 1024u * (c + 32u), s8base, s7col + 1024u * c, absmask, p.c127);
    }
    if constexpr (TAILS) {
        p.set_field(8);
        p.set_field(7);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
#if LN_LOGN == 6
// N = 64: the channel row IS stage 6 -- a register block.  The lanes of a codeword fetch its 16 chunks between them and pass them through the LDS
// column of the codeword's first lane; then EVERY lane reads the 16 chunks into its own v[190 .. 253]: the live path finds the row in its own
// registers, and so does whoever later looks it up in the lane the inherited slot word names.
template <bool CH>
__device__ __forceinline__ void op_copy6(Path& p, const RowT* chan)
{
    constexpr int NC = LC >= 16 ? 1 : 16 / LC;  // chunks per lane (32 lanes: the first 16 fetch one each)
    const RowT* rp = chan + 4 * (p.q & 15);
    uint32_t col = (uint32_t)(uintptr_t)p.lds + (uint32_t)p.g0 * 16u + (uint32_t)(p.q & 15) * 1024u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_for<NC>([&](auto I) { vtg_load<128 + 4 * decltype(I)::value, 0>(rp + 4 * LC * decltype(I)::value); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_for<NC>([&](auto I) {
        constexpr int i = decltype(I)::value;
        vtg_widen<128 + 4 * i, 0, RowT>();
        const uint32_t cc = col;  // (a copy: the asm operands of a generic lambda do not capture)
        asm volatile("s_nop 0\n\tds_write_b128 %[c], v[%[r]:%[r]+3] offset:%[o]" ::[c] "v"(cc), [r] "n"(128 + 4 * i), [o] "n"(4 * LDS_S7 + 1024 * LC * i) : "memory");
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t src = (uint32_t)(uintptr_t)p.lds + (uint32_t)p.g0 * 16u;
    static_for<16>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const uint32_t ss = src;
        asm volatile("ds_read_b128 v[%[r]:%[r]+3], %[a] offset:%[o]" ::[a] "v"(ss), [r] "n"(RS(6) + 4 * i), [o] "n"(4 * LDS_S7 + 1024 * i) : "memory");
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
#elif LN_LOGN == 7
// N = 128: the channel row IS stage 7.  The one live path sits in the first lane of its codeword; the LC lanes of the codeword fetch the
// row's 32 chunks between them (all loads in flight at once) and write them into that lane's LDS column, where the F of the root finds
// them -- and, through the slot word every path inherits, the root's G.
template <bool CH>
__device__ __forceinline__ void op_copy7(Path& p, const RowT* chan)
{
    constexpr int NC = 32 / LC;  // chunks per lane
    const RowT* rp = chan + 4 * p.q;
    uint32_t col = (uint32_t)(uintptr_t)p.lds + (uint32_t)p.g0 * 16u + (uint32_t)p.q * 1024u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_for<NC>([&](auto I) { vtg_load<128 + 4 * decltype(I)::value, 0>(rp + 4 * LC * decltype(I)::value); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_for<NC>([&](auto I) {
        constexpr int i = decltype(I)::value;
        vtg_widen<128 + 4 * i, 0, RowT>();
        const uint32_t cc = col;  // (a copy: the asm operands of a generic lambda do not capture)
        asm volatile("s_nop 0\n\tds_write_b128 %[c], v[%[r]:%[r]+3] offset:%[o]" ::[c] "v"(cc), [r] "n"(128 + 4 * i), [o] "n"(4 * LDS_S7 + 1024 * LC * i) : "memory");
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
#else
// off: bit l = level l (0 = the channel row's own f / g, i.e. the op of the root) is a g
template <bool CH>
__device__ __forceinline__ void op_vtg(Path& p, uint32_t off, const RowT* chan, uint32_t absmask, uint32_t signmask)
{
    const uint32_t gm = off & ((1u << TOPD) - 1u);
    if (gm == 0u && p.P == 1u) {
        op_vtg_first<TOPD, CH>(p, chan, absmask, signmask);
        return;
    }
    // the levels above the first g are path-independent f's: start from the virtual row of that stage
    const float* const vrow = reinterpret_cast<const float*>(p.arena + AR_VROW) + ((uint32_t)p.lane >> LCL) * VROW_FLOATS;
    const uint32_t l0 = gm ? (uint32_t)__builtin_ctz(gm) : 0u;
    switch (l0) {
#if LN_LOGN >= 12
    case 3: op_vtg_t<LOGN - 3, CH, float>(p, gm >> 3, vrow + vrow_off(LOGN - 3), absmask, signmask); break;
#endif
#if LN_LOGN >= 11
    case 2: op_vtg_t<LOGN - 2, CH, float>(p, gm >> 2, vrow + vrow_off(LOGN - 2), absmask, signmask); break;
    case 1: op_vtg_t<LOGN - 1, CH, float>(p, gm >> 1, vrow + vrow_off(LOGN - 1), absmask, signmask); break;
#endif
    default: op_vtg_t<LOGN, CH, RowT>(p, gm, chan, absmask, signmask); break;
    }
}
#endif
#if LN_LOGN >= 11 && LN_HILEAF
// RateOneDecoder / SpcDecoder (scl_avx_float.cpp:353-413, 498-621) on a block of 512 / 1024 LLRs (rates above 1/2 at N >= 2048), first
// half: the pass recomputes the block chunk by chunk; the sink collects the hard decisions -- a word per eight steps, written to
// the lane's own column of the leaf's output area, which fork_tail then rewrites in place with the chosen candidate's flips -- and
// runs the weak-LLR scan of findWeakLlrs on the magnitudes.  ws.sg[0] ends up as the XOR of all sign words (the parity SPC needs).
template <bool CH>
struct SinkScan {
    WeakScan* ws;
    uint32_t* dst;  // own column of the output area: word i at dst[i * 64]
    uint32_t word, par;
    __device__ __forceinline__ void pre(uint32_t) const {}
    template <int S, int PAR>
    __device__ __forceinline__ void take(uint32_t c)
    {
        const uint32_t e[4] = { rdu<S>(), rdu<S + 1>(), rdu<S + 2>(), rdu<S + 3>() };
        const uint32_t sh = 4u * (c & 7u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t a = e[i] & 0x7FFFFFFFu;
            if constexpr (CH) a = a < 0x42FE0000u ? a : 0x42FE0000u;  // (int8 decoder: abs(max(x, -127)))
            word |= (e[i] >> 31) << (sh + (uint32_t)i);
            if (c == 0u) ws->hv[i] = a;
            else ws->tail(a, 4u * c + (uint32_t)i);
        }
        if ((c & 7u) == 7u) {
            dst[(c >> 3) * 64u] = word;
            par ^= word;
            word = 0;
        }
    }
};
// the int8 decoder's repetition leaf at these sizes (scl_fip_char.cpp:497-547): exact sums of min(x, 0) and of x
struct SinkRep {
    float zs, xs;
    __device__ __forceinline__ void pre(uint32_t) const {}
    template <int S, int PAR>
    __device__ __forceinline__ void take(uint32_t)
    {
        const float a = rdf<S>(), b = rdf<S + 1>(), c = rdf<S + 2>(), d = rdf<S + 3>();
        zs += (min0(a) + min0(b)) + (min0(c) + min0(d));
        xs += (a + b) + (c + d);
    }
};
// the block of a leaf at stage T = 9 / 10, recomputed from the channel row through all LOGN - T levels (off bit l: the op at stage
// LOGN - l is a g) and handed to the sink chunk by chunk.  (Always from the channel row, not from a virtual row, and with two
// register sets: one instantiation of the pass per (stage, sink) -- these leaves run once or twice per codeword.)
template <int T, bool CH, typename SINK>
__device__ __forceinline__ void high_pass(Path& p, uint32_t off, const RowT* chan, uint32_t absmask, uint32_t signmask, SINK& sink)
{
    vtg_core<LOGN, T, CH, RowT, SINK, 2>(p, off & ((1u << (LOGN - T)) - 1u), chan, absmask, signmask, sink);
}
// RateZeroDecoder::decode scl_avx_float.cpp:316-337 on a block of 512 / 1024 LLRs (stage T = 9, 10: the low rates at N >= 2048).
// The block is never stored: the pass recomputes it from the channel row -- or from the highest virtual row it can use -- and
// the sink adds up min(x, 0) in the reference's order; the output bits are zeros.  off bit l: the op at stage LOGN - l is a g.
template <int T, bool CH>
__device__ __forceinline__ void leaf_r0_high(Path& p, uint32_t side, uint32_t off, const RowT* chan, uint32_t absmask, uint32_t signmask)
{
    SinkMin0 sink;
#pragma unroll
    for (int j = 0; j < 8; ++j) sink.acc[j] = 0.0f;
    high_pass<T, CH>(p, off, chan, absmask, signmask, sink);
    const float* a = sink.acc;
    p.metric = p.metric + (((((((a[0] + a[1]) + a[2]) + a[3]) + a[4]) + a[5]) + a[6]) + a[7]);
    uint32_t ln = (uint32_t)p.lane;
    asm volatile("" : "+v"(ln));
    uint32_t* o = p.arena + (side ? AR_BR(T) : AR_BL(T)) + ln;
#pragma unroll 1
    for (uint32_t i = 0; i < ar_bitwords(T); i += 4) {
        o[(i + 0) * 64] = 0u;
        o[(i + 1) * 64] = 0u;
        o[(i + 2) * 64] = 0u;
        o[(i + 3) * 64] = 0u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int T, bool CH>
__device__ __forceinline__ void leaf_scan_high(Path& p, uint32_t side, uint32_t off, const RowT* chan, uint32_t absmask, uint32_t signmask, WeakScan& ws)
{
    ws.init();
    uint32_t ln = (uint32_t)p.lane;
    asm volatile("" : "+v"(ln));
    SinkScan<CH> sink{ &ws, p.arena + (side ? AR_BR(T) : AR_BL(T)) + ln, 0u, 0u };
    high_pass<T, CH>(p, off, chan, absmask, signmask, sink);
    ws.sg[0] = sink.par;
    ws.sg[1] = ws.sg[2] = ws.sg[3] = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int T, bool CH>
__device__ __forceinline__ void leaf_rep_high(Path& p, uint32_t side, uint32_t off, const RowT* chan, uint32_t absmask, uint32_t signmask, const Stamps& sm)
{
    SinkRep sink{ 0.0f, 0.0f };
    high_pass<T, CH>(p, off, chan, absmask, signmask, sink);
    float cm[2];
    cm[0] = p.metric + sink.zs;
    cm[1] = p.metric - (sink.xs - sink.zs);
    uint32_t src, cand;
    select_fork<2>(p, cm, src, cand, sm);
    uint32_t ln = (uint32_t)p.lane;
    asm volatile("" : "+v"(ln));
    uint32_t* o = p.arena + (side ? AR_BR(T) : AR_BL(T)) + ln;
    const uint32_t w = cand ? 0xFFFFFFFFu : 0u;
#pragma unroll 1
    for (uint32_t i = 0; i < ar_bitwords(T); i += 4) {
        o[(i + 0) * 64] = w;
        o[(i + 1) * 64] = w;
        o[(i + 2) * 64] = w;
        o[(i + 3) * 64] = w;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#endif
#undef LN_VGS
#endif  // LN_LOGN != 10
#undef LN_WIDEN4
#if LN_PKFMA
#undef LN_FPAIR
#define LN_FPAIR "254"  // (walks again)
#endif

// stage 8 (arena column) -> stage 7 (LDS column): kind 0 = F, 1 = G (column of the slot owner), 2 = a + b (= G with zero
// bits).  Eight steps per batch: their sixteen 16-byte loads go out together into v[128 .. 191] (dead stages), the steps
// work on them in place as they arrive and store to the LDS.
#define LN_LD2(c) "global_load_dwordx4 v[%[d]+4*" #c ":%[d]+4*" #c "+3], %[oa], %[sb]\n\t" \
                  "global_load_dwordx4 v[%[t]+4*" #c ":%[t]+4*" #c "+3], %[ob], %[sb]\n\t" \
                  "v_add_u32 %[oa], 0x400, %[oa]\n\t" "v_add_u32 %[ob], 0x400, %[ob]\n\t"
#define LN_ST(c) "s_nop 0\n\tds_write_b128 %[la], v[%[d]+4*" #c ":%[d]+4*" #c "+3] offset:1024*" #c "\n\t"
// clamps of the int8 decoder on four results in place (see clamp_f / clamp_g)
#define LN_CF4(D_)                                                                                                   \
    "v_med3_f32 %[t0], |v[" D_ "]|, 1.0, %[c]\n\t"   "v_med3_f32 %[t1], |v[" D_ "+1]|, 1.0, %[c]\n\t"                  \
    "v_med3_f32 %[t2], |v[" D_ "+2]|, 1.0, %[c]\n\t" "v_med3_f32 %[t3], |v[" D_ "+3]|, 1.0, %[c]\n\t"                  \
    "v_bfi_b32 v[" D_ "], %[am], %[t0], v[" D_ "]\n\t"     "v_bfi_b32 v[" D_ "+1], %[am], %[t1], v[" D_ "+1]\n\t"       \
    "v_bfi_b32 v[" D_ "+2], %[am], %[t2], v[" D_ "+2]\n\t" "v_bfi_b32 v[" D_ "+3], %[am], %[t3], v[" D_ "+3]\n\t"
#define LN_CG4(D_)                                                                                                   \
    "v_med3_f32 v[" D_ "], v[" D_ "], %[m], %[c]\n\t"     "v_med3_f32 v[" D_ "+1], v[" D_ "+1], %[m], %[c]\n\t"          \
    "v_med3_f32 v[" D_ "+2], v[" D_ "+2], %[m], %[c]\n\t" "v_med3_f32 v[" D_ "+3], v[" D_ "+3], %[m], %[c]\n\t"
template <bool ISF, bool CH = false>
__device__ __forceinline__ void fg8_batch(uint32_t& oa, uint32_t& ob, const uint32_t* s8base, uint32_t la, uint32_t w, uint32_t absmask,
                                          uint32_t signmask, float c127 = 0.0f, float m128 = 0.0f)
{
    uint32_t t0, t1, t2, t3;
    if constexpr (ISF && CH) {
        asm volatile(LN_LD2(0) LN_LD2(1) LN_LD2(2) LN_LD2(3) LN_LD2(4) LN_LD2(5) LN_LD2(6) LN_LD2(7)
                     "s_waitcnt vmcnt(14)\n\t" LN_F4T("%[d]", "%[d]", "%[t]") LN_CF4("%[d]") LN_ST(0)
                     "s_waitcnt vmcnt(12)\n\t" LN_F4T("%[d]+4", "%[d]+4", "%[t]+4") LN_CF4("%[d]+4") LN_ST(1)
                     "s_waitcnt vmcnt(10)\n\t" LN_F4T("%[d]+8", "%[d]+8", "%[t]+8") LN_CF4("%[d]+8") LN_ST(2)
                     "s_waitcnt vmcnt(8)\n\t" LN_F4T("%[d]+12", "%[d]+12", "%[t]+12") LN_CF4("%[d]+12") LN_ST(3)
                     "s_waitcnt vmcnt(6)\n\t" LN_F4T("%[d]+16", "%[d]+16", "%[t]+16") LN_CF4("%[d]+16") LN_ST(4)
                     "s_waitcnt vmcnt(4)\n\t" LN_F4T("%[d]+20", "%[d]+20", "%[t]+20") LN_CF4("%[d]+20") LN_ST(5)
                     "s_waitcnt vmcnt(2)\n\t" LN_F4T("%[d]+24", "%[d]+24", "%[t]+24") LN_CF4("%[d]+24") LN_ST(6)
                     "s_waitcnt vmcnt(0)\n\t" LN_F4T("%[d]+28", "%[d]+28", "%[t]+28") LN_CF4("%[d]+28") LN_ST(7)
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [oa] "+v"(oa), [ob] "+v"(ob)
                     : [d] "n"(128), [t] "n"(160), [sb] "s"(s8base), [la] "v"(la), [am] "s"(absmask), [c] "s"(c127)
                     : "memory");
    } else if constexpr (CH) {
        asm volatile(LN_LD2(0) LN_LD2(1) LN_LD2(2) LN_LD2(3) LN_LD2(4) LN_LD2(5) LN_LD2(6) LN_LD2(7)
                     "s_waitcnt vmcnt(14)\n\t" LN_G4T("%[d]", "%[t]", 31, 30, 29, 28) LN_CG4("%[d]") LN_ST(0)
                     "s_waitcnt vmcnt(12)\n\t" LN_G4T("%[d]+4", "%[t]+4", 27, 26, 25, 24) LN_CG4("%[d]+4") LN_ST(1)
                     "s_waitcnt vmcnt(10)\n\t" LN_G4T("%[d]+8", "%[t]+8", 23, 22, 21, 20) LN_CG4("%[d]+8") LN_ST(2)
                     "s_waitcnt vmcnt(8)\n\t" LN_G4T("%[d]+12", "%[t]+12", 19, 18, 17, 16) LN_CG4("%[d]+12") LN_ST(3)
                     "s_waitcnt vmcnt(6)\n\t" LN_G4T("%[d]+16", "%[t]+16", 15, 14, 13, 12) LN_CG4("%[d]+16") LN_ST(4)
                     "s_waitcnt vmcnt(4)\n\t" LN_G4T("%[d]+20", "%[t]+20", 11, 10, 9, 8) LN_CG4("%[d]+20") LN_ST(5)
                     "s_waitcnt vmcnt(2)\n\t" LN_G4T("%[d]+24", "%[t]+24", 7, 6, 5, 4) LN_CG4("%[d]+24") LN_ST(6)
                     "s_waitcnt vmcnt(0)\n\t" LN_G4T("%[d]+28", "%[t]+28", 3, 2, 1, 0) LN_CG4("%[d]+28") LN_ST(7)
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [oa] "+v"(oa), [ob] "+v"(ob)
                     : [d] "n"(128), [t] "n"(160), [sb] "s"(s8base), [la] "v"(la), [w] "v"(w), [sm] "s"(signmask), [c] "s"(c127), [m] "v"(m128)
                     : "memory");
    } else if constexpr (ISF) {
        asm volatile(LN_LD2(0) LN_LD2(1) LN_LD2(2) LN_LD2(3) LN_LD2(4) LN_LD2(5) LN_LD2(6) LN_LD2(7)
                     "s_waitcnt vmcnt(14)\n\t" LN_F4T("%[d]", "%[d]", "%[t]") LN_ST(0)
                     "s_waitcnt vmcnt(12)\n\t" LN_F4T("%[d]+4", "%[d]+4", "%[t]+4") LN_ST(1)
                     "s_waitcnt vmcnt(10)\n\t" LN_F4T("%[d]+8", "%[d]+8", "%[t]+8") LN_ST(2)
                     "s_waitcnt vmcnt(8)\n\t" LN_F4T("%[d]+12", "%[d]+12", "%[t]+12") LN_ST(3)
                     "s_waitcnt vmcnt(6)\n\t" LN_F4T("%[d]+16", "%[d]+16", "%[t]+16") LN_ST(4)
                     "s_waitcnt vmcnt(4)\n\t" LN_F4T("%[d]+20", "%[d]+20", "%[t]+20") LN_ST(5)
                     "s_waitcnt vmcnt(2)\n\t" LN_F4T("%[d]+24", "%[d]+24", "%[t]+24") LN_ST(6)
                     "s_waitcnt vmcnt(0)\n\t" LN_F4T("%[d]+28", "%[d]+28", "%[t]+28") LN_ST(7)
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [oa] "+v"(oa), [ob] "+v"(ob)
                     : [d] "n"(128), [t] "n"(160), [sb] "s"(s8base), [la] "v"(la), [am] "s"(absmask)
                     : "memory");
    } else {
        asm volatile(LN_LD2(0) LN_LD2(1) LN_LD2(2) LN_LD2(3) LN_LD2(4) LN_LD2(5) LN_LD2(6) LN_LD2(7)
                     "s_waitcnt vmcnt(14)\n\t" LN_G4T("%[d]", "%[t]", 31, 30, 29, 28) LN_ST(0)
                     "s_waitcnt vmcnt(12)\n\t" LN_G4T("%[d]+4", "%[t]+4", 27, 26, 25, 24) LN_ST(1)
                     "s_waitcnt vmcnt(10)\n\t" LN_G4T("%[d]+8", "%[t]+8", 23, 22, 21, 20) LN_ST(2)
                     "s_waitcnt vmcnt(8)\n\t" LN_G4T("%[d]+12", "%[t]+12", 19, 18, 17, 16) LN_ST(3)
                     "s_waitcnt vmcnt(6)\n\t" LN_G4T("%[d]+16", "%[t]+16", 15, 14, 13, 12) LN_ST(4)
                     "s_waitcnt vmcnt(4)\n\t" LN_G4T("%[d]+20", "%[t]+20", 11, 10, 9, 8) LN_ST(5)
                     "s_waitcnt vmcnt(2)\n\t" LN_G4T("%[d]+24", "%[t]+24", 7, 6, 5, 4) LN_ST(6)
                     "s_waitcnt vmcnt(0)\n\t" LN_G4T("%[d]+28", "%[t]+28", 3, 2, 1, 0) LN_ST(7)
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [oa] "+v"(oa), [ob] "+v"(ob)
                     : [d] "n"(128), [t] "n"(160), [sb] "s"(s8base), [la] "v"(la), [w] "v"(w), [sm] "s"(signmask)
                     : "memory");
    }
}
#undef LN_LD2
#undef LN_ST
#undef LN_CF4
#undef LN_CG4
template <bool CH>
__device__ __forceinline__ void op_fg8(Path& p, uint32_t kind, uint32_t absmask, uint32_t signmask)
{
    const uint32_t col = kind == 1u ? p.owner(8) : (uint32_t)p.lane;
    const uint32_t* s8base = p.arena + AR_S8;
    uint32_t oa = col * 16u, ob = col * 16u + 32768u;  // byte offsets of chunk j and chunk j + 32 of the column
    const uint32_t la = (uint32_t)(uintptr_t)p.lds + (uint32_t)p.lane * 16u;
    const uint32_t w0 = rdu<RBL7>(), w1 = rdu<RBL7 + 1>(), w2 = rdu<RBL7 + 2>(), w3 = rdu<RBL7 + 3>();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
    for (uint32_t b = 0; b < 4; ++b) {
        const uint32_t w = kind != 1u ? 0u : (b == 0 ? w0 : (b == 1 ? w1 : (b == 2 ? w2 : w3)));
        if (kind == 0u) fg8_batch<true, CH>(oa, ob, s8base, la + b * 8192u, w, absmask, signmask, p.c127, p.m128);
        else fg8_batch<false, CH>(oa, ob, s8base, la + b * 8192u, w, absmask, signmask, p.c127, p.m128);
    }
    if (kind != 0u) p.set_field(8);
    p.set_field(7);
}

// stage 7 (LDS column) -> stage 6 (registers)
template <int KIND, bool CH>
__device__ __forceinline__ void op_fg7(Path& p, uint32_t absmask, uint32_t signmask)
{
    const uint32_t col = KIND == 1 ? p.owner(7) : (uint32_t)p.lane;
    const uint32_t addr = (uint32_t)(uintptr_t)p.lds + col * 16u;
    if constexpr (KIND == 0) {
        f7_round<0>(addr, absmask);
        f7_round<1>(addr, absmask);
    } else if constexpr (KIND == 1) {
        const uint32_t w0 = rdu<RBL6>(), w1 = rdu<RBL6 + 1>();
        g7_round<0, false>(addr, w0, signmask);
        g7_round<1, false>(addr, w1, signmask);
    } else {  // a + b = g with zero bits
        g7_round<0, false>(addr, 0, signmask);
        g7_round<1, false>(addr, 0, signmask);
    }
    if constexpr (CH) {
        if constexpr (KIND == 0) clamp_f<RS(6), 64>(p.c127, absmask);
        else clamp_g<RS(6), 64>(p.c127, p.m128);
    }
    if (KIND != 0) p.set_field(7);
    p.set_field(6);
}

template <int KIND, int T, bool CH>
__device__ __forceinline__ void op_fg_reg(Path& p, uint32_t absmask, uint32_t signmask, bool twice = false)
{
    if constexpr (KIND == 0) f_reg<T>(absmask);
    else if constexpr (KIND == 1) {
        g_reg<T>(p.owner(T) << 2, signmask);
        if (twice) g_reg<T>(p.owner(T) << 2, signmask);  // (diagnostics: the op is idempotent until its slot field is set)
    }
    else g0_reg<T>();
    if constexpr (CH) {
        if constexpr (KIND == 0) clamp_f<RS(T - 1), (1 << (T - 1))>(p.c127, absmask);
        else clamp_g<RS(T - 1), (1 << (T - 1))>(p.c127, p.m128);
    }
    if (KIND != 0) p.set_field(T);
    p.set_field(T - 1);
}

}  // namespace lane

// =====================================================================================================================
//  kernel
// =====================================================================================================================
// CH: the int8 list decoder SclFipChar.  LN_I8ROWS: the channel rows are int8 and are widened in the top-of-tree pass (always
// for CH; the float decoder called with int8 rows -- decode_vector(const char*) -- has its own translation units)
#if !defined(LN_OPAQUE_LANE) && !LN_GEO_DEFAULT && (LN_LOGN == 10 || (LN_LOGN >= 11 && !LN_HILEAF && !LN_CHAR))
// (N = 1024 outside the default geometry: the arena and row pointers are enough -- no scratch, + 2.4 %, A/B on one box; the float kernels
//  of N = 2048 / 4096 without the leaves above stage 8: + 0.3 ... 2 % at every number of lanes, no scratch either)
#define LN_OPAQUE_LANE 3
#endif
template <bool DIAG, bool CH = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(lane::KV)))
void scl_lane_kernel(const DecodeParams P)
{
    using namespace lane;
    // (dynamic: with a static 40 KB the compiler derives one wave per SIMD from the LDS size and then ignores the VGPR cap)
    extern __shared__ uint32_t lds[];
#if LN_LOGN == 10
    asm volatile("" ::: "v255", "a0");  // (sizes the register file: the hand-allocated registers reach up to v253; a0: see KV)
#else
    asm volatile("" ::: "v255", "a0", "a255");  // (... and the top-of-tree pass of the other block lengths lands its bit words in a[240 .. 254])
#endif
    const int lane = (int)threadIdx.x;
    const uint32_t nrows = P.rows_dev ? *P.rows_dev : P.rows;
    const uint32_t ngroups = (nrows + (uint32_t)CPW - 1u) / (uint32_t)CPW;
    const uint32_t absmask = 0x7FFFFFFFu, signmask = 0x80000000u;
    const_u64_ptr prog = (const_u64_ptr)(const void*)P.prog;
    const uint32_t N = P.N;

    Path p;
    p.lane = lane;
    p.q = lane & (LC - 1);
    p.g0 = lane & ~(LC - 1);
    p.self4 = (uint32_t)lane << 2;
    p.lds = lds;
    p.arena = reinterpret_cast<uint32_t*>(P.arena) + (size_t)blockIdx.x * AR_WORDS;
    p.L = P.L;
    p.c127 = 127.0f;
    p.m128 = -128.0f;
    asm volatile("" : "+v"(p.m128));  // (kept in a VGPR: the clamp's scalar operand is the 127)

    for (uint32_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const uint32_t ci = grp * (uint32_t)CPW + ((uint32_t)lane >> LCL);
        const bool live = ci < nrows;
        const uint32_t cj = live ? ci : grp * (uint32_t)CPW;
        const uint32_t row = P.rowidx ? P.rowidx[cj] : cj;
        const RowT* chan = reinterpret_cast<const RowT*>(P.llr) + (size_t)row * N;
        p.metric = 0.0f;
        p.pl = p.q == 0 ? 0u : PL_DEAD;
        p.P = 1;
        p.reset_slots();  // every field = own lane
        uint32_t trace_n = 0;

        uint64_t nraw = prog[0];
        for (uint32_t pc = 0;; ++pc) {
            const uint64_t raw = nraw;
            nraw = prog[pc + 1];  // (fetched one instruction ahead; the host appends a spare word behind OP_END)
#if (!LN_GEO_DEFAULT && !(defined(LN_OPAQUE_LANE) && LN_OPAQUE_LANE == 0)) || (defined(LN_OPAQUE_LANE) && LN_OPAQUE_LANE > 0)
            // (nothing derived from the lane number is carried around the interpreter loop: with the larger arenas of the other
            //  geometries the compiler otherwise keeps -- and spills to scratch -- dozens of precomputed 64-bit column addresses)
#if defined(LN_OPAQUE_LANE) && LN_OPAQUE_LANE == 1
            asm volatile("" : "+v"(p.lane));
#elif defined(LN_OPAQUE_LANE) && LN_OPAQUE_LANE == 3
            asm volatile("" : "+s"(p.arena), "+v"(chan));
#elif defined(LN_OPAQUE_LANE) && LN_OPAQUE_LANE == 4
            asm volatile("" : "+s"(p.arena));
#else
            asm volatile("" : "+v"(p.lane), "+v"(p.q), "+v"(p.g0), "+v"(p.self4));
#endif
#endif
            const uint32_t op = (uint32_t)raw & 0xFFu, st = ((uint32_t)raw >> 8) & 0xFFu, side = ((uint32_t)raw >> 16) & 0xFFu;
            const uint32_t off = (uint32_t)(raw >> 32);
            if (op == OP_END) break;
            uint64_t t_start = 0;
            if (DIAG && P.cycles && grp == 0) t_start = __builtin_amdgcn_s_memtime();
            Stamps sm{ nullptr, t_start, DIAG ? P.ablate : 0u };
            if (DIAG && P.cycles && grp == 0 && (P.ablate & 256u)) sm.sub = P.cycles + P.prog_len + 4 * pc;  // (PCD_ABLATE=256: phase stamps)
#ifdef LN_STAMP_VT
            p.vt_sub = sm.sub;
#endif
            bool is_leaf = false;
            uint32_t fkind = 0;  // forking leaf: 1 RateOne, 2 SPC
            uint32_t lk = 0, lst = st, lside = side;  // leaf to run behind the switch: kind, stage, side
            WeakScan ws;
            uint32_t opx = op;
#pragma unroll 1
          for (uint32_t again = 0; again < ((DIAG && (P.ablate & 32768u)) ? 2u : 1u); ++again, opx = 0xFFu)  // (PCD_ABLATE=32768: every dispatch twice, the second one of nothing)
            switch (opx) {
            case LN_VT:
#if LN_LOGN == 10
                if (!(DIAG && (P.ablate & 1u))) op_vt<CH>(p, off, chan, absmask, signmask);
                if (DIAG && (P.ablate & 1024u)) op_vt<CH>(p, off, chan, absmask, signmask);  // (idempotent: costed by doing it twice)
#elif LN_LOGN == 7
                op_copy7<CH>(p, chan);
#elif LN_LOGN == 6
                op_copy6<CH>(p, chan);
#else
                if (!(DIAG && (P.ablate & 1u))) op_vtg<CH>(p, off, chan, absmask, signmask);
#endif
                break;
            case LN_WALK: {
                if (DIAG && (P.ablate & 2u)) break;
                // everything between two leaves in one dispatch: COMBINEs up the right spine (stages c0 .. c1, the last one
                // with this instruction's side, the others are right children), the G of the node reached, Fs down to the next leaf
                const uint32_t c0 = off & 15u, c1 = (off >> 4) & 15u, gs = (off >> 8) & 15u, gk = (off >> 12) & 3u;
                const uint32_t f1 = (off >> 16) & 15u, f0 = (off >> 20) & 15u, r0s = (off >> 24) & 15u, r0side = (off >> 28) & 1u;
                // (plain loops over the stages with a flat switch each: fall-through switches make the compiler thread
                //  flag variables through a structured control flow graph -- dozens of scalar instructions per step)
                if (c0) {
#pragma unroll 1
                  for (uint32_t rep = 0; rep < ((DIAG && (P.ablate & 2048u)) ? 2u : 1u); ++rep)
#pragma unroll 1
                    for (uint32_t t = c0; t <= c1; ++t) {
                        const uint32_t sd = t == c1 ? side : 1u;
                        switch (t) {
                        case 2: op_combine<2>(p, sd); break;
                        case 3: op_combine<3>(p, sd); break;
                        case 4: op_combine<4>(p, sd); break;
                        case 5: op_combine<5>(p, sd); break;
                        case 6: op_combine<6>(p, sd); break;
                        case 7: op_combine<7>(p, sd); break;
                        case 8: op_combine<8>(p, sd); break;
#if LN_LOGN == 10
                        default: op_combine<9>(p, sd); break;
#else
#if LN_LOGN >= 11
                        case 9: op_combine<9>(p, sd); break;
#endif
#if LN_LOGN >= 12
                        case 10: op_combine<10>(p, sd); break;
#endif
                        default: if constexpr (LOGN >= 10) op_combine<(LOGN >= 10 ? LOGN - 1 : 9)>(p, sd); break;
#endif
                        }
                    }
                    if (c1 >= 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                sm.mark(1);
                // (ONE flat switch per step: nested if / switch ladders make the compiler thread flag registers through a structured
                //  control flow graph -- dozens of scalar instructions per step; found on the SC pair kernel, where it was worth 16 %)
                switch ((gk << 4) | gs) {
                case 0x18: op_fg8<CH>(p, 1, absmask, signmask); break;
                case 0x17: op_fg7<1, CH>(p, absmask, signmask); break;
                case 0x16: op_fg_reg<1, 6, CH>(p, absmask, signmask, DIAG && (P.ablate & 8192u)); break;
                case 0x15: op_fg_reg<1, 5, CH>(p, absmask, signmask, DIAG && (P.ablate & 8192u)); break;
                case 0x14: op_fg_reg<1, 4, CH>(p, absmask, signmask, DIAG && (P.ablate & 8192u)); break;
                case 0x13: op_fg_reg<1, 3, CH>(p, absmask, signmask, DIAG && (P.ablate & 8192u)); break;
                case 0x12: op_fg_reg<1, 2, CH>(p, absmask, signmask, DIAG && (P.ablate & 8192u)); break;
                case 0x28: op_fg8<CH>(p, 2, absmask, signmask); break;
                case 0x27: op_fg7<2, CH>(p, absmask, signmask); break;
                case 0x26: op_fg_reg<2, 6, CH>(p, absmask, signmask); break;
                case 0x25: op_fg_reg<2, 5, CH>(p, absmask, signmask); break;
                case 0x24: op_fg_reg<2, 4, CH>(p, absmask, signmask); break;
                case 0x23: op_fg_reg<2, 3, CH>(p, absmask, signmask); break;
                case 0x22: op_fg_reg<2, 2, CH>(p, absmask, signmask); break;
                default: break;
                }
                sm.mark(2);
                if (f1) {
#pragma unroll 1
                  for (uint32_t rep = 0; rep < ((DIAG && (P.ablate & 512u)) ? 2u : 1u); ++rep)
#pragma unroll 1
                    for (uint32_t t = f1; t >= f0; --t) {
                        switch (t) {
                        case 8: op_fg8<CH>(p, 0, absmask, signmask); break;
                        case 7: op_fg7<0, CH>(p, absmask, signmask); break;
                        case 6: op_fg_reg<0, 6, CH>(p, absmask, signmask); break;
                        case 5: op_fg_reg<0, 5, CH>(p, absmask, signmask); break;
                        case 4: op_fg_reg<0, 4, CH>(p, absmask, signmask); break;
                        case 3: op_fg_reg<0, 3, CH>(p, absmask, signmask); break;
                        default: op_fg_reg<0, 2, CH>(p, absmask, signmask); break;
                        }
                    }
                }
                lk = (off >> 29) & 7u;  // the leaf the descent ends at rides along: 1 all-frozen, 2 RateOne, 3 SPC, 4 repetition
                lst = r0s;
                lside = r0side;
                break;
            }
#if LN_LOGN >= 11 && LN_HILEAF
            case LN_VROWS: op_vtg_first<TOPD, CH, false>(p, chan, absmask, signmask); break;
#endif
            case LN_R0: lk = 1; break;
            case LN_R1: lk = 2; break;
            case LN_SPC: lk = 3; break;
            case LN_REP: lk = 4; break;
            default: break;
            }
            if (lk) {  // the leaf of this instruction (its own op, or the end of a walk): one instance of each leaf routine
                is_leaf = true;
                // class of the leaf: 1 all-frozen, 2 repetition, 3 RateOne / SPC (one scan serves both, the tails differ)
                const uint32_t cls = lk == 1u ? 1u : (lk == 4u ? 2u : 3u);
                fkind = cls == 3u ? (lk == 3u ? 2u : 1u) : 0u;
                uint32_t lcode = (cls << 4) | lst;
                if (DIAG && (((P.ablate & 32u) && cls == 1u) || ((P.ablate & 16u) && cls == 2u) || ((P.ablate & 8u) && cls == 3u))) {
                    lcode = 0;
                    fkind = 0;
                }
#pragma unroll 1
                for (uint32_t rep = 0; rep < ((DIAG && (P.ablate & 4096u) && cls == 3u) ? 2u : 1u); ++rep)
                    switch (lcode) {  // (flat: see the G step)
#if LN_LOGN >= 11 && LN_HILEAF
                    case 0x19: leaf_r0_high<9, CH>(p, lside, off, chan, absmask, signmask); break;
                    case 0x1A: leaf_r0_high<10, CH>(p, lside, off, chan, absmask, signmask); break;
                    case 0x29: if constexpr (CH) leaf_rep_high<9, CH>(p, lside, off, chan, absmask, signmask, sm); break;
                    case 0x2A: if constexpr (CH) leaf_rep_high<10, CH>(p, lside, off, chan, absmask, signmask, sm); break;
                    case 0x39: leaf_scan_high<9, CH>(p, lside, off, chan, absmask, signmask, ws); break;
                    case 0x3A: leaf_scan_high<10, CH>(p, lside, off, chan, absmask, signmask, ws); break;
#endif
                    case 0x18: leaf_r0<8>(p, lside); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    case 0x17: leaf_r0<7>(p, lside); break;
                    case 0x16: leaf_r0<6>(p, lside); break;
                    case 0x15: leaf_r0<5>(p, lside); break;
                    case 0x14: leaf_r0<4>(p, lside); break;
                    case 0x13: leaf_r0<3>(p, lside); break;
                    case 0x12: leaf_r0<2>(p, lside); break;
                    case 0x11: leaf_r0<1>(p, lside); break;
                    case 0x28: if constexpr (CH) leaf_rep_any<8>(p, lside, sm); break;  // (int8 decoder: repetition leaves at every size)
                    case 0x27: if constexpr (CH) leaf_rep_any<7>(p, lside, sm); break;
                    case 0x26: if constexpr (CH) leaf_rep_any<6>(p, lside, sm); break;
                    case 0x25: if constexpr (CH) leaf_rep_any<5>(p, lside, sm); break;
                    case 0x24: if constexpr (CH) leaf_rep_any<4>(p, lside, sm); break;
                    case 0x23: if constexpr (CH) leaf_rep_any<3>(p, lside, sm); break;
                    case 0x22:
                        if constexpr (CH) leaf_rep_any<2>(p, lside, sm);
                        else leaf_rep<2>(p, lside, sm);
                        break;
                    case 0x21:
                        if constexpr (CH) leaf_rep_any<1>(p, lside, sm);
                        else leaf_rep<1>(p, lside, sm);
                        break;
#if LN_BIG
                    case 0x38: leaf_scan<8, CH>(p, ws); break;
#endif
                    case 0x37: leaf_scan<7, CH>(p, ws); break;
                    case 0x36: leaf_scan<6, CH>(p, ws); break;
                    case 0x35: leaf_scan<5, CH>(p, ws); break;
                    case 0x34: leaf_scan<4, CH>(p, ws); break;
                    case 0x33: leaf_scan<3, CH>(p, ws); break;
                    case 0x32: leaf_scan<2, CH>(p, ws); break;
                    case 0x31: leaf_scan<1, CH>(p, ws); break;
                    default: break;
                    }
            }
            sm.mark(0);
            if (DIAG && (P.ablate & 4u)) fkind = 0;
            if (fkind == 2u) fork_tail<true>(p, lst, lside, ws, sm);
            else if (fkind == 1u) fork_tail<false>(p, lst, lside, ws, sm);
            if (DIAG && P.cycles && grp == 0) {
                const uint64_t t_end = __builtin_amdgcn_s_memtime();
                if (lane == 0) P.cycles[pc] = (uint32_t)(t_end - t_start);
            }
            if (DIAG && P.trace && grp == 0 && is_leaf) {
                if (trace_n < P.trace_cap && lane < LC) {
                    float* t = P.trace + (size_t)trace_n * (1 + P.L);
                    if (lane == 0) t[0] = (float)p.P;
                    if ((uint32_t)lane >= p.P && (uint32_t)lane < P.L) t[1 + lane] = 0.0f;
                    if (p.pl < p.P) t[1 + p.pl] = p.metric;
                }
                ++trace_n;
            }
        }

        if (DIAG && (P.ablate & 64u)) continue;
        // ---- SclAvxFloat::extractBestPath :711-750 -----------------------------------------------------------------
#if LN_LOGN == 10
        // code word of this path: [left ^ right | right] of the root's children (16 + 16 words from the arena)
        uint32_t xw[32];
        {
            const uint32_t* lp = p.arena + AR_BL9 + p.owner(10);
            const uint32_t* rp = p.arena + AR_BR9 + p.lane;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t l = lp[i * 64], r = rp[i * 64];
                xw[i] = l ^ r;
                xw[16 + i] = r;
            }
        }
        if (!P.systematic) {  // re-encode (:731-746): u = x G, butterflies inside the words, then between them
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                uint32_t v = xw[i];
                v ^= (v >> 1) & 0x55555555u;
                v ^= (v >> 2) & 0x33333333u;
                v ^= (v >> 4) & 0x0F0F0F0Fu;
                v ^= (v >> 8) & 0x00FF00FFu;
                v ^= (v >> 16) & 0x0000FFFFu;
                xw[i] = v;
            }
#pragma unroll
            for (int hw = 1; hw < 32; hw <<= 1) {
#pragma unroll
                for (int i = 0; i < 32; ++i)
                    if (!(i & hw)) xw[i] ^= xw[i + hw];
            }
        }
        // information bits of this path, packed like the output (bit t of the information word = bit 7 - t % 8 of byte
        // t / 8): per 32 code bits a log-step bit compress with host-made move masks (program.cpp, deposit_table), the
        // pieces strung together in a 64-bit shift register; finished words go to the LDS (stage 7 is dead now) at
        // [word * 64 + lane], byte-reversed bits first.
        uint32_t* il = lds + LDS_S7;
        {
            const uint32_t* ext = P.ext;
            uint64_t acc = 0;
#pragma unroll
            for (int w = 0; w < 32; ++w) {
                const uint32_t mask = ext[8 * w], base = ext[8 * w + 1];
                uint32_t x = xw[w] & mask;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const uint32_t t = x & ext[8 * w + 2 + i];
                    x = (x ^ t) | (t >> (1u << i));
                }
                acc |= (uint64_t)x << (base & 31u);
                const uint32_t end = base + (uint32_t)__builtin_popcount(mask);
                if ((end >> 5) != (base >> 5)) {  // (wave uniform) a word is complete
                    il[(base >> 5) * 64 + lane] = __builtin_bswap32(__brev((uint32_t)acc));
                    acc >>= 32;
                }
            }
            if (P.K & 31u) il[(P.K >> 5) * 64 + lane] = __builtin_bswap32(__brev((uint32_t)acc));
        }
#else
        // code word of this path: [left ^ right | right] of the root's children (arena), XW words at a time; the information
        // bits are extracted word by word as above, the shift register runs through the rounds
        uint32_t* il = lds + LDS_S7;
        {
            constexpr uint32_t NWT = (1u << LOGN) / 32u;   // words of the code word
            constexpr uint32_t XW = NWT < 32u ? NWT : (LOGN >= 12 ? 16u : 32u);  // words per round (N = 4096: 16, for the registers' sake)
            constexpr uint32_t NCK = NWT / XW, NWH = NWT / 2u;
            uint32_t ln = (uint32_t)lane;
            asm volatile("" : "+v"(ln));  // (opaque: see op_combine)
            const uint32_t* lp0 = p.arena + AR_BL(LOGN >= 9 ? LOGN - 1 : 8) + (LOGN >= 9 ? p.owner(LOGN) : 0u);
            const uint32_t* rp0 = p.arena + AR_BR(LOGN >= 9 ? LOGN - 1 : 8) + ln;
            const uint32_t* ext = P.ext;
            uint64_t acc = 0;
#pragma unroll 1
            for (uint32_t ck = 0; ck < NCK; ++ck) {
                const uint32_t *lp = lp0, *rp = rp0;
                asm volatile("" : "+v"(lp), "+v"(rp));
                uint32_t xw[XW];
                auto load_round = [&](uint32_t cc, bool first) {  // words [XW cc, XW cc + XW) of x, XORed into xw unless first
                    const bool lefthalf = XW * cc < NWH;         // (a round never straddles the halves unless it is the only one)
#pragma unroll
                    for (uint32_t i = 0; i < XW; ++i) {
                        const uint32_t wi = XW * cc + i;
                        uint32_t x;
                        if constexpr (LOGN <= 8) x = 0;  // (N = 256 / 128: the root's children are stage-7 / 6 blocks -- registers, below)
                        else if constexpr (NCK == 1) x = i < NWH ? (lp[i * 64] ^ rp[i * 64]) : rp[(i - NWH) * 64];
                        else x = lefthalf ? (lp[wi * 64] ^ rp[wi * 64]) : rp[(wi - NWH) * 64];
                        xw[i] = first ? x : (xw[i] ^ x);
                        if ((i & 7u) == 7u) asm volatile("" ::: "memory");  // (eight words at a time: not 64 loads in flight and their registers)
                    }
                };
                load_round(ck, true);
                if constexpr (LOGN == 8) {  // left child's words from the lane that holds them (as COMBINE at stage 8 would), right child's are own
                    const uint32_t ho = p.owner(8) << 2;
                    const uint32_t l0 = rd_lane<RBL7>(ho), l1 = rd_lane<RBL7 + 1>(ho), l2 = rd_lane<RBL7 + 2>(ho), l3 = rd_lane<RBL7 + 3>(ho);
                    const uint32_t r0 = rdu<RBR7>(), r1 = rdu<RBR7 + 1>(), r2 = rdu<RBR7 + 2>(), r3 = rdu<RBR7 + 3>();
                    xw[0] = l0 ^ r0; xw[1] = l1 ^ r1; xw[2] = l2 ^ r2; xw[3] = l3 ^ r3;
                    xw[4 % XW] = r0; xw[5 % XW] = r1; xw[6 % XW] = r2; xw[7 % XW] = r3;
                }
                if constexpr (LOGN == 6) {
                    const uint32_t l0 = rd_lane<RBL0 + 5>(p.owner(6) << 2), r0 = rdu<RBR0 + 5>();
                    xw[0] = l0 ^ r0; xw[1 % XW] = r0;
                }
                if constexpr (LOGN == 7) {
                    const uint32_t ho = p.owner(7) << 2;
                    const uint32_t l0 = rd_lane<RBL6>(ho), l1 = rd_lane<RBL6 + 1>(ho), r0 = rdu<RBR6>(), r1 = rdu<RBR6 + 1>();
                    xw[0] = l0 ^ r0; xw[1] = l1 ^ r1; xw[2 % XW] = r0; xw[3 % XW] = r1;
                }
                if (!P.systematic) {  // re-encode (:731-746): u = x G -- between the rounds, inside the words, between the words
#pragma unroll 1
                    for (uint32_t cc = ck + 1; cc < NCK; ++cc)
                        if ((cc & ck) == ck) load_round(cc, false);
#pragma unroll
                    for (uint32_t i = 0; i < XW; ++i) {
                        uint32_t v = xw[i];
                        v ^= (v >> 1) & 0x55555555u;
                        v ^= (v >> 2) & 0x33333333u;
                        v ^= (v >> 4) & 0x0F0F0F0Fu;
                        v ^= (v >> 8) & 0x00FF00FFu;
                        v ^= (v >> 16) & 0x0000FFFFu;
                        xw[i] = v;
                    }
#pragma unroll
                    for (uint32_t hw = 1; hw < XW; hw <<= 1) {
#pragma unroll
                        for (uint32_t i = 0; i < XW; ++i)
                            if (!(i & hw)) xw[i] ^= xw[i + hw];
                    }
                }
                const uint32_t* ex = ext + 8u * XW * ck;
#pragma unroll
                for (uint32_t w = 0; w < XW; ++w) {
                    const uint32_t mask = ex[8 * w], base = ex[8 * w + 1];
                    uint32_t x = xw[w] & mask;
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        const uint32_t t = x & ex[8 * w + 2 + i];
                        x = (x ^ t) | (t >> (1u << i));
                    }
                    acc |= (uint64_t)x << (base & 31u);
                    const uint32_t end = base + (uint32_t)__builtin_popcount(mask);
                    if ((end >> 5) != (base >> 5)) {  // (wave uniform) a word is complete
                        il[(base >> 5) * 64 + lane] = __builtin_bswap32(__brev((uint32_t)acc));
                        acc >>= 32;
                    }
                }
            }
            if (P.K & 31u) il[(P.K >> 5) * 64 + lane] = __builtin_bswap32(__brev((uint32_t)acc));
        }
#endif
        // detector verdict of every path: XOR of the syndrome words of its output bytes
        const bool alive = p.pl < p.P;
        uint32_t syn = 0;
        if (P.det_on) {
            const uint32_t nwords = (P.nb + 3u) >> 2;
#pragma unroll 8
            for (uint32_t j = 0; j < nwords; ++j) {  // (eight words = 32 independent table loads per wait)
                const uint32_t word = il[j * 64 + lane];
                const uint32_t* tb = P.syn_bytes + j * 1024u;
                syn ^= tb[word & 255u];
                if (4 * j + 1 < P.nb) syn ^= tb[256 + ((word >> 8) & 255u)];
                if (4 * j + 2 < P.nb) syn ^= tb[512 + ((word >> 16) & 255u)];
                if (4 * j + 3 < P.nb) syn ^= tb[768 + (word >> 24)];
            }
        }
        const uint32_t wpl = cw_min((alive && syn == 0) ? p.pl : 99u);  // first passing path in list order
        const uint32_t okflag = wpl != 99u ? 1u : 0u;
        const uint32_t want = okflag ? wpl : 0u;
        if (live) {
            if (alive && p.pl == want) {  // the winner writes its information word
                uint8_t* orow = P.out + (size_t)row * P.nb;
                if ((P.nb & 3u) == 0) {
                    uint32_t* o32 = reinterpret_cast<uint32_t*>(orow);
                    for (uint32_t j = 0; j < (P.nb >> 2); ++j) o32[j] = il[j * 64 + lane];
                } else {
                    for (uint32_t b = 0; b < P.nb; ++b) orow[b] = (uint8_t)(il[(b >> 2) * 64 + lane] >> (8u * (b & 3u)));
                }
                if (P.ok) P.ok[row] = (uint8_t)okflag;
            }
            if (P.metrics) {
                if ((uint32_t)p.q >= p.P && (uint32_t)p.q < P.L) P.metrics[(size_t)row * P.L + p.q] = 0.0f;
                if (alive) P.metrics[(size_t)row * P.L + p.pl] = p.metric;
            }
        }
        WAVE_SYNC();
    }
}

#ifdef LN_KERNEL_NAME
// other geometries (scl_lane_n*_l*.hip), the float decoder on int8 rows (scl_lane_i8*.hip): one instantiation per translation unit, named by the stub
template __global__ void scl_lane_kernel<false, LN_CHAR != 0>(const DecodeParams);

unsigned LN_ARENA_NAME() { return lane::AR_WORDS; }
hipError_t LN_LAUNCH_NAME(const DecodeParams& P, bool diag, unsigned grid, hipStream_t st)
{
    // (P.i8rows: what the caller's rows are; this kernel reads one kind only)
    if ((P.chr != 0) != (LN_CHAR != 0) || (P.i8rows != 0) != (LN_I8ROWS != 0) || P.logN != (uint32_t)lane::LOGN || P.L > (uint32_t)lane::LC)
        return hipErrorInvalidValue;
#ifdef PAMD_DIAG
    if (diag) {
        hipLaunchKernelGGL((scl_lane_kernel<true, LN_CHAR != 0>), dim3(grid), dim3(64), lane::LDS_WORDS * 4, st, P);
        return hipGetLastError();
    }
#else
    if (diag) return hipErrorInvalidValue;
#endif
    hipLaunchKernelGGL((scl_lane_kernel<false, LN_CHAR != 0>), dim3(grid), dim3(64), lane::LDS_WORDS * 4, st, P);
    return hipGetLastError();
}
#elif LN_CHAR
// SclFipChar on the lane kernel: saturating f / g, repetition leaves at every size; the rows arrive widened to fp32
template __global__ void scl_lane_kernel<false, true>(const DecodeParams);

#if LN_BIG
hipError_t launch_scl_lane_char_big(const DecodeParams& P, bool diag, unsigned grid, hipStream_t st)
#else
hipError_t launch_scl_lane_char(const DecodeParams& P, bool diag, unsigned grid, hipStream_t st)
#endif
{
#ifdef PAMD_DIAG
    if (diag) {
        hipLaunchKernelGGL((scl_lane_kernel<true, true>), dim3(grid), dim3(64), lane::LDS_WORDS * 4, st, P);
        return hipGetLastError();
    }
#else
    if (diag) return hipErrorInvalidValue;
#endif
    hipLaunchKernelGGL((scl_lane_kernel<false, true>), dim3(grid), dim3(64), lane::LDS_WORDS * 4, st, P);
    return hipGetLastError();
}
#else
#if LN_BIG
hipError_t launch_scl_lane_big(const DecodeParams& P, bool diag, unsigned grid, hipStream_t st)
#else
// (explicit instantiation first: the float kernel of the headline configuration opens the code object, see sc_pair_kernel.hip)
template __global__ void scl_lane_kernel<false, false>(const DecodeParams);

hipError_t launch_scl_lane(const DecodeParams& P, bool diag, unsigned grid, hipStream_t st)
#endif
{
    if (P.chr) return hipErrorInvalidValue;  // (launch_scl_lane_char)
#ifdef PAMD_DIAG
    if (diag) {
        hipLaunchKernelGGL((scl_lane_kernel<true, false>), dim3(grid), dim3(64), lane::LDS_WORDS * 4, st, P);
        return hipGetLastError();
    }
#else
    if (diag) return hipErrorInvalidValue;
#endif
    hipLaunchKernelGGL((scl_lane_kernel<false, false>), dim3(grid), dim3(64), lane::LDS_WORDS * 4, st, P);
    return hipGetLastError();
}
#if !LN_BIG
unsigned scl_lane_arena_words() { return lane::AR_WORDS; }
#endif
#endif  // LN_CHAR

}  // namespace pamd
