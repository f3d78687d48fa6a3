// Fast-SSC decoder, two-lanes-per-codeword form: N = 1024, fp32 rows.
//
// Reference: FastSscAvxFloat  src/polarcode/decoding/fastssc_avx_float.cpp (nodes :148-792, createDecoder :797-896,
// decode :940-955).  The kernel of sc_kernel.hip gives a codeword 16 lanes and keeps its tree in LDS; here a codeword
// has TWO lanes and the tree lives in their registers, so a wave decodes 32 codewords and f / g / COMBINE are lane-local:
//
//   code position e of a block belongs to lane (e >> 1) & 1 of the pair, local index j = ((e >> 2) << 1) | (e & 1)
//   -- the lanes alternate in PAIRS of positions.  Partners of f / g are e and e + h with h >= 4 (a block of 4 is always
//   a leaf: createDecoder :797-870), so they always sit in the same lane, h / 2 apart; a lane reads its pairs of the
//   channel row with 8-byte loads.  Every block of 2^t positions is a local block of 2^(t-1) = 2^T, T the LOCAL stage:
//
//   local stage T <= 6     VGPRs v[128..253], hand-allocated (the compiler is confined to v0..v103): asm blocks on fixed
//                          register numbers, the same map as scl_lane_kernel.hip
//   local stage 7          LDS, [16-byte chunk][lane]  (32 KB per wave)
//   local stage 8          the lane's 256 AGPRs (v_accvgpr_write / _read; the compiler uses none)
//   local stage 9          the channel row itself: read twice per codeword (f and g of the root)
//   partial sums           bit-packed per lane (bit j = local index j): stages <= 7 in v[104..127], stage 8 and the
//                          code word in LDS rows
//
// One wave per SIMD (4 single-wave workgroups per CU: 40 KB of LDS and all 512 registers of a SIMD lane each), 128
// codewords in flight per CU.  What crosses the two lanes: the reductions inside the leaves (one DPP quad_perm swap
// each), the interleave of the two half code words at the end.
//
// Bit-exactness: f exact; g is one fp32 addition (v_fma_f32 with a factor of +-1.0); the leaf sums keep the reference's
// order -- the 8 strided accumulators of fold8 are 4 per lane (acc[s] of lane p is AVX lane (s & 1) + 4 (s >> 1) + 2 p),
// every cross-lane sum the reference takes is a sum of one value per lane (commutative) or is rebuilt in full in both lanes.
#define PAMD_POLAR_F_MINF 1
// PR_LOGN (sc_pair_n9.hip): the same source for N = 512 -- the channel row IS local stage 8: it is fetched into the AGPRs once per
// codeword (whole lines, transposed through the LDS like the pass of N = 1024; int8 rows widened on the way) and the tree below is the same.
#ifndef PR_LOGN
#define PR_LOGN 10
#endif
#ifndef PR_KERNEL_NAME
#define PR_KERNEL_NAME sc_pair_kernel
#define PR_LAUNCH_NAME launch_sc_pair
#endif
#include "device_common.h"
#include "pair_runs.inc"

#include <utility>

namespace pamd {
namespace pair {

// ---- register map (per lane) ---------------------------------------------------------------------------------------
constexpr int PLOGN = PR_LOGN;             // log2 of the block length: 10, or 9 / 8 / 7 (the channel row is local stage 8 / 7 / 6)
constexpr int NWL = (1 << PLOGN) / 64;     // words of the code word per lane
constexpr int KV = 104;      // the compiler's VGPR budget: v0 .. v103
constexpr int RBL0 = 104;    // left-child / root output bits of local stage T <= 5: v[104 + T]
constexpr int RBR0 = 110;    // right-child output bits of stage T <= 5: v[110 + T]
constexpr int RBL6 = 116, RBR6 = 118;  // stage 6: 2 words each
constexpr int RBL7 = 120, RBR7 = 124;  // stage 7: 4 words each
__host__ __device__ constexpr int RS(int t) { return t == 1 ? 128 : t == 2 ? 130 : t == 3 ? 134 : t == 4 ? 142 : t == 5 ? 158 : 190; }
__host__ __device__ constexpr int RB(int side, int t) { return t <= 5 ? (side ? RBR0 : RBL0) + t : t == 6 ? (side ? RBR6 : RBL6) : (side ? RBR7 : RBL7); }
// LDS of one wave (32-bit words)
constexpr uint32_t LDS_S7 = 0;                     // [32 chunks][64 lanes][4 words]
constexpr uint32_t LDS_B8L = 8192;                 // stage-8 bits, left child: [8 words][64 lanes]
constexpr uint32_t LDS_B8R = 8192 + 512;
constexpr uint32_t LDS_X = 8192 + 1024;            // code word (local order): [16 words][64 lanes]
constexpr uint32_t LDS_WORDS = 10240;
constexpr uint32_t LDS_OUT = 0;                    // epilogue: information words [word][codeword of the wave] (stage 7 is dead)

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
// s = (s << 1) | sign(v[R]): the signs of a run of registers, taken from the highest index down, end up with element i at bit i
template <int R>
__device__ __forceinline__ void sgn_in(uint32_t& s)
{
    asm volatile("v_alignbit_b32 %0, %0, v[%1], 31" : "+v"(s) : "n"(R));
}
template <int R>
__device__ __forceinline__ void amin_in(float& m)
{
    asm volatile("v_min_f32_e64 %0, |v[%1]|, %0" : "+v"(m) : "n"(R));
}
// e = (e << 1) | (|v[R]| == m)
template <int R>
__device__ __forceinline__ void eq_in(uint32_t& e, float m)
{
    asm volatile("v_cmp_eq_f32_e64 vcc, |v[%2]|, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(e) : "v"(m), "n"(R) : "vcc");
}
template <int R>
__device__ __forceinline__ void add_in(float& a)
{
    asm volatile("v_add_f32 %0, %0, v[%1]" : "+v"(a) : "n"(R));
}
// ---- runs over consecutive registers, one asm statement each (pair_runs.inc) ----------------------------------------------------
// signs of v[R .. R+CNT), element i at bit i
template <int R, int CNT>
__device__ __forceinline__ uint32_t sign_run()
{
    uint32_t s = 0;
    if constexpr (CNT == 32) asm volatile(PR_RUN_SGN_32 : [s] "+v"(s) : [r] "n"(R));
    else if constexpr (CNT == 16) asm volatile(PR_RUN_SGN_16 : [s] "+v"(s) : [r] "n"(R));
    else if constexpr (CNT == 8) asm volatile(PR_RUN_SGN_8 : [s] "+v"(s) : [r] "n"(R));
    else if constexpr (CNT == 4) asm volatile(PR_RUN_SGN_4 : [s] "+v"(s) : [r] "n"(R));
    else asm volatile(PR_RUN_SGN_2 : [s] "+v"(s) : [r] "n"(R));
    return s;
}
// m[k & 3] = min(m[k & 3], |v[R + k]|), k < CNT
template <int R, int CNT>
__device__ __forceinline__ void amin_run4(float (&m)[4])
{
#define PR_AM_OPS [m0] "+v"(m[0]), [m1] "+v"(m[1]), [m2] "+v"(m[2]), [m3] "+v"(m[3])
    if constexpr (CNT == 64) asm volatile(PR_RUN_AMIN_64 : PR_AM_OPS : [r] "n"(R));
    else if constexpr (CNT == 32) asm volatile(PR_RUN_AMIN_32 : PR_AM_OPS : [r] "n"(R));
    else if constexpr (CNT == 16) asm volatile(PR_RUN_AMIN_16 : PR_AM_OPS : [r] "n"(R));
    else if constexpr (CNT == 8) asm volatile(PR_RUN_AMIN_8 : PR_AM_OPS : [r] "n"(R));
    else if constexpr (CNT == 4) asm volatile(PR_RUN_AMIN_4 : PR_AM_OPS : [r] "n"(R));
    else asm volatile(PR_RUN_AMIN_2 : PR_AM_OPS : [r] "n"(R));
#undef PR_AM_OPS
}
// the minimum magnitude of v[R .. R+CNT) folded into mn
template <int R, int CNT>
__device__ __forceinline__ void amin_run(float& mn)
{
    float m[4] = { mn, mn, mn, mn };
    amin_run4<R, CNT>(m);
    const float a = m[0] < m[1] ? m[0] : m[1], b = m[2] < m[3] ? m[2] : m[3];
    mn = a < b ? a : b;
}
// bit i = (|v[R + i]| == m), i < CNT <= 32
template <int R, int CNT>
__device__ __forceinline__ uint32_t eq_run(float m)
{
    uint32_t e = 0;
    if constexpr (CNT == 32) asm volatile(PR_RUN_EQ_32 : [e] "+v"(e) : [m] "v"(m), [r] "n"(R) : "vcc");
    else if constexpr (CNT == 16) asm volatile(PR_RUN_EQ_16 : [e] "+v"(e) : [m] "v"(m), [r] "n"(R) : "vcc");
    else if constexpr (CNT == 8) asm volatile(PR_RUN_EQ_8 : [e] "+v"(e) : [m] "v"(m), [r] "n"(R) : "vcc");
    else if constexpr (CNT == 4) asm volatile(PR_RUN_EQ_4 : [e] "+v"(e) : [m] "v"(m), [r] "n"(R) : "vcc");
    else asm volatile(PR_RUN_EQ_2 : [e] "+v"(e) : [m] "v"(m), [r] "n"(R) : "vcc");
    return e;
}
// ... against m[i & 3]
template <int R, int CNT>
__device__ __forceinline__ uint32_t eq_run4(const float (&m)[4])
{
    uint32_t e = 0;
#define PR_EQ_OPS [m0] "v"(m[0]), [m1] "v"(m[1]), [m2] "v"(m[2]), [m3] "v"(m[3]), [r] "n"(R)
    if constexpr (CNT == 32) asm volatile(PR_RUN_EQ4_32 : [e] "+v"(e) : PR_EQ_OPS : "vcc");
    else if constexpr (CNT == 16) asm volatile(PR_RUN_EQ4_16 : [e] "+v"(e) : PR_EQ_OPS : "vcc");
    else asm volatile(PR_RUN_EQ4_8 : [e] "+v"(e) : PR_EQ_OPS : "vcc");
#undef PR_EQ_OPS
    return e;
}
// a[k & 3] += v[R + k], k < CNT, ascending
template <int R, int CNT>
__device__ __forceinline__ void add_run4(float (&a)[4])
{
#define PR_AD_OPS [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3])
    if constexpr (CNT == 64) asm volatile(PR_RUN_ADD_64 : PR_AD_OPS : [r] "n"(R));
    else if constexpr (CNT == 32) asm volatile(PR_RUN_ADD_32 : PR_AD_OPS : [r] "n"(R));
    else if constexpr (CNT == 16) asm volatile(PR_RUN_ADD_16 : PR_AD_OPS : [r] "n"(R));
    else if constexpr (CNT == 8) asm volatile(PR_RUN_ADD_8 : PR_AD_OPS : [r] "n"(R));
    else if constexpr (CNT == 4) asm volatile(PR_RUN_ADD_4 : PR_AD_OPS : [r] "n"(R));
    else asm volatile(PR_RUN_ADD_2 : PR_AD_OPS : [r] "n"(R));
#undef PR_AD_OPS
}

// the other lane of the pair
__device__ __forceinline__ uint32_t px(uint32_t v) { return dpp_u32<0xB1>(v); }
__device__ __forceinline__ float pxf(float v) { return bitsf(dpp_u32<0xB1>(fbits(v))); }

// ---- int8 decoder (FastSscFipChar): an int8 LLR is the float of the same value, saturation is a clamp ---------------------------
// (fip_char.h:45-74; device_common.h polar_f_char / polar_g_char are the same arithmetic in C++)
struct Sat {
    float c127;  // 127.0 (scalar operand)
    float m128;  // -128.0 in a VGPR (a VALU instruction of gfx9 takes one scalar operand)
};
// f of the int8 decoder from the float f already in v[R .. R+CNT): magnitude clamped to [1, 127], sign kept (also of a -0)
template <int R, int CNT>
__device__ __forceinline__ void clamp_f(const Sat& S, uint32_t absmask)
{
    const float c = S.c127;
    static_for<CNT>([&](auto I) {
        constexpr int i = decltype(I)::value;
        uint32_t t;
        const float cc = c;
        const uint32_t am = absmask;
        asm volatile("v_med3_f32 %[t], |v[%[r]]|, 1.0, %[c]\n\tv_bfi_b32 v[%[r]], %[am], %[t], v[%[r]]" : [t] "=&v"(t) : [r] "n"(R + i), [c] "s"(cc), [am] "s"(am));
    });
}
// g / a + b of the int8 decoder: clamp to [-128, 127]
template <int R, int CNT>
__device__ __forceinline__ void clamp_g(const Sat& S)
{
    static_for<CNT>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const float cc = S.c127, mm = S.m128;
        asm volatile("v_med3_f32 v[%[r]], v[%[r]], %[m], %[c]" ::[r] "n"(R + i), [c] "s"(cc), [m] "v"(mm));
    });
}

// ---- f / g blocks on registers (local stages <= 6) -------------------------------------------------------------------
// text of f over 4 consecutive elements: v[D+k] = f(v[A+k], v[B+k]), D may equal A (the xor is taken first)
#define PR_F4T(D_, A_, B_)                                       \
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
// text of g over 4 consecutive elements: v[D+k] = g(v[A+k], v[B+k], bit (31 - %[s0] + k) of v[W]) (D may equal A)
#define PR_G4T(D_, A_, B_, W_)                                   \
    "v_lshlrev_b32 %[t0], %[s0], v[" W_ "]\n\t"                  \
    "v_lshlrev_b32 %[t1], %[s0]-1, v[" W_ "]\n\t"                \
    "v_lshlrev_b32 %[t2], %[s0]-2, v[" W_ "]\n\t"                \
    "v_lshlrev_b32 %[t3], %[s0]-3, v[" W_ "]\n\t"                \
    "v_and_or_b32 %[t0], %[t0], %[sm], 1.0\n\t"                  \
    "v_and_or_b32 %[t1], %[t1], %[sm], 1.0\n\t"                  \
    "v_and_or_b32 %[t2], %[t2], %[sm], 1.0\n\t"                  \
    "v_and_or_b32 %[t3], %[t3], %[sm], 1.0\n\t"                  \
    "v_fma_f32 v[" D_ "], v[" A_ "], %[t0], v[" B_ "]\n\t"       \
    "v_fma_f32 v[" D_ "+1], v[" A_ "+1], %[t1], v[" B_ "+1]\n\t" \
    "v_fma_f32 v[" D_ "+2], v[" A_ "+2], %[t2], v[" B_ "+2]\n\t" \
    "v_fma_f32 v[" D_ "+3], v[" A_ "+3], %[t3], v[" B_ "+3]\n\t"

template <int G, int D, int A, int B>
__device__ __forceinline__ void f_block(uint32_t absmask)
{
    static_assert(G == 1 || G == 2 || G == 4, "block size");
    uint32_t t0, t1, t2, t3;
    if constexpr (G == 4) {
        asm volatile(PR_F4T("%[d]", "%[a]", "%[b]")
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [d] "n"(D), [a] "n"(A), [b] "n"(B), [am] "s"(absmask));
    } else if constexpr (G == 2) {
        asm volatile(
            "v_xor_b32 %[t0], v[%[a]], v[%[b]]\n\t"
            "v_xor_b32 %[t1], v[%[a]+1], v[%[b]+1]\n\t"
            "v_min_f32_e64 v[%[d]], |v[%[a]]|, |v[%[b]]|\n\t"
            "v_min_f32_e64 v[%[d]+1], |v[%[a]+1]|, |v[%[b]+1]|\n\t"
            "v_bfi_b32 v[%[d]], %[am], v[%[d]], %[t0]\n\t"
            "v_bfi_b32 v[%[d]+1], %[am], v[%[d]+1], %[t1]"
            : [t0] "=&v"(t0), [t1] "=&v"(t1)
            : [d] "n"(D), [a] "n"(A), [b] "n"(B), [am] "s"(absmask));
    } else {
        asm volatile(
            "v_xor_b32 %[t0], v[%[a]], v[%[b]]\n\t"
            "v_min_f32_e64 v[%[d]], |v[%[a]]|, |v[%[b]]|\n\t"
            "v_bfi_b32 v[%[d]], %[am], v[%[d]], %[t0]"
            : [t0] "=&v"(t0)
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
// llr[T-1] = a + b (ZeroRNode :232-237), T = 2 .. 6
template <int T>
__device__ __forceinline__ void g0_reg()
{
    constexpr int h = 1 << (T - 1), A = RS(T), D = RS(T - 1);
    static_for<h>([&](auto I) {
        constexpr int i = decltype(I)::value;
        asm volatile("v_add_f32 v[%0], v[%1], v[%2]" ::"n"(D + i), "n"(A + i), "n"(A + h + i));
    });
}
// g over G consecutive elements: v[D+k] = g(v[A+k], v[B+k], bit (31 - S0 + k) of v[W])
template <int G, int D, int A, int B, int W, int S0>
__device__ __forceinline__ void g_block(uint32_t signmask)
{
    if constexpr (G == 4) {
        uint32_t t0, t1, t2, t3;
        asm volatile(PR_G4T("%[d]", "%[a]", "%[b]", "%[w]")
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [d] "n"(D), [a] "n"(A), [b] "n"(B), [w] "n"(W), [s0] "n"(S0), [sm] "s"(signmask));
    } else {
        uint32_t t0;
        asm volatile("v_lshlrev_b32 %[t0], %[s0], v[%[w]]\n\t"
                     "v_and_or_b32 %[t0], %[t0], %[sm], 1.0\n\t"
                     "v_fma_f32 v[%[d]], v[%[a]], %[t0], v[%[b]]"
                     : [t0] "=&v"(t0)
                     : [d] "n"(D), [a] "n"(A), [b] "n"(B), [w] "n"(W), [s0] "n"(S0), [sm] "s"(signmask));
    }
}
// llr[T-1] = g(llr[T], left bits), T = 2 .. 6: the left child's bits are the word v[RBL0 + T - 1]
template <int T>
__device__ __forceinline__ void g_reg(uint32_t signmask)
{
    constexpr int h = 1 << (T - 1), A = RS(T), D = RS(T - 1), W = RBL0 + (T - 1);
    constexpr int G = h >= 4 ? 4 : 1;
    static_for<h / G>([&](auto J) {
        constexpr int j = decltype(J)::value * G;
        g_block<G, D + j, A + j, A + h + j, W, 31 - j>(signmask);
    });
}

// ---- local stage 7 (LDS column, 16-byte chunks 1 KB apart) -> stage 6 (registers) --------------------------------------
// Round r loads the chunks 8r .. 8r+7 into v[D .. D+31] and their partners 16+8r .. into v[T .. T+31] (the registers of
// stage 5, dead while stage 6 is being produced), then works on them in place as they arrive.
#define PR_RD2(c) "ds_read_b128 v[%[d]+4*" #c ":%[d]+4*" #c "+3], %[ad] offset:%[oa]+1024*" #c "\n\t" \
                  "ds_read_b128 v[%[t]+4*" #c ":%[t]+4*" #c "+3], %[ad] offset:%[ob]+1024*" #c "\n\t"
#define PR_WAITL(n) "s_waitcnt lgkmcnt(" #n ")\n\t"
// chunk c of round R of the g: element 4c + k of the round is bit 4c + k of the word v[RBL6 + R]
template <int R, int C>
__device__ __forceinline__ void g7_chunk(uint32_t signmask)
{
    uint32_t u0, u1, u2, u3;
    asm volatile("s_waitcnt lgkmcnt(%[wt])\n\t" PR_G4T("%[d]", "%[d]", "%[t]", "%[w]")
                 : [t0] "=&v"(u0), [t1] "=&v"(u1), [t2] "=&v"(u2), [t3] "=&v"(u3)
                 : [d] "n"(RS(6) + 32 * R + 4 * C), [t] "n"(RS(5) + 4 * C), [w] "n"(RBL6 + R), [s0] "n"(31 - 4 * C),
                   [wt] "n"(14 - 2 * C), [sm] "s"(signmask)
                 : "memory");
}
// KIND 0: f, 1: g with the bits of word v[RBL6 + R], 2: a + b
template <int KIND, int R>
__device__ __forceinline__ void s7_round(uint32_t addr, uint32_t absmask, uint32_t signmask)
{
    uint32_t t0, t1, t2, t3;
    if constexpr (KIND == 0) {
        asm volatile(PR_RD2(0) PR_RD2(1) PR_RD2(2) PR_RD2(3) PR_RD2(4) PR_RD2(5) PR_RD2(6) PR_RD2(7)
                     PR_WAITL(14) PR_F4T("%[d]", "%[d]", "%[t]")
                     PR_WAITL(12) PR_F4T("%[d]+4", "%[d]+4", "%[t]+4")
                     PR_WAITL(10) PR_F4T("%[d]+8", "%[d]+8", "%[t]+8")
                     PR_WAITL(8) PR_F4T("%[d]+12", "%[d]+12", "%[t]+12")
                     PR_WAITL(6) PR_F4T("%[d]+16", "%[d]+16", "%[t]+16")
                     PR_WAITL(4) PR_F4T("%[d]+20", "%[d]+20", "%[t]+20")
                     PR_WAITL(2) PR_F4T("%[d]+24", "%[d]+24", "%[t]+24")
                     PR_WAITL(0) PR_F4T("%[d]+28", "%[d]+28", "%[t]+28")
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [d] "n"(RS(6) + 32 * R), [t] "n"(RS(5)), [ad] "v"(addr), [oa] "n"(4 * LDS_S7 + 8192 * R),
                       [ob] "n"(4 * LDS_S7 + 16384 + 8192 * R), [am] "s"(absmask)
                     : "memory");
    } else if constexpr (KIND == 2) {
#define PR_ADD4(o) "v_add_f32 v[%[d]+" #o "], v[%[d]+" #o "], v[%[t]+" #o "]\n\t" "v_add_f32 v[%[d]+" #o "+1], v[%[d]+" #o "+1], v[%[t]+" #o "+1]\n\t" \
                   "v_add_f32 v[%[d]+" #o "+2], v[%[d]+" #o "+2], v[%[t]+" #o "+2]\n\t" "v_add_f32 v[%[d]+" #o "+3], v[%[d]+" #o "+3], v[%[t]+" #o "+3]\n\t"
        asm volatile(PR_RD2(0) PR_RD2(1) PR_RD2(2) PR_RD2(3) PR_RD2(4) PR_RD2(5) PR_RD2(6) PR_RD2(7)
                     PR_WAITL(14) PR_ADD4(0) PR_WAITL(12) PR_ADD4(4) PR_WAITL(10) PR_ADD4(8) PR_WAITL(8) PR_ADD4(12)
                     PR_WAITL(6) PR_ADD4(16) PR_WAITL(4) PR_ADD4(20) PR_WAITL(2) PR_ADD4(24) PR_WAITL(0) PR_ADD4(28)
                     :
                     : [d] "n"(RS(6) + 32 * R), [t] "n"(RS(5)), [ad] "v"(addr), [oa] "n"(4 * LDS_S7 + 8192 * R),
                       [ob] "n"(4 * LDS_S7 + 16384 + 8192 * R)
                     : "memory");
#undef PR_ADD4
    } else {
        asm volatile(PR_RD2(0) PR_RD2(1) PR_RD2(2) PR_RD2(3) PR_RD2(4) PR_RD2(5) PR_RD2(6) PR_RD2(7)
                     :
                     : [d] "n"(RS(6) + 32 * R), [t] "n"(RS(5)), [ad] "v"(addr), [oa] "n"(4 * LDS_S7 + 8192 * R),
                       [ob] "n"(4 * LDS_S7 + 16384 + 8192 * R)
                     : "memory");
        static_for<8>([&](auto C) { g7_chunk<R, decltype(C)::value>(signmask); });
    }
    (void)t0; (void)t1; (void)t2; (void)t3;
}
#undef PR_RD2
template <int KIND>
__device__ __forceinline__ void op_s7(uint32_t lane16, uint32_t absmask, uint32_t signmask)
{
    s7_round<KIND, 0>(lane16, absmask, signmask);
    s7_round<KIND, 1>(lane16, absmask, signmask);
}

// ---- local stage 8 (AGPRs) -> stage 7 (LDS column) -----------------------------------------------------------------------
// chunk c = outputs 4c .. 4c+3 from a[4c+k] and a[128+4c+k]; temporaries rotate through v[128..247] (stages <= 6 are dead)
template <bool CH, int KIND, int c>
__device__ __forceinline__ void s8_chunk(uint32_t lane16, uint32_t absmask, uint32_t signmask, const Sat& S)
{
    constexpr int X = 128 + 8 * (c % 15);
    uint32_t t0, t1, t2, t3;
#define PR_RDA(k) "v_accvgpr_read_b32 v[%[x]+" #k "], a[%[ia]+" #k "]\n\t" "v_accvgpr_read_b32 v[%[x]+4+" #k "], a[%[ia]+128+" #k "]\n\t"
    if constexpr (KIND == 0) {
        asm volatile(PR_RDA(0) PR_RDA(1) PR_RDA(2) PR_RDA(3) PR_F4T("%[x]", "%[x]", "%[x]+4")
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [x] "n"(X), [ia] "n"(4 * c), [am] "s"(absmask));
        if constexpr (CH) clamp_f<X, 4>(S, absmask);
    } else if constexpr (KIND == 1) {
        asm volatile(PR_RDA(0) PR_RDA(1) PR_RDA(2) PR_RDA(3) PR_G4T("%[x]", "%[x]", "%[x]+4", "%[w]")
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [x] "n"(X), [ia] "n"(4 * c), [w] "n"(RBL7 + (c >> 3)), [s0] "n"(31 - ((4 * c) & 31)), [sm] "s"(signmask));
        if constexpr (CH) clamp_g<X, 4>(S);
    } else {
        asm volatile(PR_RDA(0) PR_RDA(1) PR_RDA(2) PR_RDA(3)
                     "v_add_f32 v[%[x]], v[%[x]], v[%[x]+4]\n\t" "v_add_f32 v[%[x]+1], v[%[x]+1], v[%[x]+5]\n\t"
                     "v_add_f32 v[%[x]+2], v[%[x]+2], v[%[x]+6]\n\t" "v_add_f32 v[%[x]+3], v[%[x]+3], v[%[x]+7]"
                     :
                     : [x] "n"(X), [ia] "n"(4 * c));
        if constexpr (CH) clamp_g<X, 4>(S);
    }
#undef PR_RDA
    asm volatile("ds_write_b128 %[ad], v[%[x]:%[x]+3] offset:%[of]" ::[x] "n"(X), [ad] "v"(lane16), [of] "n"(4 * LDS_S7 + 1024 * c) : "memory");
    (void)t0; (void)t1; (void)t2; (void)t3;
}
template <bool CH, int KIND>
__device__ __forceinline__ void op_s8(uint32_t lane16, uint32_t absmask, uint32_t signmask, const Sat& S)
{
    static_for<32>([&](auto C) { s8_chunk<CH, KIND, decltype(C)::value>(lane16, absmask, signmask, S); });
}

// ---- local stage 9 (the channel row) -> stage 8 (AGPRs) -----------------------------------------------------------------
// A lane reading its own pairs of its own row would touch 32 different 128-byte lines per load instruction, 16 bytes of
// each.  Instead the wave fetches whole lines -- eight lanes per row, 16 bytes each, four instructions for one line of
// each of the 32 rows -- and transposes them through the LDS (the stage-7 area, dead during this pass): line step L takes
// the lines L and L + 16 of every row (positions 32 L .. and 512 + 32 L ..), the lane pair of a row reads its 16 + 16
// values back as 8-byte pieces and produces the local outputs 16 L .. 16 L + 15 -> a[16 L ..].
// LDS unit (16 bytes) of (row, segment) = row * 8 + ((segment + row + (row >> 3)) & 7): the 64 lanes of a read spread over
// all banks twice (the minimum for 512 bytes).  Two buffers of 8 KB (line L, line L + 16), loads two steps ahead.
struct TopAddr {
    const float* rp[4];  // this lane's piece of the rows (lane >> 3) + 8 i of the group (line 0)
    uint32_t wa[4];      // LDS byte address of that piece
    uint32_t ra[8];      // LDS byte address of the k-th pair of this lane's row
};
constexpr int TOP_WORK = 192;  // v[192..207]: the pairs of line L, v[208..223]: of line L + 16
template <int L_>
__device__ __forceinline__ void top_load(const TopAddr& A)
{
    constexpr int S = 128 + 32 * (L_ & 1);
    const float *r0 = A.rp[0], *r1 = A.rp[1], *r2 = A.rp[2], *r3 = A.rp[3];
    asm volatile("global_load_dwordx4 v[%[s]:%[s]+3], %[r0], off offset:%[o]\n\t"
                 "global_load_dwordx4 v[%[s]+4:%[s]+7], %[r1], off offset:%[o]\n\t"
                 "global_load_dwordx4 v[%[s]+8:%[s]+11], %[r2], off offset:%[o]\n\t"
                 "global_load_dwordx4 v[%[s]+12:%[s]+15], %[r3], off offset:%[o]\n\t"
                 "global_load_dwordx4 v[%[s]+16:%[s]+19], %[r0], off offset:%[o]+2048\n\t"
                 "global_load_dwordx4 v[%[s]+20:%[s]+23], %[r1], off offset:%[o]+2048\n\t"
                 "global_load_dwordx4 v[%[s]+24:%[s]+27], %[r2], off offset:%[o]+2048\n\t"
                 "global_load_dwordx4 v[%[s]+28:%[s]+31], %[r3], off offset:%[o]+2048"
                 :
                 : [s] "n"(S), [r0] "v"(r0), [r1] "v"(r1), [r2] "v"(r2), [r3] "v"(r3), [o] "n"(128 * L_)
                 : "memory");
}
// KIND 0: f, 1: g (w = the left-child word that holds the bits of the outputs 16 L ..), 2: a + b
template <int KIND, int L_, int M>
__device__ __forceinline__ void top_chunk(uint32_t w, uint32_t absmask, uint32_t signmask)
{
    constexpr int X = TOP_WORK + 4 * M, Y = TOP_WORK + 16 + 4 * M, O = 16 * L_ + 4 * M;
    uint32_t t0, t1, t2, t3;
#define PR_WRA "v_accvgpr_write_b32 a[%[o]], v[%[x]]\n\t" "v_accvgpr_write_b32 a[%[o]+1], v[%[x]+1]\n\t" \
               "v_accvgpr_write_b32 a[%[o]+2], v[%[x]+2]\n\t" "v_accvgpr_write_b32 a[%[o]+3], v[%[x]+3]"
    if constexpr (KIND == 0) {
        asm volatile("s_waitcnt lgkmcnt(%[wt])\n\t" PR_F4T("%[x]", "%[x]", "%[y]") PR_WRA
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [x] "n"(X), [y] "n"(Y), [o] "n"(O), [wt] "n"(12 - 4 * M), [am] "s"(absmask)
                     : "memory");
    } else if constexpr (KIND == 1) {
        asm volatile("s_waitcnt lgkmcnt(%[wt])\n\t"
                     "v_lshlrev_b32 %[t0], %[s0], %[w]\n\t"
                     "v_lshlrev_b32 %[t1], %[s0]-1, %[w]\n\t"
                     "v_lshlrev_b32 %[t2], %[s0]-2, %[w]\n\t"
                     "v_lshlrev_b32 %[t3], %[s0]-3, %[w]\n\t"
                     "v_and_or_b32 %[t0], %[t0], %[sm], 1.0\n\t"
                     "v_and_or_b32 %[t1], %[t1], %[sm], 1.0\n\t"
                     "v_and_or_b32 %[t2], %[t2], %[sm], 1.0\n\t"
                     "v_and_or_b32 %[t3], %[t3], %[sm], 1.0\n\t"
                     "v_fma_f32 v[%[x]], v[%[x]], %[t0], v[%[y]]\n\t"
                     "v_fma_f32 v[%[x]+1], v[%[x]+1], %[t1], v[%[y]+1]\n\t"
                     "v_fma_f32 v[%[x]+2], v[%[x]+2], %[t2], v[%[y]+2]\n\t"
                     "v_fma_f32 v[%[x]+3], v[%[x]+3], %[t3], v[%[y]+3]\n\t" PR_WRA
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [x] "n"(X), [y] "n"(Y), [o] "n"(O), [wt] "n"(12 - 4 * M), [w] "v"(w), [s0] "n"(31 - (16 * (L_ & 1) + 4 * M)),
                       [sm] "s"(signmask)
                     : "memory");
    } else {
        asm volatile("s_waitcnt lgkmcnt(%[wt])\n\t"
                     "v_add_f32 v[%[x]], v[%[x]], v[%[y]]\n\t" "v_add_f32 v[%[x]+1], v[%[x]+1], v[%[y]+1]\n\t"
                     "v_add_f32 v[%[x]+2], v[%[x]+2], v[%[y]+2]\n\t" "v_add_f32 v[%[x]+3], v[%[x]+3], v[%[y]+3]\n\t" PR_WRA
                     :
                     : [x] "n"(X), [y] "n"(Y), [o] "n"(O), [wt] "n"(12 - 4 * M)
                     : "memory");
    }
#undef PR_WRA
    (void)t0; (void)t1; (void)t2; (void)t3;
}
template <int KIND, int L_>
__device__ __forceinline__ void top_step(const TopAddr& A, uint32_t w, uint32_t absmask, uint32_t signmask)
{
    constexpr int S = 128 + 32 * (L_ & 1), BUF = 8192 * (L_ & 1);
    {
        const uint32_t w0 = A.wa[0], w1 = A.wa[1], w2 = A.wa[2], w3 = A.wa[3];
        asm volatile("s_waitcnt vmcnt(%[wt])\n\t"
                     "ds_write_b128 %[w0], v[%[s]:%[s]+3] offset:%[b]\n\t"
                     "ds_write_b128 %[w1], v[%[s]+4:%[s]+7] offset:%[b]\n\t"
                     "ds_write_b128 %[w2], v[%[s]+8:%[s]+11] offset:%[b]\n\t"
                     "ds_write_b128 %[w3], v[%[s]+12:%[s]+15] offset:%[b]\n\t"
                     "ds_write_b128 %[w0], v[%[s]+16:%[s]+19] offset:%[b]+4096\n\t"
                     "ds_write_b128 %[w1], v[%[s]+20:%[s]+23] offset:%[b]+4096\n\t"
                     "ds_write_b128 %[w2], v[%[s]+24:%[s]+27] offset:%[b]+4096\n\t"
                     "ds_write_b128 %[w3], v[%[s]+28:%[s]+31] offset:%[b]+4096"
                     :
                     : [s] "n"(S), [b] "n"(BUF), [wt] "n"(L_ < 15 ? 8 : 0), [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3)
                     : "memory");
    }
    if constexpr (L_ + 2 <= 15) top_load<L_ + 2>(A);
    {
        const uint32_t r0 = A.ra[0], r1 = A.ra[1], r2 = A.ra[2], r3 = A.ra[3], r4 = A.ra[4], r5 = A.ra[5], r6 = A.ra[6], r7 = A.ra[7];
#define PR_RD(k) "ds_read_b64 v[%[x]+2*" #k ":%[x]+2*" #k "+1], %[r" #k "] offset:%[b]\n\t" \
                 "ds_read_b64 v[%[x]+16+2*" #k ":%[x]+16+2*" #k "+1], %[r" #k "] offset:%[b]+4096\n\t"
        // (issued as A0 B0 A1 B1 ...: chunk m of four outputs needs the first 4 (m + 1) reads)
        asm volatile(PR_RD(0) PR_RD(1) PR_RD(2) PR_RD(3) PR_RD(4) PR_RD(5) PR_RD(6) PR_RD(7)
                     :
                     : [x] "n"(TOP_WORK), [b] "n"(BUF), [r0] "v"(r0), [r1] "v"(r1), [r2] "v"(r2), [r3] "v"(r3), [r4] "v"(r4), [r5] "v"(r5),
                       [r6] "v"(r6), [r7] "v"(r7)
                     : "memory");
#undef PR_RD
    }
    top_chunk<KIND, L_, 0>(w, absmask, signmask);
    top_chunk<KIND, L_, 1>(w, absmask, signmask);
    top_chunk<KIND, L_, 2>(w, absmask, signmask);
    top_chunk<KIND, L_, 3>(w, absmask, signmask);
}
template <int KIND>
__device__ __forceinline__ void op_top(const TopAddr& A, const uint32_t* lds, int lane, uint32_t absmask, uint32_t signmask)
{
    top_load<0>(A);
    top_load<1>(A);
    uint32_t w[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if constexpr (KIND == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = lds[LDS_B8L + i * 64 + lane];
    }
    static_for<16>([&](auto Q) {
        constexpr int l = decltype(Q)::value;
        top_step<KIND, l>(A, w[l >> 1], absmask, signmask);
    });
}

#if PR_LOGN == 6
// ---- N = 64: the channel row IS local stage 5 (32 values per lane in v[158 .. 189]).  A row is 256 (int8: 64) bytes: both lanes of a pair
// load all of it (the same addresses: one fetch) into v[190 ..) and pick their pairs of positions -- no transposition through the LDS.
__device__ __forceinline__ void op_row_copy6(const float* r, uint64_t oddmask)
{
    static_for<16>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const float* rr = r;
        asm volatile("global_load_dwordx4 v[%0:%0+3], %1, off offset:%2" ::"n"(190 + 4 * i), "v"(rr), "n"(16 * i) : "memory");
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_for<32>([&](auto J) {  // local j = the pair j >> 1 of this lane: positions 4 (j >> 1) + 2 p + (j & 1)
        constexpr int j = decltype(J)::value;
        constexpr int e0 = ((j >> 1) << 2) | (j & 1);
        const uint64_t m = oddmask;
        asm volatile("v_cndmask_b32_e64 v[%0], v[%1], v[%2], %3" ::"n"(RS(5) + j), "n"(190 + e0), "n"(190 + e0 + 2), "s"(m));
    });
}
__device__ __forceinline__ void op_row8_copy6(const int8_t* r, uint32_t sh0, uint32_t sh1)
{
    static_for<4>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const int8_t* rr = r;
        asm volatile("global_load_dwordx4 v[%0:%0+3], %1, off offset:%2" ::"n"(190 + 4 * i), "v"(rr), "n"(16 * i) : "memory");
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_for<32>([&](auto J) {  // the dword j >> 1 holds four positions: this lane's two bytes at bit 16 p and 16 p + 8
        constexpr int j = decltype(J)::value;
        const uint32_t sh = (j & 1) ? sh1 : sh0;
        asm volatile("v_bfe_i32 v[%0], v[%1], %2, 8\n\tv_cvt_f32_i32 v[%0], v[%0]" ::"n"(RS(5) + j), "n"(190 + (j >> 1)), "v"(sh));
    });
}
#elif PR_LOGN < 10
// ---- N = 512 / 256 / 128: the channel row IS local stage 8 / 7 / 6 -- it goes, line by line through the same transposition, into the
// AGPRs / the lane's LDS column / v[190 ..) (the staging buffers of N = 256 sit in the rows of the stage-8 bits, which that length never uses)
constexpr int CP_LINES = (4 << PR_LOGN) / 128;                       // 128-byte lines per fp32 row
constexpr int STG_STRIDE = PR_LOGN == 8 ? 4096 : 8192;               // staging buffer of line L at STG_STRIDE * (L & 1)
// work registers of a line (N = 128: the destination is v[190 .. 253] itself, so they sit below it -- v[144 .. 159], and the converted
// int8 values in the bit registers v[104 .. 119], which nothing has written yet)
constexpr int CPX = PR_LOGN == 7 ? 144 : 192, CP8W = PR_LOGN == 7 ? 144 : 192, CP8A = PR_LOGN == 7 ? 104 : 208;
template <int O, int X>
__device__ __forceinline__ void copy_sink16(uint32_t col)  // v[X .. X + 15] = the local values O .. O + 15
{
    if constexpr (PR_LOGN == 9) {
        static_for<16>([&](auto I) { asm volatile("v_accvgpr_write_b32 a[%0], v[%1]" ::"n"(O + decltype(I)::value), "n"(X + decltype(I)::value)); });
    } else if constexpr (PR_LOGN == 8) {
        asm volatile("ds_write_b128 %[c], v[%[x]:%[x]+3] offset:%[o]\n\t"
                     "ds_write_b128 %[c], v[%[x]+4:%[x]+7] offset:%[o]+1024\n\t"
                     "ds_write_b128 %[c], v[%[x]+8:%[x]+11] offset:%[o]+2048\n\t"
                     "ds_write_b128 %[c], v[%[x]+12:%[x]+15] offset:%[o]+3072"
                     :
                     : [c] "v"(col), [x] "n"(X), [o] "n"(4 * LDS_S7 + (O / 4) * 1024)
                     : "memory");
    } else {
        static_for<16>([&](auto I) { asm volatile("v_mov_b32 v[%0], v[%1]" ::"n"(RS(6) + O + decltype(I)::value), "n"(X + decltype(I)::value)); });
    }
}
template <int L_>
__device__ __forceinline__ void top_copy_load(const TopAddr& A)
{
    constexpr int S = 128 + 32 * (L_ & 1);
    const float *r0 = A.rp[0], *r1 = A.rp[1], *r2 = A.rp[2], *r3 = A.rp[3];
    asm volatile("global_load_dwordx4 v[%[s]:%[s]+3], %[r0], off offset:%[o]\n\t"
                 "global_load_dwordx4 v[%[s]+4:%[s]+7], %[r1], off offset:%[o]\n\t"
                 "global_load_dwordx4 v[%[s]+8:%[s]+11], %[r2], off offset:%[o]\n\t"
                 "global_load_dwordx4 v[%[s]+12:%[s]+15], %[r3], off offset:%[o]"
                 :
                 : [s] "n"(S), [r0] "v"(r0), [r1] "v"(r1), [r2] "v"(r2), [r3] "v"(r3), [o] "n"(128 * L_)
                 : "memory");
}
template <int L_>
__device__ __forceinline__ void top_copy_step(const TopAddr& A, uint32_t col)
{
    constexpr int S = 128 + 32 * (L_ & 1), BUF = STG_STRIDE * (L_ & 1);
    {
        const uint32_t w0 = A.wa[0], w1 = A.wa[1], w2 = A.wa[2], w3 = A.wa[3];
        asm volatile("s_waitcnt vmcnt(%[wt])\n\t"
                     "ds_write_b128 %[w0], v[%[s]:%[s]+3] offset:%[b]\n\t"
                     "ds_write_b128 %[w1], v[%[s]+4:%[s]+7] offset:%[b]\n\t"
                     "ds_write_b128 %[w2], v[%[s]+8:%[s]+11] offset:%[b]\n\t"
                     "ds_write_b128 %[w3], v[%[s]+12:%[s]+15] offset:%[b]"
                     :
                     : [s] "n"(S), [b] "n"(BUF), [wt] "n"(L_ < CP_LINES - 1 ? 4 : 0), [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3)
                     : "memory");
    }
    if constexpr (L_ + 2 <= CP_LINES - 1) top_copy_load<L_ + 2>(A);
    {
        const uint32_t r0 = A.ra[0], r1 = A.ra[1], r2 = A.ra[2], r3 = A.ra[3], r4 = A.ra[4], r5 = A.ra[5], r6 = A.ra[6], r7 = A.ra[7];
#define PR_RD(k) "ds_read_b64 v[%[x]+2*" #k ":%[x]+2*" #k "+1], %[r" #k "] offset:%[b]\n\t"
        asm volatile(PR_RD(0) PR_RD(1) PR_RD(2) PR_RD(3) PR_RD(4) PR_RD(5) PR_RD(6) PR_RD(7) "s_waitcnt lgkmcnt(0)"
                     :
                     : [x] "n"(CPX), [b] "n"(BUF), [r0] "v"(r0), [r1] "v"(r1), [r2] "v"(r2), [r3] "v"(r3), [r4] "v"(r4),
                       [r5] "v"(r5), [r6] "v"(r6), [r7] "v"(r7)
                     : "memory");
#undef PR_RD
    }
    copy_sink16<16 * L_, CPX>(col);
}
__device__ __forceinline__ void op_top_copy(const TopAddr& A, uint32_t col)
{
    top_copy_load<0>(A);
    top_copy_load<1>(A);
    static_for<CP_LINES>([&](auto Q) { top_copy_step<decltype(Q)::value>(A, col); });
    if constexpr (PR_LOGN == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
#endif

// ---- the same pass for int8 rows (FastSscFipChar: a row is 1024 bytes = 8 lines) -----------------------------------------
// Line step LS = lines LS and LS + 4 of every row (positions 128 LS .. and 512 + 128 LS ..) through the same LDS image; BOTH
// lanes of a pair read the whole 128 bytes of their row (16-byte reads) and convert their own two bytes of every 32-bit word
// (v_bfe_i32 at bit 16 p / 16 p + 8, v_cvt_f32_i32); a quarter of a line (2 units) gives the local outputs
// 64 LS + 16 Q .. + 15 -> a[64 LS + 16 Q ..].  TopAddr::rp = byte pointers here (piece (lane & 7) of line 0), ra[u] = unit u.
constexpr int T8_W = 192, T8_A = 208, T8_B = 224;  // v[192..207] words (A 0..7, B 8..15), v[208..223] / v[224..239] the converted values
template <int LS>
__device__ __forceinline__ void top8_load(const TopAddr& A)
{
    constexpr int S = 128 + 32 * (LS & 1);
    const float *r0 = A.rp[0], *r1 = A.rp[1], *r2 = A.rp[2], *r3 = A.rp[3];
    asm volatile("global_load_dwordx4 v[%[s]:%[s]+3], %[r0], off offset:%[o]\n\t"
                 "global_load_dwordx4 v[%[s]+4:%[s]+7], %[r1], off offset:%[o]\n\t"
                 "global_load_dwordx4 v[%[s]+8:%[s]+11], %[r2], off offset:%[o]\n\t"
                 "global_load_dwordx4 v[%[s]+12:%[s]+15], %[r3], off offset:%[o]\n\t"
                 "global_load_dwordx4 v[%[s]+16:%[s]+19], %[r0], off offset:%[o]+512\n\t"
                 "global_load_dwordx4 v[%[s]+20:%[s]+23], %[r1], off offset:%[o]+512\n\t"
                 "global_load_dwordx4 v[%[s]+24:%[s]+27], %[r2], off offset:%[o]+512\n\t"
                 "global_load_dwordx4 v[%[s]+28:%[s]+31], %[r3], off offset:%[o]+512"
                 :
                 : [s] "n"(S), [r0] "v"(r0), [r1] "v"(r1), [r2] "v"(r2), [r3] "v"(r3), [o] "n"(128 * LS)
                 : "memory");
}
template <bool SAT, int KIND, int LS, int Q>
__device__ __forceinline__ void top8_quarter(const TopAddr& A, uint32_t sh0, uint32_t sh1, uint32_t w, uint32_t absmask, uint32_t signmask, const Sat& S)
{
    constexpr int BUF = 8192 * (LS & 1);
    {
        const uint32_t ra = A.ra[2 * Q], rb = A.ra[2 * Q + 1];
        asm volatile("ds_read_b128 v[%[w]:%[w]+3], %[ra] offset:%[b]\n\t"
                     "ds_read_b128 v[%[w]+4:%[w]+7], %[rb] offset:%[b]\n\t"
                     "ds_read_b128 v[%[w]+8:%[w]+11], %[ra] offset:%[b]+4096\n\t"
                     "ds_read_b128 v[%[w]+12:%[w]+15], %[rb] offset:%[b]+4096\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     :
                     : [w] "n"(T8_W), [b] "n"(BUF), [ra] "v"(ra), [rb] "v"(rb)
                     : "memory");
    }
    static_for<16>([&](auto K) {  // word k (0..7: line LS, 8..15: line LS + 4) -> two values each
        constexpr int k = decltype(K)::value;
        constexpr int D = (k < 8 ? T8_A : T8_B) + 2 * (k & 7);
        const uint32_t a0 = sh0, a1 = sh1;
        asm volatile("v_bfe_i32 v[%[d]], v[%[s]], %[a0], 8\n\t"
                     "v_bfe_i32 v[%[d]+1], v[%[s]], %[a1], 8\n\t"
                     "v_cvt_f32_i32 v[%[d]], v[%[d]]\n\t"
                     "v_cvt_f32_i32 v[%[d]+1], v[%[d]+1]"
                     :
                     : [d] "n"(D), [s] "n"(T8_W + k), [a0] "v"(a0), [a1] "v"(a1));
    });
    static_for<4>([&](auto M_) {
        constexpr int M = decltype(M_)::value;
        constexpr int X = T8_A + 4 * M, Y = T8_B + 4 * M, O = 64 * LS + 16 * Q + 4 * M;
        uint32_t t0, t1, t2, t3;
        const uint32_t am = absmask, sm = signmask, ww = w;
        if constexpr (KIND == 0) {
            asm volatile(PR_F4T("%[x]", "%[x]", "%[y]") : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3) : [x] "n"(X), [y] "n"(Y), [am] "s"(am));
            if constexpr (SAT) clamp_f<X, 4>(S, am);
        } else if constexpr (KIND == 1) {
            asm volatile("v_lshlrev_b32 %[t0], %[s0], %[w]\n\t"
                         "v_lshlrev_b32 %[t1], %[s0]-1, %[w]\n\t"
                         "v_lshlrev_b32 %[t2], %[s0]-2, %[w]\n\t"
                         "v_lshlrev_b32 %[t3], %[s0]-3, %[w]\n\t"
                         "v_and_or_b32 %[t0], %[t0], %[sm], 1.0\n\t"
                         "v_and_or_b32 %[t1], %[t1], %[sm], 1.0\n\t"
                         "v_and_or_b32 %[t2], %[t2], %[sm], 1.0\n\t"
                         "v_and_or_b32 %[t3], %[t3], %[sm], 1.0\n\t"
                         "v_fma_f32 v[%[x]], v[%[x]], %[t0], v[%[y]]\n\t"
                         "v_fma_f32 v[%[x]+1], v[%[x]+1], %[t1], v[%[y]+1]\n\t"
                         "v_fma_f32 v[%[x]+2], v[%[x]+2], %[t2], v[%[y]+2]\n\t"
                         "v_fma_f32 v[%[x]+3], v[%[x]+3], %[t3], v[%[y]+3]"
                         : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                         : [x] "n"(X), [y] "n"(Y), [w] "v"(ww), [s0] "n"(31 - (16 * (Q & 1) + 4 * M)), [sm] "s"(sm));
            if constexpr (SAT) clamp_g<X, 4>(S);
        } else {
            asm volatile("v_add_f32 v[%[x]], v[%[x]], v[%[y]]\n\t" "v_add_f32 v[%[x]+1], v[%[x]+1], v[%[y]+1]\n\t"
                         "v_add_f32 v[%[x]+2], v[%[x]+2], v[%[y]+2]\n\t" "v_add_f32 v[%[x]+3], v[%[x]+3], v[%[y]+3]" ::[x] "n"(X), [y] "n"(Y));
            if constexpr (SAT) clamp_g<X, 4>(S);
        }
        asm volatile("v_accvgpr_write_b32 a[%[o]], v[%[x]]\n\t" "v_accvgpr_write_b32 a[%[o]+1], v[%[x]+1]\n\t"
                     "v_accvgpr_write_b32 a[%[o]+2], v[%[x]+2]\n\t" "v_accvgpr_write_b32 a[%[o]+3], v[%[x]+3]" ::[o] "n"(O), [x] "n"(X));
        (void)t0; (void)t1; (void)t2; (void)t3;
    });
}
template <bool SAT, int KIND, int LS>
__device__ __forceinline__ void top8_step(const TopAddr& A, uint32_t sh0, uint32_t sh1, const uint32_t (&w)[8], uint32_t absmask, uint32_t signmask,
                                          const Sat& S)
{
    constexpr int SL = 128 + 32 * (LS & 1), BUF = 8192 * (LS & 1);
    {
        const uint32_t w0 = A.wa[0], w1 = A.wa[1], w2 = A.wa[2], w3 = A.wa[3];
        asm volatile("s_waitcnt vmcnt(%[wt])\n\t"
                     "ds_write_b128 %[w0], v[%[s]:%[s]+3] offset:%[b]\n\t"
                     "ds_write_b128 %[w1], v[%[s]+4:%[s]+7] offset:%[b]\n\t"
                     "ds_write_b128 %[w2], v[%[s]+8:%[s]+11] offset:%[b]\n\t"
                     "ds_write_b128 %[w3], v[%[s]+12:%[s]+15] offset:%[b]\n\t"
                     "ds_write_b128 %[w0], v[%[s]+16:%[s]+19] offset:%[b]+4096\n\t"
                     "ds_write_b128 %[w1], v[%[s]+20:%[s]+23] offset:%[b]+4096\n\t"
                     "ds_write_b128 %[w2], v[%[s]+24:%[s]+27] offset:%[b]+4096\n\t"
                     "ds_write_b128 %[w3], v[%[s]+28:%[s]+31] offset:%[b]+4096"
                     :
                     : [s] "n"(SL), [b] "n"(BUF), [wt] "n"(LS < 3 ? 8 : 0), [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3)
                     : "memory");
    }
    if constexpr (LS + 2 <= 3) top8_load<LS + 2>(A);
    top8_quarter<SAT, KIND, LS, 0>(A, sh0, sh1, w[2 * LS], absmask, signmask, S);
    top8_quarter<SAT, KIND, LS, 1>(A, sh0, sh1, w[2 * LS], absmask, signmask, S);
    top8_quarter<SAT, KIND, LS, 2>(A, sh0, sh1, w[2 * LS + 1], absmask, signmask, S);
    top8_quarter<SAT, KIND, LS, 3>(A, sh0, sh1, w[2 * LS + 1], absmask, signmask, S);
}
template <bool SAT, int KIND>
__device__ __forceinline__ void op_top8(const TopAddr& A, const uint32_t* lds, int lane, uint32_t absmask, uint32_t signmask, const Sat& S)
{
    top8_load<0>(A);
    top8_load<1>(A);
    uint32_t w[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if constexpr (KIND == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = lds[LDS_B8L + i * 64 + lane];
    }
    const uint32_t sh0 = 16u * ((uint32_t)lane & 1u), sh1 = sh0 + 8u;
    top8_step<SAT, KIND, 0>(A, sh0, sh1, w, absmask, signmask, S);
    top8_step<SAT, KIND, 1>(A, sh0, sh1, w, absmask, signmask, S);
    top8_step<SAT, KIND, 2>(A, sh0, sh1, w, absmask, signmask, S);
    top8_step<SAT, KIND, 3>(A, sh0, sh1, w, absmask, signmask, S);
}

#if PR_LOGN < 10 && PR_LOGN > 6
// ---- ... int8 rows (N bytes = 4 / 2 / 1 lines): the same, widened on the way ---------------------------------------------------------
constexpr int CP8_LINES = (1 << PR_LOGN) / 128;
template <int LS>
__device__ __forceinline__ void top8_copy_load(const TopAddr& A)
{
    constexpr int S = 128 + 32 * (LS & 1);
    const float *r0 = A.rp[0], *r1 = A.rp[1], *r2 = A.rp[2], *r3 = A.rp[3];
    asm volatile("global_load_dwordx4 v[%[s]:%[s]+3], %[r0], off offset:%[o]\n\t"
                 "global_load_dwordx4 v[%[s]+4:%[s]+7], %[r1], off offset:%[o]\n\t"
                 "global_load_dwordx4 v[%[s]+8:%[s]+11], %[r2], off offset:%[o]\n\t"
                 "global_load_dwordx4 v[%[s]+12:%[s]+15], %[r3], off offset:%[o]"
                 :
                 : [s] "n"(S), [r0] "v"(r0), [r1] "v"(r1), [r2] "v"(r2), [r3] "v"(r3), [o] "n"(128 * LS)
                 : "memory");
}
template <int LS, int Q>
__device__ __forceinline__ void top8_copy_quarter(const TopAddr& A, uint32_t sh0, uint32_t sh1, uint32_t col)
{
    constexpr int BUF = STG_STRIDE * (LS & 1);
    {
        const uint32_t ra = A.ra[2 * Q], rb = A.ra[2 * Q + 1];
        asm volatile("ds_read_b128 v[%[w]:%[w]+3], %[ra] offset:%[b]\n\t"
                     "ds_read_b128 v[%[w]+4:%[w]+7], %[rb] offset:%[b]\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     :
                     : [w] "n"(CP8W), [b] "n"(BUF), [ra] "v"(ra), [rb] "v"(rb)
                     : "memory");
    }
    static_for<8>([&](auto K) {  // word k -> this lane's two values
        constexpr int k = decltype(K)::value;
        const uint32_t a0 = sh0, a1 = sh1;
        asm volatile("v_bfe_i32 v[%[d]], v[%[s]], %[a0], 8\n\t"
                     "v_bfe_i32 v[%[d]+1], v[%[s]], %[a1], 8\n\t"
                     "v_cvt_f32_i32 v[%[d]], v[%[d]]\n\t"
                     "v_cvt_f32_i32 v[%[d]+1], v[%[d]+1]"
                     :
                     : [d] "n"(CP8A + 2 * k), [s] "n"(CP8W + k), [a0] "v"(a0), [a1] "v"(a1));
    });
    asm volatile("s_nop 0");
    copy_sink16<64 * LS + 16 * Q, CP8A>(col);
}
template <int LS>
__device__ __forceinline__ void top8_copy_step(const TopAddr& A, uint32_t sh0, uint32_t sh1, uint32_t col)
{
    constexpr int SL = 128 + 32 * (LS & 1), BUF = STG_STRIDE * (LS & 1);
    {
        const uint32_t w0 = A.wa[0], w1 = A.wa[1], w2 = A.wa[2], w3 = A.wa[3];
        asm volatile("s_waitcnt vmcnt(%[wt])\n\t"
                     "ds_write_b128 %[w0], v[%[s]:%[s]+3] offset:%[b]\n\t"
                     "ds_write_b128 %[w1], v[%[s]+4:%[s]+7] offset:%[b]\n\t"
                     "ds_write_b128 %[w2], v[%[s]+8:%[s]+11] offset:%[b]\n\t"
                     "ds_write_b128 %[w3], v[%[s]+12:%[s]+15] offset:%[b]"
                     :
                     : [s] "n"(SL), [b] "n"(BUF), [wt] "n"(LS < CP8_LINES - 1 ? 4 : 0), [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3)
                     : "memory");
    }
    if constexpr (LS + 2 <= CP8_LINES - 1) top8_copy_load<LS + 2>(A);
    top8_copy_quarter<LS, 0>(A, sh0, sh1, col);
    top8_copy_quarter<LS, 1>(A, sh0, sh1, col);
    top8_copy_quarter<LS, 2>(A, sh0, sh1, col);
    top8_copy_quarter<LS, 3>(A, sh0, sh1, col);
}
__device__ __forceinline__ void op_top8_copy(const TopAddr& A, int lane, uint32_t col)
{
    top8_copy_load<0>(A);
    if constexpr (CP8_LINES > 1) top8_copy_load<1>(A);
    const uint32_t sh0 = 16u * ((uint32_t)lane & 1u), sh1 = sh0 + 8u;
    static_for<CP8_LINES>([&](auto Q) { top8_copy_step<decltype(Q)::value>(A, sh0, sh1, col); });
    if constexpr (PR_LOGN == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
#endif

// ---- partial sums ---------------------------------------------------------------------------------------------------------
template <int T>
struct Words { static constexpr int N = T <= 5 ? 1 : (1 << (T - 5)); };
// the output bits of a block of local stage T: side 0 = left child (or root), 1 = right child
template <int T>
__device__ __forceinline__ void put_bits(uint32_t* lds, int lane, uint32_t side, const uint32_t (&w)[Words<T>::N])
{
    constexpr int NW = Words<T>::N;
    if constexpr (T <= 7) {
        if (side) {
            static_for<NW>([&](auto I) { wru<RB(1, T) + decltype(I)::value>(w[decltype(I)::value]); });
        } else {
            static_for<NW>([&](auto I) { wru<RB(0, T) + decltype(I)::value>(w[decltype(I)::value]); });
        }
    } else if constexpr (T == 8) {
        uint32_t* o = lds + (side ? LDS_B8R : LDS_B8L) + lane;
#pragma unroll
        for (int i = 0; i < NW; ++i) o[i * 64] = w[i];
    } else {
        uint32_t* o = lds + LDS_X + lane;
#pragma unroll
        for (int i = 0; i < NW; ++i) o[i * 64] = w[i];
    }
}
// the children's bits of a block of local stage T (T - 1 <= 7: registers, 8: LDS rows)
template <int T, int SIDE>
__device__ __forceinline__ void get_child(const uint32_t* lds, int lane, uint32_t (&w)[Words<T - 1>::N])
{
    constexpr int NW = Words<T - 1>::N;
    if constexpr (T - 1 <= 7) {
        static_for<NW>([&](auto I) { w[decltype(I)::value] = rdu<RB(SIDE, T - 1) + decltype(I)::value>(); });
    } else {
        const uint32_t* s = lds + (SIDE ? LDS_B8R : LDS_B8L) + lane;
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] = s[i * 64];
    }
}
// parent bits = [l ^ r | r] from the children's words (h = 2^(T-1) bits each)
template <int T>
__device__ __forceinline__ void join(const uint32_t (&l)[Words<T - 1>::N], const uint32_t (&r)[Words<T - 1>::N], uint32_t (&w)[Words<T>::N])
{
    constexpr int h = 1 << (T - 1);
    if constexpr (T <= 5) {
        w[0] = ((l[0] ^ r[0]) & ((1u << h) - 1u)) | (r[0] << h);
    } else {
        constexpr int NC = Words<T - 1>::N;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            w[i] = l[i] ^ r[i];
            w[NC + i] = r[i];
        }
    }
}
// COMBINE (avx_float.h:190-197) / COMBINE_0R (:199-204, ZERO: the left child was never decoded, its bits count as 0)
template <int T, bool ZERO>
__device__ __forceinline__ void op_combine(uint32_t* lds, int lane, uint32_t side)
{
    uint32_t l[Words<T - 1>::N], r[Words<T - 1>::N], w[Words<T>::N];
    get_child<T, 1>(lds, lane, r);
    if constexpr (ZERO) {
#pragma unroll
        for (int i = 0; i < Words<T - 1>::N; ++i) l[i] = 0;
    } else {
        get_child<T, 0>(lds, lane, l);
    }
    join<T>(l, r, w);
    put_bits<T>(lds, lane, side, w);
}
// signs of the registers v[R .. R+cnt), element i at bit i (cnt <= 32)
template <int R, int CNT>
__device__ __forceinline__ uint32_t sign_word()
{
    return sign_run<R, CNT>();
}
// signs of the block of local stage T held in v[R ..)
template <int T, int R>
__device__ __forceinline__ void sign_words(uint32_t (&w)[Words<T>::N])
{
    constexpr int m = 1 << T;
    if constexpr (m <= 32) {
        w[0] = sign_word<R, m>();
    } else {
        static_for<m / 32>([&](auto I) { w[decltype(I)::value] = sign_word<R + 32 * decltype(I)::value, 32>(); });
    }
}
// ROneNode::rightDecode :205-219 (T = 3 .. 6): lam = g(llr[T], left bits); bits = [left ^ sign(lam) | sign(lam)]
template <int T>
__device__ __forceinline__ void op_rone(uint32_t* lds, int lane, uint32_t side, uint32_t signmask)
{
    g_reg<T>(signmask);
    uint32_t l[Words<T - 1>::N], r[Words<T - 1>::N], w[Words<T>::N];
    sign_words<T - 1, RS(T - 1)>(r);
    get_child<T, 0>(lds, lane, l);
    join<T>(l, r, w);
    put_bits<T>(lds, lane, side, w);
}

// ... of a block whose LLRs sit in the LDS (T = 7): lam -> the registers of stage 6
__device__ __forceinline__ void op_rone7(uint32_t* lds, int lane, uint32_t lane16, uint32_t side, uint32_t absmask, uint32_t signmask)
{
    op_s7<1>(lane16, absmask, signmask);
    uint32_t l[2], r[2], w[4];
    sign_words<6, RS(6)>(r);
    get_child<7, 0>(lds, lane, l);
    join<7>(l, r, w);
    put_bits<7>(lds, lane, side, w);
}
// signs of the 128 values of the LDS column (stage 7), 32 per word: eight chunks at a time through v[128..159]
__device__ __forceinline__ void sign_words_s7(uint32_t lane16, uint32_t (&w)[4])
{
    static_for<4>([&](auto Wd) {
        constexpr int wd = decltype(Wd)::value;
        const uint32_t ad = lane16;
#define PR_RDC(c) "ds_read_b128 v[128+4*" #c ":128+4*" #c "+3], %[ad] offset:%[o]+1024*" #c "\n\t"
        asm volatile(PR_RDC(0) PR_RDC(1) PR_RDC(2) PR_RDC(3) PR_RDC(4) PR_RDC(5) PR_RDC(6) PR_RDC(7) "s_waitcnt lgkmcnt(0)"
                     :
                     : [ad] "v"(ad), [o] "n"(4 * LDS_S7 + 8192 * wd)
                     : "memory");
#undef PR_RDC
        w[wd] = sign_word<128, 32>();
    });
}
// ... in the AGPRs (T = 8): lam goes to the LDS column like the g of that stage, its signs are read back from there
template <bool CH>
__device__ __forceinline__ void op_rone8(uint32_t* lds, int lane, uint32_t lane16, uint32_t side, uint32_t absmask, uint32_t signmask, const Sat& S)
{
    op_s8<CH, 1>(lane16, absmask, signmask, S);
    uint32_t l[4], r[4], w[8];
    sign_words_s7(lane16, r);
    get_child<8, 0>(lds, lane, l);
    join<8>(l, r, w);
    put_bits<8>(lds, lane, side, w);
}
// s = (s << 1) | sign(a[I])
template <int I>
__device__ __forceinline__ void sgn_in_agpr(uint32_t& s)
{
    uint32_t t;
    asm volatile("v_accvgpr_read_b32 %[t], a[%[i]]\n\tv_alignbit_b32 %[s], %[s], %[t], 31" : [s] "+v"(s), [t] "=&v"(t) : [i] "n"(I));
}
// signs of the 256 AGPRs (stage 8)
__device__ __forceinline__ void sign_words_s8(uint32_t (&w)[8])
{
    static_for<8>([&](auto Wd) {
        constexpr int wd = decltype(Wd)::value;
        uint32_t s = 0;
        static_for<32>([&](auto I) { sgn_in_agpr<32 * wd + 31 - decltype(I)::value>(s); });
        w[wd] = s;
    });
}

// ---- leaves (local stage T <= 6: the block sits in v[RS(T) ..)) ----------------------------------------------------------
struct Lane {
    uint32_t* lds;
    int lane;
    uint32_t p;  // 0 / 1: which lane of the pair
};
template <int T>
__device__ __forceinline__ uint32_t maskw(int i)
{
    constexpr int m = 1 << T;
    return m >= 32 ? 0xFFFFFFFFu : (i == 0 ? ((1u << (m & 31)) - 1u) : 0u);
}
// the 8 accumulators of fold8 ("acc = 0; acc += x[8k + v]", avx_float.h:238-243): this lane's four -- acc[s] belongs to
// AVX lane (s & 1) + 4 (s >> 1) + 2 p
// Leaves above 128 positions (T = 7: the block is the LDS column, T = 8: the AGPRs; only in the BIG instantiations of the kernel,
// which the host picks for the codes that need them -- in the others their mere presence cost 10-13 %) work
// on 64 values at a time, brought into the registers of stage 6 (dead under a leaf): part Q = local elements 64 Q .. 64 Q + 63
template <int T>
struct Parts { static constexpr int N = T <= 6 ? 1 : (1 << (T - 6)); };
template <int T, int Q>
__device__ __forceinline__ void load_part(int lane)
{
    static_assert(T == 7 || T == 8, "stage");
    if constexpr (T == 7) {
        const uint32_t ad = (uint32_t)lane << 4;
#define PR_RDC(c) "ds_read_b128 v[%[d]+4*" #c ":%[d]+4*" #c "+3], %[ad] offset:%[o]+1024*" #c "\n\t"
        asm volatile(PR_RDC(0) PR_RDC(1) PR_RDC(2) PR_RDC(3) PR_RDC(4) PR_RDC(5) PR_RDC(6) PR_RDC(7) PR_RDC(8) PR_RDC(9) PR_RDC(10) PR_RDC(11)
                     PR_RDC(12) PR_RDC(13) PR_RDC(14) PR_RDC(15) "s_waitcnt lgkmcnt(0)"
                     :
                     : [d] "n"(RS(6)), [ad] "v"(ad), [o] "n"(4 * LDS_S7 + 16384 * Q)
                     : "memory");
#undef PR_RDC
    } else {
        static_for<64>([&](auto I) {
            constexpr int i = decltype(I)::value;
            asm volatile("v_accvgpr_read_b32 v[%0], a[%1]" ::"n"(RS(6) + i), "n"(64 * Q + i));
        });
    }
}
template <int T>
__device__ __forceinline__ void fold4(const Lane& L, float (&acc)[4])
{
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[s] = 0.0f;
    if constexpr (T <= 6) {
        add_run4<RS(T), (1 << T)>(acc);
    } else {
        static_for<Parts<T>::N>([&](auto Qc) {
            load_part<T, decltype(Qc)::value>(L.lane);
            add_run4<RS(6), 64>(acc);
        });
    }
}
// ... all eight, in both lanes, in the reference's order
__device__ __forceinline__ void gather8(const Lane& L, const float (&acc)[4], float (&a)[8])
{
    float pa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) pa[s] = pxf(acc[s]);
    const bool odd = L.p != 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float lo = odd ? pa[s] : acc[s], hi = odd ? acc[s] : pa[s];  // the even lane's value, the odd lane's value
        a[(s & 1) + 4 * (s >> 1)] = lo;
        a[(s & 1) + 4 * (s >> 1) + 2] = hi;
    }
}
// _mm256_spc_right4_ps avx_float.h:289-302 on four values, two per lane (t0 = in[2p], t1 = in[2p+1]): the output signs of
// this lane's two; flips EVERY position tied for the minimum
__device__ __forceinline__ void spc4_pair(float t0, float t1, uint32_t& r0, uint32_t& r1)
{
    const float a0 = absf(t0), a1 = absf(t1);
    float mm = a0 < a1 ? a0 : a1;
    const float om = pxf(mm);
    mm = om < mm ? om : mm;
    uint32_t par = sgn(t0) ^ sgn(t1);
    par ^= px(par);
    r0 = sgn(t0) ^ ((par && a0 == mm) ? 1u : 0u);
    r1 = sgn(t1) ^ ((par && a1 == mm) ? 1u : 0u);
}
// position e of a block for the local index j of lane p
__device__ __forceinline__ uint32_t real_index(uint32_t j, uint32_t p) { return ((j >> 1) << 2) | (p << 1) | (j & 1u); }

template <int T>
__device__ __forceinline__ void leaf_r0(const Lane& L, uint32_t side)
{
    uint32_t w[Words<T>::N];
#pragma unroll
    for (int i = 0; i < Words<T>::N; ++i) w[i] = 0;
    put_bits<T>(L.lds, L.lane, side, w);
}
template <int T>
__device__ __forceinline__ void leaf_r1(const Lane& L, uint32_t side)
{
    uint32_t w[Words<T>::N];
    sign_words<T, RS(T)>(w);
    put_bits<T>(L.lds, L.lane, side, w);
}
// RepetitionDecoder :273-287: every bit = the sign of the block's sum (fold8, then x0 + ... + x7 left to right)
template <int T>
__device__ __forceinline__ void leaf_rep(const Lane& L, uint32_t side)
{
    float acc[4], a[8];
    fold4<T>(L, acc);
    gather8(L, acc, a);
    const float s = ((((((a[0] + a[1]) + a[2]) + a[3]) + a[4]) + a[5]) + a[6]) + a[7];
    uint32_t w[Words<T>::N];
#pragma unroll
    for (int i = 0; i < Words<T>::N; ++i) w[i] = sgn(s) ? maskw<T>(i) : 0u;
    put_bits<T>(L.lds, L.lane, side, w);
}
// hard decisions w of v[R .. R+m), their parity over the pair, and -- when it is odd -- the flip of the FIRST position of the
// minimum magnitude (SpcDecoder :342-373, strict '<' = first index); returns the flip mask words
template <int T, int R>
__device__ __forceinline__ void spc_core(const Lane& L, uint32_t (&w)[Words<T>::N], uint32_t (&flip)[Words<T>::N])
{
    constexpr int m = 1 << T, NW = Words<T>::N;
    sign_words<T, R>(w);
    float mn = bitsf(0x7F800000u);
    amin_run<R, m>(mn);
    uint32_t par = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) par ^= (uint32_t)__builtin_popcount(w[i]);
    par &= 1u;
    par ^= px(par);
    const float om = pxf(mn);
    const float mall = om < mn ? om : mn;
    uint32_t e[NW];
    if constexpr (m <= 32) {
        e[0] = eq_run<R, m>(mall);
    } else {
        e[0] = eq_run<R, 32>(mall);
        e[1] = eq_run<R + 32, 32>(mall);
    }
    uint32_t j = 0xFFFFu;
    if constexpr (NW == 2) {
        if (e[1]) j = 32u + (uint32_t)__builtin_ctz(e[1]);
    }
    if (e[0]) j = (uint32_t)__builtin_ctz(e[0]);
    const uint32_t mine = j == 0xFFFFu ? 0xFFFFu : real_index(j, L.p);
    const uint32_t other = px(mine);
    const bool doit = par && mine < other;
#pragma unroll
    for (int i = 0; i < NW; ++i) flip[i] = (doit && (j >> 5) == (uint32_t)i) ? (1u << (j & 31u)) : 0u;
}
template <int T>
__device__ __forceinline__ void leaf_spc(const Lane& L, uint32_t side)
{
    uint32_t w[Words<T>::N], f[Words<T>::N];
    spc_core<T, RS(T)>(L, w, f);
#pragma unroll
    for (int i = 0; i < Words<T>::N; ++i) w[i] ^= f[i];
    put_bits<T>(L.lds, L.lane, side, w);
}
// ZeroSpcDecoder :503-546 (block of 2^(T+1) positions > 8): SPC on y = a + b, the output is the RIGHT half's signs, twice
template <int T>
__device__ __forceinline__ void leaf_zerospc(const Lane& L, uint32_t side)
{
    constexpr int hl = 1 << (T - 1);
    g0_reg<T>();  // y -> the registers of stage T - 1
    uint32_t ys[Words<T - 1>::N], f[Words<T - 1>::N], r[Words<T - 1>::N], w[Words<T>::N];
    spc_core<T - 1, RS(T - 1)>(L, ys, f);
    sign_words<T - 1, RS(T) + hl>(r);
#pragma unroll
    for (int i = 0; i < Words<T - 1>::N; ++i) r[i] ^= f[i];
    if constexpr (T <= 5) {
        w[0] = r[0] | (r[0] << hl);
    } else {
        w[0] = r[0];
        w[1] = r[0];
    }
    put_bits<T>(L.lds, L.lane, side, w);
}
// DoubleRepetitionDecoder :303-332: even positions = sign of the even sum, odd positions = sign of the odd sum
template <int T>
__device__ __forceinline__ void leaf_drep(const Lane& L, uint32_t side)
{
    float acc[4];
    fold4<T>(L, acc);
    float e, o;
    if constexpr (T >= 2) {  // e = (a0 + a4) + (a2 + a6): one bracket per lane
        const float eo = acc[0] + acc[2], oo = acc[1] + acc[3];
        e = eo + pxf(eo);
        o = oo + pxf(oo);
    } else {  // block of 4: e = ((a0 + a2) + a4) + a6 with a4 = a6 = +0
        e = ((acc[0] + pxf(acc[0])) + 0.0f) + 0.0f;
        o = ((acc[1] + pxf(acc[1])) + 0.0f) + 0.0f;
    }
    const uint32_t pat = (sgn(e) ? 0x55555555u : 0u) | (sgn(o) ? 0xAAAAAAAAu : 0u);
    uint32_t w[Words<T>::N];
#pragma unroll
    for (int i = 0; i < Words<T>::N; ++i) w[i] = pat & maskw<T>(i);
    put_bits<T>(L.lds, L.lane, side, w);
}
// DoubleSpcDecoderShort8 :473-488 (T = 2): per parity class, flip EVERY member tied for the class minimum
__device__ __forceinline__ void leaf_dspc8(const Lane& L, uint32_t side)
{
    const float x0 = rdf<RS(2)>(), x1 = rdf<RS(2) + 1>(), x2 = rdf<RS(2) + 2>(), x3 = rdf<RS(2) + 3>();
    auto cls = [&](float a, float b, uint32_t& fa, uint32_t& fb) {
        const float aa = absf(a), ab = absf(b);
        float mm = aa < ab ? aa : ab;
        const float om = pxf(mm);
        mm = om < mm ? om : mm;
        uint32_t par = sgn(a) ^ sgn(b);
        par ^= px(par);
        fa = sgn(a) ^ ((par && aa == mm) ? 1u : 0u);
        fb = sgn(b) ^ ((par && ab == mm) ? 1u : 0u);
    };
    uint32_t b0, b1, b2, b3;
    cls(x0, x2, b0, b2);
    cls(x1, x3, b1, b3);
    const uint32_t w[1] = { b0 | (b1 << 1) | (b2 << 2) | (b3 << 3) };
    put_bits<2>(L.lds, L.lane, side, w);
}
// DoubleSpcDecoder :425-466 (T >= 3): per AVX lane the LAST row with the lane minimum, per parity class the LOWEST AVX lane
// with the class minimum; that one position is flipped when the class parity is odd
template <int T>
__device__ __forceinline__ void leaf_dspc(const Lane& L, uint32_t side)
{
    constexpr int m = 1 << T, NW = Words<T>::N;
    uint32_t w[NW];
    sign_words<T, RS(T)>(w);
    float mv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) mv[s] = bitsf(0x7F800000u);
    static_for<m>([&](auto I) {
        constexpr int i = decltype(I)::value;
        amin_in<RS(T) + i>(mv[i & 3]);
    });
    uint32_t e[NW];
    if constexpr (m <= 32) {
        e[0] = 0;
        static_for<m>([&](auto I) {
            constexpr int i = m - 1 - decltype(I)::value;
            eq_in<RS(T) + i>(e[0], mv[i & 3]);
        });
    } else {
        static_for<NW>([&](auto Wd) {
            constexpr int wd = decltype(Wd)::value;
            e[wd] = 0;
            static_for<32>([&](auto I) {
                constexpr int i = 32 * wd + 31 - decltype(I)::value;
                eq_in<RS(T) + i>(e[wd], mv[i & 3]);
            });
        });
    }
    // last local index of slot s holding its minimum
    auto last = [&](int s) -> uint32_t {
        const uint32_t sm = 0x11111111u << s;
        uint32_t j = 31u - (uint32_t)__builtin_clz((e[0] & sm) | (1u << s));
        if constexpr (NW == 2) {
            if (e[1] & sm) j = 63u - (uint32_t)__builtin_clz(e[1] & sm);
        }
        return j;
    };
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        // class c: AVX lanes c + 2p (slot c) and c + 4 + 2p (slot c + 2) of this lane, the partner's in between
        float cm = mv[c] < mv[c + 2] ? mv[c] : mv[c + 2];
        const float om = pxf(cm);
        cm = om < cm ? om : cm;
        const uint32_t mine = mv[c] == cm ? (uint32_t)c + 2u * L.p : (mv[c + 2] == cm ? (uint32_t)c + 4u + 2u * L.p : 99u);
        const uint32_t other = px(mine);
        const uint32_t cmask = c ? 0xAAAAAAAAu : 0x55555555u;
        uint32_t par = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) par ^= (uint32_t)__builtin_popcount(w[i] & cmask);
        par &= 1u;
        par ^= px(par);
        if (par && mine < other) {
            const uint32_t j = mine >= 4u ? last(c + 2) : last(c);
#pragma unroll
            for (int i = 0; i < NW; ++i)
                if ((j >> 5) == (uint32_t)i) w[i] ^= 1u << (j & 31u);
        }
    }
    put_bits<T>(L.lds, L.lane, side, w);
}
// TripleRepetitionDecoder :572-589 (T >= 3): t[j] = a[j] + a[j+4], spc4, bit i = sign of r[i & 3]
template <int T>
__device__ __forceinline__ void leaf_trep(const Lane& L, uint32_t side)
{
    float acc[4];
    fold4<T>(L, acc);
    uint32_t r0, r1;
    spc4_pair(acc[0] + acc[2], acc[1] + acc[3], r0, r1);
    const uint32_t pat = (r0 ? 0x55555555u : 0u) | (r1 ? 0xAAAAAAAAu : 0u);
    uint32_t w[Words<T>::N];
#pragma unroll
    for (int i = 0; i < Words<T>::N; ++i) w[i] = pat & maskw<T>(i);
    put_bits<T>(L.lds, L.lane, side, w);
}
// TypeFiveDecoder :762-792 (FOLD: fold8 of the block, T >= 2) and RepetitionRateOneDecoderShort8 :718-739 (the block of 8
// itself, no spc4): q = f(l[j], l[j+4]); rho = (q0 + q1) + (q2 + q3); gg = g(l[j], l[j+4], sign rho); s = spc4(gg);
// bits 8i + j = sign rho ^ sign s[j], bits 8i + 4 + j = sign s[j]
template <int T, bool FOLD>
__device__ __forceinline__ void leaf_type5(const Lane& L, uint32_t side)
{
    float l4[4];
    if constexpr (FOLD) {
        fold4<T>(L, l4);
    } else {
        static_assert(T == 2, "block of 8");
        static_for<4>([&](auto I) { l4[decltype(I)::value] = rdf<RS(2) + decltype(I)::value>(); });
    }
    const float q0 = polar_f(l4[0], l4[2]), q1 = polar_f(l4[1], l4[3]);
    const float qs = q0 + q1;
    const float rho = qs + pxf(qs);
    const uint32_t sr = sgn(rho);
    const float g0 = polar_g(l4[0], l4[2], sr), g1 = polar_g(l4[1], l4[3], sr);
    uint32_t s0, s1;
    if constexpr (FOLD) {
        spc4_pair(g0, g1, s0, s1);
    } else {
        s0 = sgn(g0);
        s1 = sgn(g1);
    }
    const uint32_t nib = (sr ^ s0) | ((sr ^ s1) << 1) | (s0 << 2) | (s1 << 3);
    const uint32_t pat = nib * 0x11111111u;
    uint32_t w[Words<T>::N];
#pragma unroll
    for (int i = 0; i < Words<T>::N; ++i) w[i] = pat & maskw<T>(i);
    put_bits<T>(L.lds, L.lane, side, w);
}
// ZeroSpcDecoderShort8 :556-565 (T = 2): t[j] = x[j] + x[j+4], spc4, the four signs twice
__device__ __forceinline__ void leaf_zspc8(const Lane& L, uint32_t side)
{
    const float x0 = rdf<RS(2)>(), x1 = rdf<RS(2) + 1>(), x2 = rdf<RS(2) + 2>(), x3 = rdf<RS(2) + 3>();
    uint32_t r0, r1;
    spc4_pair(x0 + x2, x1 + x3, r0, r1);
    const uint32_t w[1] = { r0 | (r1 << 1) | (r0 << 2) | (r1 << 3) };
    put_bits<2>(L.lds, L.lane, side, w);
}

// ---- leaves of the int8 decoder (fastssc_fip_char.cpp:196-395; saturating sums in the reference's order) ------------------
__device__ __forceinline__ float sat8f(float v) { return __builtin_amdgcn_fmed3f(v, -128.0f, 127.0f); }
// the 32 byte-lane accumulators "LlrSum = adds(LlrSum, load(LlrIn + i))" (:219-234, a block of at most 32 positions is one
// zero-padded vector): this lane's sixteen -- v[s] belongs to byte lane 4 (s >> 1) + 2 p + (s & 1)
#define PR_V16_OPS [v0] "+v"(v[0]), [v1] "+v"(v[1]), [v2] "+v"(v[2]), [v3] "+v"(v[3]), [v4] "+v"(v[4]), [v5] "+v"(v[5]), [v6] "+v"(v[6]), [v7] "+v"(v[7]), \
                   [v8] "+v"(v[8]), [v9] "+v"(v[9]), [v10] "+v"(v[10]), [v11] "+v"(v[11]), [v12] "+v"(v[12]), [v13] "+v"(v[13]), [v14] "+v"(v[14]), [v15] "+v"(v[15])
// v[k & 15] = sat(v[k & 15] + v[RS(6) + k]), k < 64
__device__ __forceinline__ void satadd_part(float (&v)[16], float mm, float cc)
{
    asm volatile(PR_RUN_SATADD_ALL_64 : PR_V16_OPS : [r] "n"(RS(6)), [m] "v"(mm), [c] "s"(cc));
}
template <int T>
__device__ __forceinline__ void lane_sums16(const Lane& L, const Sat& S, float (&v)[16])
{
    constexpr int m = 1 << T;
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = 0.0f;
    const float cc = S.c127, mm = S.m128;
    if constexpr (m < 16) {
        static_for<m>([&](auto I) { v[decltype(I)::value] = rdf<RS(T) + decltype(I)::value>(); });
    } else if constexpr (T <= 6) {
        asm volatile(PR_RUN_LOAD_16 : PR_V16_OPS : [r] "n"(RS(T)));
        if constexpr (m == 32) asm volatile(PR_RUN_SATADD_16 : PR_V16_OPS : [r] "n"(RS(T)), [m] "v"(mm), [c] "s"(cc));
        if constexpr (m == 64) asm volatile(PR_RUN_SATADD_48 : PR_V16_OPS : [r] "n"(RS(T)), [m] "v"(mm), [c] "s"(cc));
    } else {  // above 128 positions: over the parts (sat(0 + x) = x, so the first sixteen need no special case)
        static_for<Parts<T>::N>([&](auto Qc) {
            load_part<T, decltype(Qc)::value>(L.lane);
            satadd_part(v, mm, cc);
        });
    }
}
// RepetitionDecoder :219-234 / ShortRepetitionDecoder :259-266: the sign of reduce_adds_epi8 (avxconvenience.h:92-101) over the
// lane sums -- pairs (l, l + 16), (l, l + 8), (l, l + 4) are local (slots s + 8, s + 4, s + 2), (l, l + 2) is the other lane's
// same slot, (0, 1) local again
template <int T>
__device__ __forceinline__ void leaf_rep_ch(const Lane& L, uint32_t side, const Sat& S)
{
    float v[16];
    lane_sums16<T>(L, S, v);
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = sat8f(v[s] + v[s + 8]);
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = sat8f(v[s] + v[s + 4]);
#pragma unroll
    for (int s = 0; s < 2; ++s) v[s] = sat8f(v[s] + v[s + 2]);
    const float d0 = sat8f(v[0] + pxf(v[0])), d1 = sat8f(v[1] + pxf(v[1]));
    const float r = sat8f(d0 + d1);
    uint32_t w[Words<T>::N];
#pragma unroll
    for (int i = 0; i < Words<T>::N; ++i) w[i] = sgn(r) ? maskw<T>(i) : 0u;
    put_bits<T>(L.lds, L.lane, side, w);
}
// DoubleRepetitionDecoder :242-257 (blocks of 32 positions and more): half_reduce_adds_epi8 (avxconvenience.h:202-210) -- every
// byte lane ends as the saturating butterfly sum of its parity class; even positions take the even class's sign
template <int T>
__device__ __forceinline__ void leaf_drep_ch(const Lane& L, uint32_t side, const Sat& S)
{
    float v[16], a[16], b[16];
    lane_sums16<T>(L, S, v);
#pragma unroll
    for (int s = 0; s < 16; ++s) a[s] = sat8f(v[s] + v[s ^ 8]);
#pragma unroll
    for (int s = 0; s < 16; ++s) b[s] = sat8f(a[s] + a[s ^ 4]);
    const float c0 = sat8f(b[0] + b[2]), c1 = sat8f(b[1] + b[3]);
    const float e = sat8f(c0 + pxf(c0)), o = sat8f(c1 + pxf(c1));
    const uint32_t pat = (sgn(e) ? 0x55555555u : 0u) | (sgn(o) ? 0xAAAAAAAAu : 0u);
    uint32_t w[Words<T>::N];
#pragma unroll
    for (int i = 0; i < Words<T>::N; ++i) w[i] = pat & maskw<T>(i);
    put_bits<T>(L.lds, L.lane, side, w);
}
// SpcDecoder :268-297 / ShortSpcDecoder :299-313 on the block in v[R ..) (M = log2 of its local length), NREAL = positions of
// the reference's block: hard decisions, and for odd parity the char negation of the first position of the minimum magnitude
// (|-128| = 128 sorts last; blocks above 32 positions search vector-wise from 127 -- nothing below 127: position 0; a single
// padded vector: a minimum above 127 means the 127 of the padding wins, no flip; 0 and -128 keep their sign)
template <int M, int R, int NREAL>
__device__ __forceinline__ void spc_core_ch(const Lane& L, uint32_t (&w)[Words<M>::N], uint32_t (&flip)[Words<M>::N])
{
    constexpr int m = 1 << M, NW = Words<M>::N;
    sign_words<M, R>(w);
    float mn = bitsf(0x7F800000u);
    amin_run<R, m>(mn);
    uint32_t par = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) par ^= (uint32_t)__builtin_popcount(w[i]);
    par &= 1u;
    par ^= px(par);
    const float om = pxf(mn);
    const float mall = om < mn ? om : mn;
    uint32_t e[NW];
    if constexpr (m <= 32) {
        e[0] = eq_run<R, m>(mall);
    } else {
        e[0] = eq_run<R, 32>(mall);
        e[1] = eq_run<R + 32, 32>(mall);
    }
    uint32_t j = 0xFFFFu;
    if constexpr (NW == 2) {
        if (e[1]) j = 32u + (uint32_t)__builtin_ctz(e[1]);
    }
    if (e[0]) j = (uint32_t)__builtin_ctz(e[0]);
    uint32_t mine = j == 0xFFFFu ? 0xFFFFu : real_index(j, L.p);
    bool doit = par != 0;
    if constexpr (NREAL > 32) {
        if (mall >= 127.0f) {  // position 0: local index 0 of the even lane
            j = 0;
            mine = L.p ? 0xFFFFu : 0u;
        }
    } else {
        if (mall > 127.0f) doit = false;
    }
    const uint32_t other = px(mine);
    doit = doit && mine < other;
    // the char negation keeps the sign of 0 and of -128
    const uint32_t sj = (w[NW == 2 ? (j >> 5) & 1u : 0] >> (j & 31u)) & 1u;
    float vj = mall;
    if constexpr (NREAL > 32) vj = (mall >= 127.0f) ? absf(rdf<R>()) : mall;
    doit = doit && vj != 0.0f && !(vj == 128.0f && sj);
#pragma unroll
    for (int i = 0; i < NW; ++i) flip[i] = (doit && (j >> 5) == (uint32_t)i) ? (1u << (j & 31u)) : 0u;
}
template <int T>
__device__ __forceinline__ void leaf_spc_ch(const Lane& L, uint32_t side)
{
    uint32_t w[Words<T>::N], f[Words<T>::N];
    spc_core_ch<T, RS(T), (2 << T)>(L, w, f);
#pragma unroll
    for (int i = 0; i < Words<T>::N; ++i) w[i] ^= f[i];
    put_bits<T>(L.lds, L.lane, side, w);
}
// ZeroSpcDecoder :315-353 / ShortZeroSpcDecoder :355-384: SPC on y = sat(a + b), the result twice
template <int T>
__device__ __forceinline__ void leaf_zerospc_ch(const Lane& L, uint32_t side, const Sat& S)
{
    constexpr int hl = 1 << (T - 1);
    g0_reg<T>();
    clamp_g<RS(T - 1), hl>(S);
    uint32_t ys[Words<T - 1>::N], f[Words<T - 1>::N], w[Words<T>::N];
    spc_core_ch<T - 1, RS(T - 1), (2 << T)>(L, ys, f);
#pragma unroll
    for (int i = 0; i < Words<T - 1>::N; ++i) ys[i] ^= f[i];
    if constexpr (T <= 5) {
        w[0] = ys[0] | (ys[0] << hl);
    } else {
        w[0] = ys[0];
        w[1] = ys[0];
    }
    put_bits<T>(L.lds, L.lane, side, w);
}
// SpcDecoder on a block above 128 positions (T = 7, 8), float (SpcDecoder :342-373) or int8 (:268-297, vector-wise search): two
// passes over the parts -- hard decisions and the minimum magnitude, then the first position of that minimum
template <bool CH, int T>
__device__ __forceinline__ void leaf_spc_big(const Lane& L, uint32_t side)
{
    constexpr int NW = Words<T>::N, NP = Parts<T>::N;
    uint32_t w[NW];
    float mn = bitsf(0x7F800000u), x0 = 0.0f;
    static_for<NP>([&](auto Qc) {
        constexpr int q = decltype(Qc)::value;
        load_part<T, q>(L.lane);
        if constexpr (q == 0) x0 = rdf<RS(6)>();
        w[2 * q] = sign_run<RS(6), 32>();
        w[2 * q + 1] = sign_run<RS(6) + 32, 32>();
        amin_run<RS(6), 64>(mn);
    });
    uint32_t par = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) par ^= (uint32_t)__builtin_popcount(w[i]);
    par &= 1u;
    par ^= px(par);
    const float om = pxf(mn);
    const float mall = om < mn ? om : mn;
    uint32_t j = 0xFFFFu;
    static_for<NP>([&](auto Qc) {
        constexpr int q = decltype(Qc)::value;
        load_part<T, q>(L.lane);
        const uint32_t e0 = eq_run<RS(6), 32>(mall), e1 = eq_run<RS(6) + 32, 32>(mall);
        const uint32_t jq = e0 ? (uint32_t)(64 * q) + (uint32_t)__builtin_ctz(e0) : (e1 ? (uint32_t)(64 * q + 32) + (uint32_t)__builtin_ctz(e1) : 0xFFFFu);
        j = j == 0xFFFFu ? jq : j;
    });
    uint32_t mine = j == 0xFFFFu ? 0xFFFFu : real_index(j, L.p);
    bool doit = par != 0;
    float vj = mall;
    if constexpr (CH) {
        if (mall >= 127.0f) {  // (vector-wise search from 127: nothing below -> position 0)
            j = 0;
            mine = L.p ? 0xFFFFu : 0u;
            vj = absf(x0);
        }
    }
    const uint32_t other = px(mine);
    doit = doit && mine < other;
    if constexpr (CH) {  // the char negation keeps the sign of 0 and of -128
        uint32_t sj = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) sj |= ((j >> 5) == (uint32_t)i) ? ((w[i] >> (j & 31u)) & 1u) : 0u;
        doit = doit && vj != 0.0f && !(vj == 128.0f && sj);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] ^= (doit && (j >> 5) == (uint32_t)i) ? (1u << (j & 31u)) : 0u;
    put_bits<T>(L.lds, L.lane, side, w);
}
// ShortZeroOneDecoder :386-395 on a block of 4 positions (G0R / RateOne / COMBINE0R of the plain program): both halves =
// sat(a + b), a and b sit in the two lanes
__device__ __forceinline__ void leaf_zr1(const Lane& L, uint32_t side)
{
    const float x0 = rdf<RS(1)>(), x1 = rdf<RS(1) + 1>();
    const float y0 = x0 + pxf(x0), y1 = x1 + pxf(x1);  // (the clamp does not change a sign)
    const uint32_t w[1] = { sgn(y0) | (sgn(y1) << 1) };
    put_bits<1>(L.lds, L.lane, side, w);
}

}  // namespace pair

namespace pair {
// the steps of the way down with the int8 decoder's saturation where CH is set
template <bool CH, int T>
__device__ __forceinline__ void dn_f(uint32_t absmask, const Sat& S)
{
    f_reg<T>(absmask);
    if constexpr (CH) clamp_f<RS(T - 1), (1 << (T - 1))>(S, absmask);
}
template <bool CH, int T>
__device__ __forceinline__ void dn_g(uint32_t signmask, const Sat& S)
{
    g_reg<T>(signmask);
    if constexpr (CH) clamp_g<RS(T - 1), (1 << (T - 1))>(S);
}
template <bool CH, int T>
__device__ __forceinline__ void dn_g0(const Sat& S)
{
    g0_reg<T>();
    if constexpr (CH) clamp_g<RS(T - 1), (1 << (T - 1))>(S);
}
template <bool CH, int KIND>
__device__ __forceinline__ void dn_s7(uint32_t lane16, uint32_t absmask, uint32_t signmask, const Sat& S)
{
    op_s7<KIND>(lane16, absmask, signmask);
    if constexpr (CH) {
        if constexpr (KIND == 0) clamp_f<RS(6), 64>(S, absmask);
        else clamp_g<RS(6), 64>(S);
    }
}
template <bool CH, bool I8, int KIND>
__device__ __forceinline__ void dn_top(const TopAddr& A, const uint32_t* lds, int lane, uint32_t absmask, uint32_t signmask, const Sat& S)
{
    if constexpr (I8) op_top8<CH, KIND>(A, lds, lane, absmask, signmask, S);  // (int8 rows; CH: with the int8 decoder's saturation)
    else op_top<KIND>(A, lds, lane, absmask, signmask);
}
// ROneNode at the root (T = 9): lam = g over the channel row -> the AGPRs, the code word = [left ^ sign(lam) | sign(lam)]
template <bool CH, bool I8>
__device__ __forceinline__ void op_rone9(const TopAddr& A, uint32_t* lds, int lane, uint32_t side, uint32_t absmask, uint32_t signmask, const Sat& S)
{
    dn_top<CH, I8, 1>(A, lds, lane, absmask, signmask, S);
    uint32_t l[8], r[8], w[16];
    sign_words_s8(r);
    get_child<9, 0>(lds, lane, l);
    join<9>(l, r, w);
    put_bits<9>(lds, lane, side, w);
}
}  // namespace pair

// =====================================================================================================================
//  kernel
// =====================================================================================================================
// CH: the int8 decoder's arithmetic (FastSscFipChar); I8: the rows are int8 (always with CH; without it the float decoder fed the
// decode_vector(const char*) way: the values converted on the way in, float arithmetic)
// BIG: with the leaves above 128 positions (a property of the code: the host picks the instantiation per decoder)
template <bool CH, bool I8, bool BIG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(pair::KV)))
void PR_KERNEL_NAME(const DecodeParams P)
{
    using namespace pair;
    // (dynamic: with a static 40 KB the compiler derives one wave per SIMD from the LDS size and then ignores the VGPR cap)
    extern __shared__ uint32_t lds[];
    asm volatile("" ::: "v255", "a255");  // (sizes the register file: all 256 + 256 registers of the lane)
    const int lane = (int)threadIdx.x;
    const uint32_t ngroups = (P.rows + 31u) >> 5;
    const uint32_t absmask = 0x7FFFFFFFu, signmask = 0x80000000u;
    const_u64_ptr prog = (const_u64_ptr)(const void*)P.prog;
    const uint32_t N = P.N;
    Lane L;
    L.lds = lds;
    L.lane = lane;
    L.p = (uint32_t)lane & 1u;
    const uint32_t lane16 = (uint32_t)lane << 4;
    Sat S;
    S.c127 = 127.0f;
    S.m128 = -128.0f;
    asm volatile("" : "+v"(S.m128));  // (kept in a VGPR: the clamp's scalar operand is the 127)
    TopAddr TA;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t r = ((uint32_t)lane >> 3) + 8u * (uint32_t)i, seg = (uint32_t)lane & 7u;
        TA.wa[i] = 4u * (PLOGN == 8 ? LDS_B8L : LDS_S7) + r * 128u + ((seg + r + (r >> 3)) & 7u) * 16u;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t r = (uint32_t)lane >> 1;
        TA.ra[k] = 4u * (PLOGN == 8 ? LDS_B8L : LDS_S7) + r * 128u + (((uint32_t)k + r + (r >> 3)) & 7u) * 16u + (I8 ? 0u : 8u * L.p);  // (int8 rows: both lanes read the whole unit)
    }

    for (uint32_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const uint32_t ci = grp * 32u + ((uint32_t)lane >> 1);
        const bool live = ci < P.rows;
        const uint32_t cj = live ? ci : grp * 32u;
        const uint32_t row = P.rowidx ? P.rowidx[cj] : cj;
        // the pieces of the channel rows this lane fetches for the wave (whole 128-byte lines, eight lanes per row)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t ri = grp * 32u + ((uint32_t)lane >> 3) + 8u * (uint32_t)i;
            const uint32_t rj = ri < P.rows ? ri : grp * 32u;
            const uint32_t rr = P.rowidx ? P.rowidx[rj] : rj;
            if constexpr (I8)  // (int8 rows: 16 positions per 16-byte piece)
                TA.rp[i] = reinterpret_cast<const float*>(reinterpret_cast<const int8_t*>(P.llr) + (size_t)rr * N + 16u * ((uint32_t)lane & 7u));
            else
                TA.rp[i] = reinterpret_cast<const float*>(P.llr) + (size_t)rr * N + 4u * ((uint32_t)lane & 7u);
        }

#if PR_LOGN == 6
        if constexpr (I8) op_row8_copy6(reinterpret_cast<const int8_t*>(P.llr) + (size_t)row * N, 16u * L.p, 16u * L.p + 8u);
        else op_row_copy6(reinterpret_cast<const float*>(P.llr) + (size_t)row * N, 0xAAAAAAAAAAAAAAAAull);
#elif PR_LOGN < 10
        if constexpr (I8) op_top8_copy(TA, lane, lane16);  // the channel row = the top local stage
        else op_top_copy(TA, lane16);
#endif
        uint64_t nraw = prog[0];
        for (uint32_t pc = 0;; ++pc) {
            const uint64_t raw = nraw;
            nraw = prog[pc + 1];  // (fetched one instruction ahead; the host appends a spare word behind OP_END)
            const uint32_t lo = (uint32_t)raw, hi = (uint32_t)(raw >> 32);
            if ((lo & 0xFFu) == OP_END) break;
#ifdef PAMD_DIAG
            uint64_t t_start = 0;
            if (P.cycles && grp == 0) t_start = __builtin_amdgcn_s_memtime();
#endif
            // PR_WALK (pair_program.cpp): everything between two leaves in one instruction -- the blocks finished on the way up
            // (COMBINE / COMBINE0R / RONE at consecutive stages c0 ..), the way down to the next leaf (G or G0R or F first, then
            // F / G0R at consecutive stages d1, d1 - 1, ...), the leaf.  Plain loops with a flat switch per step.
            uint32_t lk = (lo >> 8) & 15u, nup = (lo >> 16) & 15u, ndn = (lo >> 24) & 15u;
            const uint32_t c0 = (lo >> 12) & 15u, d1 = (lo >> 20) & 15u;
#ifdef PAMD_DIAG
            if (P.ablate & 8u) lk = 0;    // (cost hunt: no leaves / no way up / no way down -- results are wrong)
            if (P.ablate & 16u) nup = 0;
            if (P.ablate & 32u) ndn = 0;
#endif
            const uint32_t first = (lo >> 28) & 3u;  // first step down: 0 F, 1 G, 2 G0R
            const uint32_t upk = hi & 0xFFFFu;       // 2 bits per step up: 1 COMBINE, 2 COMBINE0R, 3 RONE
            const uint32_t dnk = hi >> 16;           // bit i: step i (> 0) down is a G0R (else F)
#pragma unroll 1
            for (uint32_t i = 0; i < nup; ++i) {
                // the finished block is a right child unless what follows takes it as a left child (RONE, the G, the end)
                const uint32_t side = (i + 1u < nup && ((upk >> (2u * i + 2u)) & 3u) != 3u) ? 1u : 0u;
                // (ONE flat switch per step: nested if / switch ladders make the compiler thread flag registers through a
                //  structured control flow graph -- dozens of scalar instructions per step)
                switch ((((upk >> (2u * i)) & 3u) << 4) | (c0 + i)) {
#if PR_LOGN == 10
                case 0x19: op_combine<9, false>(lds, lane, side); break;
                case 0x29: op_combine<9, true>(lds, lane, side); break;
                case 0x39: op_rone9<CH, I8>(TA, lds, lane, side, absmask, signmask, S); break;
#endif
                case 0x18: op_combine<8, false>(lds, lane, side); break;
                case 0x17: op_combine<7, false>(lds, lane, side); break;
                case 0x16: op_combine<6, false>(lds, lane, side); break;
                case 0x15: op_combine<5, false>(lds, lane, side); break;
                case 0x14: op_combine<4, false>(lds, lane, side); break;
                case 0x13: op_combine<3, false>(lds, lane, side); break;
                case 0x12: op_combine<2, false>(lds, lane, side); break;
                case 0x28: op_combine<8, true>(lds, lane, side); break;
                case 0x27: op_combine<7, true>(lds, lane, side); break;
                case 0x26: op_combine<6, true>(lds, lane, side); break;
                case 0x25: op_combine<5, true>(lds, lane, side); break;
                case 0x24: op_combine<4, true>(lds, lane, side); break;
                case 0x23: op_combine<3, true>(lds, lane, side); break;
                case 0x22: op_combine<2, true>(lds, lane, side); break;
                case 0x38: op_rone8<CH>(lds, lane, lane16, side, absmask, signmask, S); break;
                case 0x37: op_rone7(lds, lane, lane16, side, absmask, signmask); break;
                case 0x36: op_rone<6>(lds, lane, side, signmask); break;
                case 0x35: op_rone<5>(lds, lane, side, signmask); break;
                case 0x34: op_rone<4>(lds, lane, side, signmask); break;
                case 0x33: op_rone<3>(lds, lane, side, signmask); break;
                case 0x32: op_rone<2>(lds, lane, side, signmask); break;
                default: break;
                }
            }
            uint32_t lastk = 0;
#pragma unroll 1
            for (uint32_t i = 0; i < ndn; ++i) {
                const uint32_t kind = i == 0 ? first : (((dnk >> i) & 1u) ? 2u : 0u);
                lastk = kind;
#ifdef PAMD_DIAG
                if ((P.ablate & 64u) && d1 - i == 9u) continue;   // (cost hunt: no pass over the channel row / no stage 8 / no stage 7)
                if ((P.ablate & 128u) && d1 - i == 8u) continue;
                if ((P.ablate & 256u) && d1 - i == 7u) continue;
                if ((P.ablate & 512u) && d1 - i <= 6u) continue;
#endif
                switch ((kind << 4) | (d1 - i)) {
#if PR_LOGN == 10
                case 0x09: dn_top<CH, I8, 0>(TA, lds, lane, absmask, signmask, S); break;
                case 0x19: dn_top<CH, I8, 1>(TA, lds, lane, absmask, signmask, S); break;
                case 0x29: dn_top<CH, I8, 2>(TA, lds, lane, absmask, signmask, S); break;
#endif
                case 0x08: op_s8<CH, 0>(lane16, absmask, signmask, S); break;
                case 0x07: dn_s7<CH, 0>(lane16, absmask, signmask, S); break;
                case 0x06: dn_f<CH, 6>(absmask, S); break;
                case 0x05: dn_f<CH, 5>(absmask, S); break;
                case 0x04: dn_f<CH, 4>(absmask, S); break;
                case 0x03: dn_f<CH, 3>(absmask, S); break;
                case 0x02: dn_f<CH, 2>(absmask, S); break;
                case 0x18: op_s8<CH, 1>(lane16, absmask, signmask, S); break;
                case 0x17: dn_s7<CH, 1>(lane16, absmask, signmask, S); break;
                case 0x16: dn_g<CH, 6>(signmask, S); break;
                case 0x15: dn_g<CH, 5>(signmask, S); break;
                case 0x14: dn_g<CH, 4>(signmask, S); break;
                case 0x13: dn_g<CH, 3>(signmask, S); break;
                case 0x12: dn_g<CH, 2>(signmask, S); break;
                case 0x28: op_s8<CH, 2>(lane16, absmask, signmask, S); break;
                case 0x27: dn_s7<CH, 2>(lane16, absmask, signmask, S); break;
                case 0x26: dn_g0<CH, 6>(S); break;
                case 0x25: dn_g0<CH, 5>(S); break;
                case 0x24: dn_g0<CH, 4>(S); break;
                case 0x23: dn_g0<CH, 3>(S); break;
                case 0x22: dn_g0<CH, 2>(S); break;
                default: break;
                }
            }
            if (lk) {  // the leaf the way down ends at: its stage is one below the last step, it is a left child behind an F
                const uint32_t side = lastk != 0u ? 1u : 0u;
#define PR_LF(K_, T_) (((K_) - PR_R0 + 1) << 3 | (T_))
                const uint32_t lst = d1 - ndn, lsd = side;
                bool done = false;
                if constexpr (BIG) {
                    if (lst >= 7u) {  // leaves of 256 / 512 positions: the block is the LDS column / the AGPRs
                        done = true;
                        const uint32_t bcode = (lk << 1) | (lst & 1u);
#define PR_BL(K_, T_) ((((K_) - PR_R0 + 1) << 1) | ((T_) & 1))
                        if constexpr (CH) {
                            switch (bcode) {
                            case PR_BL(PR_REP, 7): leaf_rep_ch<7>(L, lsd, S); break;
                            case PR_BL(PR_REP, 8): leaf_rep_ch<8>(L, lsd, S); break;
                            case PR_BL(PR_SPC, 7): leaf_spc_big<true, 7>(L, lsd); break;
                            case PR_BL(PR_SPC, 8): leaf_spc_big<true, 8>(L, lsd); break;
                            case PR_BL(PR_DREP, 7): leaf_drep_ch<7>(L, lsd, S); break;
                            case PR_BL(PR_DREP, 8): leaf_drep_ch<8>(L, lsd, S); break;
                            default: break;
                            }
                        } else {
                            switch (bcode) {
                            case PR_BL(PR_REP, 7): leaf_rep<7>(L, lsd); break;
                            case PR_BL(PR_REP, 8): leaf_rep<8>(L, lsd); break;
                            case PR_BL(PR_SPC, 7): leaf_spc_big<false, 7>(L, lsd); break;
                            case PR_BL(PR_SPC, 8): leaf_spc_big<false, 8>(L, lsd); break;
                            case PR_BL(PR_DREP, 7): leaf_drep<7>(L, lsd); break;
                            case PR_BL(PR_DREP, 8): leaf_drep<8>(L, lsd); break;
                            case PR_BL(PR_TREP, 7): leaf_trep<7>(L, lsd); break;
                            case PR_BL(PR_TREP, 8): leaf_trep<8>(L, lsd); break;
                            case PR_BL(PR_TYPE5, 7): leaf_type5<7, true>(L, lsd); break;
                            case PR_BL(PR_TYPE5, 8): leaf_type5<8, true>(L, lsd); break;
                            default: break;
                            }
                        }
#undef PR_BL
                    }
                }
                if (!done) {
                if constexpr (CH) {
                    switch ((lk << 3) | lst) {
                    case PR_LF(PR_R0, 1): leaf_r0<1>(L, lsd); break;
                    case PR_LF(PR_R0, 2): leaf_r0<2>(L, lsd); break;
                    case PR_LF(PR_R1, 1): leaf_r1<1>(L, lsd); break;
                    case PR_LF(PR_R1, 2): leaf_r1<2>(L, lsd); break;
#define PR_LEAF_1_6(K_, FN, ...) case PR_LF(K_, 1): FN<1>(L, lsd, ##__VA_ARGS__); break; case PR_LF(K_, 2): FN<2>(L, lsd, ##__VA_ARGS__); break; \
                                 case PR_LF(K_, 3): FN<3>(L, lsd, ##__VA_ARGS__); break; case PR_LF(K_, 4): FN<4>(L, lsd, ##__VA_ARGS__); break; \
                                 case PR_LF(K_, 5): FN<5>(L, lsd, ##__VA_ARGS__); break; case PR_LF(K_, 6): FN<6>(L, lsd, ##__VA_ARGS__); break;
                    PR_LEAF_1_6(PR_REP, leaf_rep_ch, S)
                    PR_LEAF_1_6(PR_SPC, leaf_spc_ch)
#undef PR_LEAF_1_6
                    case PR_LF(PR_DREP, 4): leaf_drep_ch<4>(L, lsd, S); break;
                    case PR_LF(PR_DREP, 5): leaf_drep_ch<5>(L, lsd, S); break;
                    case PR_LF(PR_DREP, 6): leaf_drep_ch<6>(L, lsd, S); break;
                    case PR_LF(PR_ZEROSPC, 2): leaf_zerospc_ch<2>(L, lsd, S); break;
                    case PR_LF(PR_ZEROSPC, 3): leaf_zerospc_ch<3>(L, lsd, S); break;
                    case PR_LF(PR_ZEROSPC, 4): leaf_zerospc_ch<4>(L, lsd, S); break;
                    case PR_LF(PR_ZEROSPC, 5): leaf_zerospc_ch<5>(L, lsd, S); break;
                    case PR_LF(PR_ZEROSPC, 6): leaf_zerospc_ch<6>(L, lsd, S); break;
                    case PR_LF(PR_ZR1, 1): leaf_zr1(L, lsd); break;
                    default: break;
                    }
                } else {
                switch ((lk << 3) | lst) {
                case PR_LF(PR_R0, 1): leaf_r0<1>(L, side); break;
                case PR_LF(PR_R0, 2): leaf_r0<2>(L, side); break;
                case PR_LF(PR_R1, 1): leaf_r1<1>(L, side); break;
                case PR_LF(PR_R1, 2): leaf_r1<2>(L, side); break;
#define PR_LEAF_1_6(K_, FN) case PR_LF(K_, 1): FN<1>(L, side); break; case PR_LF(K_, 2): FN<2>(L, side); break; case PR_LF(K_, 3): FN<3>(L, side); break; \
                            case PR_LF(K_, 4): FN<4>(L, side); break; case PR_LF(K_, 5): FN<5>(L, side); break; case PR_LF(K_, 6): FN<6>(L, side); break;
#define PR_LEAF_3_6(K_, FN) case PR_LF(K_, 3): FN<3>(L, side); break; case PR_LF(K_, 4): FN<4>(L, side); break; case PR_LF(K_, 5): FN<5>(L, side); break; \
                            case PR_LF(K_, 6): FN<6>(L, side); break;
                PR_LEAF_1_6(PR_REP, leaf_rep)
                PR_LEAF_1_6(PR_SPC, leaf_spc)
                PR_LEAF_1_6(PR_DREP, leaf_drep)
                PR_LEAF_3_6(PR_DSPC, leaf_dspc)
                PR_LEAF_3_6(PR_TREP, leaf_trep)
                PR_LEAF_3_6(PR_ZEROSPC, leaf_zerospc)
                case PR_LF(PR_DSPC8, 2): leaf_dspc8(L, side); break;
                case PR_LF(PR_TYPE5, 2): leaf_type5<2, true>(L, side); break;
                case PR_LF(PR_TYPE5, 3): leaf_type5<3, true>(L, side); break;
                case PR_LF(PR_TYPE5, 4): leaf_type5<4, true>(L, side); break;
                case PR_LF(PR_TYPE5, 5): leaf_type5<5, true>(L, side); break;
                case PR_LF(PR_TYPE5, 6): leaf_type5<6, true>(L, side); break;
                case PR_LF(PR_REPONE8, 2): leaf_type5<2, false>(L, side); break;
                case PR_LF(PR_ZSPC8, 2): leaf_zspc8(L, side); break;
                default: break;
                }
                }
#undef PR_LEAF_1_6
#undef PR_LEAF_3_6
#undef PR_LF
                }
            }
#ifdef PAMD_DIAG
            if (P.cycles && grp == 0) {  // diagnostics: cycles this instruction took for the first group
                const uint64_t t_end = __builtin_amdgcn_s_memtime();
                if (lane == 0) P.cycles[pc] = (uint32_t)(t_end - t_start);
            }
#endif
        }

#ifdef PAMD_DIAG
        if (P.ablate & 4u) continue;  // (cost hunt: no epilogue)
#endif
        // ---- output: FastSscAvxFloat::decode :940-955 -------------------------------------------------------------------
        // this lane's half of the code word, local order: 16 words (N = 512: 8, the root's block is a stage-8 left block)
        uint32_t x[NWL];
        if constexpr (PLOGN >= 9) {
#pragma unroll
            for (int i = 0; i < NWL; ++i) x[i] = lds[(PLOGN == 10 ? LDS_X : LDS_B8L) + i * 64 + lane];
        } else {  // (N = 256 / 128: the root's block is a register block -- 4 / 2 words)
            static_for<NWL>([&](auto I) { x[decltype(I)::value] = rdu<RB(0, PLOGN - 1) + decltype(I)::value>(); });
        }
        if (!P.systematic) {  // re-encode u = x G (butterfly_fip.cpp:15-63) in the local order: position distance 1 = local bit
                              // distance 1, 2 = the other lane, 2^k (k >= 2) = local distance 2^(k-1)
#pragma unroll
            for (int i = 0; i < NWL; ++i) {
                uint32_t v = x[i];
                v ^= (v >> 1) & 0x55555555u;
                const uint32_t o = px(v);
                v ^= L.p ? 0u : o;
                v ^= (v >> 2) & 0x33333333u;
                v ^= (v >> 4) & 0x0F0F0F0Fu;
                v ^= (v >> 8) & 0x00FF00FFu;
                v ^= (v >> 16) & 0x0000FFFFu;
                x[i] = v;
            }
#pragma unroll
            for (int hw = 1; hw < NWL; hw <<= 1) {
#pragma unroll
                for (int i = 0; i < NWL; ++i)
                    if (!(i & hw)) x[i] ^= x[i + hw];
            }
        }
        // Interleave: lane p takes the code words 16 p .. 16 p + 15 (positions 512 p ..; N = 512: 8 p .., 256 p ..): word 2k + hf of them =
        // the half hf of the local word 8 p + k of BOTH lanes, pairs of bits alternating
        uint32_t rw[NWL];
        {
            const bool odd = L.p != 0;
            auto spread = [](uint32_t v) {  // 16 bits -> pairs of bits to the low half of every nibble
                v = (v | (v << 8)) & 0x00FF00FFu;
                v = (v | (v << 4)) & 0x0F0F0F0Fu;
                return (v | (v << 2)) & 0x33333333u;
            };
            if constexpr (NWL == 1) {  // (N = 64: lane p takes code word p = the half p of the one local word of both lanes)
                const uint32_t recv = px(x[0]);
                const uint32_t ev = odd ? recv : x[0], ov = odd ? x[0] : recv;
                rw[0] = odd ? (spread(ev >> 16) | (spread(ov >> 16) << 2)) : (spread(ev & 0xFFFFu) | (spread(ov & 0xFFFFu) << 2));
            }
#pragma unroll
            for (int k = 0; k < NWL / 2; ++k) {
                const uint32_t keep = odd ? x[NWL / 2 + k] : x[k], send = odd ? x[k] : x[NWL / 2 + k];
                const uint32_t recv = px(send);
                const uint32_t ev = odd ? recv : keep, ov = odd ? keep : recv;  // the even lane's word, the odd lane's word
                rw[2 * k] = spread(ev & 0xFFFFu) | (spread(ov & 0xFFFFu) << 2);
                rw[2 * k + 1] = spread(ev >> 16) | (spread(ov >> 16) << 2);
            }
        }
        // information bits, packed like the output (bit t of the information word = bit 7 - t % 8 of byte t / 8): per 32 code
        // bits a log-step bit compress with host-made move masks (program.cpp, deposit_table), strung together in a 64-bit
        // shift register; both lanes OR their words into the LDS rows [word][codeword of the wave] (the lanes meet in the
        // word where the second half of the code word starts)
        WAVE_SYNC();
        uint32_t* il = lds + LDS_OUT;
        const uint32_t cw = (uint32_t)lane >> 1;
        const uint32_t nwords = (P.nb + 3u) >> 2;
        for (uint32_t j = L.p; j <= nwords; j += 2) il[j * 32 + cw] = 0;
        WAVE_SYNC();
        {
            const uint4* ext = reinterpret_cast<const uint4*>(P.ext) + 2u * (uint32_t)NWL * L.p;  // (two uint4 per code word of 32 bits)
            uint64_t acc = 0;
            uint32_t cur = 0;
#pragma unroll
            for (int i = 0; i < NWL; ++i) {
                const uint4 ea = ext[2 * i], eb = ext[2 * i + 1];  // mask, base, mv0, mv1 | mv2, mv3, mv4, pad
                const uint32_t mask = ea.x, base = ea.y;
                const uint32_t mv[5] = { ea.z, ea.w, eb.x, eb.y, eb.z };
                uint32_t v = rw[i] & mask;
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    const uint32_t t = v & mv[s];
                    v = (v ^ t) | (t >> (1u << s));
                }
                if (i == 0) cur = base;
                acc |= (uint64_t)v << (base & 31u);
                const uint32_t end = base + (uint32_t)__builtin_popcount(mask);
                if ((end >> 5) != (base >> 5)) {
                    atomicOr(&il[(base >> 5) * 32 + cw], __builtin_bswap32(__brev((uint32_t)acc)));
                    acc >>= 32;
                }
                cur = end;
            }
            if (cur & 31u) atomicOr(&il[(cur >> 5) * 32 + cw], __builtin_bswap32(__brev((uint32_t)acc)));
        }
        WAVE_SYNC();
        // detector verdict (XOR of the syndrome words of the output bytes) and the output row: the lanes take alternate words
        uint32_t syn = 0;
        uint8_t* orow = P.out + (size_t)row * P.nb;
        for (uint32_t j = L.p; j < nwords; j += 2) {
            const uint32_t word = il[j * 32 + cw];
            if (P.det_on) {
                const uint32_t* tb = P.syn_bytes + j * 1024u;
                syn ^= tb[word & 255u];
                if (4 * j + 1 < P.nb) syn ^= tb[256 + ((word >> 8) & 255u)];
                if (4 * j + 2 < P.nb) syn ^= tb[512 + ((word >> 16) & 255u)];
                if (4 * j + 3 < P.nb) syn ^= tb[768 + (word >> 24)];
            }
#ifdef PAMD_DIAG
            if (P.ablate & 1u) continue;  // (fault hunt: no output)
#endif
            if (live) {
                if (4 * j + 4 <= P.nb && (P.nb & 3u) == 0) {
                    reinterpret_cast<uint32_t*>(orow)[j] = word;
                } else {
                    for (uint32_t b = 0; b < 4 && 4 * j + b < P.nb; ++b) orow[4 * j + b] = (uint8_t)(word >> (8u * b));
                }
            }
        }
        syn ^= px(syn);
#ifdef PAMD_DIAG
        if (!(P.ablate & 1u))
#endif
        if (live && L.p == 0 && P.ok) P.ok[row] = (syn == 0) ? 1 : 0;
        WAVE_SYNC();
    }
}

#if PR_LOGN == 10
// (explicit instantiations in this order: the code object then starts with the kernels of the BASELINE configurations, and what is
//  added behind them does not move them -- their speed depends on where their code sits)
template __global__ void sc_pair_kernel<false, false, false>(const DecodeParams);
template __global__ void sc_pair_kernel<true, true, false>(const DecodeParams);
template __global__ void sc_pair_kernel<false, true, false>(const DecodeParams);
template __global__ void sc_pair_kernel<false, false, true>(const DecodeParams);
template __global__ void sc_pair_kernel<true, true, true>(const DecodeParams);
template __global__ void sc_pair_kernel<false, true, true>(const DecodeParams);

template <bool BIG>
static void launch_sc_pair_t(const DecodeParams& P, bool i8rows, unsigned grid, hipStream_t st)
{
    if (P.chr) hipLaunchKernelGGL((sc_pair_kernel<true, true, BIG>), dim3(grid), dim3(64), pair::LDS_WORDS * 4, st, P);
    else if (i8rows) hipLaunchKernelGGL((sc_pair_kernel<false, true, BIG>), dim3(grid), dim3(64), pair::LDS_WORDS * 4, st, P);
    else hipLaunchKernelGGL((sc_pair_kernel<false, false, BIG>), dim3(grid), dim3(64), pair::LDS_WORDS * 4, st, P);
}
hipError_t launch_sc_pair(const DecodeParams& P, bool i8rows, bool big, unsigned grid, hipStream_t st)
{
    if (big) launch_sc_pair_t<true>(P, i8rows, grid, st);
    else launch_sc_pair_t<false>(P, i8rows, grid, st);
    return hipGetLastError();
}

#else
// N = 512 / 256 / 128 / 64
template __global__ void PR_KERNEL_NAME<false, false, false>(const DecodeParams);
template __global__ void PR_KERNEL_NAME<true, true, false>(const DecodeParams);
template __global__ void PR_KERNEL_NAME<false, true, false>(const DecodeParams);
template __global__ void PR_KERNEL_NAME<false, false, true>(const DecodeParams);
template __global__ void PR_KERNEL_NAME<true, true, true>(const DecodeParams);
template __global__ void PR_KERNEL_NAME<false, true, true>(const DecodeParams);
template <bool BIG>
static void launch_sc_pair_small_t(const DecodeParams& P, bool i8rows, unsigned grid, hipStream_t st)
{
    if (P.chr) hipLaunchKernelGGL((PR_KERNEL_NAME<true, true, BIG>), dim3(grid), dim3(64), pair::LDS_WORDS * 4, st, P);
    else if (i8rows) hipLaunchKernelGGL((PR_KERNEL_NAME<false, true, BIG>), dim3(grid), dim3(64), pair::LDS_WORDS * 4, st, P);
    else hipLaunchKernelGGL((PR_KERNEL_NAME<false, false, BIG>), dim3(grid), dim3(64), pair::LDS_WORDS * 4, st, P);
}
hipError_t PR_LAUNCH_NAME(const DecodeParams& P, bool i8rows, bool big, unsigned grid, hipStream_t st)
{
    if (P.logN != (uint32_t)pair::PLOGN) return hipErrorInvalidValue;
    if (big) launch_sc_pair_small_t<true>(P, i8rows, grid, st);
    else launch_sc_pair_small_t<false>(P, i8rows, grid, st);
    return hipGetLastError();
}
#endif

}  // namespace pamd
