// Batched polar decoders for MI355X (gfx950, wave64): device helpers shared by the SC and the list kernels.
// A wavefront decodes a few codewords side by side:
// every wave interprets the same flat decode program (program.h), so control flow is uniform
// across the batch and there is no inter-wave synchronisation at all (no __syncthreads).
//
//   sc_decode_kernel   Fast-SSC  (reference: src/polarcode/decoding/fastssc_avx_float.cpp), sc_kernel.hip
//                      16 lanes work on one codeword (four codewords per wave); LLR stage buffers + packed
//                      partial sums in LDS, the upper stages of long blocks in a global arena.
//   scl_decode_kernel  Fast-SSCL (reference: src/polarcode/decoding/scl_avx_float.cpp), scl_kernel.h
//                      the wave is split into 64/LP path groups of LP lanes, shared by 1, 2 or 4 codewords;
//                      per-path LLR stage buffers in LDS (upper stages in a global arena), partial sums
//                      bit-packed, the reference's DataPool lazy copy replaced by one "slot word" per path
//                      that is moved with a single cross-lane shuffle at every fork.
//
// Bit-exactness notes: f is exact; g is one fp32 add; the metric sums follow the
// reference's "8 strided accumulators, then x0+..+x7" order; every selection reproduces the
// reference's swap-mutating selection sorts (arrayfuncs.h:161-231) including their tie behaviour.
// Compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kernels.h"

namespace pamd {

#define WAVE_SYNC()                                            \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

// the decode program is read-only for the whole launch: reading it through the constant address space lets
// the compiler use scalar loads (s_load_dwordx2 through the scalar cache) instead of per-lane vector loads
typedef const uint64_t __attribute__((address_space(4))) * const_u64_ptr;

__device__ __forceinline__ uint32_t fbits(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float bitsf(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t sgn(float x) { return __float_as_uint(x) >> 31; }
__device__ __forceinline__ float absf(float x) { return __uint_as_float(__float_as_uint(x) & 0x7FFFFFFFu); }

// _mm256_polarf_ps avx_float.h:55-63 : sign = xor of signs, magnitude = min of magnitudes
#ifdef PAMD_POLAR_F_MINF
// Three VALU instructions: v_min_f32 with |.| source modifiers (exact on magnitudes: the kernels run with fp32
// denormals preserved), v_xor_b32, v_and_or_b32.  Measured on one box: cfg 3 (4 lanes per path x 2 codewords)
// +1.3 %, but SC -2.3 % and cfg 5 -3.8 % (the asm statement keeps the compiler from merging the |.| with its other
// users there), so only the translation units that define the macro take this form.
__device__ __forceinline__ float polar_f(float a, float b)
{
    float m;
    asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(m) : "v"(a), "v"(b));
    return bitsf(((fbits(a) ^ fbits(b)) & 0x80000000u) | fbits(m));
}
#else
__device__ __forceinline__ float polar_f(float a, float b)
{
    const uint32_t ua = fbits(a), ub = fbits(b);
    const uint32_t ma = ua & 0x7FFFFFFFu, mb = ub & 0x7FFFFFFFu;
    const uint32_t m = ma < mb ? ma : mb;  // magnitudes of non-NaN floats order like their bits
    return bitsf(((ua ^ ub) & 0x80000000u) | m);
}
#endif
// _mm256_polarg_ps avx_float.h:71-81 : (a ^ sign(u)) + b
__device__ __forceinline__ float polar_g(float a, float b, uint32_t u)
{
    return bitsf(fbits(a) ^ (u << 31)) + b;
}
// ---- int8 ("char") decoders: FastSscFipChar / SclFipChar (include/polarcode/decoding/fip_char.h) ---------------
// An int8 LLR is held as the float of the same value: every operation below is exact (small integers), saturation is a
// clamp (v_med3_f32), and a zero is always +0.0 (converted from int8, a + (-a), or b - b), so the IEEE sign bit is the
// char's sign bit.
// F_function_calc fip_char.h:45-66: magnitudes clamped to [1, 127] (abs(-128) counts as 127, a zero as 1), sign = xor
__device__ __forceinline__ float polar_f_char(float a, float b)
{
    const float ma = __builtin_amdgcn_fmed3f(absf(a), 1.0f, 127.0f), mb = __builtin_amdgcn_fmed3f(absf(b), 1.0f, 127.0f);
    const float m = ma < mb ? ma : mb;
    return bitsf(((fbits(a) ^ fbits(b)) & 0x80000000u) | fbits(m));
}
// G_function_calc fip_char.h:68-74: _mm_adds_epi8 / _mm_subs_epi8 = add, then clamp to [-128, 127]
__device__ __forceinline__ float polar_g_char(float a, float b, uint32_t u)
{
    return __builtin_amdgcn_fmed3f(bitsf(fbits(a) ^ (u << 31)) + b, -128.0f, 127.0f);
}
template <bool CH>
__device__ __forceinline__ float polar_fx(float a, float b)
{
    if constexpr (CH) return polar_f_char(a, b);
    else return polar_f(a, b);
}
template <bool CH>
__device__ __forceinline__ float polar_gx(float a, float b, uint32_t u)
{
    if constexpr (CH) return polar_g_char(a, b, u);
    else return polar_g(a, b, u);
}
// G_function_0R (all-frozen left child): a + b, saturating for the int8 decoders (fip_char.h:118-140)
template <bool CH>
__device__ __forceinline__ float addx(float a, float b)
{
    if constexpr (CH) return __builtin_amdgcn_fmed3f(a + b, -128.0f, 127.0f);
    else return a + b;
}
// |x| as an unsigned-comparable bit pattern; the int8 leaves use abs(max(x, -127)) (scl_fip_char.cpp:413-417,587-590)
template <bool CH>
__device__ __forceinline__ uint32_t magbits(float x)
{
    const uint32_t a = fbits(x) & 0x7FFFFFFFu;
    if constexpr (CH) return a < 0x42FE0000u ? a : 0x42FE0000u;  // 127.0f
    else return a;
}

// _mm256_min_ps(x, 0) / _mm256_max_ps(x, 0): second operand on ties
__device__ __forceinline__ float min0(float x) { return x < 0.0f ? x : 0.0f; }
__device__ __forceinline__ float max0(float x) { return x > 0.0f ? x : 0.0f; }

__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }

__device__ __forceinline__ uint32_t shfl_u32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
__device__ __forceinline__ float shfl_f32(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src)
{
    const uint32_t lo = shfl_u32((uint32_t)v, src), hi = shfl_u32((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t shflx_u32(uint32_t v, int m) { return (uint32_t)__shfl_xor((int)v, m, 64); }
__device__ __forceinline__ float shflx_f32(float v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ uint64_t shflx_u64(uint64_t v, int m)
{
    const uint32_t lo = shflx_u32((uint32_t)v, m), hi = shflx_u32((uint32_t)(v >> 32), m);
    return ((uint64_t)hi << 32) | lo;
}

// ---- cross-lane primitives that stay on the VALU (DPP / readlane) -------------------------------------
// ds_bpermute (__shfl) takes an LDS round trip (~100+ cycles on a dependent chain, with 3-4 waves per SIMD
// to hide it); DPP modifiers and v_readlane cost a VALU slot.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    // every control used here (quad_perm, row_mirror, row_half_mirror) reads a valid lane, so the
    // bound_ctrl form is exact and lets the compiler fold the move into the consuming VALU op (v_max_u32_dpp ...)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
// partner value for an all-reduce step over 2M-lane groups.  M = 1, 2: lane ^ M (quad_perm);
// M = 4: row_half_mirror, M = 8: row_mirror -- the mirrored lane sits in the other half, which is all an
// all-reduce needs once the halves are uniform; M = 16, 32: ds_bpermute.
template <int M>
__device__ __forceinline__ uint32_t partner_u32(uint32_t v)
{
    if constexpr (M == 1)
        return dpp_u32<0xB1>(v);
    else if constexpr (M == 2)
        return dpp_u32<0x4E>(v);
    else if constexpr (M == 4)
        return dpp_u32<0x141>(v);
    else if constexpr (M == 8)
        return dpp_u32<0x140>(v);
    else
        return shflx_u32(v, M);
}
#define PAMD_GROUP_REDUCE(T, NAME, PARTNER, EXPR)                          \
    template <int G>                                                       \
    __device__ __forceinline__ T NAME(T v)                                 \
    {                                                                      \
        if constexpr (G > 1) { const T o = PARTNER<1>(v); v = (EXPR); }    \
        if constexpr (G > 2) { const T o = PARTNER<2>(v); v = (EXPR); }    \
        if constexpr (G > 4) { const T o = PARTNER<4>(v); v = (EXPR); }    \
        if constexpr (G > 8) { const T o = PARTNER<8>(v); v = (EXPR); }    \
        if constexpr (G > 16) { const T o = PARTNER<16>(v); v = (EXPR); }  \
        if constexpr (G > 32) { const T o = PARTNER<32>(v); v = (EXPR); }  \
        return v;                                                          \
    }
PAMD_GROUP_REDUCE(uint32_t, group_xor_u32, partner_u32, v ^ o)
PAMD_GROUP_REDUCE(uint32_t, group_or_u32, partner_u32, v | o)
PAMD_GROUP_REDUCE(uint32_t, group_max_u32, partner_u32, o > v ? o : v)
PAMD_GROUP_REDUCE(uint32_t, group_min_u32, partner_u32, o < v ? o : v)

// all-reduce minimum / maximum over G = 32 or 64 lanes without the LDS: DPP inside the 16-lane rows, then the gfx950 row /
// half swaps (v_permlane16_swap, v_permlane32_swap with both operands = v: the two results are "my side" and "the other
// side" in some order, so their minimum / maximum needs no lane predicate)
template <int G>
__device__ __forceinline__ uint32_t group_min_u32x(uint32_t v)
{
    v = group_min_u32<(G < 16 ? G : 16)>(v);
    if constexpr (G > 16) {
        const auto s = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        v = s[0] < s[1] ? s[0] : s[1];
    }
    if constexpr (G > 32) {
        const auto s = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        v = s[0] < s[1] ? s[0] : s[1];
    }
    return v;
}
template <int G>
__device__ __forceinline__ uint32_t group_max_u32x(uint32_t v)
{
    v = group_max_u32<(G < 16 ? G : 16)>(v);
    if constexpr (G > 16) {
        const auto s = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        v = s[0] > s[1] ? s[0] : s[1];
    }
    if constexpr (G > 32) {
        const auto s = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        v = s[0] > s[1] ? s[0] : s[1];
    }
    return v;
}
// wave-wide maximum of a u32 as a wave-uniform value: DPP inside the 16-lane rows, then 4 readlanes
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    v = group_max_u32<16>(v);
    const uint32_t a = readlane_u32(v, 0), b = readlane_u32(v, 16), c = readlane_u32(v, 32), d = readlane_u32(v, 48);
    const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
// value held by lane I of each 8-lane group, delivered to all 8 lanes (two DPP moves and a select)
template <int I>
__device__ __forceinline__ uint32_t group8_bcast(uint32_t v, int j)
{
    const uint32_t t = dpp_u32<(I & 3) * 0x55>(v);  // lane I&3 of each quad
    const uint32_t u = dpp_u32<0x141>(t);           // ... of the other quad (row_half_mirror)
    if constexpr (I < 4)
        return (j < 4) ? t : u;
    else
        return (j < 4) ? u : t;
}
// order-preserving map float -> u32 (and back); -0.0 is folded onto +0.0 (they compare equal)
__device__ __forceinline__ uint32_t okey(float f)
{
    uint32_t b = fbits(f);
    b = (b == 0x80000000u) ? 0u : b;
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unkey(uint32_t k) { return bitsf((k >> 31) ? (k ^ 0x80000000u) : ~k); }

// _mm256_spc_right4_ps avx_float.h:289-302 on four wave-uniform values: returns the 4 output signs
__device__ __forceinline__ uint32_t spc4_signs(const float in[4])
{
    float m = absf(in[0]);
    uint32_t p = 0, out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = absf(in[i]);
        m = a < m ? a : m;
        p ^= sgn(in[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t flip = (p && absf(in[i]) == m) ? 1u : 0u;
        out |= (sgn(in[i]) ^ flip) << i;
    }
    return out;
}

}  // namespace pamd
