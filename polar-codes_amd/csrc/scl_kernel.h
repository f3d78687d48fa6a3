// Fast-SSCL list decoder kernel templates and their launch helpers.  Included by the scl_inst_*.hip translation
// units, each of which instantiates the kernels of one (lanes per path, codewords per wave) pair, so that the
// instantiations compile in parallel.
#pragma once
#include "device_common.h"

namespace pamd {

// =============================================================================================
//  Fast-SSCL
// =============================================================================================
// A wave holds 64 / LP path groups of LP lanes; they are split into 1 << CWL codewords of LCW groups each (SclCtx).
// LDS per wave (all its codewords):
//   llr   : one path buffer of path_stride floats per group; stage t (block 2^t) at offset 1<<t, t < s0
//           (stages s0 .. logN-1 live in the per-wave arena in global memory, the channel row -- stage logN -- is
//           read straight from global memory / L2, and stage logN-1 may be recomputed from it: "virtual top")
//   bitsL : groups x wppL words -- outputs of LEFT children (and of the root), per stage
//   bitsR : groups x wppR words -- outputs of RIGHT children
//           stage t lives at word woff(t): t <= 5 -> word t (low 2^t bits), else (1 << (t-5)) + 4
//           (stages >= sb may live in the arena instead, kernel flag HB)
//   scratch: candidate keys / rank tables of the selection routines
// Per lane: p = lane / LP (own group = path buffer), pl = its path index inside the codeword, the path metric
// (replicated over the path's lanes) and the slot word:
//   field s-1 (FB bits) = which path buffer of the codeword holds  LLR[s]  while the path is inside the left
//   subtree of its stage-s ancestor, or the LEFT child's bits of that ancestor while inside the
//   right subtree.  Writers always use their own index (all paths sit on the same tree node, so
//   the previous contents are dead for everyone); forks copy the word (PathList::duplicatePath).

__device__ __forceinline__ uint32_t woff(uint32_t t) { return t <= 5 ? t : ((1u << (t - 5)) + 4u); }

// LLR sources / destinations by memory space.  Stage buffers live in LDS for stages < s0, in the per-wave
// global "arena" (L2-resident scratch) for s0 <= stage < logN, and stage logN is the channel row itself.
typedef float __attribute__((address_space(3))) lds_f32;
enum { K_LDS = 0, K_GF32 = 1, K_GI8 = 2, K_ARENA = 3 };
typedef float v4f32 __attribute__((ext_vector_type(4)));
typedef v4f32 __attribute__((address_space(3))) lds_f32x4;
template <int KIND>
struct Rd {
    const void* p;
    __device__ __forceinline__ float operator[](uint32_t i) const
    {
        if constexpr (KIND == K_LDS) return ((const lds_f32*)(uintptr_t)p)[i];
        else if constexpr (KIND == K_GI8) return (float)reinterpret_cast<const int8_t*>(p)[i];
        // K_ARENA: written earlier by other lanes of THIS wave; the stores were waited for (vmcnt(0)) and a CU's
        // L1 is coherent for its own waves' accesses, so a plain load sees them
        else return reinterpret_cast<const float*>(p)[i];
    }
    // four consecutive LLRs, i % 4 == 0 and 16-byte aligned storage
    __device__ __forceinline__ float4 ld4(uint32_t i) const
    {
        if constexpr (KIND == K_LDS) {
            const v4f32 v = *(const lds_f32x4*)(uintptr_t)(reinterpret_cast<const float*>(p) + i);
            return make_float4(v.x, v.y, v.z, v.w);
        } else if constexpr (KIND == K_GI8) {
            const uint32_t w = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const int8_t*>(p) + i);
            return make_float4((float)(int8_t)(w & 0xFF), (float)(int8_t)((w >> 8) & 0xFF), (float)(int8_t)((w >> 16) & 0xFF),
                               (float)(int8_t)(w >> 24));
        } else {
            return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
        }
    }
};
template <int KIND>
struct Wr {
    void* p;
    __device__ __forceinline__ void set(uint32_t i, float v) const
    {
        if constexpr (KIND == K_LDS) ((lds_f32*)(uintptr_t)p)[i] = v;
        else reinterpret_cast<float*>(p)[i] = v;
    }
    __device__ __forceinline__ void st4(uint32_t i, float4 v) const
    {
        if constexpr (KIND == K_LDS) {
            v4f32 t;
            t.x = v.x;
            t.y = v.y;
            t.z = v.z;
            t.w = v.w;
            *(lds_f32x4*)(uintptr_t)(reinterpret_cast<float*>(p) + i) = t;
        }
        else *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i) = v;
    }
};

// CWL: log2 of the codewords one wave decodes side by side.  The 64 >> LPL path groups of the wave are split into
// 1 << CWL codewords of LCW groups each; a group is BOTH a path buffer (LDS / arena / bit areas, index p) and, at any
// moment, the home of the path with index pl inside its codeword.  Everything per path is unchanged by CWL; what
// becomes per codeword is the list management (candidate selection, fork sources), the channel row and the output.
template <int LPL, bool S64, bool HB, int CWL = 0, bool CHAR = false>
struct SclCtx {
    static constexpr bool CH = CHAR;  // int8 ("char") arithmetic: SclFipChar
    using SlotT = std::conditional_t<S64, uint64_t, uint32_t>;
    static constexpr int LP = 1 << LPL;     // lanes per path
    static constexpr int LMAX = 64 >> LPL;  // path groups per wave
    static constexpr int CW = 1 << CWL;     // codewords per wave
    static constexpr int LCW = LMAX >> CWL; // path groups (= list size limit) per codeword
    static constexpr int FB = 6 - LPL - CWL;  // bits per slot field: a path buffer of the same codeword
    static constexpr int NACC = LP >= 8 ? 1 : 8 / LP;  // AVX-lane accumulators per lane

    float* llr;
    uint32_t* bitsL;
    uint32_t* bitsR;
    uint32_t path_stride, wppL, wppR;
    uint32_t ablate;
    uint32_t* scratch;  // per-wave LDS scratch (candidate keys + rank table): 64 words at LP = 8, 64 * 8 / LP + 32 below
    float* arena;  // this wave's global scratch (stages >= s0)
    uint32_t arena_path_stride, s0, logN;
    // bit blocks of stages >= sb live in the global arena too (they are touched by a handful of ops per codeword
    // but would otherwise take most of the LDS at large N x L): per path gwl words of left-child / root outputs
    // followed by the right-child outputs, stage t at woff(t) - woff(sb)
    uint32_t* gbits;
    uint32_t sb, wsb, gwl, gwpp;
    const void* chan;
    int lane, p, j, base;
    int pl, g0;      // path index inside its codeword; first group of the codeword (p = g0 + pl); CWL = 0: pl = p, g0 = 0
    uint32_t P;      // live paths per codeword (wave uniform: every codeword follows the same program)
    uint32_t L;      // list size limit
    float metric;
    SlotT slot;

    // (24-bit multiplies are full rate; v_mul_lo_u32 is not)
    __device__ __forceinline__ float* path_llr(uint32_t q) const { return llr + __umul24(q, path_stride); }
    __device__ __forceinline__ uint32_t* area(uint32_t side, uint32_t q) const
    {
        return side ? bitsR + __umul24(q, wppR) : bitsL + __umul24(q, wppL);
    }
    // stage-t bit block of path buffer q (side 0: left children and the root, 1: right children), wherever it lives
    __device__ __forceinline__ uint32_t* bits(uint32_t side, uint32_t q, uint32_t t) const
    {
        // (HB = false: everything is in LDS and the compiler keeps ds_* instructions; with HB the pointer is
        // generic and these accesses become flat_* -- only worth it where the LDS saved buys resident waves)
        if constexpr (HB) {
            if (t >= sb) return gbits + __umul24(q, gwpp) + (side ? gwl : 0u) + (woff(t) - wsb);
        }
        return area(side, q) + woff(t);
    }
    __device__ __forceinline__ float* arena_stage(uint32_t q, uint32_t t) const
    {
        return arena + __umul24(q, arena_path_stride) + ((1u << t) - (1u << s0));
    }
    // call fn(accessor) with the reader for the stage-t LLRs of path buffer q
    template <bool I8, typename F>
    __device__ __forceinline__ void with_src(uint32_t t, uint32_t q, F&& fn) const
    {
        // (fp32 channel rows and arena stages are both plain global floats: one instantiation of the caller's body
        // serves both -- the code of this kernel has to stay well inside the instruction cache)
        if (I8 && t == logN) {
            fn(Rd<K_GI8>{ chan });
        } else if (t >= s0) {
            fn(Rd<K_ARENA>{ t == logN ? chan : (const void*)arena_stage(q, t) });
        } else {
            fn(Rd<K_LDS>{ path_llr(q) + (1u << t) });
        }
    }
    // compile-time memory kind of stage t (wave uniform), and the per-path base pointer for that kind
    template <bool I8, typename F>
    __device__ __forceinline__ void with_src_kind(uint32_t t, F&& fn) const
    {
        if (I8 && t == logN) {
            fn(std::integral_constant<int, K_GI8>{});
        } else if (t >= s0) {
            fn(std::integral_constant<int, K_ARENA>{});
        } else {
            fn(std::integral_constant<int, K_LDS>{});
        }
    }
    template <int KIND>
    __device__ __forceinline__ const void* src_base(uint32_t t, uint32_t q) const
    {
        if constexpr (KIND == K_LDS) return path_llr(q) + (1u << t);
        else if constexpr (KIND == K_ARENA) return arena_stage(q, t);
        else return chan;
    }
    template <typename F>
    __device__ __forceinline__ void with_dst(uint32_t t, uint32_t q, F&& fn) const
    {
        if (t >= s0)
            fn(Wr<K_GF32>{ arena_stage(q, t) });
        else
            fn(Wr<K_LDS>{ path_llr(q) + (1u << t) });
    }
    // the path buffer named by field s of the slot word sl of a path that lives in group gq
    __device__ __forceinline__ uint32_t field(SlotT sl, uint32_t s, uint32_t gq) const
    {
        return (gq & ~(uint32_t)(LCW - 1)) | ((uint32_t)(sl >> ((s - 1) * FB)) & (uint32_t)(LCW - 1));
    }
    __device__ __forceinline__ void set_field(uint32_t s)
    {
        const uint32_t sh = (s - 1) * FB;
        slot = (slot & ~((SlotT)(LCW - 1) << sh)) | ((SlotT)(p & (LCW - 1)) << sh);
    }
    // the slot word held by lane src (per-lane source)
    __device__ __forceinline__ SlotT slot_from_lane(int src) const
    {
        if constexpr (S64)
            return shfl_u64(slot, src);
        else
            return shfl_u32(slot, src);
    }

    // sum over the path of v(i), i < n, in the reference's order: 8 strided accumulators
    // (acc[v] += x[8k+v], k ascending) then acc0+acc1+...+acc7 (avxconvenience.h:267).
    // acc[] holds this lane's accumulators: virtual AVX lane v = j + LP*a  (LP < 8) or v = j (LP >= 8)
    // The same sum when only AVX lanes 0..3 can be non-zero (blocks of at most 4 LLRs, the REP leaves) and no term is
    // -0.0: x + 0.0 == x then, so the four trailing additions of the reference's order are dropped.
    __device__ __forceinline__ float lanesum_low4(const float acc[NACC]) const
    {
        if constexpr (LP == 4 || LP == 8) {
            const uint32_t a = fbits(acc[0]);
            uint32_t b0 = dpp_u32<0x00>(a), b1 = dpp_u32<0x55>(a), b2 = dpp_u32<0xAA>(a), b3 = dpp_u32<0xFF>(a);
            if constexpr (LP == 8) {  // lanes 4..7 of the group take the first quad's values (row_half_mirror)
                const bool hi = (j & 4) != 0;
                const uint32_t o0 = dpp_u32<0x141>(b0), o1 = dpp_u32<0x141>(b1), o2 = dpp_u32<0x141>(b2), o3 = dpp_u32<0x141>(b3);
                b0 = hi ? o0 : b0;
                b1 = hi ? o1 : b1;
                b2 = hi ? o2 : b2;
                b3 = hi ? o3 : b3;
            }
            return ((bitsf(b0) + bitsf(b1)) + bitsf(b2)) + bitsf(b3);
        } else {
            return lanesum_finish(acc);
        }
    }
    __device__ __forceinline__ float lanesum_finish(const float acc[NACC]) const
    {
        if constexpr (LP == 8) {
            // own quad's four accumulators by quad_perm broadcast, the other quad's by row_half_mirror
            const uint32_t a = fbits(acc[0]);
            const uint32_t b0 = dpp_u32<0x00>(a), b1 = dpp_u32<0x55>(a), b2 = dpp_u32<0xAA>(a), b3 = dpp_u32<0xFF>(a);
            const uint32_t o0 = dpp_u32<0x141>(b0), o1 = dpp_u32<0x141>(b1), o2 = dpp_u32<0x141>(b2),
                           o3 = dpp_u32<0x141>(b3);
            const bool hi = (j & 4) != 0;
            const float x0 = bitsf(hi ? o0 : b0), x1 = bitsf(hi ? o1 : b1), x2 = bitsf(hi ? o2 : b2),
                        x3 = bitsf(hi ? o3 : b3), x4 = bitsf(hi ? b0 : o0), x5 = bitsf(hi ? b1 : o1),
                        x6 = bitsf(hi ? b2 : o2), x7 = bitsf(hi ? b3 : o3);
            return ((((((x0 + x1) + x2) + x3) + x4) + x5) + x6) + x7;
        }
        if constexpr (LP == 4) {
            // lane j holds the accumulators of AVX lanes j and j + 4: quad_perm broadcasts, no LDS round trips
            const uint32_t a = fbits(acc[0]), b = fbits(acc[1]);
            const float x0 = bitsf(dpp_u32<0x00>(a)), x1 = bitsf(dpp_u32<0x55>(a)), x2 = bitsf(dpp_u32<0xAA>(a)),
                        x3 = bitsf(dpp_u32<0xFF>(a)), x4 = bitsf(dpp_u32<0x00>(b)), x5 = bitsf(dpp_u32<0x55>(b)),
                        x6 = bitsf(dpp_u32<0xAA>(b)), x7 = bitsf(dpp_u32<0xFF>(b));
            return ((((((x0 + x1) + x2) + x3) + x4) + x5) + x6) + x7;
        }
        float s = 0.0f;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const int a = LP >= 8 ? 0 : v / LP;
            const int src = base + (LP >= 8 ? v : v % LP);
            const float t = shfl_f32(acc[a], src);
            s = (v == 0) ? t : s + t;
        }
        return s;
    }
};

// ---- weakest-LLR search: findWeakLlrs(idx, |x|, n, CNT) arrayfuncs.h:209-231 ----------------------
// The reference runs a swap-mutating selection: iteration i takes the first minimum among positions >= i
// and SWAPS it with position i.  Only the elements that start at positions 0..CNT-1 ("head") can ever be
// moved (to the position of a winner); everything else ("tail") stays where it is until it is selected.
// So: head elements live in path-uniform registers together with their CURRENT position, each lane keeps
// the CNT smallest (value, position) of its strided tail positions, and an iteration is
//   winner = lexicographic min over {alive head (value, current position)} U {lane fronts};
//   the head element sitting at position i moves to the winner's position; the winner leaves.
template <int CNT>
struct Weak {
    uint32_t v[CNT];   // value bits (|x|), ascending selection result
    uint32_t ix[CNT];  // original indices
};

// signs (optional): receives the XOR of the sign bits of this lane's share of the block (bit 0), so that the SPC parity
// comes out of the same scan.
template <typename CTX, int CNT, typename SRC>
__device__ __forceinline__ void find_weak(const CTX& c, const SRC& x, uint32_t n, Weak<CNT>& out, uint32_t* signs = nullptr)
{
    constexpr int LP = CTX::LP;
    constexpr uint32_t INFB = 0xFFFFFFFFu;  // larger than any |x| bit pattern / any position
    uint32_t hv[CNT], hp[CNT];
#pragma unroll
    for (int e = 0; e < CNT; ++e) {
        const bool in = (uint32_t)e < n;
        hv[e] = in ? magbits<CTX::CH>(x[in ? e : 0]) : INFB;
        hp[e] = in ? (uint32_t)e : INFB;
    }
    uint32_t tv[CNT], tp[CNT];
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
        tv[k] = INFB;
        tp[k] = INFB;
    }
    const bool has_tail = n > (uint32_t)CNT;  // wave uniform (then n >= 4)
    uint32_t sx = 0;
    if (has_tail) {
        // 16-byte loads: lane j takes the chunks j, j + LP, ... of four consecutive LLRs (one load latency per four
        // elements: the scan of a large leaf is bound by it); the head positions in chunk 0 stay out of the tail lists
        const uint32_t nch = n >> 2;
        for (uint32_t ch = (uint32_t)c.j; ch < nch; ch += LP) {
            const float4 q = x.ld4(ch << 2);
            const float e4[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t fb = fbits(e4[e]);
                sx ^= fb;
                uint32_t ap = (ch << 2) + (uint32_t)e;
                uint32_t a = ap >= (uint32_t)CNT ? magbits<CTX::CH>(e4[e]) : INFB;
                bool placed = false;
#pragma unroll
                for (int k = 0; k < CNT; ++k) {  // positions arrive ascending: strict '<' keeps earlier ones first
                    const bool lt = placed || (a < tv[k]);
                    const uint32_t sv = lt ? tv[k] : a, sp = lt ? tp[k] : ap;
                    tv[k] = lt ? a : tv[k];
                    tp[k] = lt ? ap : tp[k];
                    a = sv;
                    ap = sp;
                    placed = lt;
                }
            }
        }
    } else if (c.j == 0) {
#pragma unroll
        for (int e = 0; e < CNT; ++e)
            if ((uint32_t)e < n) sx ^= fbits(x[e]);
    }
    if (signs) *signs = sx >> 31;
    const uint32_t lim = (n - 1 < (uint32_t)CNT) ? n - 1 : (uint32_t)CNT;
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        out.v[i] = hv[i];  // (only reached for i >= lim: what is left at position i)
        out.ix[i] = i;
        if ((uint32_t)i < lim) {
            uint32_t tvmin = INFB, tpmin = INFB;
            if (has_tail) {
                tvmin = group_min_u32<LP>(tv[0]);
                tpmin = group_min_u32<LP>(tv[0] == tvmin ? tp[0] : INFB);
            }
            uint32_t bv = hv[0], bp = hp[0], be = 0;
#pragma unroll
            for (int e = 1; e < CNT; ++e) {
                const bool lt = (hv[e] < bv) || (hv[e] == bv && hp[e] < bp);
                bv = lt ? hv[e] : bv;
                bp = lt ? hp[e] : bp;
                be = lt ? (uint32_t)e : be;
            }
            const bool tail_wins = (tvmin < bv) || (tvmin == bv && tpmin < bp);
            const uint32_t wv = tail_wins ? tvmin : bv, wp = tail_wins ? tpmin : bp;
            out.v[i] = wv;
            out.ix[i] = tail_wins ? wp : be;  // tail elements never move: original index = position
#pragma unroll
            for (int e = 0; e < CNT; ++e) {
                const bool is_w = !tail_wins && be == (uint32_t)e;
                const bool at_i = hp[e] == (uint32_t)i;
                hp[e] = is_w ? INFB : (at_i ? wp : hp[e]);
                hv[e] = is_w ? INFB : hv[e];
            }
            if (has_tail) {
                const bool pop = tail_wins && tp[0] == wp;
#pragma unroll
                for (int k = 0; k < CNT - 1; ++k) {
                    tv[k] = pop ? tv[k + 1] : tv[k];
                    tp[k] = pop ? tp[k + 1] : tp[k];
                }
                tv[CNT - 1] = pop ? INFB : tv[CNT - 1];
                tp[CNT - 1] = pop ? INFB : tp[CNT - 1];
            }
        }
    }
    // positions >= lim keep whatever sits there: the head element whose current position is i
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        if ((uint32_t)i >= lim && (uint32_t)i < n) {
#pragma unroll
            for (int e = 0; e < CNT; ++e)
                if (hp[e] == (uint32_t)i) {
                    out.v[i] = hv[e];
                    out.ix[i] = e;
                }
        }
    }
}

// ---- LP == 8: block of n = 2^S <= 8 LLRs, one per lane (lane j holds |x_j| bits, INFB beyond n) ----------
// findWeakLlrs emulated literally with "array position = lane": iteration i finds the first minimum among
// lanes >= i and swaps it with lane i, all with DPP reductions / broadcasts inside the 8-lane path group.
template <int S, int CNT>
__device__ __forceinline__ void find_weak_small(uint32_t a, int j, Weak<CNT>& out)
{
    constexpr uint32_t INFB = 0xFFFFFFFFu;
    constexpr int n = 1 << S;
    constexpr int lim = (n - 1 < CNT) ? n - 1 : CNT;
    uint32_t o = (uint32_t)j;  // original index travelling with the value
    auto step = [&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (i < lim) {
            const bool el = j >= i;
            const uint32_t m = group_min_u32<8>(el ? a : INFB);
            const uint32_t pk = group_min_u32<8>((el && a == m) ? (((uint32_t)j << 3) | o) : INFB);
            const int k = (int)(pk >> 3);
            const uint32_t ok = pk & 7u;
            const uint32_t ai = group8_bcast<i>(a, j), oi = group8_bcast<i>(o, j);
            if (j == k) {
                a = ai;
                o = oi;
            }
            if (j == i) {
                a = m;
                o = ok;
            }
            out.v[i] = m;
            out.ix[i] = ok;
        } else if constexpr (i < n) {
            out.v[i] = group8_bcast<i>(a, j);
            out.ix[i] = group8_bcast<i>(o, j);
        } else {
            out.v[i] = 0;
            out.ix[i] = 0;
        }
    };
    step(std::integral_constant<int, 0>{});
    if constexpr (CNT > 1) step(std::integral_constant<int, 1>{});
    if constexpr (CNT > 2) step(std::integral_constant<int, 2>{});
    if constexpr (CNT > 3) step(std::integral_constant<int, 3>{});
}

// ---- LP == 4: block of n = 2^S <= 8 LLRs, positions j (a0) and j + 4 (a1, n = 8 only) in lane j -----------------
// Same literal emulation of findWeakLlrs as find_weak_small, reductions and broadcasts inside the quad.
template <int I>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v)
{
    return dpp_u32<I * 0x55>(v);
}
template <int S, int CNT>
__device__ __forceinline__ void find_weak_small4(uint32_t a0, uint32_t a1, int j, Weak<CNT>& out)
{
    constexpr uint32_t INFB = 0xFFFFFFFFu;
    constexpr int n = 1 << S;
    constexpr int lim = (n - 1 < CNT) ? n - 1 : CNT;
    uint32_t o0 = (uint32_t)j, o1 = (uint32_t)j + 4u;  // original indices travelling with the values
    auto step = [&](auto I) {
        constexpr int i = decltype(I)::value;  // (i <= 3: position i is register 0 of lane i)
        if constexpr (i < lim) {
            const uint32_t v0 = (j >= i) ? a0 : INFB, v1 = (S == 3) ? a1 : INFB;
            const bool second = v1 < v0;  // position j precedes j + 4: ties go to the first
            const uint32_t lv = second ? v1 : v0;
            const uint32_t lpo = second ? ((((uint32_t)j + 4u) << 3) | o1) : (((uint32_t)j << 3) | o0);
            const uint32_t m = group_min_u32<4>(lv);
            const uint32_t pk = group_min_u32<4>((lv == m) ? lpo : INFB);
            const uint32_t k = pk >> 3, ok = pk & 7u;
            const uint32_t ai = quad_bcast<i>(a0), oi = quad_bcast<i>(o0);
            if ((uint32_t)j == k) {
                a0 = ai;
                o0 = oi;
            }
            if (S == 3 && (uint32_t)j + 4u == k) {
                a1 = ai;
                o1 = oi;
            }
            if (j == i) {
                a0 = m;
                o0 = ok;
            }
            out.v[i] = m;
            out.ix[i] = ok;
        } else if constexpr (i < n) {
            out.v[i] = quad_bcast<i>(a0);
            out.ix[i] = quad_bcast<i>(o0);
        } else {
            out.v[i] = 0;
            out.ix[i] = 0;
        }
    };
    step(std::integral_constant<int, 0>{});
    if constexpr (CNT > 1) step(std::integral_constant<int, 1>{});
    if constexpr (CNT > 2) step(std::integral_constant<int, 2>{});
    if constexpr (CNT > 3) step(std::integral_constant<int, 3>{});
}

// Fast form of the same search when no two LLR magnitudes of a block are equal (the usual case for real-valued
// input): the reference's selection then simply yields ascending order, so every element is ranked against the other
// n - 1 (quad rotations), the four smallest go through the LDS scratch by rank, and every lane reads them back as
// two 16-byte broadcasts.  Any exact tie inside a block of the wave sends the whole wave to the literal emulation.
template <typename CTX, int S, int CNT>
__device__ __forceinline__ void weak_small4(const CTX& c, uint32_t a0, uint32_t a1, Weak<CNT>& out)
{
    constexpr uint32_t INFB = 0xFFFFFFFFu;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 __attribute__((address_space(3))) lds_u32x4;
    typedef uint32_t __attribute__((address_space(3))) lds_u32;
    const uint32_t j = (uint32_t)c.j;
    // values of lanes j+1, j+2, j+3 (mod 4) of the quad
    const uint32_t p1 = dpp_u32<0x39>(a0), p2 = dpp_u32<0x4E>(a0), p3 = dpp_u32<0x93>(a0);
    uint32_t r0 = ((p1 < a0) ? 1u : 0u) + ((p2 < a0) ? 1u : 0u) + ((p3 < a0) ? 1u : 0u), r1 = 8u;
    bool tie = (a0 != INFB) && (p1 == a0 || p2 == a0 || p3 == a0);
    if constexpr (S == 3) {
        const uint32_t q1 = dpp_u32<0x39>(a1), q2 = dpp_u32<0x4E>(a1), q3 = dpp_u32<0x93>(a1);
        r0 += ((a1 < a0) ? 1u : 0u) + ((q1 < a0) ? 1u : 0u) + ((q2 < a0) ? 1u : 0u) + ((q3 < a0) ? 1u : 0u);
        r1 = ((a0 < a1) ? 1u : 0u) + ((p1 < a1) ? 1u : 0u) + ((p2 < a1) ? 1u : 0u) + ((p3 < a1) ? 1u : 0u) +
             ((q1 < a1) ? 1u : 0u) + ((q2 < a1) ? 1u : 0u) + ((q3 < a1) ? 1u : 0u);
        tie = tie || a1 == a0 || q1 == a0 || q2 == a0 || q3 == a0 || p1 == a1 || p2 == a1 || p3 == a1 || q1 == a1 ||
              q2 == a1 || q3 == a1;
    }
    if (ballot64(tie && (uint32_t)c.pl < c.P) != 0) {  // (stale buffers of dead paths do not count)
        find_weak_small4<S, CNT>(a0, a1, c.j, out);
        return;
    }
    // per path: words 0..3 = the four smallest values, 4..7 = their indices (lanes beyond n hold INFB and rank last)
    lds_u32* slotp = (lds_u32*)(uintptr_t)c.scratch + (uint32_t)c.p * 8u;
    if (r0 < 4u) {
        slotp[r0] = a0;
        slotp[4u + r0] = j;
    }
    if (S == 3 && r1 < 4u) {
        slotp[r1] = a1;
        slotp[4u + r1] = j + 4u;
    }
    WAVE_SYNC();
    const u32x4 v = *(const lds_u32x4*)(uintptr_t)(c.scratch + (uint32_t)c.p * 8u);
    const u32x4 x = *(const lds_u32x4*)(uintptr_t)(c.scratch + (uint32_t)c.p * 8u + 4u);
    WAVE_SYNC();  // (the scratch is reused by the candidate ranking)
    constexpr int n = 1 << S;
    const uint32_t vv[4] = { v.x, v.y, v.z, v.w }, xx[4] = { x.x, x.y, x.z, x.w };
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        out.v[i] = i < n ? vv[i] : 0u;
        out.ix[i] = i < n ? xx[i] : 0u;
    }
}

// ---- LP == 8 candidate selection by rank -------------------------------------------------------------------
// When no two candidates that matter carry the same metric, the reference's selection sort simply yields the
// newP largest in descending order, so candidate c becomes new path  rank(c) = #{d : metric_d > metric_c}.
// The rank is counted against every candidate slot with fully unrolled v_readlane / v_cmp / v_addc triples
// (constant lane numbers, no scalar loop bookkeeping).  Candidates with equal metrics get equal ranks and
// collide in the scatter below; that is detected and handed to the literal emulation (select_literal_cw), which
// reproduces the reference's swap-order tie behaviour.
template <typename CTX, int CPP>
__device__ __forceinline__ bool select_ranked8(const CTX& c, float val, uint32_t newP, float& nval,
                                               uint32_t& nidx)
{
    constexpr int CS = CPP == 8 ? 3 : (CPP == 4 ? 2 : 1);
    constexpr int LCW = CTX::LCW;
    const bool valid = ((uint32_t)c.j < (uint32_t)CPP) && ((uint32_t)c.pl < c.P);
    const uint32_t key = valid ? okey(val) : 0u;  // 0 is below every real key
    // every lane compares its key with all candidate keys of ITS codeword: the keys go through a 64-word LDS scratch
    // and come back as broadcast 16-byte reads (pure VALU compares, no v_readlane -> SGPR -> v_cmp hazards)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 __attribute__((address_space(3))) lds_u32x4;
    typedef uint32_t __attribute__((address_space(3))) lds_u32;
    ((lds_u32*)(uintptr_t)c.scratch)[c.lane] = key;
    WAVE_SYNC();
    const uint32_t* mykeys = c.scratch + c.g0 * 8;
    uint32_t gt = 0;
#pragma unroll
    for (int q = 0; q < LCW; ++q) {
        if ((uint32_t)q < c.P) {  // wave uniform
            const u32x4 k0 = *(const lds_u32x4*)(uintptr_t)(mykeys + q * 8);
            gt += (k0.x > key) ? 1u : 0u;
            if constexpr (CPP > 1) gt += (k0.y > key) ? 1u : 0u;
            if constexpr (CPP > 2) {
                gt += (k0.z > key) ? 1u : 0u;
                gt += (k0.w > key) ? 1u : 0u;
            }
            if constexpr (CPP > 4) {
                const u32x4 k1 = *(const lds_u32x4*)(uintptr_t)(mykeys + q * 8 + 4);
                gt += (k1.x > key) ? 1u : 0u;
                gt += (k1.y > key) ? 1u : 0u;
                gt += (k1.z > key) ? 1u : 0u;
                gt += (k1.w > key) ? 1u : 0u;
            }
        }
    }
    const bool sel = valid && gt < newP;
    // new path r inherits the candidate of rank r: scatter the lane ids (ds_permute), spread over the group
    const uint32_t dst = sel ? (((uint32_t)c.g0 + gt) << 3) : (((uint32_t)c.g0 << 3) + 1u);  // losers dump into lane 1 of the codeword, which nobody reads
    const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), c.lane);
    const uint32_t srcl = group8_bcast<0>(got, c.j);
    // a selected lane whose id did not arrive lost a collision: two winners share a rank = an exact tie
    const uint32_t back = shfl_u32(got, (int)dst);
    if (ballot64(sel && back != (uint32_t)c.lane) != 0) return false;
    nval = unkey(shfl_u32(key, (int)srcl));
    nidx = (((srcl >> 3) - (uint32_t)c.g0) << CS) + (srcl & 7u);
    return true;
}

// ---- candidate selection by rank for fewer than 8 lanes per path (list sizes above 8) ------------------------
// Same idea as select_ranked8 with CPL = CPP / LP candidates per lane: every candidate key goes to the LDS scratch
// (slot = lane * CPL + register), every lane counts how many keys beat each of its own, the winners write their
// slot id into a rank table, and new path r reads entry r.  Returns false when two winners share a rank (an exact
// tie between candidates that matter): the caller then runs the literal emulation.
template <typename CTX, int LPL, int CPP>
__device__ __forceinline__ bool select_ranked(const CTX& c, const float (&val)[(CPP >> LPL) ? (CPP >> LPL) : 1],
                                              uint32_t newP, float& nval, uint32_t& nidx)
{
    constexpr int LP = CTX::LP;
    constexpr int CPL = (CPP >> LPL) ? (CPP >> LPL) : 1;
    // fewer candidates than lanes per path (REP: 2): the keys are stored densely, slot = group * CPP + q, so that the
    // ranking does not wade through the empty slots of the idle lanes
    constexpr bool COMPACT = CPL == 1 && CPP < LP;
    constexpr uint32_t PER_GROUP = COMPACT ? (uint32_t)CPP : (uint32_t)(LP * CPL);  // key slots per path group
    constexpr uint32_t SLOTS = 64u * CPL;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 __attribute__((address_space(3))) lds_u32x4;
    typedef uint32_t __attribute__((address_space(3))) lds_u32;
    lds_u32* keys = (lds_u32*)(uintptr_t)c.scratch;
    lds_u32* rk = keys + SLOTS + (uint32_t)c.g0;  // one rank table per codeword (CTX::LCW entries)
    const uint32_t slot0 = (uint32_t)c.g0 * PER_GROUP;  // first key slot of this lane's codeword
    uint32_t key[CPL], id[CPL];
    bool valid[CPL];
#pragma unroll
    for (int r = 0; r < CPL; ++r) {
        const uint32_t q = (uint32_t)c.j * CPL + r;
        valid[r] = (q < (uint32_t)CPP) && ((uint32_t)c.pl < c.P);
        key[r] = valid[r] ? okey(val[r]) : 0u;  // 0 is below every real key
        id[r] = COMPACT ? (uint32_t)c.p * CPP + q : (uint32_t)c.lane * CPL + r;
        if (!COMPACT || q < (uint32_t)CPP) keys[id[r]] = key[r];
    }
    WAVE_SYNC();
    uint32_t gt[CPL];
#pragma unroll
    for (int r = 0; r < CPL; ++r) gt[r] = 0;
    const uint32_t nslots = (c.P * PER_GROUP + 3u) & ~3u;  // (the slots of dead paths hold 0)
    for (uint32_t s4 = 0; s4 < nslots; s4 += 4) {
        const u32x4 k = *(const lds_u32x4*)(uintptr_t)(c.scratch + slot0 + s4);
#pragma unroll
        for (int r = 0; r < CPL; ++r) {
            gt[r] += (k.x > key[r]) ? 1u : 0u;
            gt[r] += (k.y > key[r]) ? 1u : 0u;
            gt[r] += (k.z > key[r]) ? 1u : 0u;
            gt[r] += (k.w > key[r]) ? 1u : 0u;
        }
    }
    bool sel[CPL];
#pragma unroll
    for (int r = 0; r < CPL; ++r) {
        sel[r] = valid[r] && gt[r] < newP;
        if (sel[r]) rk[gt[r]] = id[r];
    }
    WAVE_SYNC();
    bool lost = false;
#pragma unroll
    for (int r = 0; r < CPL; ++r)
        if (sel[r]) lost = lost || (rk[gt[r]] != id[r]);
    if (ballot64(lost) != 0) return false;
    nval = 0.0f;
    nidx = 0;
    if ((uint32_t)c.pl < newP) {
        const uint32_t w = rk[c.pl];
        nval = unkey(keys[w]);
        if constexpr (COMPACT) {
            nidx = w - slot0;  // (source path inside the codeword) * CPP + q
        } else {
            const uint32_t sl = w / CPL, r = w % CPL;
            nidx = ((sl >> LPL) - (uint32_t)c.g0) * CPP + (sl & (uint32_t)(LP - 1)) * CPL + r;
        }
    }
    return true;
}

// ---- candidate selection by threshold for long lists (several candidates per lane, one codeword per wave) -----
// Ranking every candidate against all others costs (candidates)^2 / 64 compare pairs per lane (2 048 VALU instructions
// for 256 SPC candidates at L = 32).  Here the newP-th largest key T is found first by a bit-wise search (32 rounds of
// one compare per candidate register + scalar popcounts), the candidates >= T -- exactly newP unless several tie at
// T -- are compacted into the scratch by prefix count, and only those are ranked against each other (one survivor per
// lane).  Returns false on any tie that matters; the caller then runs the literal emulation.
template <typename CTX, int LPL, int CPP>
__device__ __forceinline__ bool select_threshold(const CTX& c, const float (&val)[(CPP >> LPL) ? (CPP >> LPL) : 1],
                                                 uint32_t newP, float& nval, uint32_t& nidx)
{
    static_assert(CTX::CW == 1, "wave-wide ballots: one codeword per wave");
    constexpr int LP = CTX::LP;
    constexpr int CPL = (CPP >> LPL) ? (CPP >> LPL) : 1;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 __attribute__((address_space(3))) lds_u32x4;
    typedef uint32_t __attribute__((address_space(3))) lds_u32;
    lds_u32* skey = (lds_u32*)(uintptr_t)c.scratch;  // [0, 40): keys of the survivors, zero padded
    lds_u32* sid = skey + 40;                        // [40, 80): their candidate slot ids
    lds_u32* rk = skey + 80;                         // [80, 112): rank -> survivor number
    uint32_t key[CPL];
#pragma unroll
    for (int r = 0; r < CPL; ++r) {
        const uint32_t q = (uint32_t)c.j * CPL + r;
        const bool valid = (q < (uint32_t)CPP) && ((uint32_t)c.pl < c.P);
        key[r] = valid ? okey(val[r]) : 0u;  // 0 is below every real key
    }
    uint32_t T = 0;
    for (int b = 31; b >= 0; --b) {
        const uint32_t cand = T | (1u << b);
        uint32_t cnt = 0;
#pragma unroll
        for (int r = 0; r < CPL; ++r) cnt += (uint32_t)__popcll(ballot64(key[r] >= cand));
        T = cnt >= newP ? cand : T;
    }
    uint32_t base = 0, pos[CPL];
    bool sel[CPL];
#pragma unroll
    for (int r = 0; r < CPL; ++r) {
        sel[r] = key[r] >= T;
        const uint64_t m = ballot64(sel[r]);
        pos[r] = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        base += (uint32_t)__popcll(m);
    }
    if (base != newP) return false;  // several candidates tie at the threshold
    if (c.lane < 4) skey[newP + (uint32_t)c.lane] = 0u;
#pragma unroll
    for (int r = 0; r < CPL; ++r)
        if (sel[r]) {
            skey[pos[r]] = key[r];
            sid[pos[r]] = (uint32_t)c.lane * CPL + r;
        }
    WAVE_SYNC();
    const bool mine = (uint32_t)c.lane < newP;  // lane i ranks survivor i
    const uint32_t myk = mine ? skey[c.lane] : 0xFFFFFFFFu;
    uint32_t gt = 0;
    for (uint32_t s4 = 0; s4 < newP; s4 += 4) {
        const u32x4 k = *(const lds_u32x4*)(uintptr_t)(c.scratch + s4);
        gt += (k.x > myk) ? 1u : 0u;
        gt += (k.y > myk) ? 1u : 0u;
        gt += (k.z > myk) ? 1u : 0u;
        gt += (k.w > myk) ? 1u : 0u;
    }
    if (mine) rk[gt] = (uint32_t)c.lane;
    WAVE_SYNC();
    if (ballot64(mine && rk[gt] != (uint32_t)c.lane) != 0) return false;  // two survivors share a rank: exact tie
    nval = 0.0f;
    nidx = 0;
    if ((uint32_t)c.pl < newP) {
        const uint32_t si = rk[c.pl];
        nval = unkey(skey[si]);
        const uint32_t id = sid[si], sl = id / CPL, r = id % CPL;
        nidx = (sl >> LPL) * CPP + (sl & (uint32_t)(LP - 1)) * CPL + r;
    }
    return true;
}

// ---- candidate selection, literal: simplePartialSortDescending(idx, metrics, P', size) arrayfuncs.h:161-183 ----------
// Candidate position c = path * CPP + q starts in lane path * LP + q / CPL, register q % CPL.  All codewords of the wave
// run at once (reductions over the lanes of a codeword).  On return every lane knows, for ITS OWN new path index, the
// winning candidate's metric (nval) and original position (nidx).  This is the fallback of the rank / merge selections
// when metrics are equal, and THE selection of the int8 decoders, where they always are.
template <typename CTX, int LPL, int CPP>
__device__ __forceinline__ void select_literal_cw(const CTX& c, const float (&val)[(CPP >> LPL) ? (CPP >> LPL) : 1],
                                                  uint32_t newP, float& nval, uint32_t& nidx)
{
    // Every candidate stays in its lane and carries its CURRENT position in the reference's array; a round of the
    // selection sort (arrayfuncs.h:161-183) is then one reduction and two local updates: the winner is the lowest current
    // position among the maxima (strict `>`: first occurrence), it leaves for its rank, and whoever sits at position i
    // takes over the winner's old place (the swap).  Winners drop (value, original index) into the LDS scratch at their
    // rank; the new paths read them back at the end.  No cross-lane fetch inside the loop.
    constexpr int CPL = (CPP >> LPL) ? (CPP >> LPL) : 1;
    constexpr int G = 64 / CTX::CW;  // lanes per codeword
    constexpr uint32_t INF = 0xFFFFFFFFu;
    const uint32_t size = c.P * CPP;
    const uint32_t lim = (size - 1 < newP) ? size - 1 : newP;
    uint32_t key[CPL], cur[CPL];  // cur = INF: no candidate, or already picked
#pragma unroll
    for (int r = 0; r < CPL; ++r) {
        const uint32_t q = (uint32_t)c.j * CPL + r;
        const bool valid = (q < (uint32_t)CPP) && ((uint32_t)c.pl < c.P);
        cur[r] = valid ? (uint32_t)c.pl * CPP + q : INF;
        if constexpr (CTX::CH)
            key[r] = (uint32_t)(0.0f - val[r]);  // int8 decoder: -metric, an integer <= 127 N (smaller = better)
        else
            key[r] = valid ? okey(val[r]) : 0u;
    }
    typedef uint32_t __attribute__((address_space(3))) lds_u32;
    lds_u32* out = (lds_u32*)(uintptr_t)c.scratch + (uint32_t)c.g0 * 2u;  // 2 words per rank, LCW ranks per codeword
    for (uint32_t i = 0; i < lim; ++i) {
        uint32_t wp;
        if constexpr (CTX::CH) {  // one reduction: (-metric) << 8 | position  (<= 32 x 8 positions, |metric| < 2^22)
            uint32_t lk = INF;
#pragma unroll
            for (int r = 0; r < CPL; ++r) {
                const uint32_t kk = (key[r] << 8) | (cur[r] & 255u);
                lk = (cur[r] != INF && kk < lk) ? kk : lk;
            }
            wp = group_min_u32x<G>(lk) & 255u;
        } else {
            uint32_t lk = 0, lp = INF;
#pragma unroll
            for (int r = 0; r < CPL; ++r) {
                const bool take = cur[r] != INF && (key[r] > lk || (key[r] == lk && cur[r] < lp));
                lk = take ? key[r] : lk;
                lp = take ? cur[r] : lp;
            }
            const uint32_t M = group_max_u32x<G>(lk);
            wp = group_min_u32x<G>((lk == M) ? lp : INF);
        }
#pragma unroll
        for (int r = 0; r < CPL; ++r) {
            const bool win = cur[r] == wp;
            if (win) {
                out[2u * i] = key[r];
                out[2u * i + 1u] = (uint32_t)c.pl * CPP + (uint32_t)c.j * CPL + (uint32_t)r;
            }
            cur[r] = win ? INF : (cur[r] == i ? wp : cur[r]);
        }
    }
    if (lim < newP) {  // every candidate survives: the one that is left sits at position lim and stays there
#pragma unroll
        for (int r = 0; r < CPL; ++r)
            if (cur[r] != INF) {
                out[2u * lim] = key[r];
                out[2u * lim + 1u] = (uint32_t)c.pl * CPP + (uint32_t)c.j * CPL + (uint32_t)r;
            }
    }
    WAVE_SYNC();
    const uint32_t me = (uint32_t)c.pl < newP ? (uint32_t)c.pl : 0u;
    const uint32_t rkey = out[2u * me], ridx = out[2u * me + 1u];
    WAVE_SYNC();  // (the scratch is reused by the next selection)
    if constexpr (CTX::CH)
        nval = 0.0f - (float)rkey;
    else
        nval = unkey(rkey);
    nidx = ridx;
}

// ---- candidate selection by merging sorted lists (4 lanes per path x 8 path groups x 2 codewords per wave) ----------
// The candidates of ONE path come out of the leaf formulas already ordered (R1: m >= m-v0 >= m-v1 >= (m-v0)-v1 because
// v0 <= v1 and fp32 subtraction is monotone; SPC: all but one pair, fixed by the caller; REP: one compare), so the
// reference's partial selection sort over the L x CPP candidates is a merge of 8 sorted lists that keeps the best 8.
// Path metrics are never positive, so the raw fp32 bit pattern read as an unsigned integer orders them the other way
// round: the BEST candidate has the SMALLEST key, and 0xFFFFFFFF (no metric) marks an empty slot.
// Layout: list element i of a path sits in register i / 4, lane i % 4 of the path's quad.  Three merge levels (quad
// pairs inside 8 lanes, 8-lane groups inside a DPP row, the two rows of the codeword), each = one "keep the smaller of
// a_i and b_(7-i)" step followed by a 3-stage bitonic clean-up over 8 lanes; the sort direction alternates between
// sibling groups so that the partner of the next level is a plain lane ^ 8 / lane ^ 16 exchange.  Inside an 8-lane
// group, rank r ends up at lane phi(r) = r < 4 ? r : 11 - r (all four groups of a codeword hold the same result).
// Everything stays on the VALU (DPP moves, v_permlane16_swap); exact ties that matter are detected afterwards and
// handed to the literal emulation by the caller (return value false).
constexpr uint64_t merge_keepmin_mask(int dist, int dbit)
{
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l) {
        const int b2 = (l >> 2) & 1;
        const bool base = dist == 4 ? (b2 == 0) : (dist == 2 ? (((l >> 1) & 1) == b2) : ((l & 1) == b2));
        const bool d = dbit >= 0 ? ((l >> dbit) & 1) != 0 : false;
        if (base != d) m |= 1ull << l;
    }
    return m;
}
// One compare-exchange stage on (key, payload) pairs with the partner lane named by a DPP control; KEEPMIN marks the
// lanes that keep the smaller key (32-bit pattern, the same for both halves of the wave).  Written in assembly: the
// compiler's form keeps nine 64-bit lane masks in SGPRs, which this kernel does not have (they were spilled to VGPR
// lanes and reloaded in every stage); here the mask is a literal of two s_xor_b32.  DPP reads need two wait states
// after the VALU write of their source (s_nop 1; the compiler does not see inside the statement).
//   vcc = (partner >= own) ^ KEEPMIN... : "keep own" = !(partner < own) on KEEPMIN lanes, (partner < own) on the others
#define PAMD_MERGE_STAGE(DPPSTR)                                                                          \
    asm volatile("s_nop 1\n\t"                                                                            \
                 "v_mov_b32_dpp %2, %0 " DPPSTR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"             \
                 "v_cmp_lt_u32_e32 vcc, %2, %0\n\t"                                                        \
                 "s_xor_b32 vcc_lo, vcc_lo, %3\n\t"                                                        \
                 "s_xor_b32 vcc_hi, vcc_hi, %3\n\t"                                                        \
                 "v_cndmask_b32_e32 %0, %2, %0, vcc\n\t"                                                   \
                 "v_cndmask_b32_dpp %1, %1, %1, vcc " DPPSTR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                 "s_nop 1"  /* (both outputs are VALU-written here: a DPP read by the code behind the statement is safe) */ \
                 : "+v"(k), "+v"(x), "=&v"(pk)                                                            \
                 : "i"(KEEPMIN)                                                                           \
                 : "vcc", "scc")  /* (s_xor_b32 writes SCC) */
template <int DIST, uint32_t KEEPMIN>
__device__ __forceinline__ void merge_cmpex(uint32_t& k, uint32_t& x)
{
    uint32_t pk;
    if constexpr (DIST == 4)
        PAMD_MERGE_STAGE("row_half_mirror");
    else if constexpr (DIST == 2)
        PAMD_MERGE_STAGE("quad_perm:[2,3,0,1]");
    else
        PAMD_MERGE_STAGE("quad_perm:[1,0,3,2]");
}
#undef PAMD_MERGE_STAGE
template <int DBIT>
__device__ __forceinline__ void merge_sort8(uint32_t& k, uint32_t& x)
{
    merge_cmpex<4, (uint32_t)merge_keepmin_mask(4, DBIT)>(k, x);  // sequence distance 4: lane <-> 7 - lane
    merge_cmpex<2, (uint32_t)merge_keepmin_mask(2, DBIT)>(k, x);
    merge_cmpex<1, (uint32_t)merge_keepmin_mask(1, DBIT)>(k, x);
}
template <typename CTX, int CPP>
__device__ __forceinline__ bool select_merge(const CTX& c, uint32_t k0, uint32_t x0, uint32_t k1, uint32_t x1,
                                             uint32_t newP, float& nval, uint32_t& nidx)
{
    static_assert(CTX::LP == 4 && (CTX::LCW == 8 || CTX::LCW == 4 || CTX::LCW == 2), "8 / 4 / 2 path groups of 4 lanes per codeword");
    constexpr int LCW = CTX::LCW;
    constexpr uint32_t INFK = 0xFFFFFFFFu;
    uint32_t k = k0, x = x0, pk;
    if constexpr (CPP == 8) {  // a_i against b_(7-i): the other quad's second register, mirrored; keep the smaller
        asm volatile("s_nop 1\n\t"
                     "v_mov_b32_dpp %2, %3 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_cmp_ge_u32_e32 vcc, %2, %0\n\t"
                     "v_cndmask_b32_e32 %0, %2, %0, vcc\n\t"
                     "v_cndmask_b32_dpp %1, %4, %1, vcc row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1"
                     : "+v"(k), "+v"(x), "=&v"(pk)
                     : "v"(k1), "v"(x1)
                     : "vcc");
    }
    merge_sort8<(LCW > 2 ? 3 : -1)>(k, x);
    if constexpr (LCW > 2) {
    // the 8-lane neighbour group of the row (lane ^ 8) holds its list the other way round
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %2, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_cmp_ge_u32_e32 vcc, %2, %0\n\t"
                 "v_cndmask_b32_e32 %0, %2, %0, vcc\n\t"
                 "v_cndmask_b32_dpp %1, %1, %1, vcc row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "+v"(k), "+v"(x), "=&v"(pk)
                 :
                 : "vcc");
    merge_sort8<(LCW > 4 ? 4 : -1)>(k, x);
    }
    if constexpr (LCW > 4) {
        // lane ^ 16: v_permlane16_swap exchanges the odd rows of its first operand with the even rows of the second.
        // With both operands = k the results are {row 0, row 0, row 2, row 2} and {row 1, row 1, row 3, row 3}: in every
        // lane one of them is the partner's key and the other the lane's own, so no lane predicate is needed
        const auto sk = __builtin_amdgcn_permlane16_swap(k, k, false, false);
        const auto sx = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        const bool ta = sk[0] < k, tb = sk[1] < k;
        x = ta ? sx[0] : (tb ? sx[1] : x);
        k = ta ? sk[0] : (tb ? sk[1] : k);
        merge_sort8<-1>(k, x);
    }
    // new path pl inherits the candidate of rank pl
    const uint32_t pl = (uint32_t)c.pl;
    const uint32_t src = ((uint32_t)c.lane & ~7u) | (pl < 4u ? pl : 11u - pl);
    const uint32_t nx = dpp_u32<0x39>(k), n4 = dpp_u32<0x124>(k);
    const uint32_t rl = newP - 1u, tl = rl < 4u ? rl : 11u - rl;
    if constexpr (LCW < 8) {
        // four or eight codewords per wave: the checks of the two-codeword form below, without per-codeword scalars -- the
        // cut key comes from the lane's own 8-lane group (every group of a codeword holds the sorted winners), the count
        // "candidates at or below it" is the population of the codeword's segment of one ballot
        constexpr uint32_t G = 4u * (uint32_t)LCW;  // lanes per codeword
        const bool tie = k != INFK && (k == nx || k == n4);
        const uint32_t rk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)k);
        const uint32_t rx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)x);
        const uint32_t tv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((uint32_t)c.lane & ~7u) | tl) << 2), (int)k);
        const uint32_t sh = (uint32_t)c.lane & ~(G - 1u);
        uint32_t cnt = (uint32_t)__popc((uint32_t)(ballot64(k0 <= tv) >> sh) & ((1u << G) - 1u));
        if constexpr (CPP == 8) cnt += (uint32_t)__popc((uint32_t)(ballot64(k1 <= tv) >> sh) & ((1u << G) - 1u));
        if (ballot64(tie || cnt != newP) != 0) return false;
        nval = bitsf(rk);
        nidx = rx;
        return true;
    } else {
        // exact ties among the winners: ranks r | r + 1 are quad neighbours (rotation by one lane; its wrap-around pair and
        // the pairs of the second comparison that are not neighbours compare two ranks of the same list, so an equality
        // there is a tie as well), ranks 3 | 4 sit in lanes 3 | 7 (rotation of the row by four lanes) ...
        const bool tie = k != INFK && (k == nx || k == n4);
        // (the winners are fetched first: the LDS round trip overlaps the scalar chain of the checks)
        const uint32_t rk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)k);
        const uint32_t rx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)x);
        // ... and at the cut: exactly newP candidates of each codeword may be at or below the key of rank newP - 1
        const uint32_t t0 = readlane_u32(k, (int)tl), t1 = readlane_u32(k, (int)(32u + tl));
        uint32_t n0 = (uint32_t)__popc((uint32_t)ballot64(k0 <= t0)), n1 = (uint32_t)__popc((uint32_t)(ballot64(k0 <= t1) >> 32));
        if constexpr (CPP == 8) {
            n0 += (uint32_t)__popc((uint32_t)ballot64(k1 <= t0));
            n1 += (uint32_t)__popc((uint32_t)(ballot64(k1 <= t1) >> 32));
        }
        if (ballot64(tie) != 0 || n0 != newP || n1 != newP) return false;
        nval = bitsf(rk);
        nidx = rx;
        return true;
    }
}

template <typename CTX, int LPL, int CPP>
__device__ __forceinline__ void select_dispatch(const CTX& c, float (&val)[(CPP >> LPL) ? (CPP >> LPL) : 1],
                                                float pathbest, uint32_t newP, float& nval, uint32_t& nidx)
{
    if (c.ablate & 1u) {  // perf experiment (diagnostic build only): no selection
        nval = val[0];
        nidx = (uint32_t)c.pl * CPP;
        return;
    }
    nval = 0.0f;
    nidx = 0;
    if constexpr (CTX::CH) {
        // int8 decoder: equal metrics are the rule, the rank-based selections would fall back almost every time
    } else if constexpr (LPL == 3) {
        if (select_ranked8<CTX, CPP>(c, val[0], newP, nval, nidx)) return;
    } else if (c.P * CPP > newP) {
        // (when every candidate survives, the reference leaves the last position unsorted: literal code below)
        constexpr int CPL = (CPP >> LPL) ? (CPP >> LPL) : 1;
        if constexpr (CTX::CW == 1 && CPL >= 2) {
            if (select_threshold<CTX, LPL, CPP>(c, val, newP, nval, nidx)) return;
        } else {
            if (select_ranked<CTX, LPL, CPP>(c, val, newP, nval, nidx)) return;
        }
    }
    // literal emulation of the reference's swap order
    select_literal_cw<CTX, LPL, CPP>(c, val, newP, nval, nidx);
}

// write the n leaf bits of every (new) path: sign(x_src[i]) ^ flips, packed, into area(side)[stage]
// flips: up to 4 positions (fmask bit k enables fpos[k]);  REP passes n_from_llr = false
template <typename CTX, typename SRC>
__device__ __forceinline__ void write_leaf_bits(const CTX& c, const SRC& x, uint32_t s, uint32_t side,
                                                bool active, uint32_t fmask, const uint32_t fpos[4])
{
    constexpr int LP = CTX::LP;
    const uint32_t n = 1u << s;
    uint32_t* dst = c.bits(side, c.p, s);
    uint32_t acc = 0;
    constexpr uint32_t E = 4u * LP;  // LLRs per path and trip with 16-byte loads
    if (LP <= 8 && n >= E) {
        // every lane takes four consecutive LLRs, the sign nibbles of the path's lanes are OR-ed together (DPP), and
        // the flips are applied to the finished words
        for (uint32_t i0 = 0; i0 < n; i0 += E) {
            uint32_t nib = 0;
            if (active) {
                const float4 a = x.ld4(i0 + 4u * (uint32_t)c.j);
                nib = sgn(a.x) | (sgn(a.y) << 1) | (sgn(a.z) << 2) | (sgn(a.w) << 3);
            }
            acc |= group_or_u32<LP>(nib << (4u * (uint32_t)c.j)) << (i0 & 31u);
            if (((i0 + E) & 31u) == 0 || i0 + E >= n) {
                const uint32_t wi = i0 >> 5;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    acc ^= (((fmask >> k) & 1u) && (fpos[k] >> 5) == wi) ? (1u << (fpos[k] & 31u)) : 0u;
                if (active && c.j == 0) dst[wi] = acc;
                acc = 0;
            }
        }
        return;
    }
    for (uint32_t i0 = 0; i0 < n; i0 += LP) {
        const uint32_t i = i0 + c.j;
        uint32_t b = 0;
        if (active && i < n) {
            b = sgn(x[i]);
#pragma unroll
            for (int k = 0; k < 4; ++k) b ^= (((fmask >> k) & 1u) && fpos[k] == i) ? 1u : 0u;
        }
        const uint64_t bal = ballot64(b != 0);
        const uint32_t chunk = (uint32_t)(bal >> c.base) & (LP == 32 ? 0xFFFFFFFFu : ((1u << LP) - 1u));
        acc |= chunk << (i0 & 31u);
        if (((i0 + LP) & 31u) == 0 || i0 + LP >= n) {
            if (active && c.j == 0) dst[i0 >> 5] = acc;
            acc = 0;
        }
    }
}

// "Virtual" stage logN-1: the LLRs right below the root are never stored.  Left child of the root:
// f(x_i, x_{i+N/2}); right child: g(x_i, x_{i+N/2}, u_i) with u = the root's left-child bits of the path (held by
// the path buffer its slot word names).  Four consecutive values from two 16-byte channel loads.
template <bool I8, bool CH = false>
__device__ __forceinline__ float4 virt_top4(const void* chan, uint32_t halfN, uint32_t i, bool right, const uint32_t* rootleft)
{
    const Rd<I8 ? K_GI8 : K_GF32> x{ chan };
    const float4 a = x.ld4(i), b = x.ld4(halfN + i);
    if (right) {
        const uint32_t u = (rootleft[i >> 5] >> (i & 31u)) & 0xFu;
        return make_float4(polar_gx<CH>(a.x, b.x, u & 1u), polar_gx<CH>(a.y, b.y, (u >> 1) & 1u), polar_gx<CH>(a.z, b.z, (u >> 2) & 1u),
                           polar_gx<CH>(a.w, b.w, u >> 3));
    }
    return make_float4(polar_fx<CH>(a.x, b.x), polar_fx<CH>(a.y, b.y), polar_fx<CH>(a.z, b.z), polar_fx<CH>(a.w, b.w));
}

#ifndef PAMD_WAVES_PER_EU
#define PAMD_WAVES_PER_EU 4
#endif
// 4 waves per SIMD (<= 128 VGPRs) for the L <= 8 production kernels: occupancy is what hides the LDS / arena latency
// IN: 0 = fp32 channel rows, 1 = int8 rows widened to float (the float decoder's decode_vector(const char*)), 2 = int8
// rows AND int8 arithmetic (SclFipChar)
template <int LPL, int IN, bool S64, bool DIAG, bool HB, int CWL>
// (at most 128 VGPRs for the kernels that reach 16 waves per CU; capping at 96 for 20 waves measured slower: spills)
__global__ __launch_bounds__(256, ((LPL == 3 || CWL > 0) && !DIAG) ? PAMD_WAVES_PER_EU : 1) void scl_decode_kernel(DecodeParams P)
{
    constexpr bool I8 = IN != 0, CH = IN == 2;
    using Ctx = SclCtx<LPL, S64, HB, CWL, CH>;
    constexpr int LCW = Ctx::LCW;  // path groups per codeword
    constexpr int CW = Ctx::CW;    // codewords per wave
    constexpr int LP = Ctx::LP;
    constexpr int LMAX = Ctx::LMAX;
    constexpr int NACC = Ctx::NACC;
    extern __shared__ uint32_t smem[];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const uint32_t wpb = blockDim.x >> 6;
    uint32_t* wl = smem + (size_t)wib * P.wave_lds_words;

    Ctx c;
    c.llr = reinterpret_cast<float*>(wl);
    c.bitsL = wl + (size_t)LMAX * P.path_stride;
    c.bitsR = c.bitsL + (size_t)LMAX * P.wppL;
    c.scratch = wl + (((size_t)LMAX * P.path_stride + (size_t)LMAX * (P.wppL + P.wppR) + 3) & ~(size_t)3);  // 16-byte aligned
    c.path_stride = P.path_stride;
    c.wppL = P.wppL;
    c.wppR = P.wppR;
    c.arena = P.arena + (size_t)(blockIdx.x * wpb + wib) * P.arena_wave_words;
    c.arena_path_stride = P.arena_path_stride;
    c.gbits = reinterpret_cast<uint32_t*>(c.arena + P.arena_bits_off);
    c.sb = P.sb;
    c.wsb = woff(P.sb);
    c.gwl = P.gbit_left_words;
    c.gwpp = P.gbit_path_words;
    c.s0 = P.s0;
    c.logN = P.logN;
    c.ablate = DIAG ? P.ablate : 0u;
    c.lane = lane;
    c.p = lane >> LPL;
    c.pl = c.p & (LCW - 1);
    c.g0 = c.p & ~(LCW - 1);
    c.j = lane & (LP - 1);
    c.base = lane & ~(LP - 1);
    c.L = P.L;
    const uint32_t N = P.N, logN = P.logN;
    const uint32_t wave0 = blockIdx.x * wpb + wib, nwaves = gridDim.x * wpb;
    const const_u64_ptr prog64 = (const_u64_ptr)(uintptr_t)P.prog;

    // ("mixed" decoder: the number of rows the SC pass failed on is only known on the device)
    const uint32_t nrows = P.rows_dev ? (uint32_t)__builtin_amdgcn_readfirstlane((int)*P.rows_dev) : P.rows;
    const uint32_t ngroups = (nrows + CW - 1) / CW;
    const uint32_t mycw = (uint32_t)c.p >> (6 - LPL - CWL);  // which of the wave's codewords this lane works for
    for (uint32_t cw = wave0; cw < ngroups; cw += nwaves) {
        // channel rows of the wave's codewords (wave uniform); the last group may be ragged: idle slots decode the
        // group's first row again and store nothing
        const void* chanv[CW];
#pragma unroll
        for (int t = 0; t < CW; ++t) {
            const uint32_t ci = cw * CW + t, cl = ci < nrows ? ci : cw * CW;
            const uint32_t r = P.rowidx ? P.rowidx[cl] : cl;
            chanv[t] = I8 ? (const void*)(reinterpret_cast<const int8_t*>(P.llr) + (size_t)r * N)
                          : (const void*)(reinterpret_cast<const float*>(P.llr) + (size_t)r * N);
        }
        auto chan_of = [&](uint32_t t) -> const void* {  // (t: per lane)
            const void* r = chanv[0];
#pragma unroll
            for (int k = 1; k < CW; ++k) r = (t == (uint32_t)k) ? chanv[k] : r;
            return r;
        };
        const uint32_t cwi = cw * CW + mycw;
        const bool live = cwi < nrows;
        const uint32_t row = P.rowidx ? P.rowidx[live ? cwi : cw * CW] : (live ? cwi : cw * CW);
        const void* chan = chan_of(mycw);
        c.chan = chan;
        // makeInitialPathList :705-709 -- one path, metric 0 (fresh decoder)
        c.P = 1;
        c.metric = 0.0f;
        c.slot = 0;
        uint32_t trace_n = 0;

        const_u64_ptr pp = prog64;
        uint64_t iw_next = pp[0];
        int lane_v = threadIdx.x & 63;
        // (P.prog_len = instructions in front of the OP_END word, which is still there: the fetch runs one ahead)
        for (uint32_t pc = 0; pc < P.prog_len; ++pc) {
            // The lane number is made opaque once per instruction so that nothing derived from it (addresses, lane
            // predicates as 64-bit SGPR masks) is hoisted out of the interpreter loop: the kernel is at its register
            // limits (128 VGPRs, 102 SGPRs) and hoisted values were spilled; re-deriving them costs a few VALU ops.
            asm volatile("" : "+v"(lane_v));
            const int lane = lane_v;
            c.lane = lane;
            c.p = lane >> LPL;
            c.pl = c.p & (LCW - 1);
            c.g0 = c.p & ~(LCW - 1);
            c.j = lane & (LP - 1);
            c.base = lane & ~(LP - 1);
            const uint64_t iw = iw_next;
            Ins ins;
            ins.op = (uint8_t)(iw & 0xFF);
            ++pp;
            iw_next = pp[0];  // scalar load, in flight while this instruction executes
            ins.stage = (uint8_t)((iw >> 8) & 0xFF);
            ins.side = (uint8_t)((iw >> 16) & 0xFF);
            const uint32_t s = ins.stage, n = 1u << s, h = n >> 1, side = ins.side & 1u;
            // host flags (capi.cpp, specialize_scl): 2 = the output stage is virtual (nothing to store),
            // 4 = the input stage is virtual (recomputed from the channel row), 8 = ... of the root's RIGHT child
            const uint32_t vflags = ins.side & 0xEu;
            const bool active = (uint32_t)c.pl < c.P;
            // wide ops walk the live paths of all codewords of the wave: path number ql in [0, CW * P) -> its group
            // (= path buffer) and codeword t
            auto item_group = [&](uint32_t ql, uint32_t& t) -> uint32_t {
                if constexpr (CWL == 0) {
                    t = 0;
                    return ql;
                } else {
                    t = 0;
#pragma unroll
                    for (uint32_t m = 1; m < (uint32_t)CW; ++m) t += (ql >= m * c.P) ? 1u : 0u;
                    return t * (uint32_t)LCW + (ql - t * c.P);
                }
            };
            uint64_t t_start = 0;
            if (DIAG && P.cycles && cw == 0) t_start = __builtin_amdgcn_s_memtime();
#define PAMD_STAMP(k)                                                                             \
    if (DIAG && P.cycles && cw == 0) {                                                            \
        const uint64_t t_ = __builtin_amdgcn_s_memtime();                                         \
        if (lane == 0) P.cycles[P.prog_len + pc * 4 + (k)] = (uint32_t)(t_ - t_start);            \
    }

            const bool is_leaf = (ins.op >= SCL_R0 && ins.op <= SCL_SPC) || ins.op == SCL_SPC_S || ins.op == SCL_R1_S;
            if (DIAG && ((P.ablate & 128u) || ((P.ablate & 64u) && is_leaf))) continue;
            // dispatch ordered by frequency (F / G / COMBINE are three quarters of the program); the two forking
            // leaves share one generic body instantiated separately, so each variant is straight-line code
            // Host-fused neighbours of a small leaf (capi.cpp, fuse_small_leaves; 4 lanes per path, n <= 8): the F / G
            // of the parent that produces the leaf's input runs as the leaf's prologue (its output is consumed in
            // registers and never stored), the COMBINEs that follow a right leaf (or a COMBINE) as its epilogue.
            //   bits 0-1: input is 1 = F of the parent's LLRs, 2 = G, 3 = G after an all-frozen left sibling (whose
            //             F + RateZero step runs here too);  bits 2-4: COMBINEs that follow;  bit 5: side of the last
            constexpr bool kFuse = LPL == 2 || LPL == 3;  // (8 lanes per path: lane j holds LLR j, x1 is unused)
            constexpr bool kMerge = LPL == 2 && CWL >= 1;  // candidate selection by select_merge (4 lanes per path, 8 / 4 / 2 paths)
            constexpr bool kTop8 = LPL == 2 && CWL == 1;   // the N = 1024 chunk-major ops (SCL_VTOP, SCL_VG2): 8 paths x 2 codewords
            const uint32_t fuse = kFuse ? ((uint32_t)(iw >> 24) & 0xFFu) : 0u;
            auto leaf_input4 = [&](float& x0, float& x1) {  // LLRs j and j + 4 of this lane's path
                const bool in0 = (uint32_t)c.j < n, in1 = LP == 4 && n == 8;
                const uint32_t i0 = in0 ? (uint32_t)c.j : 0u, i1 = in1 ? (uint32_t)c.j + 4u : 0u;
                const uint32_t pro = fuse & 3u;
                if (pro == 0) {
                    const float* xb = c.path_llr(c.p) + n;
                    x0 = xb[i0];
                    x1 = xb[i1];
                } else {
                    constexpr typename Ctx::SlotT FM = (typename Ctx::SlotT)(LCW - 1);
                    const uint32_t psh = s * (uint32_t)Ctx::FB;  // slot field of the parent stage s + 1
                    const uint32_t grand = LPL == 2 ? (((uint32_t)(iw >> 56)) & 3u) : 0u;
                    if (grand) {
                        // n = 4, pro = 1, and the stage-3 block the F reads is itself the output of the F_S / G_S / G0_S
                        // (grand = 1 / 2 / 3) of stage 4 right before: that op runs here too -- lane j produces LLRs j
                        // and j + 4 of stage 3 (stored: the parent's G needs them later), exactly the pair its F takes
                        const uint32_t gsh = 3u * (uint32_t)Ctx::FB;  // slot field of stage 4
                        const uint32_t srcq = grand == 2 ? (uint32_t)c.g0 + ((uint32_t)(c.slot >> gsh) & (uint32_t)(LCW - 1)) : (uint32_t)c.p;
                        const float* X = c.path_llr(srcq) + 16;
                        float* d = c.path_llr(c.p) + 8;
                        const float a0 = X[c.j], b0 = X[8 + c.j], a1 = X[4 + c.j], b1 = X[12 + c.j];
                        float v0, v1;
                        if (grand == 1) {
                            v0 = polar_fx<CH>(a0, b0);
                            v1 = polar_fx<CH>(a1, b1);
                        } else if (grand == 2) {
                            const uint32_t lw = (c.area(0, c.p) + woff(3))[0];
                            v0 = polar_gx<CH>(a0, b0, (lw >> c.j) & 1u);
                            v1 = polar_gx<CH>(a1, b1, (lw >> (c.j + 4)) & 1u);
                        } else {
                            float acc[NACC];
#pragma unroll
                            for (int a = 0; a < NACC; ++a) acc[a] = 0.0f;
                            acc[0] = acc[0] + min0(polar_fx<CH>(a0, b0));
                            if constexpr (NACC > 1) acc[1] = acc[1] + min0(polar_fx<CH>(a1, b1));
                            const float sum = c.lanesum_finish(acc);
                            if (active) {
                                c.metric = c.metric + sum;
                                if (c.j == 0) (c.area(0, c.p) + woff(3))[0] = 0u;
                            }
                            v0 = addx<CH>(a0, b0);
                            v1 = addx<CH>(a1, b1);
                        }
                        if (active) {
                            d[c.j] = v0;
                            d[c.j + 4] = v1;
                        }
                        c.slot = (c.slot & ~(FM << gsh)) | ((typename Ctx::SlotT)c.pl << gsh);
                        x0 = polar_fx<CH>(v0, v1);
                        x1 = 0.0f;
                    } else
                    if (pro == 1) {  // SCL_F_S
                        const float* X = c.path_llr(c.p) + 2 * n;
                        x0 = polar_fx<CH>(X[i0], X[n + i0]);
                        x1 = polar_fx<CH>(X[i1], X[n + i1]);
                    } else if (pro == 2) {  // SCL_G_S
                        const uint32_t srcq = (uint32_t)c.g0 + ((uint32_t)(c.slot >> psh) & (uint32_t)(LCW - 1));
                        const float* X = c.path_llr(srcq) + 2 * n;
                        const uint32_t lw = (c.area(0, c.p) + woff(s))[0];
                        x0 = polar_gx<CH>(X[i0], X[n + i0], (lw >> i0) & 1u);
                        x1 = polar_gx<CH>(X[i1], X[n + i1], (lw >> i1) & 1u);
                    } else {  // SCL_F_S, SCL_R0 on its output (:316-337, no fork), SCL_G_S with all-zero left bits
                        const float* X = c.path_llr(c.p) + 2 * n;
                        const float a0 = X[i0], b0 = X[n + i0], a1 = X[i1], b1 = X[n + i1];
                        float acc[NACC];
#pragma unroll
                        for (int a = 0; a < NACC; ++a) acc[a] = 0.0f;
                        acc[0] = acc[0] + (in0 ? min0(polar_fx<CH>(a0, b0)) : 0.0f);
                        if constexpr (NACC > 1) acc[1] = acc[1] + (in1 ? min0(polar_fx<CH>(a1, b1)) : 0.0f);
                        const float sum = c.lanesum_finish(acc);
                        if (active) c.metric = c.metric + sum;
                        x0 = addx<CH>(a0, b0);
                        x1 = addx<CH>(a1, b1);
                    }
                    c.slot = (c.slot & ~(FM << psh)) | ((typename Ctx::SlotT)c.pl << psh);
                }
            };
            // k COMBINEs on top of the block r of stage st (right child of its parent each time), h < 32 throughout
            // (fuse bit 6: the chain reaches the 32-bit block of stage 5 and the two-word COMBINE of stage 6 follows)
            auto combine_chain = [&](uint32_t r, uint32_t st, uint32_t k, uint32_t fside, bool left0) {
                for (uint32_t t = 0; t < k; ++t) {
                    const uint32_t lq = (uint32_t)c.g0 + ((uint32_t)(c.slot >> ((st + t) * (uint32_t)Ctx::FB)) & (uint32_t)(LCW - 1));
                    const uint32_t nb = 1u << (st + t), mask = (1u << nb) - 1u;
                    const uint32_t left = (left0 && t == 0) ? 0u : (c.area(0, lq) + woff(st + t))[0];
                    r &= mask;
                    r = ((left ^ r) & mask) | (r << nb);
                }
                if (fuse & 64u) {
                    const uint32_t lq = (uint32_t)c.g0 + ((uint32_t)(c.slot >> (5u * (uint32_t)Ctx::FB)) & (uint32_t)(LCW - 1));
                    uint32_t* ob = c.area(fside, c.p) + woff(6);
                    ob[0] = (c.area(0, lq) + woff(5))[0] ^ r;
                    ob[1] = r;
                } else {
                    (c.area(fside, c.p) + woff(st + k))[0] = r;
                }
            };
            auto leaf_bits_out = [&](bool wr, uint32_t bw) {  // wr: lane 0 of a live path; bw: the leaf's n bits
                const uint32_t k = (fuse >> 2) & 7u;
                if (k == 0) {
                    if (wr) (c.area(side, c.p) + woff(s))[0] = bw;
                } else if (wr) {
                    combine_chain(bw, s, k, (fuse >> 5) & 1u, (fuse & 3u) == 3u);
                }
            };
            auto fork_leaf = [&](auto SPC_T, auto SMALL_T) {  // SpcDecoder::decode     :498-621
                constexpr bool spc = decltype(SPC_T)::value;
                Weak<4> wk;
                uint32_t par = 0;
                auto search = [&](auto x) {
                    if (spc) {
                        uint32_t pb = 0;
                        find_weak<Ctx, 4>(c, x, n, wk, &pb);
                        par = group_xor_u32<LP>(pb) & 1u;
                    } else {
                        Weak<2> w2;
                        find_weak<Ctx, 2>(c, x, n, w2);
                        wk.v[0] = w2.v[0];
                        wk.v[1] = w2.v[1];
                        wk.ix[0] = w2.ix[0];
                        wk.ix[1] = w2.ix[1];
                        wk.v[2] = wk.v[3] = 0;
                        wk.ix[2] = wk.ix[3] = 0;
                    }
                };
                // n <= 8 with the block in LDS and 8 lanes per path: decided on the host (SCL_SPC_S / SCL_R1_S)
                constexpr bool kSmallOp = decltype(SMALL_T)::value;
                bool small = kSmallOp;
                uint32_t signword = 0;  // (4 lanes per path, small leaf) the signs of the path's block, bit i = LLR i
                if (DIAG && (P.ablate & 2u)) {  // perf experiment: no weak search
                    small = true;
                    wk.v[0] = wk.v[1] = wk.v[2] = wk.v[3] = 0x3F800000u;
                    wk.ix[0] = 0; wk.ix[1] = 1; wk.ix[2] = 2; wk.ix[3] = 3;
                } else
                if constexpr (LPL == 2) {
                    if (small && !(DIAG && (P.ablate & 2u))) {  // n <= 8: positions j and j + 4 of the block in lane j
                        constexpr uint32_t INFB = 0xFFFFFFFFu;
                        const bool in0 = (uint32_t)c.j < n, in1 = n == 8;
                        float x0, x1;
                        leaf_input4(x0, x1);
                        const uint32_t a0 = in0 ? magbits<CH>(x0) : INFB, a1 = in1 ? magbits<CH>(x1) : INFB;
                        const uint64_t sg0 = ballot64(in0 && sgn(x0)), sg1 = ballot64(in1 && sgn(x1));
                        signword = ((uint32_t)(sg0 >> c.base) & 0xFu) | (((uint32_t)(sg1 >> c.base) & 0xFu) << 4);
                        if (spc) {
                            if (s == 2)
                                weak_small4<Ctx, 2, 4>(c, a0, a1, wk);
                            else
                                weak_small4<Ctx, 3, 4>(c, a0, a1, wk);
                            par = (uint32_t)__popc(signword) & 1u;
                        } else {
                            Weak<2> w2;
                            if (s == 1)
                                weak_small4<Ctx, 1, 2>(c, a0, a1, w2);
                            else if (s == 2)
                                weak_small4<Ctx, 2, 2>(c, a0, a1, w2);
                            else
                                weak_small4<Ctx, 3, 2>(c, a0, a1, w2);
                            wk.v[0] = w2.v[0];
                            wk.v[1] = w2.v[1];
                            wk.ix[0] = w2.ix[0];
                            wk.ix[1] = w2.ix[1];
                            wk.v[2] = wk.v[3] = 0;
                            wk.ix[2] = wk.ix[3] = 0;
                        }
                    }
                }
                if constexpr (LPL == 3) {
                    if (small && !(DIAG && (P.ablate & 2u))) {
                        constexpr uint32_t INFB = 0xFFFFFFFFu;
                        const bool in = (uint32_t)c.j < n;
                        float xj, xunused;
                        leaf_input4(xj, xunused);
                        const uint32_t a = in ? magbits<CH>(xj) : INFB;
                        const uint64_t sg = ballot64(in && sgn(xj));
                        signword = (uint32_t)(sg >> c.base) & 0xFFu;
                        if (spc) {
                            if (s == 2)
                                find_weak_small<2, 4>(a, c.j, wk);
                            else
                                find_weak_small<3, 4>(a, c.j, wk);
                            par = (uint32_t)__popc(signword) & 1u;
                        } else {
                            Weak<2> w2;
                            if (s == 1)
                                find_weak_small<1, 2>(a, c.j, w2);
                            else if (s == 2)
                                find_weak_small<2, 2>(a, c.j, w2);
                            else
                                find_weak_small<3, 2>(a, c.j, w2);
                            wk.v[0] = w2.v[0];
                            wk.v[1] = w2.v[1];
                            wk.ix[0] = w2.ix[0];
                            wk.ix[1] = w2.ix[1];
                            wk.v[2] = wk.v[3] = 0;
                            wk.ix[2] = wk.ix[3] = 0;
                        }
                    }
                }
                if constexpr (!kSmallOp) {
                    if (!small) c.template with_src<I8>(s, c.p, search);
                }
                PAMD_STAMP(0)
                const float v0 = bitsf(wk.v[0]), v1 = bitsf(wk.v[1]), v2 = bitsf(wk.v[2]), v3 = bitsf(wk.v[3]);
                uint32_t nidx = 0;
                float nval = 0.0f;
                uint32_t newP;
                if (spc) {
                    constexpr int CPL = (8 >> LPL) ? (8 >> LPL) : 1;
                    const float m = par ? c.metric - v0 : c.metric;
                    const float beta = par ? m - 0.0f : m - v0;
                    newP = (c.P * 8 < c.L) ? c.P * 8 : c.L;
                    auto literal = [&]() {  // candidates in the reference's order: position q of path pl in lane q / CPL
                        float val[CPL];
#pragma unroll
                        for (int r = 0; r < CPL; ++r) {
                            const uint32_t q = (uint32_t)c.j * CPL + r;
                            float t;
                            switch (q) {
                            case 0: t = m; break;
                            case 1: t = beta - v1; break;
                            case 2: t = beta - v2; break;
                            case 3: t = beta - v3; break;
                            case 4: t = (m - v1) - v2; break;
                            case 5: t = (m - v1) - v3; break;
                            case 6: t = (m - v2) - v3; break;
                            default: t = ((beta - v1) - v2) - v3; break;
                            }
                            val[r] = t;
                        }
                        if constexpr (kMerge)
                            select_literal_cw<Ctx, LPL, 8>(c, val, newP, nval, nidx);
                        else
                            select_dispatch<Ctx, LPL, 8>(c, val, m, newP, nval, nidx);
                    };
                    if constexpr (kMerge) {
                        bool done = false;
                        if (!CH && c.P * 8 > newP && !(c.ablate & 1u)) {
                            // the path's candidates best first: c0 c1 c2 {c3, c4} c5 c6 c7 -- element i in register
                            // i / 4 of lane i % 4 (select_merge); branch-free per-lane selects
                            const uint32_t j = (uint32_t)c.j;
                            const float t1 = m - v1, t2 = m - v2;
                            const float c3 = beta - v3, c4 = t1 - v2;
                            const bool sw = fbits(c4) < fbits(c3);  // (keys: smaller = better)
                            const float vs = j == 1 ? v1 : (j == 2 ? v2 : v3);
                            float e0 = j == 0 ? m : beta - vs;
                            e0 = (j == 3 && sw) ? c4 : e0;
                            const float a3 = (beta - v1) - v2;
                            const float A = j == 3 ? a3 : (j == 2 ? t2 : t1);
                            float e1 = A - (j == 0 ? v2 : v3);
                            e1 = (j == 0 && sw) ? c3 : e1;
                            const uint32_t q0 = (j == 3 && sw) ? 4u : j, q1 = (j == 0 && sw) ? 3u : 4u + j;
                            const uint32_t pb = (uint32_t)c.pl << 3;
                            done = select_merge<Ctx, 8>(c, active ? fbits(e0) : 0xFFFFFFFFu, pb | q0,
                                                        active ? fbits(e1) : 0xFFFFFFFFu, pb | q1, newP, nval, nidx);
                        }
                        if (!done) literal();
                    } else {
                        literal();
                    }
                } else {
                    constexpr int CPL = (4 >> LPL) ? (4 >> LPL) : 1;
                    const float m = c.metric;
                    newP = (c.P * 4 < c.L) ? c.P * 4 : c.L;
                    auto literal = [&]() {
                        float val[CPL];
#pragma unroll
                        for (int r = 0; r < CPL; ++r) {
                            const uint32_t q = (uint32_t)c.j * CPL + r;
                            float t;
                            switch (q) {
                            case 0: t = m; break;
                            case 1: t = m - v0; break;
                            case 2: t = m - v1; break;
                            default: t = (m - v0) - v1; break;
                            }
                            val[r] = t;
                        }
                        if constexpr (kMerge)
                            select_literal_cw<Ctx, LPL, 4>(c, val, newP, nval, nidx);
                        else
                            select_dispatch<Ctx, LPL, 4>(c, val, m, newP, nval, nidx);
                    };
                    if constexpr (kMerge) {
                        bool done = false;
                        if (!CH && c.P * 4 > newP && !(c.ablate & 1u)) {  // m >= m - v0 >= m - v1 >= (m - v0) - v1: lane j holds candidate j
                            const uint32_t j = (uint32_t)c.j;
                            const float d0 = m - v0;
                            const float e0 = j == 0 ? m : (j == 1 ? d0 : (j == 2 ? m - v1 : d0 - v1));
                            done = select_merge<Ctx, 4>(c, active ? fbits(e0) : 0xFFFFFFFFu, ((uint32_t)c.pl << 2) | j,
                                                        0xFFFFFFFFu, 0u, newP, nval, nidx);
                        }
                        if (!done) literal();
                    } else {
                        literal();
                    }
                }
                PAMD_STAMP(1)
                const uint32_t srcp = (uint32_t)c.g0 + (spc ? (nidx >> 3) : (nidx >> 2));  // the source path's group
                const uint32_t q = spc ? (nidx & 7u) : (nidx & 3u);
                const int sl = (int)(srcp << LPL);
                const typename Ctx::SlotT nslot = c.slot_from_lane(sl);
                // flip positions (< 2^15 each) and the parity travel packed in two registers
                // (small leaves at 4 lanes per path: positions < 8, and the source path's sign word rides along in pk0)
                constexpr bool kSignWord = kSmallOp && (LPL == 2 || LPL == 3);
                const uint32_t pk0 = shfl_u32(wk.ix[0] | (kSignWord ? (signword << 8) : 0u) | (wk.ix[1] << 16), sl);
                const uint32_t pk1 = shfl_u32(wk.ix[2] | (wk.ix[3] << 16) | (par << 31), sl);
                const uint32_t fpos[4] = { kSignWord ? (pk0 & 0xFFu) : (pk0 & 0xFFFFu), pk0 >> 16, pk1 & 0xFFFFu,
                                           (pk1 >> 16) & 0x7FFFu };
                const uint32_t spar = pk1 >> 31;
                uint32_t fmask;
                if (spc) {
                    // flip sets :533-585 as bit masks over {i0,i1,i2,i3}; odd parity toggles i0
                    const uint32_t even = (0xFCA69530u >> (4 * q)) & 0xFu;
                    fmask = even ^ (spar ? 1u : 0u);
                } else {
                    fmask = q;
                }
                c.P = newP;
                const bool nact = (uint32_t)c.pl < c.P;
                if (nact) {
                    c.metric = nval;
                    c.slot = nslot;
                }
                PAMD_STAMP(2)
                // bits of the new path = signs of the SOURCE path's LLRs (identity buffer) with flips
                if constexpr (kSignWord) {
                    // bits of the new path = the source path's sign word with the flip positions toggled: no LLR reads
                    uint32_t fw = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) fw |= ((fmask >> k) & 1u) << fpos[k];
                    leaf_bits_out(nact && c.j == 0, (((pk0 >> 8) & 0xFFu) ^ fw) & ((1u << n) - 1u));
                } else if (small && LPL == 2) {  // n <= 8: LLRs j and j + 4 of the SOURCE path per lane, two ballot nibbles per path
                    const float* xs = c.path_llr(srcp) + n;
                    const bool in0 = (uint32_t)c.j < n, in1 = n == 8;
                    uint32_t b0 = 0, b1 = 0;
                    if (nact && in0) {
                        b0 = sgn(xs[c.j]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) b0 ^= (((fmask >> k) & 1u) && fpos[k] == (uint32_t)c.j) ? 1u : 0u;
                    }
                    if (nact && in1) {
                        b1 = sgn(xs[c.j + 4]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) b1 ^= (((fmask >> k) & 1u) && fpos[k] == (uint32_t)c.j + 4u) ? 1u : 0u;
                    }
                    const uint64_t bal0 = ballot64(b0 != 0), bal1 = ballot64(b1 != 0);
                    if (nact && c.j == 0)
                        (c.area(side, c.p) + woff(s))[0] = ((uint32_t)(bal0 >> c.base) & 0xFu) | (((uint32_t)(bal1 >> c.base) & 0xFu) << 4);
                } else if (small) {  // n <= 8: one LLR of the SOURCE path per lane, one ballot byte per path
                    const bool in = (uint32_t)c.j < n;
                    uint32_t b = 0;
                    if (nact && in) {
                        b = sgn((c.path_llr(srcp) + n)[c.j]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) b ^= (((fmask >> k) & 1u) && fpos[k] == (uint32_t)c.j) ? 1u : 0u;
                    }
                    const uint64_t bal = ballot64(b != 0);
                    if (nact && c.j == 0) (c.area(side, c.p) + woff(s))[0] = (uint32_t)(bal >> c.base) & 0xFFu;
                } else if constexpr (!kSmallOp) {
                    c.template with_src<I8>(s, srcp,
                                            [&](auto x) { write_leaf_bits<Ctx>(c, x, s, side, nact, fmask, fpos); });
                }
            };
            const uint32_t aux = (uint32_t)(iw >> 32);
            const uint32_t wo_s = aux & 0x1FFu, wo_c = (aux >> 9) & 0x1FFu, fsh = (aux >> 18) & 0x3Fu;
            const bool skip_small = DIAG && (P.ablate & 16u);
            if constexpr (LPL == 3 || LPL == 2) {
                if (ins.op == SCL_SPC_S) {
                    fork_leaf(std::true_type{}, std::true_type{});
                    goto op_done;
                }
                if (ins.op == SCL_R1_S) {
                    fork_leaf(std::false_type{}, std::true_type{});
                    goto op_done;
                }
            }
            if (ins.op == SCL_F_S) {  // F, both stages in LDS, h < 64 (specialised on the host)
                if (active && !skip_small) {
                    const float* x = c.path_llr(c.p) + n;
                    float* d = c.path_llr(c.p) + h;
                    if (h >= 4) {  // 16-byte LDS accesses: a lane takes four consecutive outputs per trip
                        const Rd<K_LDS> xr{ x };
                        const Wr<K_LDS> dw{ d };
                        for (uint32_t i = 4u * (uint32_t)c.j; i < h; i += 4u * LP) {
                            const float4 a = xr.ld4(i), b = xr.ld4(h + i);
                            dw.st4(i, make_float4(polar_fx<CH>(a.x, b.x), polar_fx<CH>(a.y, b.y), polar_fx<CH>(a.z, b.z), polar_fx<CH>(a.w, b.w)));
                        }
                    } else {
                        for (uint32_t i = c.j; i < h; i += LP) d[i] = polar_fx<CH>(x[i], x[h + i]);
                    }
                }
                c.slot = (c.slot & ~((typename Ctx::SlotT)(LCW - 1) << fsh)) | ((typename Ctx::SlotT)c.pl << fsh);
            } else if (ins.op == SCL_G_S) {  // G, both stages in LDS, h <= 32: one left-bit word
                if (active && !skip_small) {
                    const uint32_t srcq = (uint32_t)c.g0 + ((uint32_t)(c.slot >> fsh) & (uint32_t)(LCW - 1));
                    const float* x = c.path_llr(srcq) + n;
                    float* d = c.path_llr(c.p) + h;
                    const uint32_t lw = (c.area(0, c.p) + wo_c)[0];
                    if (h >= 4) {
                        const Rd<K_LDS> xr{ x };
                        const Wr<K_LDS> dw{ d };
                        for (uint32_t i = 4u * (uint32_t)c.j; i < h; i += 4u * LP) {
                            const float4 a = xr.ld4(i), b = xr.ld4(h + i);
                            const uint32_t u = lw >> i;
                            dw.st4(i, make_float4(polar_gx<CH>(a.x, b.x, u & 1u), polar_gx<CH>(a.y, b.y, (u >> 1) & 1u),
                                                  polar_gx<CH>(a.z, b.z, (u >> 2) & 1u), polar_gx<CH>(a.w, b.w, (u >> 3) & 1u)));
                        }
                    } else {
                        for (uint32_t i = c.j; i < h; i += LP) d[i] = polar_gx<CH>(x[i], x[h + i], (lw >> i) & 1u);
                    }
                }
                c.slot = (c.slot & ~((typename Ctx::SlotT)(LCW - 1) << fsh)) | ((typename Ctx::SlotT)c.pl << fsh);
            } else if (ins.op == SCL_G0_S) {  // all-frozen left child: F_S + RateZeroDecoder::decode :316-337 + G_S (u = 0)
                float acc[NACC];
#pragma unroll
                for (int a = 0; a < NACC; ++a) acc[a] = 0.0f;
                const float* x = c.path_llr(c.p) + n;
                float* d = c.path_llr(c.p) + h;
                if (active && (LP < 8 || c.j < 8)) {
#pragma unroll
                    for (int a = 0; a < NACC; ++a) {
                        const uint32_t v = (uint32_t)c.j + (uint32_t)a * LP;
                        for (uint32_t i = v; i < h; i += 8) {
                            const float xa = x[i], xb = x[h + i];
                            acc[a] = acc[a] + min0(polar_fx<CH>(xa, xb));
                            if constexpr (LP <= 8) d[i] = addx<CH>(xa, xb);
                        }
                        if (v >= h) acc[a] = acc[a] + 0.0f;
                    }
                }
                if constexpr (LP > 8) {
                    if (active)
                        for (uint32_t i = c.j; i < h; i += LP) d[i] = addx<CH>(x[i], x[h + i]);
                }
                const float sum = c.lanesum_finish(acc);
                if (active) {
                    c.metric = c.metric + sum;
                    if (c.j == 0) (c.area(0, c.p) + wo_c)[0] = 0u;  // the left child's bits, for the COMBINE
                }
                c.slot = (c.slot & ~((typename Ctx::SlotT)(LCW - 1) << fsh)) | ((typename Ctx::SlotT)c.pl << fsh);
            } else if (ins.op == SCL_C_S) {  // COMBINE, h < 32
                if (kFuse && fuse) {  // ... and the COMBINEs right above it
                    if (active && c.j == 0) combine_chain((c.area(1, c.p) + wo_c)[0], s - 1, (fuse >> 2) & 7u, (fuse >> 5) & 1u, false);
                } else
                if (active && c.j == 0 && !skip_small) {
                    const uint32_t lq = (uint32_t)c.g0 + ((uint32_t)(c.slot >> fsh) & (uint32_t)(LCW - 1));
                    const uint32_t mask = (1u << h) - 1u;
                    const uint32_t r = (c.area(1, c.p) + wo_c)[0] & mask;
                    (c.area(side, c.p) + wo_s)[0] = (((c.area(0, lq) + wo_c)[0] ^ r) & mask) | (r << h);
                }
            } else if (ins.op == SCL_C_W) {  // COMBINE, h >= 32
                if (active && !skip_small) {
                    const uint32_t lq = (uint32_t)c.g0 + ((uint32_t)(c.slot >> fsh) & (uint32_t)(LCW - 1));
                    const uint32_t* lb = c.bits(0, lq, s - 1);
                    const uint32_t* rb = c.bits(1, c.p, s - 1);
                    uint32_t* ob = c.bits(side, c.p, s);
                    const uint32_t hw = h >> 5;
                    for (uint32_t w = c.j; w < hw; w += LP) {
                        const uint32_t r = rb[w];
                        ob[w] = lb[w] ^ r;
                        ob[hw + w] = r;
                    }
                }
            } else if (ins.op == SCL_F) {  // RateRNode::decode :233-238 -- reads the path's own (identity) buffer
                if (DIAG && (P.ablate & 8u) && h >= 64) {
                } else if (DIAG && (P.ablate & 16u) && h < 64) {
                } else if (vflags & 2u) {
                    // the output stage is virtual: its consumers recompute it from the channel row
                } else if (vflags & 4u) {
                    const uint32_t ish = s - 3, ipp = h >> 2, total = (c.P * CW) << ish;
                    const bool right = (vflags & 8u) != 0, dst_arena = (s - 1) >= c.s0;
#pragma unroll 2
                    for (uint32_t e0 = 0; e0 < total; e0 += 64) {
                        const uint32_t e = e0 + lane;
                        const bool in = e < total;
                        uint32_t ct;
                        const uint32_t q = item_group(in ? (e >> ish) : 0u, ct), i = (e & (ipp - 1)) << 2;
                        const uint32_t holder = c.field(c.slot_from_lane((int)(q << LPL)), logN, q);
                        if (in) {
                            const uint32_t* rl = c.bits(0, holder, logN - 1);
                            const float4 a = virt_top4<I8, CH>(chan_of(ct), N >> 1, i, right, rl), b = virt_top4<I8, CH>(chan_of(ct), N >> 1, h + i, right, rl);
                            const float4 r = make_float4(polar_fx<CH>(a.x, b.x), polar_fx<CH>(a.y, b.y), polar_fx<CH>(a.z, b.z),
                                                         polar_fx<CH>(a.w, b.w));
                            if (dst_arena)
                                Wr<K_GF32>{ c.arena_stage(q, s - 1) }.st4(i, r);
                            else
                                Wr<K_LDS>{ c.path_llr(q) + h }.st4(i, r);
                        }
                    }
                } else if (h >= 64) {
                    // 16-byte items: item e = (path q, 4 consecutive LLRs); several loads in flight per lane
                    const uint32_t ish = s - 3, ipp = h >> 2, total = (c.P * CW) << ish;
                    c.template with_src_kind<I8>(s, [&](auto KS) {
                        constexpr int SK = decltype(KS)::value;
                        const bool dst_arena = (s - 1) >= c.s0;
#pragma unroll 2
                        for (uint32_t e = lane; e < total; e += 64) {
                            uint32_t ct;
                            const uint32_t q = item_group(e >> ish, ct), i = (e & (ipp - 1)) << 2;
                            const Rd<SK> x{ (SK == K_GI8 || s == logN) ? chan_of(ct) : c.template src_base<SK>(s, q) };
                            const float4 a = x.ld4(i), b = x.ld4(h + i);
                            const float4 r = make_float4(polar_fx<CH>(a.x, b.x), polar_fx<CH>(a.y, b.y), polar_fx<CH>(a.z, b.z),
                                                         polar_fx<CH>(a.w, b.w));
                            if (dst_arena)
                                Wr<K_GF32>{ c.arena_stage(q, s - 1) }.st4(i, r);
                            else
                                Wr<K_LDS>{ c.path_llr(q) + h }.st4(i, r);
                        }
                    });
                } else if (active) {  // (stages in LDS normally run as SCL_F_S)
                    c.template with_src<I8>(s, c.p, [&](auto x) {
                        c.with_dst(s - 1, c.p, [&](auto d) {
                            for (uint32_t i = c.j; i < h; i += LP) d.set(i, polar_fx<CH>(x[i], x[h + i]));
                        });
                    });
                }
                c.set_field(s);
            } else if (ins.op == SCL_G) {  // :242-249 -- LLR[s] through the slot word, left bits at identity
                const uint32_t wo = woff(s - 1);
                if (DIAG && (P.ablate & 8u) && h >= 64) {
                } else if (DIAG && (P.ablate & 16u) && h < 64) {
                } else if (vflags & 2u) {
                    // virtual output stage: only the slot-word update below remains
                } else if (vflags & 4u) {
                    const uint32_t ish = s - 3, ipp = h >> 2, total = (c.P * CW) << ish;
                    const bool right = (vflags & 8u) != 0, dst_arena = (s - 1) >= c.s0;
#pragma unroll 2
                    for (uint32_t e0 = 0; e0 < total; e0 += 64) {
                        const uint32_t e = e0 + lane;
                        const bool in = e < total;
                        uint32_t ct;
                        const uint32_t q = item_group(in ? (e >> ish) : 0u, ct), i = (e & (ipp - 1)) << 2;
                        const uint32_t holder = c.field(c.slot_from_lane((int)(q << LPL)), logN, q);
                        if (in) {
                            const uint32_t* rl = c.bits(0, holder, logN - 1);
                            const float4 a = virt_top4<I8, CH>(chan_of(ct), N >> 1, i, right, rl), b = virt_top4<I8, CH>(chan_of(ct), N >> 1, h + i, right, rl);
                            const uint32_t u = (c.bits(0, q, s - 1)[i >> 5] >> (i & 31u)) & 0xFu;
                            const float4 r = make_float4(polar_gx<CH>(a.x, b.x, u & 1u), polar_gx<CH>(a.y, b.y, (u >> 1) & 1u),
                                                         polar_gx<CH>(a.z, b.z, (u >> 2) & 1u), polar_gx<CH>(a.w, b.w, u >> 3));
                            if (dst_arena)
                                Wr<K_GF32>{ c.arena_stage(q, s - 1) }.st4(i, r);
                            else
                                Wr<K_LDS>{ c.path_llr(q) + h }.st4(i, r);
                        }
                    }
                } else if (h >= 64) {
                    const uint32_t ish = s - 3, ipp = h >> 2, total = (c.P * CW) << ish;
                    c.template with_src_kind<I8>(s, [&](auto KS) {
                        constexpr int SK = decltype(KS)::value;
                        const bool dst_arena = (s - 1) >= c.s0;
#pragma unroll 2
                        for (uint32_t e0 = 0; e0 < total; e0 += 64) {
                            const uint32_t e = e0 + lane;
                            const bool in = e < total;
                            uint32_t ct;
                        const uint32_t q = item_group(in ? (e >> ish) : 0u, ct), i = (e & (ipp - 1)) << 2;
                            // LLR[s] of path q lives in the buffer its slot word names (all lanes take part in the move)
                            const uint32_t srcq = c.field(c.slot_from_lane((int)(q << LPL)), s, q);
                            if (in) {
                                const Rd<SK> x{ (SK == K_GI8 || s == logN) ? chan_of(ct) : c.template src_base<SK>(s, srcq) };
                                const float4 a = x.ld4(i), b = x.ld4(h + i);
                                const uint32_t u = (c.bits(0, q, s - 1)[i >> 5] >> (i & 31u)) & 0xFu;
                                const float4 r = make_float4(polar_gx<CH>(a.x, b.x, u & 1u), polar_gx<CH>(a.y, b.y, (u >> 1) & 1u),
                                                             polar_gx<CH>(a.z, b.z, (u >> 2) & 1u), polar_gx<CH>(a.w, b.w, u >> 3));
                                if (dst_arena)
                                    Wr<K_GF32>{ c.arena_stage(q, s - 1) }.st4(i, r);
                                else
                                    Wr<K_LDS>{ c.path_llr(q) + h }.st4(i, r);
                            }
                        }
                    });
                } else if (active) {  // (stages in LDS normally run as SCL_G_S)
                    const uint32_t srcq = c.field(c.slot, s, (uint32_t)c.p);
                    const uint32_t* lb = c.area(0, c.p) + wo;
                    c.template with_src<I8>(s, srcq, [&](auto x) {
                        c.with_dst(s - 1, c.p, [&](auto d) {
                            for (uint32_t i = c.j; i < h; i += LP)
                                d.set(i, polar_gx<CH>(x[i], x[h + i], (lb[i >> 5] >> (i & 31u)) & 1u));
                        });
                    });
                }
                c.set_field(s);  // from now on the field names the holder of the LEFT bits
            } else if (ins.op == SCL_COMBINE) {  // :253-260 CombineBitsLong / CombineBitsShort
                if (active && !(DIAG && (P.ablate & 16u))) {
                    const uint32_t lq = c.field(c.slot, s, (uint32_t)c.p);
                    const uint32_t* lb = c.bits(0, lq, s - 1);
                    const uint32_t* rb = c.bits(1, c.p, s - 1);
                    uint32_t* ob = c.bits(side, c.p, s);
                    if (h >= 32) {
                        const uint32_t hw = h >> 5;
                        for (uint32_t w = c.j; w < hw; w += LP) {
                            const uint32_t r = rb[w];
                            ob[w] = lb[w] ^ r;
                            ob[hw + w] = r;
                        }
                    } else if (c.j == 0) {
                        const uint32_t mask = (1u << h) - 1u;
                        const uint32_t r = rb[0] & mask;
                        ob[0] = ((lb[0] ^ r) & mask) | (r << h);
                    }
                }
            } else if (ins.op == SCL_R0) {  // RateZeroDecoder::decode :316-337
                float acc[NACC];
#pragma unroll
                for (int a = 0; a < NACC; ++a) acc[a] = 0.0f;
                if (kFuse && fuse) {  // n <= 8, input straight from the fused F / G
                    float x0, x1;
                    leaf_input4(x0, x1);
                    if (active) {
#pragma unroll
                        for (int a = 0; a < NACC; ++a) {
                            const uint32_t v = (uint32_t)c.j + (uint32_t)a * LP;
                            acc[a] = acc[a] + (v < n ? min0(a == 0 ? x0 : x1) : 0.0f);
                        }
                    }
                } else if (active && (LP < 8 || c.j < 8)) {
                    auto body = [&](auto x) {
#pragma unroll
                        for (int a = 0; a < NACC; ++a) {
                            const uint32_t v = (uint32_t)c.j + (uint32_t)a * LP;
                            if (n >= 32) {  // four loads in flight per accumulator (arena blocks: one global latency per 4 adds)
                                for (uint32_t i = v; i < n; i += 32) {
                                    const float x0 = x[i], x1 = x[i + 8], x2 = x[i + 16], x3 = x[i + 24];
                                    acc[a] = (((acc[a] + min0(x0)) + min0(x1)) + min0(x2)) + min0(x3);
                                }
                            } else {
                                for (uint32_t i = v; i < n; i += 8) acc[a] = acc[a] + min0(x[i]);
                            }
                            if (v >= n) acc[a] = acc[a] + 0.0f;
                        }
                    };
                    if (s < c.s0 && s < logN)
                        body(Rd<K_LDS>{ c.path_llr(c.p) + n });
                    else
                        c.template with_src<I8>(s, c.p, body);
                }
                const float sum = c.lanesum_finish(acc);
                if (active) {
                    c.metric = c.metric + sum;
                    if (kFuse && fuse) {
                        leaf_bits_out(c.j == 0, 0u);
                    } else {
                        uint32_t* ob = c.bits(side, c.p, s);
                        const uint32_t nw = n >= 32 ? (n >> 5) : 1u;
                        for (uint32_t w = c.j; w < nw; w += LP) ob[w] = 0u;
                    }
                }
            } else if (ins.op == SCL_REP) {  // RepetitionDecoder::decode :428-481 (n = 2 or 4)
                float az[NACC], ao[NACC];
#pragma unroll
                for (int a = 0; a < NACC; ++a) az[a] = ao[a] = 0.0f;
                if (kFuse && fuse) {  // n <= 4 (lanesum_low4 below), input straight from the fused F / G
                    float x0, x1;
                    leaf_input4(x0, x1);
                    if (active) {
                        const float xv = (uint32_t)c.j < n ? x0 : 0.0f;
                        az[0] = az[0] + min0(xv);
                        ao[0] = ao[0] + max0(xv);
#pragma unroll
                        for (int a = 1; a < NACC; ++a) {
                            az[a] = az[a] + min0(0.0f);
                            ao[a] = ao[a] + max0(0.0f);
                        }
                    }
                } else if (active && (LP < 8 || c.j < 8)) {
                    auto body = [&](auto x) {
#pragma unroll
                        for (int a = 0; a < NACC; ++a) {
                            const uint32_t v = (uint32_t)c.j + (uint32_t)a * LP;
                            if constexpr (CH) {  // SclFipChar: repetition leaves of every size (scl_fip_char.cpp:762-764), exact sums
                                for (uint32_t i = v; i < n; i += 8) {
                                    const float xv = x[i];
                                    az[a] = az[a] + min0(xv);
                                    ao[a] = ao[a] + max0(xv);
                                }
                            } else {
                                const float xv = v < n ? x[v < n ? v : 0] : 0.0f;
                                az[a] = az[a] + min0(xv);
                                ao[a] = ao[a] + max0(xv);
                            }
                        }
                    };
                    if (s < c.s0 && s < logN)
                        body(Rd<K_LDS>{ c.path_llr(c.p) + n });
                    else
                        c.template with_src<I8>(s, c.p, body);
                }
                // (float: n <= 4, and min0 / max0 never yield -0.0, so the four trailing additions of the reference's order
                //  are dropped; int8: sums of integers, any order)
                const float z = (CH && n > 4) ? c.lanesum_finish(az) : c.lanesum_low4(az);
                const float o = (CH && n > 4) ? c.lanesum_finish(ao) : c.lanesum_low4(ao);
                PAMD_STAMP(0)
                constexpr int CPL = (2 >> LPL) ? (2 >> LPL) : 1;
                float val[CPL];
#pragma unroll
                for (int r = 0; r < CPL; ++r) {
                    const uint32_t q = (uint32_t)c.j * CPL + r;
                    val[r] = (q == 0) ? c.metric + z : c.metric - o;
                }
                const uint32_t newP = (c.P * 2 < c.L) ? c.P * 2 : c.L;
                float nval = 0.0f;
                uint32_t nidx = 0;
                bool done = false;
                if constexpr (kMerge) {
                    if (!CH && c.P * 2 > newP && !(c.ablate & 1u)) {  // the better of the two in lane 0, the other in lane 1
                        const float ca = c.metric + z, cb = c.metric - o;
                        const bool sw = fbits(cb) < fbits(ca);
                        const bool first = c.j == 0;
                        const float e0 = (first != sw) ? ca : cb;
                        const uint32_t q0 = (first != sw) ? 0u : 1u;
                        done = select_merge<Ctx, 2>(c, (active && c.j < 2) ? fbits(e0) : 0xFFFFFFFFu, ((uint32_t)c.pl << 1) | q0,
                                                    0xFFFFFFFFu, 0u, newP, nval, nidx);
                    }
                    if (!done) select_literal_cw<Ctx, LPL, 2>(c, val, newP, nval, nidx);
                } else
                select_dispatch<Ctx, LPL, 2>(c, val, (c.metric + z > c.metric - o) ? c.metric + z : c.metric - o, newP, nval, nidx);
                PAMD_STAMP(1)
                const uint32_t srcp = (uint32_t)c.g0 + (nidx >> 1), q = nidx & 1u;
                const typename Ctx::SlotT nslot = c.slot_from_lane((int)(srcp << LPL));
                PAMD_STAMP(2)
                c.P = newP;
                if ((uint32_t)c.pl < c.P) {
                    c.metric = nval;
                    c.slot = nslot;
                    if (CH && !(kFuse && fuse)) {  // blocks of any size: n identical bits
                        uint32_t* ob = c.bits(side, c.p, s);
                        const uint32_t nw = n >= 32 ? (n >> 5) : 1u, pat = q ? (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u)) : 0u;
                        for (uint32_t w = c.j; w < nw; w += LP) ob[w] = pat;
                    } else {
                        leaf_bits_out(c.j == 0, q ? ((1u << n) - 1u) : 0u);
                    }
                }
            } else if (ins.op == SCL_SPC) {
                fork_leaf(std::true_type{}, std::false_type{});
            } else if (ins.op == SCL_R1) {
                fork_leaf(std::false_type{}, std::false_type{});
            } else if (ins.op == SCL_WFG) {
                const bool isg = (aux & 1u) != 0;
                const uint32_t hh = h >> 1, ish = s - 4, ipp = h >> 3, total = (c.P * CW) << ish;
                c.template with_src_kind<I8>(s, [&](auto KS) {
                    constexpr int SK = decltype(KS)::value;
#pragma unroll 2
                    for (uint32_t e0 = 0; e0 < total; e0 += 64) {
                        const uint32_t e = e0 + (uint32_t)lane;
                        const bool in = e < total;
                        const uint32_t ee = in ? e : 0u;  // (idle lanes load item 0 and store nothing)
                        uint32_t ct;
                        const uint32_t q = item_group(ee >> ish, ct), i = (ee & (ipp - 1)) << 2;
                        uint32_t srcq = q, u0 = 0, u1 = 0;
                        if (isg) {  // LLR[s] through the slot word (all lanes take part in the move), left bits at identity
                            srcq = c.field(c.slot_from_lane((int)(q << LPL)), s, q);
                            const uint32_t* lb = c.bits(0, q, s - 1);
                            u0 = (lb[i >> 5] >> (i & 31u)) & 0xFu;
                            u1 = (lb[(hh + i) >> 5] >> (i & 31u)) & 0xFu;
                        }
                        const Rd<SK> x{ (SK == K_GI8 || s == logN) ? chan_of(ct) : c.template src_base<SK>(s, srcq) };
                        const float4 a0 = x.ld4(i), b0 = x.ld4(h + i), a1 = x.ld4(hh + i), b1 = x.ld4(h + hh + i);
                        float4 r0, r1;
                        if (isg) {
                            r0 = make_float4(polar_gx<CH>(a0.x, b0.x, u0 & 1u), polar_gx<CH>(a0.y, b0.y, (u0 >> 1) & 1u),
                                             polar_gx<CH>(a0.z, b0.z, (u0 >> 2) & 1u), polar_gx<CH>(a0.w, b0.w, u0 >> 3));
                            r1 = make_float4(polar_gx<CH>(a1.x, b1.x, u1 & 1u), polar_gx<CH>(a1.y, b1.y, (u1 >> 1) & 1u),
                                             polar_gx<CH>(a1.z, b1.z, (u1 >> 2) & 1u), polar_gx<CH>(a1.w, b1.w, u1 >> 3));
                        } else {
                            r0 = make_float4(polar_fx<CH>(a0.x, b0.x), polar_fx<CH>(a0.y, b0.y), polar_fx<CH>(a0.z, b0.z), polar_fx<CH>(a0.w, b0.w));
                            r1 = make_float4(polar_fx<CH>(a1.x, b1.x), polar_fx<CH>(a1.y, b1.y), polar_fx<CH>(a1.z, b1.z), polar_fx<CH>(a1.w, b1.w));
                        }
                        const float4 r2 = make_float4(polar_fx<CH>(r0.x, r1.x), polar_fx<CH>(r0.y, r1.y), polar_fx<CH>(r0.z, r1.z),
                                                      polar_fx<CH>(r0.w, r1.w));
                        if (in) {
                            c.with_dst(s - 1, q, [&](auto d) {
                                d.st4(i, r0);
                                d.st4(hh + i, r1);
                            });
                            c.with_dst(s - 2, q, [&](auto d) { d.st4(i, r2); });
                        }
                    }
                });
                c.set_field(s);
                c.set_field(s - 1);
            } else if (ins.op == SCL_VG2) {
                if constexpr (kTop8 && !S64 && HB) {
                    // s = logN - 2 = 8: 128 outputs per path = 32 chunks: lane l serves codeword l / 32, chunk l % 32 and loops
                    // over the paths of its codeword.  Output chunk c needs the chunks c, c + 32 of the (never stored)
                    // stage-8 block, those need the chunks c, c + 32, c + 64, c + 96 of the stage-9 block, and those the
                    // channel chunks at the same places and 128 chunks further on: eight 16-byte channel loads per lane,
                    // once for all paths.  What does not depend on the path (f of the channel when the root child is
                    // the left one, f of that when the grandchild is a left child too) is computed once.
                    const bool r9 = (aux & 1u) != 0, r8 = (aux & 2u) != 0;
                    const uint32_t kf = (aux >> 2) & 3u;
                    const uint32_t t = (uint32_t)lane >> 5, c4 = 4u * ((uint32_t)lane & 31u);
                    const uint32_t wi = c4 >> 5, wsh = c4 & 31u;  // (all chunk distances are multiples of 32 elements)
                    const uint32_t g0q = t * (uint32_t)LCW;
                    auto f4 = [](float4 a, float4 b) {
                        return make_float4(polar_fx<CH>(a.x, b.x), polar_fx<CH>(a.y, b.y), polar_fx<CH>(a.z, b.z), polar_fx<CH>(a.w, b.w));
                    };
                    auto g4 = [](float4 a, float4 b, uint32_t u) {
                        return make_float4(polar_gx<CH>(a.x, b.x, u & 1u), polar_gx<CH>(a.y, b.y, (u >> 1) & 1u),
                                           polar_gx<CH>(a.z, b.z, (u >> 2) & 1u), polar_gx<CH>(a.w, b.w, (u >> 3) & 1u));
                    };
                    const Rd<I8 ? K_GI8 : K_GF32> x{ chan_of(t) };
                    const uint32_t halfN = N >> 1;
                    float4 v0 = x.ld4(c4), v1 = x.ld4(c4 + 128u), v2 = x.ld4(c4 + 256u), v3 = x.ld4(c4 + 384u);
                    float4 w0 = x.ld4(halfN + c4), w1 = x.ld4(halfN + c4 + 128u), w2 = x.ld4(halfN + c4 + 256u), w3 = x.ld4(halfN + c4 + 384u);
                    if (!r9) {  // stage-9 chunks, the same for every path
                        v0 = f4(v0, w0);
                        v1 = f4(v1, w1);
                        v2 = f4(v2, w2);
                        v3 = f4(v3, w3);
                        if (!r8) {  // ... and so are the stage-8 chunks c (from v0, v2) and c + 32 (from v1, v3)
                            v0 = f4(v0, v2);
                            v1 = f4(v1, v3);
                        }
                    }
                    const uint32_t wo_root = woff(logN - 1) - c.wsb + wi, wo_9 = woff(logN - 2) - c.wsb + wi, wo_8 = woff(s - 1) - c.wsb + wi;
                    // the seven 4-bit groups that steer the g's of one path, packed: bits 0-15 the root's left-child bits at
                    // the four stage-9 chunks, 16-23 the stage-9 node's left-child bits at the two stage-8 chunks, 24-27 the
                    // left-child bits of this node.  Fetched one path ahead (plain global loads, counted waits).
                    auto fetch = [&](uint32_t ql) -> uint32_t {
                        const uint32_t q = g0q + ql;  // (per lane: the path group of this lane's codeword)
                        const uint32_t sq = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(q << (LPL + 2)), (int)(uint32_t)c.slot);
                        uint32_t nb = (((c.gbits + __umul24(q, c.gwpp) + wo_8)[0] >> wsh) & 0xFu) << 24;  // left-child bits at identity
                        if (r9) {
                            const uint32_t holder = g0q + ((sq >> ((logN - 1u) * (uint32_t)Ctx::FB)) & (uint32_t)(LCW - 1));
                            const uint32_t* rl = c.gbits + __umul24(holder, c.gwpp) + wo_root;
                            nb |= ((rl[0] >> wsh) & 0xFu) | (((rl[4] >> wsh) & 0xFu) << 4) | (((rl[8] >> wsh) & 0xFu) << 8) |
                                  (((rl[12] >> wsh) & 0xFu) << 12);
                        }
                        if (r8) {
                            const uint32_t h9 = g0q + ((sq >> ((logN - 2u) * (uint32_t)Ctx::FB)) & (uint32_t)(LCW - 1));
                            const uint32_t* b9 = c.gbits + __umul24(h9, c.gwpp) + wo_9;
                            nb |= (((b9[0] >> wsh) & 0xFu) << 16) | (((b9[4] >> wsh) & 0xFu) << 20);
                        }
                        return nb;
                    };
                    uint32_t nxt = fetch(0);
                    for (uint32_t ql = 0; ql < c.P; ++ql) {
                        const uint32_t q = g0q + ql;
                        const uint32_t nb = nxt;
                        if (ql + 1u < c.P) nxt = fetch(ql + 1u);
                        float4 a, b;  // stage-8 chunks c and c + 32 of path q
                        if (r9) {
                            const float4 s0_ = g4(v0, w0, nb & 0xFu), s1_ = g4(v1, w1, (nb >> 4) & 0xFu),
                                         s2_ = g4(v2, w2, (nb >> 8) & 0xFu), s3_ = g4(v3, w3, (nb >> 12) & 0xFu);
                            a = r8 ? g4(s0_, s2_, (nb >> 16) & 0xFu) : f4(s0_, s2_);
                            b = r8 ? g4(s1_, s3_, (nb >> 20) & 0xFu) : f4(s1_, s3_);
                        } else if (r8) {
                            a = g4(v0, v2, (nb >> 16) & 0xFu);
                            b = g4(v1, v3, (nb >> 20) & 0xFu);
                        } else {
                            a = v0;
                            b = v1;
                        }
                        const uint32_t u8 = (nb >> 24) & 0xFu;
                        const float4 r = g4(a, b, u8);
                        Wr<K_GF32>{ c.arena_stage(q, s - 1) }.st4(c4, r);
                        if (kf >= 1u) {  // F of stage 7: chunk c with c + 16 (the other DPP row of the codeword)
                            const auto tx = __builtin_amdgcn_permlane16_swap(fbits(r.x), fbits(r.x), false, false);
                            const auto ty = __builtin_amdgcn_permlane16_swap(fbits(r.y), fbits(r.y), false, false);
                            const auto tz = __builtin_amdgcn_permlane16_swap(fbits(r.z), fbits(r.z), false, false);
                            const auto tw = __builtin_amdgcn_permlane16_swap(fbits(r.w), fbits(r.w), false, false);
                            const float4 r2 = f4(r, make_float4(bitsf(tx[1]), bitsf(ty[1]), bitsf(tz[1]), bitsf(tw[1])));
                            if (((uint32_t)lane & 31u) < 16u) Wr<K_LDS>{ c.path_llr(q) + (1u << (s - 2)) }.st4(c4, r2);
                        }
                    }
                    c.set_field(s);
                    if (kf >= 1u) c.set_field(s - 1);
                }
            } else if (ins.op == SCL_VTOP) {
                if constexpr (kTop8 && !S64 && HB) {
                    // Stage s = logN - 1 block of every live path from the channel row: h = 256 outputs per path = one
                    // 16-byte chunk per lane, so the four channel chunks a lane needs are loaded ONCE per codeword and
                    // serve all its paths (they differ only in the bits that steer the g's).  The F of the next stages
                    // pairs chunk l with l + 32 (the other half of the wave: v_permlane32_swap), then l with l + 16
                    // (the other DPP row: v_permlane16_swap): the left descent continues in registers, every level
                    // is still stored for the G that comes back to it later.  The bit words of path ql + 1 are fetched
                    // before path ql is computed and stored (plain global loads -- the host only emits the op when
                    // these bit blocks live in the arena -- so that the waits can be counted: a flat load would have to
                    // wait for the stores issued behind it as well).
                    const bool right = (vflags & 8u) != 0, isg = (aux & 1u) != 0;
                    const uint32_t kf = (aux >> 1) & 3u;
                    const uint32_t ci = 4u * (uint32_t)lane;
                    const uint32_t wi = ci >> 5, wsh = ci & 31u;  // (h = 256 is a multiple of 32: chunk ci + h has the same shift)
                    const uint32_t wo_root = woff(logN - 1) - c.wsb + wi, wo_u = woff(s - 1) - c.wsb + wi;
                    auto f4 = [](float4 a, float4 b) {
                        return make_float4(polar_fx<CH>(a.x, b.x), polar_fx<CH>(a.y, b.y), polar_fx<CH>(a.z, b.z), polar_fx<CH>(a.w, b.w));
                    };
                    auto g4 = [](float4 a, float4 b, uint32_t u) {
                        return make_float4(polar_gx<CH>(a.x, b.x, u & 1u), polar_gx<CH>(a.y, b.y, (u >> 1) & 1u),
                                           polar_gx<CH>(a.z, b.z, (u >> 2) & 1u), polar_gx<CH>(a.w, b.w, (u >> 3) & 1u));
                    };
                    for (uint32_t t = 0; t < (uint32_t)CW; ++t) {
                        const void* chp = chanv[0];
#pragma unroll
                        for (int k = 1; k < CW; ++k) chp = (t == (uint32_t)k) ? chanv[k] : chp;
                        const Rd<I8 ? K_GI8 : K_GF32> x{ chp };
                        const float4 c0 = x.ld4(ci), c1 = x.ld4(ci + h), c2 = x.ld4(ci + 2u * h), c3 = x.ld4(ci + 3u * h);
                        auto fetch = [&](uint32_t ql, uint32_t& wa, uint32_t& wb, uint32_t& wu) {
                            const uint32_t q = t * (uint32_t)LCW + ql;
                            wa = wb = wu = 0;
                            if (right) {  // the root's left-child bits of path q sit in the buffer its slot word names
                                const uint32_t sq = readlane_u32((uint32_t)c.slot, (int)(q << LPL));
                                const uint32_t holder = t * (uint32_t)LCW + ((sq >> ((logN - 1u) * (uint32_t)Ctx::FB)) & (uint32_t)(LCW - 1));
                                const uint32_t* rl = c.gbits + __umul24(holder, c.gwpp) + wo_root;
                                wa = rl[0];
                                wb = rl[h >> 5];
                            }
                            if (isg) wu = (c.gbits + __umul24(q, c.gwpp) + wo_u)[0];
                        };
                        uint32_t na, nb, nu;
                        fetch(0, na, nb, nu);
                        for (uint32_t ql = 0; ql < c.P; ++ql) {
                            const uint32_t q = t * (uint32_t)LCW + ql;
                            const uint32_t ua = (na >> wsh) & 0xFu, ub = (nb >> wsh) & 0xFu, u = (nu >> wsh) & 0xFu;
                            if (ql + 1u < c.P) fetch(ql + 1u, na, nb, nu);
                            const float4 a = right ? g4(c0, c2, ua) : f4(c0, c2), b = right ? g4(c1, c3, ub) : f4(c1, c3);
                            const float4 r = isg ? g4(a, b, u) : f4(a, b);
                            if (!(aux & 32u)) Wr<K_GF32>{ c.arena_stage(q, s - 1) }.st4(ci, r);  // (bit 5: nobody reads it back, SCL_VG2)
                            if (kf >= 1u) {
                                // lanes 0..31: chunk l of stage s - 2 = f(chunk l, chunk l + 32)
                                const auto sx = __builtin_amdgcn_permlane32_swap(fbits(r.x), fbits(r.x), false, false);
                                const auto sy = __builtin_amdgcn_permlane32_swap(fbits(r.y), fbits(r.y), false, false);
                                const auto sz = __builtin_amdgcn_permlane32_swap(fbits(r.z), fbits(r.z), false, false);
                                const auto sw = __builtin_amdgcn_permlane32_swap(fbits(r.w), fbits(r.w), false, false);
                                const float4 r2 = f4(r, make_float4(bitsf(sx[1]), bitsf(sy[1]), bitsf(sz[1]), bitsf(sw[1])));
                                if (lane < 32) Wr<K_GF32>{ c.arena_stage(q, s - 2) }.st4(ci, r2);
                                if (kf >= 2u) {
                                    // lanes 0..15: chunk l of stage s - 3 = f(chunk l, chunk l + 16)
                                    const auto tx = __builtin_amdgcn_permlane16_swap(fbits(r2.x), fbits(r2.x), false, false);
                                    const auto ty = __builtin_amdgcn_permlane16_swap(fbits(r2.y), fbits(r2.y), false, false);
                                    const auto tz = __builtin_amdgcn_permlane16_swap(fbits(r2.z), fbits(r2.z), false, false);
                                    const auto tw = __builtin_amdgcn_permlane16_swap(fbits(r2.w), fbits(r2.w), false, false);
                                    const float4 r3 = f4(r2, make_float4(bitsf(tx[1]), bitsf(ty[1]), bitsf(tz[1]), bitsf(tw[1])));
                                    if (lane < 16) Wr<K_LDS>{ c.path_llr(q) + (1u << (s - 3)) }.st4(ci, r3);
                                }
                            }
                        }
                    }
                    c.set_field(s);
                    if (kf >= 1u) c.set_field(s - 1);
                    if (kf >= 2u) c.set_field(s - 2);
                }
            }
        op_done:
            if ((ins.op <= SCL_G || ins.op == SCL_VTOP || ins.op == SCL_WFG || ins.op == SCL_VG2) && s - 1 >= c.s0 && !(vflags & 2u)) {
                // this op stored LLRs to the global arena: wait for the stores (L2 is the meeting point; the
                // arena is read back with sc1 loads)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            WAVE_SYNC();
            if (DIAG && P.cycles && cw == 0) {
                const uint64_t t_end = __builtin_amdgcn_s_memtime();
                if (lane == 0) P.cycles[pc] = (uint32_t)(t_end - t_start);
            }
            if (DIAG && P.trace && cw == 0 && is_leaf) {
                if (trace_n < P.trace_cap) {
                    float* t = P.trace + (size_t)trace_n * (1 + P.L);
                    if (lane == 0) t[0] = (float)c.P;
                    if (mycw == 0 && c.j == 0 && (uint32_t)c.pl < P.L) t[1 + c.pl] = ((uint32_t)c.pl < c.P) ? c.metric : 0.0f;
                }
                ++trace_n;
            }
        }

        if (DIAG && (P.ablate & 32u)) continue;
        // ---- SclAvxFloat::extractBestPath :711-750 ---------------------------------------------
        const bool active = (uint32_t)c.pl < c.P;
        uint32_t* cwbits = c.bits(0, c.p, logN);
        const uint32_t nwords = N >= 32 ? (N >> 5) : 1u;
        if (!P.systematic) {  // re-encode every candidate (:731-746)
            if (active) {
                for (uint32_t w = c.j; w < nwords; w += LP) {
                    uint32_t v = cwbits[w];
                    if (N >= 2) v ^= (v >> 1) & 0x55555555u;
                    if (N >= 4) v ^= (v >> 2) & 0x33333333u;
                    if (N >= 8) v ^= (v >> 4) & 0x0F0F0F0Fu;
                    if (N >= 16) v ^= (v >> 8) & 0x00FF00FFu;
                    if (N >= 32) v ^= (v >> 16) & 0x0000FFFFu;
                    cwbits[w] = v;
                }
            }
            WAVE_SYNC();
            for (uint32_t hw = 1; hw < nwords; hw <<= 1) {
                if (active)
                    for (uint32_t w = c.j; w < nwords; w += LP)
                        if (!(w & hw)) cwbits[w] ^= cwbits[w + hw];
                WAVE_SYNC();
            }
        }
        // detector verdict of every path in parallel (syndrome XOR), first passing path of each codeword wins
        constexpr int LCWL = 64 >> CWL;                 // lanes per codeword
        const uint32_t lane0 = (uint32_t)c.g0 << LPL;   // first lane of this lane's codeword
        uint32_t winner = (uint32_t)c.g0, okflag = 1;   // (group of the winning path; per codeword)
        bool best_passes = false;
        if (P.det_on) {
            // the best path passes for almost every codeword: check it alone with all lanes of the codeword first
            const uint32_t* b0 = c.bits(0, (uint32_t)c.g0, logN);
            uint32_t syn0 = 0;
#pragma unroll 4
            for (uint32_t t = (uint32_t)lane - lane0; t < P.K; t += LCWL) {  // (branch-free: the table loads of several trips overlap)
                const uint32_t pos = P.lut[t], sy = P.syn[t];
                syn0 ^= ((b0[pos >> 5] >> (pos & 31u)) & 1u) ? sy : 0u;
            }
            syn0 = group_xor_u32<LCWL>(syn0);
            best_passes = ballot64(syn0 != 0) == 0;  // ... of every codeword of the wave
        }
        if (P.det_on && !best_passes) {
            uint32_t syn = 0;
            if (active) {
                for (uint32_t t = c.j; t < P.K; t += LP) {
                    const uint32_t pos = P.lut[t];
                    if ((cwbits[pos >> 5] >> (pos & 31u)) & 1u) syn ^= P.syn[t];
                }
            }
#pragma unroll
            for (int m = 1; m < LP; m <<= 1) syn ^= shflx_u32(syn, m);
            const uint64_t pass = ballot64(active && c.j == 0 && syn == 0) >> lane0;
            const uint64_t mine = CWL == 0 ? pass : (pass & ((1ull << (LCWL & 63)) - 1ull));
            if (mine) {
                winner = (uint32_t)c.g0 + ((uint32_t)(__ffsll((long long)mine) - 1) >> LPL);
            } else {
                winner = (uint32_t)c.g0;
                okflag = 0;
            }
        }
#pragma unroll
        for (int t = 0; t < CW; ++t) {  // the information bits of each codeword's winner, 64 at a time with the whole wave
            const uint32_t ci = cw * CW + t;
            const bool live_t = ci < nrows;
            const uint32_t row_t = P.rowidx ? P.rowidx[live_t ? ci : cw * CW] : ci;
            const uint32_t wg = CWL == 0 ? winner : readlane_u32(winner, t * LCWL);
            const uint32_t* wb = c.bits(0, wg, logN);
            uint8_t* orow = P.out + (size_t)row_t * P.nb;
#pragma unroll 2
            for (uint32_t t0 = 0; t0 < P.K; t0 += 64) {
                const uint32_t tt = t0 + lane;
                uint32_t bit = 0;
                if (tt < P.K) {
                    const uint32_t pos = P.lut[tt];
                    bit = (wb[pos >> 5] >> (pos & 31u)) & 1u;
                }
                const uint64_t m = ballot64(bit != 0);
                if (lane < 8 && live_t) {
                    const uint32_t byte = (t0 >> 3) + lane;
                    if (byte < P.nb) orow[byte] = (uint8_t)(__brev((uint32_t)((m >> (8 * lane)) & 0xFFu)) >> 24);
                }
            }
        }
        if ((uint32_t)lane == lane0 && live && P.ok) P.ok[row] = (uint8_t)okflag;
        if (P.metrics && live) {
            if (c.j == 0 && (uint32_t)c.pl < P.L) P.metrics[(size_t)row * P.L + c.pl] = active ? c.metric : 0.0f;
        }
        WAVE_SYNC();
    }
}

// =============================================================================================
//  launchers
// =============================================================================================
// occ != nullptr: do not launch, report the resident workgroups per CU for this block size / LDS size in *occ
template <int LPL, int IN, bool S64, bool DIAG, bool HB, int CWL>
static hipError_t launch_scl_k(const DecodeParams& P, dim3 grid, dim3 block, size_t lds, hipStream_t st, int* occ)
{
    auto kern = scl_decode_kernel<LPL, IN, S64, DIAG, HB, CWL>;
    if (occ) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, kern, (int)block.x, lds);
        if (e != hipSuccess) return e;
        // the API over-reports for single-wave workgroups: clamp by the register file (512 VGPRs per SIMD lane,
        // allocation granule 8, at most 8 waves per SIMD; MI355X_MICROARCH.md "Register files")
        hipFuncAttributes fa;
        e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
        if (e != hipSuccess) return e;
        const int regs = ((fa.numRegs > 0 ? fa.numRegs : 1) + 7) & ~7;
        int waves_per_simd = 512 / regs;
        if (waves_per_simd > 8) waves_per_simd = 8;
        if (waves_per_simd < 1) waves_per_simd = 1;
        const int by_regs = (4 * waves_per_simd) / (int)(block.x / 64);
        if (*occ > by_regs) *occ = by_regs;
        return hipSuccess;
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, block, lds, st, P);
    return hipGetLastError();
}

// Instantiations per (lanes per path, codewords per wave): production kernels with a 64-bit slot word, with and
// without the high-stage bit blocks in the arena; a 32-bit slot word where log2(N) * (bits per slot field) <= 32
// (the headline configuration among them); and a diagnostic variant (trace / cycle stamps / ablation switches)
// for float input only.
template <int LPL, int CWL>
static hipError_t launch_scl_t(const DecodeParams& P, bool i8, bool diag, dim3 grid, dim3 block, size_t lds,
                               hipStream_t st, int* occ)
{
    constexpr unsigned fb = 6u - LPL - CWL;    // bits per slot field
    const bool slot64 = P.logN * fb > 32u;
    const bool hb = P.sb <= P.logN;            // high-stage bit blocks in the arena
#ifdef PAMD_DIAG
    if (diag && !i8) {  // (diagnostic variant: the slot word of the production kernel where it exists)
        constexpr bool DS64 = fb > 3;
        if (slot64 && !DS64) return hipErrorInvalidValue;
        return launch_scl_k<LPL, 0, DS64, true, true, CWL>(P, grid, block, lds, st, occ);
    }
#else
    if (diag) return hipErrorInvalidValue;  // (trace / cycle stamps / ablation: diagnostic build only)
#endif
    // input kind (template parameter IN of the kernel): 0 fp32 rows, 1 int8 rows widened, 2 int8 rows + int8 arithmetic
    const int in = P.chr ? 2 : (i8 ? 1 : 0);
    if (P.chr && !i8) return hipErrorInvalidValue;  // (the int8 decoders read int8 rows; float rows are converted first)
#define PAMD_SCL_LAUNCH(S64_, HB_)                                                                       \
    return in == 2 ? launch_scl_k<LPL, 2, S64_, false, HB_, CWL>(P, grid, block, lds, st, occ)           \
         : in == 1 ? launch_scl_k<LPL, 1, S64_, false, HB_, CWL>(P, grid, block, lds, st, occ)           \
                   : launch_scl_k<LPL, 0, S64_, false, HB_, CWL>(P, grid, block, lds, st, occ)
    if constexpr (fb <= 3) {
        if (!slot64 && !hb) PAMD_SCL_LAUNCH(false, false);
        if (!slot64) PAMD_SCL_LAUNCH(false, true);
    }
    if constexpr (fb <= 2) {  // log2(N) * fb <= 32 for every supported N: no 64-bit variant needed
        return hipErrorInvalidValue;
    } else {
        if (hb) PAMD_SCL_LAUNCH(true, true);
        PAMD_SCL_LAUNCH(true, false);
    }
#undef PAMD_SCL_LAUNCH
}

}  // namespace pamd
