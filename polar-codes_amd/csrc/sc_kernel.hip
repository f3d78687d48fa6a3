// Fast-SSC kernel, the "mixed" helper kernel, and the launch dispatch of both decoder families.
#include <algorithm>

#include "device_common.h"
#include "scl_launch.h"

namespace pamd {

// =============================================================================================
//  Fast-SSC
// =============================================================================================
// LC = 2^LCL lanes work on one codeword, so a wave decodes 64/LC codewords side by side (LC = 16 for
// N <= 1024: the tree ops below stage 6 have at most 16 elements, and every wave-uniform cost -- program
// fetch, dispatch, address arithmetic -- is shared by the four codewords).  Control flow is identical for
// all codewords, so nothing diverges.
// LDS per codeword: llr[N] floats (stage s < logN at offset 1<<s; the channel row is read from global
// memory by the root ops) + max(1, N/32) words of packed partial sums (code bit i = word i>>5, bit i&31).
template <int LCL>
struct ScLanes {
    static constexpr int LC = 1 << LCL;
    static constexpr uint64_t MASK = LC == 64 ? ~0ull : ((1ull << LC) - 1ull);
    int l;       // lane inside the codeword group
    int lbase;   // first wave lane of the group
    __device__ __forceinline__ uint64_t slice(uint64_t ballot) const { return (ballot >> lbase) & MASK; }
    // store `cnt` (<= LC) bits of m at code-bit offset off (off % cnt == 0 when cnt < 32)
    __device__ __forceinline__ void put_bits(uint32_t* bits, uint32_t off, uint32_t cnt, uint64_t m) const
    {
        if (cnt >= 64) {
            if (l < 2) bits[(off >> 5) + l] = (uint32_t)(m >> (32 * l));
        } else if (cnt == 32) {
            if (l == 0) bits[off >> 5] = (uint32_t)m;
        } else if (l == 0) {
            const uint32_t sh = off & 31u, mask = ((1u << cnt) - 1u) << sh;
            const uint32_t w = bits[off >> 5];
            bits[off >> 5] = (w & ~mask) | (((uint32_t)m << sh) & mask);
        }
    }
    __device__ __forceinline__ void flip_bit(uint32_t* bits, uint32_t pos) const
    {
        if (l == 0) bits[pos >> 5] ^= 1u << (pos & 31u);
    }
    // fill n bits at off with a repeating 32-bit pattern (pattern period divides 32 and n)
    __device__ __forceinline__ void fill_pattern(uint32_t* bits, uint32_t off, uint32_t n, uint32_t pat) const
    {
        if (n >= 32) {
            for (uint32_t w = l; w < (n >> 5); w += LC) bits[(off >> 5) + w] = pat;
        } else {
            put_bits(bits, off, n, pat & ((1u << n) - 1u));
        }
    }
    // the reference's fold8 (8 strided AVX accumulators, acc = 0; acc += x[8k+v]).  The eight sums arrive in LANE 0 of
    // the group (DPP row shifts; a group never straddles a 16-lane row) -- the other lanes hold don't-care values.
    template <typename X>
    __device__ __forceinline__ void fold8(const X& x, uint32_t n, float a[8]) const
    {
        float acc = 0.0f;
        if (l < 8)
            for (uint32_t i = l; i < n; i += 8) acc = acc + x(i);
        const uint32_t u = fbits(acc);
        a[0] = acc;
        a[1] = bitsf(dpp_u32<0x101>(u));
        a[2] = bitsf(dpp_u32<0x102>(u));
        a[3] = bitsf(dpp_u32<0x103>(u));
        a[4] = bitsf(dpp_u32<0x104>(u));
        a[5] = bitsf(dpp_u32<0x105>(u));
        a[6] = bitsf(dpp_u32<0x106>(u));
        a[7] = bitsf(dpp_u32<0x107>(u));
    }
    // value of lane l ^ 4 (same parity class; row_half_mirror would cross the classes).  Both moves run in all lanes
    // BEFORE the select: a DPP read from a lane that is masked out returns 0, so they must not sit in the arms of a branch.
    __device__ __forceinline__ uint32_t xor4_u32(uint32_t v) const
    {
        const uint32_t dn = dpp_u32<0x114>(v), up = dpp_u32<0x104>(v);  // row_shr:4 (lane l - 4), row_shl:4 (lane l + 4)
        return (l & 4) ? dn : up;
    }
    // the value lane 0 of the group holds, in every lane of the group (LC <= 16: quad broadcast + row shifts)
    __device__ __forceinline__ uint32_t from_lane0(uint32_t v) const
    {
        if constexpr (LC > 16) {
            return shfl_u32(v, lbase);
        } else {
            const uint32_t q0 = dpp_u32<0x00>(v);  // lane 0 of every quad
            if constexpr (LC <= 4) return q0;
            const uint32_t q1 = dpp_u32<0x114>(q0);  // row_shr:4
            uint32_t r = (l & 4) ? q1 : q0;
            if constexpr (LC > 8) {
                const uint32_t q2 = dpp_u32<0x118>(q0), q3 = dpp_u32<0x11C>(q0);  // row_shr:8, :12
                r = (l & 8) ? ((l & 4) ? q3 : q2) : r;
            }
            return r;
        }
    }
    // sign of the reference's lanesum (fold8 + x0+..+x7 in that order), uniform in the group: lane 0 of the group pulls
    // the accumulators of lanes 1..7 with DPP row shifts (a group never straddles a 16-lane row), one ballot spreads it
    template <typename X>
    __device__ __forceinline__ bool lanesum_negative(const X& x, uint32_t n) const
    {
        float acc = 0.0f;
        if (l < 8)
            for (uint32_t i = l; i < n; i += 8) acc = acc + x(i);
        const uint32_t a = fbits(acc);
        float s = acc;
        s = s + bitsf(dpp_u32<0x101>(a));
        s = s + bitsf(dpp_u32<0x102>(a));
        s = s + bitsf(dpp_u32<0x103>(a));
        s = s + bitsf(dpp_u32<0x104>(a));
        s = s + bitsf(dpp_u32<0x105>(a));
        s = s + bitsf(dpp_u32<0x106>(a));
        s = s + bitsf(dpp_u32<0x107>(a));
        return (slice(ballot64(l == 0 && sgn(s))) & 1ull) != 0;
    }
    // ---- int8 decoder (FastSscFipChar) reductions: saturating adds in the reference's order ------------------------
    // value of lane l + d of the group (l + d < LC), by DPP row shift inside a 16-lane group, ds_bpermute otherwise
    __device__ __forceinline__ uint32_t pull_u32(uint32_t v, int d) const
    {
        if constexpr (LC == 16) {
            switch (d) {
            case 1: return dpp_u32<0x101>(v);
            case 2: return dpp_u32<0x102>(v);
            case 4: return dpp_u32<0x104>(v);
            default: return dpp_u32<0x128>(v);  // 8: rotation of the row by 8
            }
        } else {
            return shfl_u32(v, lbase + ((l + d) & (LC - 1)));
        }
    }
    __device__ __forceinline__ static float sat8f(float v) { return __builtin_amdgcn_fmed3f(v, -128.0f, 127.0f); }
    // byte lane v of the 32 accumulators "LlrSum = adds(LlrSum, load(LlrIn + i))": elements v, v + 32, ... one after the
    // other, saturating; a block of at most 32 elements is one zero-padded vector
    template <typename X>
    __device__ __forceinline__ float char_lane_sum(const X& x, uint32_t n, uint32_t v) const
    {
        float acc = 0.0f;
        for (uint32_t i = v; i < n; i += 32) acc = sat8f(acc + x(i));
        return acc;
    }
    // first step of reduce_adds_epi8 / half_reduce_adds_epi8: lane v + lane v + 16, v < 16 (lanes 0..15 of the group)
    template <typename X>
    __device__ __forceinline__ float char_lane_pairs(const X& x, uint32_t n) const
    {
        if constexpr (LC <= 16) {
            return sat8f(char_lane_sum(x, n, (uint32_t)l) + char_lane_sum(x, n, (uint32_t)l + 16u));
        } else {
            const float a = l < 32 ? char_lane_sum(x, n, (uint32_t)l) : 0.0f;
            return sat8f(a + bitsf(pull_u32(fbits(a), 16)));
        }
    }
    // reduce_adds_epi8 avxconvenience.h:92-101 from the 16 pair sums: (v, v + 8), (v, v + 4), (v, v + 2), (0, 1); lane 0
    __device__ __forceinline__ float char_reduce16(float t) const
    {
        t = sat8f(t + bitsf(pull_u32(fbits(t), 8)));
        t = sat8f(t + bitsf(pull_u32(fbits(t), 4)));
        t = sat8f(t + bitsf(pull_u32(fbits(t), 2)));
        t = sat8f(t + bitsf(pull_u32(fbits(t), 1)));
        return t;
    }
    // half_reduce_adds_epi8 avxconvenience.h:202-210 from the 16 pair sums: butterflies over lane ^ 8, ^ 4, ^ 2
    __device__ __forceinline__ float char_class_sums(float t) const
    {
        if constexpr (LC == 16) {
            t = sat8f(t + bitsf(dpp_u32<0x128>(fbits(t))));
            t = sat8f(t + bitsf(xor4_u32(fbits(t))));
            t = sat8f(t + bitsf(dpp_u32<0x4E>(fbits(t))));
        } else {
            t = sat8f(t + bitsf(shflx_u32(fbits(t), 8)));
            t = sat8f(t + bitsf(shflx_u32(fbits(t), 4)));
            t = sat8f(t + bitsf(shflx_u32(fbits(t), 2)));
        }
        return t;
    }
    // lexicographic (value, index) minimum over the group; key of a lane without candidates = (inf, ~0)
    __device__ __forceinline__ uint32_t min_index(uint32_t best, uint32_t bidx, bool any) const
    {
        const uint32_t idx = (uint32_t)min_key(best, bidx, any);
        return idx == 0xFFFFFFFFu ? 0u : idx;
    }
    // ... the winning (value << 32 | index) itself
    __device__ __forceinline__ uint64_t min_key(uint32_t best, uint32_t bidx, bool any) const
    {
        uint64_t key = any ? (((uint64_t)best << 32) | bidx) : (((uint64_t)0x7F800000u << 32) | 0xFFFFFFFFu);
        // all-reduce inside the group: DPP partners inside a 16-lane row (no LDS round trips), ds_bpermute beyond
        auto step = [&](auto M) {
            constexpr int m = decltype(M)::value;
            if constexpr (m < LC) {
                const uint64_t o = ((uint64_t)partner_u32<m>((uint32_t)(key >> 32)) << 32) | partner_u32<m>((uint32_t)key);
                key = o < key ? o : key;
            }
        };
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 8>{});
        step(std::integral_constant<int, 16>{});
        step(std::integral_constant<int, 32>{});
        return key;
    }
    // first index of the minimum of |x[0..n)| (SpcDecoder::decode :342-373)
    template <typename X>
    __device__ __forceinline__ uint32_t firstmin_abs(const X& x, uint32_t n) const
    {
        uint32_t best = 0x7F800000u, bidx = 0;  // +inf, index 0 (reference start values)
        bool any = false;
        for (uint32_t i = l; i < n; i += LC) {
            const uint32_t a = fbits(x(i)) & 0x7FFFFFFFu;
            if (a < best) {
                best = a;
                bidx = i;
                any = true;
            }
        }
        return min_index(best, bidx, any);
    }
};
__device__ __forceinline__ float reduce_add8(const float a[8])
{
    return ((((((a[0] + a[1]) + a[2]) + a[3]) + a[4]) + a[5]) + a[6]) + a[7];
}

// ARENA: LLR stages s0 <= t < logN - 1 live in a per-codeword slice of global memory (cache resident) instead of LDS,
// so that long blocks keep four codewords per wave and 16 waves per CU (the list kernels do the same, scl_kernel.h).
// B2: the wide F / G loops take two elements per lane and trip (their loads are in flight together).  Worth 2-3 % from
// N = 1024 on; it costs registers (86 -> 115 VGPRs, 4 instead of 5 waves per SIMD), which shorter blocks -- whose LDS
// footprint would allow the fifth wave -- pay with 15 %, so the host picks by block length.
// CH: int8 arithmetic (FastSscFipChar): saturating f / g on int8-valued floats and the SCC_* leaves
// Blocks up to N = 512 (the plain variant): the LDS slice is small, resident waves are bounded by registers, and each
// instruction does little work -- so this variant is compiled for 8 waves per SIMD (<= 64 VGPRs) and keeps lane-derived
// values hoisted (no per-instruction re-derivation, see the interpreter loop).  Measured N = 64 / 128 / 256 / 512:
// 1.65 / 1.05 / 0.63 / 0.35 G cw/s with the re-derivation -> 1.78 / 1.17 / 0.77 / 0.43 G cw/s.
#ifndef PAMD_SC_SMALL_WAVES
#define PAMD_SC_SMALL_WAVES 8
#endif
template <int LCL, bool I8, bool DIAG = false, bool ARENA = false, bool B2 = false, bool CH = false>
__global__ __launch_bounds__(256, (!B2 && !ARENA && !CH && !DIAG && LCL <= 4) ? PAMD_SC_SMALL_WAVES : 1) void sc_decode_kernel(DecodeParams P)
{
    constexpr int LC = 1 << LCL;
    constexpr int CW = 64 >> LCL;  // codewords per wave
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const uint32_t wpb = blockDim.x >> 6;
    ScLanes<LCL> g;
    g.l = lane & (LC - 1);
    g.lbase = lane & ~(LC - 1);
    const uint32_t slot = (uint32_t)lane >> LCL;
    const uint32_t N = P.N, logN = P.logN;
    const uint32_t nwords = N >= 32 ? (N >> 5) : 1u;
    // this codeword's LDS slice
    uint32_t* wl = smem + ((size_t)wib * CW + slot) * P.wave_lds_words;
    float* llr = reinterpret_cast<float*>(wl);
    const bool virt = P.sc_virtual_top != 0;          // stage logN-1 recomputed from the channel, not stored
    const uint32_t s0 = ARENA ? P.s0 : logN;          // first LLR stage kept in the arena
    uint32_t* bits = wl + (ARENA ? (1u << s0) : (virt ? (N >> 1) : N));
    // this codeword's arena slice: stage t at offset (1 << t) - (1 << s0)
    float* ar = ARENA ? P.arena + ((size_t)(blockIdx.x * wpb + wib) * CW + slot) * P.arena_wave_words - (1u << s0) : nullptr;
    const uint32_t wave0 = blockIdx.x * wpb + wib, nwaves = gridDim.x * wpb;
    const const_u64_ptr prog64 = (const_u64_ptr)(uintptr_t)P.prog;
    const uint32_t ngroups = (P.rows + CW - 1) / CW;

    for (uint32_t grp = wave0; grp < ngroups; grp += nwaves) {
        const uint32_t cwi = grp * CW + slot;
        const bool live = cwi < P.rows;  // the last group may be ragged: idle slots decode row 0 and store nothing
        const uint32_t row = live ? (P.rowidx ? P.rowidx[cwi] : cwi) : (P.rowidx ? P.rowidx[0] : 0u);
        const void* chan = I8 ? (const void*)(reinterpret_cast<const int8_t*>(P.llr) + (size_t)row * N)
                              : (const void*)(reinterpret_cast<const float*>(P.llr) + (size_t)row * N);

        const_u64_ptr pp = prog64;
        uint64_t iw_next = pp[0];
        int lane_v = lane;
        // (P.prog_len = instructions in front of the OP_END word, which is still there: the fetch runs one ahead)
        for (uint32_t pc = 0; pc < P.prog_len; ++pc) {
            // the lane number is made opaque once per instruction: lane predicates (64-bit SGPR masks) and other values
            // derived from it are then re-derived per instruction instead of being hoisted out of the interpreter loop
            // (the kernel sits at the SGPR limit: hoisted masks were spilled to VGPR lanes).  Not for the short-block
            // variant, which does too little per instruction to pay for it.
            if constexpr (B2 || ARENA || CH) {
                asm volatile("" : "+v"(lane_v));
                g.l = lane_v & (LC - 1);
                g.lbase = lane_v & ~(LC - 1);
            }
            const int l = g.l;
            const uint64_t iw = iw_next;
            const uint32_t op = (uint32_t)(iw & 0xFF);
            ++pp;
            iw_next = pp[0];  // scalar load, in flight while this instruction executes
            const uint32_t s = (uint32_t)((iw >> 8) & 0xFF), n = 1u << s, h = n >> 1, off = (uint32_t)(iw >> 32);
            uint64_t t_start = 0;
            if (DIAG && P.cycles && grp == 0) t_start = __builtin_amdgcn_s_memtime();

            // Host-fused neighbours (capi.cpp, fuse_sc) -- the SC kernel is bound by the scalar unit (dispatch), so an
            // instruction saved counts more than its body:
            //   pad  bits 0-1: the F / G / G0R (1 / 2 / 3) of the parent stage s + 1 (an LDS stage) that produces this leaf's
            //                  input runs as a prologue;  bits 2-3: the parent's COMBINE / COMBINE0R (1 / 2) runs as an
            //                  epilogue of this (right) leaf;  bits 4-6: plain COMBINEs of the stages above that follow
            //   side bit 3:    the block this instruction finishes is a left child -- the parent's G runs here too
            //   side bits 0-2: F ops of the stages below that follow this F / G / G0R or that tail G (left descent)
            const uint32_t fzz = (uint32_t)(iw >> 16) & 0xFFFFu;
            const uint32_t fz = fzz >> 8;
            // fn(i) for i = l, l + LC, ... < cnt (cnt a power of two).  Written with a wave-uniform trip count: a loop that
            // starts at the lane number is compiled as an exec-mask loop (three scalar instructions per trip for the mask,
            // and this kernel is bound by the scalar unit)
            auto lanes = [&](uint32_t cnt, auto&& fn) {
                if (cnt >= (uint32_t)LC) {
                    for (uint32_t i0 = 0; i0 < cnt; i0 += LC) fn(i0 + (uint32_t)l);
                } else if ((uint32_t)l < cnt) {
                    fn((uint32_t)l);
                }
            };
            if (__builtin_expect(fz != 0u, 0) && (fz & 3u)) {
                const float* X = llr + 2 * n;
                float* d = llr + n;
                const uint32_t pro = fz & 3u, lo = off - n;  // (G: the left sibling's bits sit right below this block)
                lanes(n, [&](uint32_t i) {
                    const float a = X[i], b = X[n + i];
                    float r;
                    if (pro == 1u) {
                        r = polar_fx<CH>(a, b);
                    } else if (pro == 3u) {
                        r = addx<CH>(a, b);
                    } else {
                        const uint32_t bi = lo + i;
                        r = polar_gx<CH>(a, b, (bits[bi >> 5] >> (bi & 31u)) & 1u);
                    }
                    d[i] = r;
                });
                WAVE_SYNC();
            }
            auto combine = [&](uint32_t coff, uint32_t ch_, bool copy) {  // Combine / Combine_0R avx_float.h:190-204
                if (ch_ >= 32) {
                    const uint32_t w0 = coff >> 5, hw = ch_ >> 5;
                    for (uint32_t w = l; w < hw; w += LC) {
                        const uint32_t r = bits[w0 + hw + w];
                        bits[w0 + w] = copy ? r : (bits[w0 + w] ^ r);
                    }
                } else if (l == 0) {
                    const uint32_t sh = coff & 31u, mask = (1u << ch_) - 1u;
                    uint32_t w = bits[coff >> 5];
                    const uint32_t r = (w >> (sh + ch_)) & mask;
                    if (copy)
                        w = (w & ~(mask << sh)) | (r << sh);
                    else
                        w ^= r << sh;
                    bits[coff >> 5] = w;
                }
            };
            // the op body, instantiated for the two kinds of input: an LDS stage buffer, or (root) the channel row
            auto run = [&](auto x) {
                switch (op) {
                case SC_F:
                    if (ARENA && s - 1 >= s0)
                        lanes(h, [&](uint32_t i) { ar[h + i] = polar_fx<CH>(x(i), x(h + i)); });
                    else if (B2 && h >= 2u * LC) {  // two elements per lane and trip: their loads are in flight together
                        for (uint32_t b0_ = 0; b0_ < h; b0_ += 2u * LC) {
                            const uint32_t i0 = b0_ + (uint32_t)l;
                            const float a0 = x(i0), b0 = x(h + i0), a1 = x(i0 + LC), b1 = x(h + i0 + LC);
                            llr[h + i0] = polar_fx<CH>(a0, b0);
                            llr[h + i0 + LC] = polar_fx<CH>(a1, b1);
                        }
                    } else
                        lanes(h, [&](uint32_t i) { llr[h + i] = polar_fx<CH>(x(i), x(h + i)); });
                    break;
                case SC_G:
                    if (ARENA && s - 1 >= s0) {
                        lanes(h, [&](uint32_t i) {
                            const uint32_t b = off + i;
                            ar[h + i] = polar_gx<CH>(x(i), x(h + i), (bits[b >> 5] >> (b & 31u)) & 1u);
                        });
                    } else if (B2 && h >= 2u * LC) {
                        for (uint32_t b0_ = 0; b0_ < h; b0_ += 2u * LC) {
                            const uint32_t i0 = b0_ + (uint32_t)l;
                            const uint32_t bi0 = off + i0, bi1 = bi0 + LC;
                            const float a0 = x(i0), b0 = x(h + i0), a1 = x(i0 + LC), b1 = x(h + i0 + LC);
                            const uint32_t u0 = (bits[bi0 >> 5] >> (bi0 & 31u)) & 1u, u1 = (bits[bi1 >> 5] >> (bi1 & 31u)) & 1u;
                            llr[h + i0] = polar_gx<CH>(a0, b0, u0);
                            llr[h + i0 + LC] = polar_gx<CH>(a1, b1, u1);
                        }
                    } else {
                        lanes(h, [&](uint32_t i) {
                            const uint32_t b = off + i;
                            llr[h + i] = polar_gx<CH>(x(i), x(h + i), (bits[b >> 5] >> (b & 31u)) & 1u);
                        });
                    }
                    break;
                case SC_G0R:
                    if (ARENA && s - 1 >= s0)
                        lanes(h, [&](uint32_t i) { ar[h + i] = addx<CH>(x(i), x(h + i)); });
                    else
                        lanes(h, [&](uint32_t i) { llr[h + i] = addx<CH>(x(i), x(h + i)); });
                    break;
                case SC_COMBINE:
                case SC_COMBINE0R:
                    combine(off, h, op == SC_COMBINE0R);
                    break;
                case SC_RONE: {  // h >= 8
                    for (uint32_t i0 = 0; i0 < h; i0 += LC) {
                        const uint32_t i = i0 + l;
                        const bool act = i < h;
                        uint32_t u = 0, sl = 0;
                        if (act) {
                            const uint32_t b = off + i;
                            u = (bits[b >> 5] >> (b & 31u)) & 1u;
                            sl = sgn(polar_gx<CH>(x(i), x(h + i), u));
                        }
                        const uint64_t mr = g.slice(ballot64(act && sl));
                        const uint64_t ml = g.slice(ballot64(act && (sl ^ u)));
                        WAVE_SYNC();
                        const uint32_t cnt = h < (uint32_t)LC ? h : (uint32_t)LC;
                        g.put_bits(bits, off + i0, cnt, ml);
                        g.put_bits(bits, off + h + i0, cnt, mr);
                    }
                    break;
                }
                case SC_R0:
                    g.fill_pattern(bits, off, n, 0u);
                    break;
                case SC_R1:
                    for (uint32_t i0 = 0; i0 < n; i0 += LC) {
                        const uint32_t i = i0 + l;
                        const uint64_t m = g.slice(ballot64(i < n && sgn(x(i < n ? i : 0))));
                        g.put_bits(bits, off + i0, n < (uint32_t)LC ? n : (uint32_t)LC, m);
                    }
                    break;
                case SC_REP:
                    g.fill_pattern(bits, off, n, g.lanesum_negative(x, n) ? 0xFFFFFFFFu : 0u);
                    break;
                case SC_SPC: {  // one pass: hard decisions, parity and the first minimum of |x| (SpcDecoder::decode :342-373)
                    uint32_t par = 0, best = 0x7F800000u, bidx = 0;
                    bool any = false;
                    for (uint32_t i0 = 0; i0 < n; i0 += LC) {
                        const uint32_t i = i0 + l;
                        const bool in = i < n;
                        const uint32_t xb = fbits(x(in ? i : 0));
                        const uint64_t m = g.slice(ballot64(in && (xb >> 31)));
                        par ^= (uint32_t)__popcll(m) & 1u;
                        g.put_bits(bits, off + i0, n < (uint32_t)LC ? n : (uint32_t)LC, m);
                        const uint32_t a = xb & 0x7FFFFFFFu;
                        if (in && a < best) {
                            best = a;
                            bidx = i;
                            any = true;
                        }
                    }
                    const uint32_t im = g.min_index(best, bidx, any);
                    if (par) g.flip_bit(bits, off + im);
                    break;
                }
                case SC_DREP: {
                    float a[8];
                    g.fold8(x, n, a);
                    float e, o;
                    if (n >= 8) {
                        e = (a[0] + a[4]) + (a[2] + a[6]);
                        o = (a[1] + a[5]) + (a[3] + a[7]);
                    } else {
                        e = ((a[0] + a[2]) + a[4]) + a[6];
                        o = ((a[1] + a[3]) + a[5]) + a[7];
                    }
                    g.fill_pattern(bits, off, n, g.from_lane0((sgn(e) ? 0x55555555u : 0u) | (sgn(o) ? 0xAAAAAAAAu : 0u)));
                    break;
                }
                case SC_DSPC8: {  // flip EVERY class member tied for the class minimum
                    const int l8 = l & 7;
                    const float v = x(l8);
                    float m = absf(v);
                    uint32_t p = sgn(v);
                    {  // (lanes l ^ 2 and l ^ 4 by DPP: both keep the parity class of l)
                        float om = bitsf(partner_u32<2>(fbits(m)));
                        m = om < m ? om : m;
                        p ^= partner_u32<2>(p);
                        om = bitsf(g.xor4_u32(fbits(m)));
                        m = om < m ? om : m;
                        p ^= g.xor4_u32(p);
                    }
                    const uint32_t bit = sgn(v) ^ ((p && absf(v) == m) ? 1u : 0u);
                    g.put_bits(bits, off, 8, g.slice(ballot64(l < 8 && bit)));
                    break;
                }
                case SC_DSPC: {  // n > 8: per AVX lane the LAST row with the lane minimum, lowest lane wins
                    float minv = 3.402823466e+38f;
                    uint32_t mini = 0;
                    if (l < 8) {
                        for (uint32_t i = l; i < n; i += 8) {
                            const float a = absf(x(i));
                            if (!(a > minv)) {
                                minv = a;
                                mini = i;
                            }
                        }
                    }
                    float cm = minv;  // class minimum over lanes of equal parity (within lanes 0..7 of the group)
                    {
                        float om = bitsf(partner_u32<2>(fbits(cm)));
                        cm = om < cm ? om : cm;
                        om = bitsf(g.xor4_u32(fbits(cm)));
                        cm = om < cm ? om : cm;
                    }
                    const uint64_t tie = g.slice(ballot64(l < 8 && minv == cm));
                    const int le = __ffsll((long long)(tie & 0x55ull)) - 1;
                    const int lo = __ffsll((long long)(tie & 0xAAull)) - 1;
                    const uint32_t ie = shfl_u32(mini, g.lbase + (le < 0 ? 0 : le)), io = shfl_u32(mini, g.lbase + (lo < 0 ? 1 : lo));
                    uint32_t pe = 0, po = 0;
                    for (uint32_t i0 = 0; i0 < n; i0 += LC) {
                        const uint32_t i = i0 + l;
                        const uint64_t m = g.slice(ballot64(i < n && sgn(x(i < n ? i : 0))));
                        pe ^= (uint32_t)__popcll(m & 0x5555555555555555ull) & 1u;
                        po ^= (uint32_t)__popcll(m & 0xAAAAAAAAAAAAAAAAull) & 1u;
                        g.put_bits(bits, off + i0, n < (uint32_t)LC ? n : (uint32_t)LC, m);
                    }
                    if (pe) g.flip_bit(bits, off + ie);
                    if (po) g.flip_bit(bits, off + io);
                    break;
                }
                case SC_TREP: {
                    float a[8], t[4];
                    g.fold8(x, n, a);
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = a[j] + a[j + 4];
                    g.fill_pattern(bits, off, n, g.from_lane0(spc4_signs(t) * 0x11111111u));
                    break;
                }
                case SC_TYPE5:
                case SC_REPONE8: {
                    float v8[8], q[4], gg[4];
                    if (op == SC_TYPE5) {
                        g.fold8(x, n, v8);
                    } else {
#pragma unroll
                        for (int v = 0; v < 8; ++v) v8[v] = x(v);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) q[j] = polar_fx<CH>(v8[j], v8[j + 4]);
                    const float rho = (q[0] + q[1]) + (q[2] + q[3]);
                    const uint32_t sr = sgn(rho);
#pragma unroll
                    for (int j = 0; j < 4; ++j) gg[j] = polar_gx<CH>(v8[j], v8[j + 4], sr);
                    uint32_t ss;
                    if (op == SC_TYPE5)
                        ss = spc4_signs(gg);
                    else
                        ss = sgn(gg[0]) | (sgn(gg[1]) << 1) | (sgn(gg[2]) << 2) | (sgn(gg[3]) << 3);
                    const uint32_t byte = ((sr ? (ss ^ 0xFu) : ss) & 0xFu) | (ss << 4);
                    g.fill_pattern(bits, off, n, g.from_lane0(byte * 0x01010101u));
                    break;
                }
                case SC_ZSPC8: {
                    float t[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = x(j) + x(j + 4);
                    const uint32_t r = spc4_signs(t);
                    g.put_bits(bits, off, 8, (uint64_t)(r | (r << 4)));
                    break;
                }
                case SC_ZEROSPC: {  // n > 8; outputs the RIGHT half's signs
                    uint32_t par = 0;
                    uint32_t best = 0x7F800000u, bidx = 0;
                    bool any = false;
                    for (uint32_t i0 = 0; i0 < h; i0 += LC) {
                        const uint32_t i = i0 + l;
                        const bool act = i < h;
                        float y = 0.0f, r = 0.0f;
                        if (act) {
                            r = x(h + i);
                            y = x(i) + r;
                            const uint32_t a = fbits(y) & 0x7FFFFFFFu;
                            if (a < best) {
                                best = a;
                                bidx = i;
                                any = true;
                            }
                        }
                        par ^= (uint32_t)__popcll(g.slice(ballot64(act && sgn(y)))) & 1u;
                        const uint64_t m = g.slice(ballot64(act && sgn(r)));
                        const uint32_t cnt = h < (uint32_t)LC ? h : (uint32_t)LC;
                        g.put_bits(bits, off + i0, cnt, m);
                        g.put_bits(bits, off + h + i0, cnt, m);
                    }
                    const uint32_t im = g.min_index(best, bidx, any);
                    if (par) {
                        g.flip_bit(bits, off + im);
                        g.flip_bit(bits, off + h + im);
                    }
                    break;
                }
                case SCC_REP: {  // sign of reduce_adds_epi8 over the 32 byte-lane sums (fastssc_fip_char.cpp:219-234,259-266)
                    if constexpr (CH) {
                        const float t = g.char_lane_pairs(x, n);
                        g.fill_pattern(bits, off, n, g.from_lane0(sgn(g.char_reduce16(t)) ? 0xFFFFFFFFu : 0u));
                    }
                    break;
                }
                case SCC_DREP: {  // half_reduce_adds_epi8: even / odd positions take the saturating sum of their class (:242-257)
                    if constexpr (CH) {
                        float t = g.char_lane_pairs(x, n);
                        t = g.char_class_sums(t);
                        const uint32_t ev = sgn(t) ? 0x55555555u : 0u;  // (lane 0: even class, lane 1: odd class)
                        const uint32_t od = g.pull_u32(sgn(t) ? 0xAAAAAAAAu : 0u, 1);
                        g.fill_pattern(bits, off, n, g.from_lane0(ev | od));
                    }
                    break;
                }
                case SCC_SPC:        // SpcDecoder :268-297 / ShortSpcDecoder :299-313
                case SCC_ZEROSPC: {  // ZeroSpcDecoder :315-353 / ShortZeroSpcDecoder :355-384 (on y = sat(a + b), output twice)
                    if constexpr (CH) {
                        const bool zs = op == SCC_ZEROSPC;
                        const uint32_t m_ = zs ? h : n;  // elements searched
                        auto y = [&](uint32_t i) { return zs ? addx<true>(x(i), x(h + i)) : x(i); };
                        uint32_t par = 0, best = 0x7F800000u, bidx = 0;
                        bool any = false;
                        for (uint32_t i0 = 0; i0 < m_; i0 += LC) {
                            const uint32_t i = i0 + l;
                            const bool in = i < m_;
                            const uint32_t yb = fbits(y(in ? i : 0));
                            const uint64_t m = g.slice(ballot64(in && (yb >> 31)));
                            par ^= (uint32_t)__popcll(m) & 1u;
                            const uint32_t cnt = m_ < (uint32_t)LC ? m_ : (uint32_t)LC;
                            g.put_bits(bits, off + i0, cnt, m);
                            if (zs) g.put_bits(bits, off + h + i0, cnt, m);
                            const uint32_t a = yb & 0x7FFFFFFFu;  // fi_abs_epi8 read as unsigned: |-128| = 128 sorts last
                            if (in && a < best) {
                                best = a;
                                bidx = i;
                                any = true;
                            }
                        }
                        const uint64_t key = g.min_key(best, bidx, any);  // first index of the minimum (minpos_epu8)
                        const uint32_t mv = (uint32_t)(key >> 32);
                        uint32_t im = (uint32_t)key;
                        bool flip = par != 0;
                        if (n > 32) {  // vector-wise search, `testAbs < minAbs` from minAbs = 127: nothing below 127 -> index 0
                            if (mv >= 0x42FE0000u) im = 0;
                        } else if (mv > 0x42FE0000u) {  // one padded vector: the 127 padding wins against all -128
                            flip = false;
                        }
                        // the flip is the char negation: 0 and -128 keep their sign
                        const float v = y(im);
                        if (flip && v != 0.0f && v != -128.0f) {
                            g.flip_bit(bits, off + im);
                            if (zs) g.flip_bit(bits, off + h + im);
                        }
                    }
                    break;
                }
                default: break;
                }
            };
            uint32_t fskip = 0;  // F steps of the left descent already done by the op itself (the fused top op)
            auto ch = [&](uint32_t i) -> float {
                if constexpr (I8)
                    return (float)reinterpret_cast<const int8_t*>(chan)[i];
                else
                    return reinterpret_cast<const float*>(chan)[i];
            };
            if (s == logN) {
                // the stage below the root is never stored (see below), so the root's F / G / G0R have nothing to do
                if (!(virt && (op == SC_F || op == SC_G || op == SC_G0R))) run(ch);
            } else if (virt && s + 1 == logN) {
                // "virtual" stage logN-1: its LLRs are recomputed from the channel row (L2-resident) on every use
                // instead of living in LDS -- left half: f(x_i, x_{i+N/2}); right half: g(.., root-left bit i) where
                // the root's left output still sits untouched in bits[0, N/2)
                const uint32_t hN = N >> 1;
                const bool fgop = op == SC_F || op == SC_G || op == SC_G0R;
                if (LCL == 4 && fgop && h >= 4u * LC) {
                    // the root's grandchild block, four consecutive LLRs per lane and trip: four 16-byte channel loads (one
                    // 4-byte load each for int8 rows), one 16-byte store, the steering bits four at a time
                    auto ch4 = [&](uint32_t j) -> float4 {
                        if constexpr (I8) {
                            const uint32_t w = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const int8_t*>(chan) + j);
                            return make_float4((float)(int8_t)(w & 0xFFu), (float)(int8_t)((w >> 8) & 0xFFu),
                                               (float)(int8_t)((w >> 16) & 0xFFu), (float)(int8_t)(w >> 24));
                        } else {
                            return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(chan) + j);
                        }
                    };
                    auto f4 = [](const float4& a, const float4& b) {
                        return make_float4(polar_fx<CH>(a.x, b.x), polar_fx<CH>(a.y, b.y), polar_fx<CH>(a.z, b.z), polar_fx<CH>(a.w, b.w));
                    };
                    auto a4 = [](const float4& a, const float4& b) {
                        return make_float4(addx<CH>(a.x, b.x), addx<CH>(a.y, b.y), addx<CH>(a.z, b.z), addx<CH>(a.w, b.w));
                    };
                    auto g4 = [](const float4& a, const float4& b, uint32_t u) {
                        return make_float4(polar_gx<CH>(a.x, b.x, u & 1u), polar_gx<CH>(a.y, b.y, (u >> 1) & 1u),
                                           polar_gx<CH>(a.z, b.z, (u >> 2) & 1u), polar_gx<CH>(a.w, b.w, (u >> 3) & 1u));
                    };
                    auto bits4 = [&](uint32_t j) { return (bits[j >> 5] >> (j & 31u)) & 0xFu; };
                    // N = 1024 (four trips): the first two F steps of the left descent that follows (side bits 0-2) run on the
                    // results in registers: with four consecutive LLRs per lane and 64 per trip, the partner i + h / 2 (then
                    // i + h / 4) of a chunk is the same lane's chunk two trips (then one trip) further on
                    auto fused = [&](auto x4, auto T_) {
                        constexpr int T = decltype(T_)::value;
                        float4 r[T];
#pragma unroll
                        for (int t = 0; t < T; ++t) {
                            const uint32_t i = 64u * (uint32_t)t + 4u * (uint32_t)l;
                            const float4 a = x4(i), b = x4(h + i);
                            if (op == SC_F)
                                r[t] = f4(a, b);
                            else if (op == SC_G0R)
                                r[t] = a4(a, b);
                            else
                                r[t] = g4(a, b, bits4(off + i));
                            *reinterpret_cast<float4*>(llr + h + i) = r[t];
                        }
                        const uint32_t nf = fzz & 7u;
                        if (nf >= 1u) {
                            float4 s7[T / 2];
#pragma unroll
                            for (int t = 0; t < T / 2; ++t) {
                                s7[t] = f4(r[t], r[t + T / 2]);
                                *reinterpret_cast<float4*>(llr + (h >> 1) + 64u * (uint32_t)t + 4u * (uint32_t)l) = s7[t];
                            }
                            fskip = 1;
                            if constexpr (T >= 4) {
                                if (nf >= 2u) {
#pragma unroll
                                    for (int t = 0; t < T / 4; ++t)
                                        *reinterpret_cast<float4*>(llr + (h >> 2) + 64u * (uint32_t)t + 4u * (uint32_t)l) = f4(s7[t], s7[t + T / 4]);
                                    fskip = 2;
                                }
                            }
                        }
                    };
                    auto trips = [&](auto x4) {
                        if constexpr (B2 && !ARENA) {  // (N = 1024; the short-block variant has no registers to spare)
                            if (h == 256u) {
                                fused(x4, std::integral_constant<int, 4>{});
                                return;
                            }
                        }
                        for (uint32_t i0 = 0; i0 < h; i0 += 4u * LC) {
                            const uint32_t i = i0 + 4u * (uint32_t)l;
                            const float4 a = x4(i), b = x4(h + i);
                            float4 r;
                            if (op == SC_F)
                                r = f4(a, b);
                            else if (op == SC_G0R)
                                r = a4(a, b);
                            else
                                r = g4(a, b, bits4(off + i));
                            if (ARENA && s - 1 >= s0)
                                *reinterpret_cast<float4*>(ar + h + i) = r;
                            else
                                *reinterpret_cast<float4*>(llr + h + i) = r;
                        }
                    };
                    if (off == 0)
                        trips([&](uint32_t j) { return f4(ch4(j), ch4(hN + j)); });
                    else if (P.prog_root_g0r)
                        trips([&](uint32_t j) { return a4(ch4(j), ch4(hN + j)); });
                    else
                        trips([&](uint32_t j) { return g4(ch4(j), ch4(hN + j), bits4(j)); });
                } else
                if (off == 0)
                    run([&](uint32_t i) { return polar_fx<CH>(ch(i), ch(hN + i)); });
                else if (P.prog_root_g0r)
                    run([&](uint32_t i) { return addx<CH>(ch(i), ch(hN + i)); });
                else
                    run([&](uint32_t i) { return polar_gx<CH>(ch(i), ch(hN + i), (bits[i >> 5] >> (i & 31u)) & 1u); });
            } else if (ARENA && s >= s0) {
                const float* xs = ar + n;  // written earlier by this wave, waited for below
                run([&](uint32_t i) { return xs[i]; });
            } else {
                const float* xs = llr + n;
                run([&](uint32_t i) { return xs[i]; });
            }
            if (__builtin_expect((fzz & 0xFC0Fu) != 0u, 0)) {  // (anything but a prologue)
                uint32_t coff = off, ch_ = h;  // (head = COMBINE / COMBINE0R / RONE: the finished block is its own)
                if (fz & 12u) {  // the parent's COMBINE: this leaf is its right child, the block starts n bits below
                    WAVE_SYNC();
                    coff = off - n;
                    ch_ = n;
                    combine(coff, ch_, (fz & 12u) == 8u);
                }
                for (uint32_t t = (fz >> 4) & 7u; t; --t) {  // ... and it is the right child of the next one up
                    coff -= 2 * ch_;
                    ch_ *= 2;
                    WAVE_SYNC();
                    combine(coff, ch_, false);
                }
                uint32_t fbase = h;  // length of the stage the first follower F reads
                if (fzz & 8u) {  // G of the parent of the finished block [coff, coff + 2 ch_) (RateRNode::decode :148-155)
                    const uint32_t gh = 2 * ch_;
                    WAVE_SYNC();
                    const float* X = llr + 2 * gh;
                    float* d = llr + gh;
                    lanes(gh, [&](uint32_t i) {
                        const uint32_t bi = coff + i;
                        d[i] = polar_gx<CH>(X[i], X[gh + i], (bits[bi >> 5] >> (bi & 31u)) & 1u);
                    });
                    fbase = gh;
                }
                for (uint32_t t = fskip; t < (fzz & 7u); ++t) {  // F of the stages below (left descent)
                    const uint32_t fn = fbase >> t, fh = fn >> 1;
                    WAVE_SYNC();
                    const float* X = llr + fn;
                    float* d = llr + fh;
                    lanes(fh, [&](uint32_t i) { d[i] = polar_fx<CH>(X[i], X[fh + i]); });
                }
            }
            if (ARENA && (op == SC_F || op == SC_G || op == SC_G0R) && s - 1 >= s0) {
                // the op stored LLRs to the arena: wait for the stores before other lanes of the wave read them
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            WAVE_SYNC();
            if (DIAG && P.cycles && grp == 0) {  // diagnostics: cycles this instruction took for the first group
                const uint64_t t_end = __builtin_amdgcn_s_memtime();
                if (lane == 0) P.cycles[pc] = (uint32_t)(t_end - t_start);
            }
        }

        // ---- output: FastSscAvxFloat::decode :940-955 ------------------------------------------
        const int l = g.l;
        if (!P.systematic) {  // re-encode (butterfly_fip.cpp:15-63) on the packed words
            for (uint32_t w = l; w < nwords; w += LC) {
                uint32_t v = bits[w];
                if (N >= 2) v ^= (v >> 1) & 0x55555555u;
                if (N >= 4) v ^= (v >> 2) & 0x33333333u;
                if (N >= 8) v ^= (v >> 4) & 0x0F0F0F0Fu;
                if (N >= 16) v ^= (v >> 8) & 0x00FF00FFu;
                if (N >= 32) v ^= (v >> 16) & 0x0000FFFFu;
                bits[w] = v;
            }
            WAVE_SYNC();
            for (uint32_t hw = 1; hw < nwords; hw <<= 1) {
                for (uint32_t w = l; w < nwords; w += LC)
                    if (!(w & hw)) bits[w] ^= bits[w + hw];
                WAVE_SYNC();
            }
        }
        uint32_t syn = 0;
        uint8_t* orow = P.out + (size_t)row * P.nb;
        for (uint32_t t0 = 0; t0 < P.K; t0 += LC) {
            const uint32_t t = t0 + l;
            uint32_t bit = 0;
            if (t < P.K) {
                const uint32_t pos = P.lut[t];
                bit = (bits[pos >> 5] >> (pos & 31u)) & 1u;
                if (bit) syn ^= P.syn[t];
            }
            const uint64_t m = g.slice(ballot64(bit != 0));
            if (l < LC / 8) {
                const uint32_t byte = (t0 >> 3) + l;
                if (live && byte < P.nb) orow[byte] = (uint8_t)(__brev((uint32_t)((m >> (8 * l)) & 0xFFu)) >> 24);
            }
        }
#pragma unroll
        for (int m = 1; m < LC; m <<= 1) syn ^= shflx_u32(syn, m);
        if (live && l == 0 && P.ok) P.ok[row] = (syn == 0) ? 1 : 0;
        WAVE_SYNC();
    }
}

// =============================================================================================
//  small helper kernels
// =============================================================================================
// "mixed" (AdaptiveFloat::decode adaptive_float.cpp:33-45): gather the rows the SC pass failed on
__global__ void collect_failed_kernel(const uint8_t* ok, uint32_t rows, uint32_t* list, uint32_t* count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows && !ok[i]) {
        const uint32_t k = atomicAdd(count, 1u);
        list[k] = i;
    }
}

// float rows for an int8 decoder: _mm256_cvtps_epi32 (round to nearest even) + saturating packs (bitcontainer.cpp:468-503).
// Finite values: round to nearest even, saturate to [-128, 127] -- as the reference.  NaN -> -128, as the reference's
// conversion gives (cvtps yields INT_MIN, the packs saturate it).  Deliberate deviation: +inf and values >= 2^31 give +127
// here (the reference's cvtps turns them into INT_MIN, i.e. -128: an artefact of the x86 conversion, not a decision).
__global__ void f32_to_i8_kernel(const float* in, int8_t* out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = in[i];
        const float v = rintf(x);
        out[i] = (x != x) ? (int8_t)-128 : (int8_t)(int)(v < -128.0f ? -128.0f : (v > 127.0f ? 127.0f : v));
    }
}

// =============================================================================================
//  launchers
// =============================================================================================
template <int LCL, bool I8>
static hipError_t launch_sc_k(const DecodeParams& P, dim3 g, dim3 b, size_t lds, hipStream_t st, int* occupancy)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc_decode_kernel<LCL, I8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (occupancy) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, sc_decode_kernel<LCL, I8>, (int)b.x, lds);
    hipLaunchKernelGGL((sc_decode_kernel<LCL, I8>), g, b, lds, st, P);
    return hipGetLastError();
}

// lcl: log2 of the lanes that share one codeword (4 -> four codewords per wave, 6 -> one)
hipError_t launch_sc(const DecodeParams& P, int lcl, bool i8, unsigned grid, unsigned waves_per_block, size_t lds,
                     hipStream_t st, int* occupancy)
{
    dim3 g(grid), b(64 * waves_per_block);
    if (P.chr) {  // FastSscFipChar: int8 rows, int8 arithmetic; 16 or 64 lanes per codeword
        if (!i8 || P.cycles) return hipErrorInvalidValue;
#define PAMD_SCC_LAUNCH(LCL_, ARENA_, B2_)                                                                            \
    do {                                                                                                             \
        auto kern = sc_decode_kernel<LCL_, true, false, ARENA_, B2_, true>;                                          \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (occupancy) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, kern, (int)b.x, lds);          \
        hipLaunchKernelGGL(kern, g, b, lds, st, P);                                                                  \
        return hipGetLastError();                                                                                    \
    } while (0)
        if (lcl == 4 && P.s0 < P.logN) PAMD_SCC_LAUNCH(4, true, true);
        if (lcl == 4 && P.logN >= 10) PAMD_SCC_LAUNCH(4, false, true);
        if (lcl == 4) PAMD_SCC_LAUNCH(4, false, false);
        if (lcl == 6) PAMD_SCC_LAUNCH(6, false, false);
#undef PAMD_SCC_LAUNCH
        return hipErrorInvalidValue;
    }
    if (lcl == 3)
        return i8 ? launch_sc_k<3, true>(P, g, b, lds, st, occupancy) : launch_sc_k<3, false>(P, g, b, lds, st, occupancy);
    if (lcl == 4 && P.s0 < P.logN && !P.cycles) {  // long blocks: upper LLR stages in the arena
        if (i8) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc_decode_kernel<4, true, false, true, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (occupancy) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, sc_decode_kernel<4, true, false, true, true>, (int)b.x, lds);
            hipLaunchKernelGGL((sc_decode_kernel<4, true, false, true, true>), g, b, lds, st, P);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc_decode_kernel<4, false, false, true, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (occupancy) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, sc_decode_kernel<4, false, false, true, true>, (int)b.x, lds);
            hipLaunchKernelGGL((sc_decode_kernel<4, false, false, true, true>), g, b, lds, st, P);
        }
        return hipGetLastError();
    }
#ifdef PAMD_DIAG
    if (lcl == 4 && P.cycles && !i8) {  // per-instruction cycle stamps (diagnostic build only)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc_decode_kernel<4, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        // (an occupancy query must not launch: the caller caches the answer as the grid size of every later call)
        if (occupancy) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, sc_decode_kernel<4, false, true>, (int)b.x, lds);
        hipLaunchKernelGGL((sc_decode_kernel<4, false, true>), g, b, lds, st, P);
        return hipGetLastError();
    }
#else
    if (P.cycles) return hipErrorInvalidValue;
#endif
    if (lcl == 4 && P.logN >= 10) {  // (N = 1024 without arena: the two-deep variant)
        if (i8) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc_decode_kernel<4, true, false, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (occupancy) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, sc_decode_kernel<4, true, false, false, true>, (int)b.x, lds);
            hipLaunchKernelGGL((sc_decode_kernel<4, true, false, false, true>), g, b, lds, st, P);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sc_decode_kernel<4, false, false, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (occupancy) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occupancy, sc_decode_kernel<4, false, false, false, true>, (int)b.x, lds);
            hipLaunchKernelGGL((sc_decode_kernel<4, false, false, false, true>), g, b, lds, st, P);
        }
        return hipGetLastError();
    }
    if (lcl == 4)
        return i8 ? launch_sc_k<4, true>(P, g, b, lds, st, occupancy) : launch_sc_k<4, false>(P, g, b, lds, st, occupancy);
    return i8 ? launch_sc_k<6, true>(P, g, b, lds, st, occupancy) : launch_sc_k<6, false>(P, g, b, lds, st, occupancy);
}

hipError_t launch_scl(const DecodeParams& P, int lpl, bool i8, unsigned grid, unsigned waves_per_block,
                      size_t lds, hipStream_t st, int* occupancy)
{
    dim3 g(grid), b(64 * waves_per_block);
    const bool diag = P.trace || P.cycles || P.ablate;
    switch (lpl * 4 + (int)P.cwl) {  // (log2 lanes per path, log2 codewords per wave)
    case 1 * 4 + 0: return launch_scl_inst_1_0(P, i8, diag, g, b, lds, st, occupancy);
    case 2 * 4 + 0: return launch_scl_inst_2_0(P, i8, diag, g, b, lds, st, occupancy);
    case 3 * 4 + 0: return launch_scl_inst_3_0(P, i8, diag, g, b, lds, st, occupancy);
    case 4 * 4 + 0: return launch_scl_inst_4_0(P, i8, diag, g, b, lds, st, occupancy);
    case 5 * 4 + 0: return launch_scl_inst_5_0(P, i8, diag, g, b, lds, st, occupancy);
    case 2 * 4 + 1: return launch_scl_inst_2_1(P, i8, diag, g, b, lds, st, occupancy);  // L <= 8: 4 lanes per path, 2 codewords
    case 3 * 4 + 1: return launch_scl_inst_3_1(P, i8, diag, g, b, lds, st, occupancy);  // L <= 4: 8 lanes per path, 2 codewords
    case 3 * 4 + 2: return launch_scl_inst_3_2(P, i8, diag, g, b, lds, st, occupancy);  // L <= 2: 8 lanes per path, 4 codewords
    case 2 * 4 + 2: return launch_scl_inst_2_2(P, i8, diag, g, b, lds, st, occupancy);  // L <= 4: 4 lanes per path, 4 codewords
    case 2 * 4 + 3: return launch_scl_inst_2_3(P, i8, diag, g, b, lds, st, occupancy);  // L <= 2: 4 lanes per path, 8 codewords
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_f32_to_i8(const float* in, int8_t* out, size_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const size_t blocks = std::min<size_t>((n + 255) / 256, (size_t)1 << 20);  // (grid-stride loop: any n)
    hipLaunchKernelGGL(f32_to_i8_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, n);
    return hipGetLastError();
}

hipError_t launch_collect_failed(const uint8_t* ok, uint32_t rows, uint32_t* list, uint32_t* count,
                                 hipStream_t st)
{
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(collect_failed_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, ok, rows, list, count);
    return hipGetLastError();
}

}  // namespace pamd
