// Batched polar encoder for gfx950: the reference's ButterflyFipPacked::encode (butterfly_fip_packed.cpp:45-59)
//   generate checksum in place -> place the K information bits on the non-frozen positions -> x = u G
//   -> (systematic) clear the frozen positions and transform again -> packed code word.
// A code word is N/32 words of 32 code bits; a lane owns WPL consecutive words (16 bytes of the packed row for
// WPL = 4) and 2^GLL lanes share a code word, so a wave encodes 64 >> GLL code words at a time (8 at N = 1024).
// Butterfly stages inside a word are shift-xor, stages between a lane's words are register xors, stages between
// lanes are DPP / ds_swizzle exchanges.  The checksum is 8 nibble-table lookups per information dword (all
// supported detectors are GF(2)-linear), XOR-reduced across the lanes of the code word.
// HBM-bound by design: per code word nb bytes in, N/8 bytes out (+ the checksum field written back).
#include "encoder.h"

namespace pamd {

#define ENC_WAVE_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

// packed bytes are MSB first (bit i of a byte string = byte i/8, bit 7 - i%8); the kernel works LSB first
// (bit i = word i/32, bit i%32): the two differ by a bit reversal inside every byte
__device__ __forceinline__ uint32_t flip_bytes(uint32_t w) { return __builtin_bswap32(__brev(w)); }

template <int CTRL>
__device__ __forceinline__ uint32_t enc_dpp(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
// value held by lane (l ^ M)
template <int M>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v)
{
    if constexpr (M == 1) return enc_dpp<0xB1>(v);                       // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return enc_dpp<0x4E>(v);                  // quad_perm [2,3,0,1]
    else if constexpr (M == 4) return enc_dpp<0x1B>(enc_dpp<0x141>(v));  // row_half_mirror (^7) then quad reverse (^3)
    else if constexpr (M == 8) return enc_dpp<0x141>(enc_dpp<0x140>(v)); // row_mirror (^15) then row_half_mirror (^7)
    else if constexpr (M == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (16 << 10) | 0x1F);
    else return (uint32_t)__shfl_xor((int)v, 32, 64);
}
// XOR over the 2^GLL lanes of a group (any pairing that mixes the halves will do)
template <int GLL>
__device__ __forceinline__ uint32_t group_xor(uint32_t v)
{
    if constexpr (GLL > 0) v ^= enc_dpp<0xB1>(v);
    if constexpr (GLL > 1) v ^= enc_dpp<0x4E>(v);
    if constexpr (GLL > 2) v ^= enc_dpp<0x141>(v);
    if constexpr (GLL > 3) v ^= enc_dpp<0x140>(v);
    if constexpr (GLL > 4) v ^= lane_xor<16>(v);
    if constexpr (GLL > 5) v ^= lane_xor<32>(v);
    return v;
}

// Plain C for the bit logic: forcing v_bitop3_b32 (__builtin_amdgcn_bitop3_b32) measured 10 % SLOWER here, and
// buffer loads for the table lookups 7 % slower than plain global loads (MI355X, N = 1024, round 1).
__device__ __forceinline__ uint32_t xor_and(uint32_t a, uint32_t b, uint32_t c) { return a ^ (b & c); }
// (x & ~m) | (y & m)
__device__ __forceinline__ uint32_t bit_select(uint32_t m, uint32_t y, uint32_t x) { return (x & ~m) | (y & m); }

struct DepRegs {
    uint32_t mask, base, mv0, mv1, mv2, mv3, mv4;
};
__device__ __forceinline__ DepRegs load_dep(const DepositWord* dep, uint32_t w)
{
    const uint4* dp = reinterpret_cast<const uint4*>(dep + w);
    const uint4 d0 = dp[0], d1 = dp[1];
    return DepRegs{ d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z };
}

// WPL: words of 32 code bits per lane (word w of the code block = lane w / WPL, register w % WPL);
// GLL: log2(lanes per code word)
template <int WPL, int GLL, bool ALIGNED>
__global__ __launch_bounds__(256) void encode_kernel(EncodeParams P)
{
    extern __shared__ uint32_t lds[];
    constexpr uint32_t GL = 1u << GLL, G = 64u >> GLL;
    constexpr bool HOIST = WPL <= 4;  // keep the placement tables in registers across code words
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const uint32_t l = (uint32_t)lane & (GL - 1u), slot = (uint32_t)lane >> GLL;
    uint32_t* iw = lds + ((uint32_t)wib * G + slot) * P.iw_words;
    const uint32_t groups = (P.rows + G - 1u) / G;
    const uint32_t code_bytes = P.N >> 3, nbw = (P.nb + 3u) >> 2;
    // checksum field as a 64-bit window over dwords k0, k0 + 1 of the information word
    const uint32_t fsh = 8u * (P.field_off & 3u), k0 = P.field_off >> 2;
    const uint64_t fmask = (P.field_bytes >= 4 ? 0xFFFFFFFFull : ((1ull << (8u * P.field_bytes)) - 1ull)) << fsh;

    uint32_t low[6];  // all-ones in the lanes that hold the lower word of a butterfly at lane distance 2^i
#pragma unroll
    for (int i = 0; i < 6; ++i) low[i] = (l >> i) & 1u ? 0u : 0xFFFFFFFFu;

    DepRegs dh[HOIST ? WPL : 1];
    if constexpr (HOIST) {
#pragma unroll
        for (int r = 0; r < WPL; ++r) dh[r] = load_dep(P.dep, l * WPL + (uint32_t)r);
    }

    for (uint32_t grp = blockIdx.x * wpb + wib; grp < groups; grp += gridDim.x * wpb) {
        const uint32_t cw = grp * G + slot;
        const bool live = cw < P.rows;
        const uint32_t row = live ? cw : 0u;
        uint8_t* irow = P.info + (size_t)row * P.nb;

        // ---- information dwords (memory order): dword k = r * GL + l in register r ---------------------------
        uint32_t raw[WPL];
#pragma unroll
        for (int r = 0; r < WPL; ++r) {
            const uint32_t k = (uint32_t)r * GL + l;
            uint32_t w = 0;
            if (k < nbw) {
                if constexpr (ALIGNED) {
                    w = reinterpret_cast<const uint32_t*>(irow)[k];
                } else {
#pragma unroll
                    for (uint32_t b = 0; b < 4; ++b)
                        if (4u * k + b < P.nb) w |= (uint32_t)irow[4u * k + b] << (8u * b);
                }
            }
            raw[r] = w;
        }

        // ---- Detector::generate: checksum = XOR of the generator words of the set payload bits ---------------
        if (P.gen4) {
            uint32_t acc = 0;
#pragma unroll
            for (int r = 0; r < WPL; ++r) {
                if ((uint32_t)r * GL < nbw) {  // wave uniform
                    const uint32_t k = (uint32_t)r * GL + l;
                    const uint32_t kb = (k < nbw ? k : 0u) * 512u;  // (lanes past the information word read row 0 of the table with raw = 0: entry 0 is 0)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t nib4 = j == 0 ? (raw[r] << 2) : (raw[r] >> (4 * j - 2));
                        acc ^= *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(P.gen4) +
                                                                  (bit_select(0x3Cu, nib4, kb) + (uint32_t)j * 64u));
                    }
                }
            }
            acc = group_xor<GLL>(acc);
            const uint64_t f64 = (uint64_t)acc << fsh;
            // dword k0 (and k0 + 1 if the field straddles it) is register k0 >> GLL of lane k0 % GL
#pragma unroll
            for (int r = 0; r < WPL; ++r) {
                const bool lo_r = (k0 >> GLL) == (uint32_t)r, hi_r = ((k0 + 1u) >> GLL) == (uint32_t)r && (uint32_t)(fmask >> 32) != 0;
                if (lo_r || hi_r) {  // wave uniform
                    const uint32_t k = (uint32_t)r * GL + l;
                    const bool lo = lo_r && k == k0, hi = hi_r && k == k0 + 1u;
                    if (lo || hi) {
                        const uint32_t m = lo ? (uint32_t)fmask : (uint32_t)(fmask >> 32);
                        const uint32_t f = lo ? (uint32_t)f64 : (uint32_t)(f64 >> 32);
                        raw[r] = bit_select(m, f, raw[r]);
                        if (live) {  // the caller's buffer receives the checksum, as in the reference
                            if constexpr (ALIGNED) {
                                reinterpret_cast<uint32_t*>(irow)[k] = raw[r];
                            } else {
                                for (uint32_t b = 0; b < 4; ++b)
                                    if (((m >> (8u * b)) & 0xFFu) && 4u * k + b < P.nb) irow[4u * k + b] = (uint8_t)(raw[r] >> (8u * b));
                            }
                        }
                    }
                }
            }
        }

        // ---- information bits, LSB first, to LDS so that every lane can fetch its windows ----------------------
#pragma unroll
        for (int r = 0; r < WPL; ++r) {
            const uint32_t k = (uint32_t)r * GL + l;
            if (k < nbw) iw[k] = flip_bytes(raw[r]);
        }
        ENC_WAVE_SYNC();

        // ---- place the information bits: u[w] = deposit(next popcount(mask) bits, mask) ------------------------
        uint32_t v[WPL], mask[WPL];
#pragma unroll
        for (int r = 0; r < WPL; ++r) {
            DepRegs d;
            if constexpr (HOIST) d = dh[r];
            else d = load_dep(P.dep, l * WPL + (uint32_t)r);
            const uint32_t wi = d.base >> 5, sh = d.base & 31u;
            const uint32_t lo = iw[wi], hi = iw[wi + 1];  // (words past the information word are never selected)
            uint32_t x = __builtin_amdgcn_alignbit(hi, lo, sh);
            x = bit_select(d.mv4, x << 16, x);
            x = bit_select(d.mv3, x << 8, x);
            x = bit_select(d.mv2, x << 4, x);
            x = bit_select(d.mv1, x << 2, x);
            x = bit_select(d.mv0, x << 1, x);
            mask[r] = d.mask;
            v[r] = x & d.mask;
        }

        // ---- x = u G: parent = [a ^ b | b] at every stage (butterfly_fip.cpp:15-63) ---------------------------
        auto transform = [&]() {
#pragma unroll
            for (int r = 0; r < WPL; ++r) {
                uint32_t x = v[r];
                x = xor_and(x, x >> 1, 0x55555555u);
                x = xor_and(x, x >> 2, 0x33333333u);
                x = xor_and(x, x >> 4, 0x0F0F0F0Fu);
                if (P.N >= 16) x = xor_and(x, x >> 8, 0x00FF00FFu);
                if (P.N >= 32) x ^= (x >> 16);
                v[r] = x;
            }
#pragma unroll
            for (int hk = 1; hk < WPL; hk <<= 1)
#pragma unroll
                for (int r = 0; r < WPL; ++r)
                    if (!(r & hk)) v[r] ^= v[r | hk];
#pragma unroll
            for (int r = 0; r < WPL; ++r) {
                // (the exchange runs in ALL lanes before the select: a DPP read of a lane that is masked off returns 0)
                if constexpr (GLL > 0) { const uint32_t o = lane_xor<1>(v[r]); v[r] = xor_and(v[r], o, low[0]); }
                if constexpr (GLL > 1) { const uint32_t o = lane_xor<2>(v[r]); v[r] = xor_and(v[r], o, low[1]); }
                if constexpr (GLL > 2) { const uint32_t o = lane_xor<4>(v[r]); v[r] = xor_and(v[r], o, low[2]); }
                if constexpr (GLL > 3) { const uint32_t o = lane_xor<8>(v[r]); v[r] = xor_and(v[r], o, low[3]); }
                if constexpr (GLL > 4) { const uint32_t o = lane_xor<16>(v[r]); v[r] = xor_and(v[r], o, low[4]); }
                if constexpr (GLL > 5) { const uint32_t o = lane_xor<32>(v[r]); v[r] = xor_and(v[r], o, low[5]); }
            }
        };
        transform();
        if (P.systematic) {  // PackedContainer::resetFrozenBits, then the second transform
#pragma unroll
            for (int r = 0; r < WPL; ++r) v[r] &= mask[r];
            transform();
        }

        // ---- PackedContainer::getPackedBits: WPL consecutive dwords of the packed row per lane -----------------
        if (live) {
            uint8_t* crow = P.code + (size_t)cw * code_bytes;
            if (P.N >= 32) {
                uint32_t* dst = reinterpret_cast<uint32_t*>(crow) + l * WPL;
                if constexpr (WPL >= 4) {
#pragma unroll
                    for (int r = 0; r < WPL; r += 4)
                        *reinterpret_cast<uint4*>(dst + r) =
                            make_uint4(flip_bytes(v[r]), flip_bytes(v[r + 1]), flip_bytes(v[r + 2]), flip_bytes(v[r + 3]));
                } else if constexpr (WPL == 2) {
                    *reinterpret_cast<uint2*>(dst) = make_uint2(flip_bytes(v[0]), flip_bytes(v[1]));
                } else {
                    dst[0] = flip_bytes(v[0]);
                }
            } else {
                const uint32_t o = flip_bytes(v[0]);
                for (uint32_t b = 0; b < code_bytes; ++b) crow[b] = (uint8_t)(o >> (8u * b));
            }
        }
        ENC_WAVE_SYNC();  // iw[] is rewritten by the next iteration
    }
}

template <int WPL, int GLL>
static hipError_t launch_k(const EncodeParams& P, unsigned grid, unsigned wpb, hipStream_t st)
{
    const size_t lds = (size_t)wpb * (64u >> GLL) * P.iw_words * sizeof(uint32_t);
    if (P.info_aligned)
        hipLaunchKernelGGL((encode_kernel<WPL, GLL, true>), dim3(grid), dim3(64 * wpb), lds, st, P);
    else
        hipLaunchKernelGGL((encode_kernel<WPL, GLL, false>), dim3(grid), dim3(64 * wpb), lds, st, P);
    return hipGetLastError();
}

unsigned encode_codewords_per_wave(unsigned N)
{
    const unsigned nwords = N >= 32 ? N / 32 : 1;
    const unsigned wpl = nwords >= 512 ? nwords / 64 : (nwords >= 4 ? 4 : nwords);
    return 64 / (nwords / wpl);
}

hipError_t launch_encode(const EncodeParams& P, unsigned grid, unsigned wpb, hipStream_t st)
{
    const uint32_t nwords = P.N >= 32 ? (P.N >> 5) : 1u;
    switch (nwords) {
    case 1: return launch_k<1, 0>(P, grid, wpb, st);
    case 2: return launch_k<2, 0>(P, grid, wpb, st);
    case 4: return launch_k<4, 0>(P, grid, wpb, st);
    case 8: return launch_k<4, 1>(P, grid, wpb, st);
    case 16: return launch_k<4, 2>(P, grid, wpb, st);
    case 32: return launch_k<4, 3>(P, grid, wpb, st);
    case 64: return launch_k<4, 4>(P, grid, wpb, st);
    case 128: return launch_k<4, 5>(P, grid, wpb, st);
    case 256: return launch_k<4, 6>(P, grid, wpb, st);
    case 512: return launch_k<8, 6>(P, grid, wpb, st);
    case 1024: return launch_k<16, 6>(P, grid, wpb, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace pamd
