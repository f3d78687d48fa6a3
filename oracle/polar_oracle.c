/* TEST INFRASTRUCTURE ONLY -- see polar_oracle.h.
 *
 * Plain-C restatement of the reference's float decoders.  Every routine cites the reference
 * file:line it follows (paths relative to /root/reference).  IEEE fp32, every + and - rounds once;
 * compile WITHOUT -ffast-math and with -ffp-contract=off (oracle/Makefile does).
 */
#include "polar_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * fp32 helpers
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t fbits(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    return u;
}
static inline float bitsf(uint32_t u)
{
    float x;
    memcpy(&x, &u, 4);
    return x;
}
static inline unsigned sb(float x) { return fbits(x) >> 31; }
static inline float fabs32(float x) { return bitsf(fbits(x) & 0x7FFFFFFFu); }
/* _mm256_min_ps(a,b): a<b ? a : b (second operand on ties / unordered) */
static inline float minps(float a, float b) { return a < b ? a : b; }
static inline float maxps(float a, float b) { return a > b ? a : b; }

/* _mm256_polarf_ps  include/polarcode/decoding/avx_float.h:55-63 */
static inline float polar_f(float a, float b)
{
    const uint32_t sgn = (fbits(a) ^ fbits(b)) & 0x80000000u;
    const float m = minps(fabs32(a), fabs32(b));
    return bitsf(sgn | fbits(m));
}
/* _mm256_polarg_ps  avx_float.h:71-81 ; u = 0/1 sign bit of the partial sum */
static inline float polar_g(float a, float b, unsigned u)
{
    const float sa = bitsf(fbits(a) ^ (u ? 0x80000000u : 0u));
    return sa + b;
}

void po_f(const float* in, float* out, unsigned h)
{
    for (unsigned i = 0; i < h; ++i) out[i] = polar_f(in[i], in[i + h]);
}
void po_g(const float* in, const float* bits, float* out, unsigned h)
{
    for (unsigned i = 0; i < h; ++i) out[i] = polar_g(in[i], in[i + h], sb(bits[i]));
}

/* fold8: the 8 strided AVX lane accumulators, "acc = setzero; acc += load(x+8k)".
 * Elements >= n (n < 8) are treated as +0 (RepetitionPrepare, avx_float.h:238-243). */
static void fold8(const float* x, unsigned n, float acc[8])
{
    for (unsigned j = 0; j < 8; ++j) acc[j] = 0.0f;
    for (unsigned k = 0; k < n; k += 8)
        for (unsigned j = 0; j < 8; ++j) acc[j] = acc[j] + ((k + j < n) ? x[k + j] : 0.0f);
}
/* reduce_add_ps  include/polarcode/avxconvenience.h:256-272: x0+x1+...+x7 left to right */
static float reduce_add8(const float a[8])
{
    return ((((((a[0] + a[1]) + a[2]) + a[3]) + a[4]) + a[5]) + a[6]) + a[7];
}
static float lanesum(const float* x, unsigned n)
{
    float acc[8];
    fold8(x, n, acc);
    return reduce_add8(acc);
}

/* _mm256_spc_right4_ps  avx_float.h:289-302: flips EVERY position tied for the minimum */
static void spc4(const float in[4], float out[4])
{
    float m = fabs32(in[0]);
    unsigned p = 0;
    for (unsigned i = 0; i < 4; ++i) {
        m = minps(m, fabs32(in[i]));
        p ^= sb(in[i]);
    }
    for (unsigned i = 0; i < 4; ++i) {
        const int tied = fabs32(in[i]) == m;
        out[i] = (p && tied) ? bitsf(fbits(in[i]) ^ 0x80000000u) : in[i];
    }
}

static unsigned firstmin_abs(const float* x, unsigned n)
{
    /* SpcDecoder::decode fastssc_avx_float.cpp:342-373 : per 8-vector first-index minimum
     * (_mm256_minidx_ps avxconvenience.h:371-388), strict '<' across vectors => first index */
    unsigned idx = 0;
    float best = INFINITY;
    for (unsigned i = 0; i < n; ++i) {
        const float a = fabs32(x[i]);
        if (a < best) {
            best = a;
            idx = i;
        }
    }
    return idx;
}

/* splitFrozenBits  src/polarcode/polarcode.cpp:14-34 */
static void split_frozen(const uint32_t* F, unsigned c, unsigned h, uint32_t* FL, unsigned* cl,
                         uint32_t* FR, unsigned* cr)
{
    *cl = *cr = 0;
    for (unsigned i = 0; i < c; ++i) {
        if (F[i] < h)
            FL[(*cl)++] = F[i];
        else
            FR[(*cr)++] = F[i] - h;
    }
}

static int is_prefix(const uint32_t* F, unsigned c)
{
    for (unsigned i = 0; i < c; ++i)
        if (F[i] != i) return 0;
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * Fast-SSC (L = 1)      FastSscAvx::createDecoder  fastssc_avx_float.cpp:797-896
 * bits[] = sign bits (the reference keeps float words whose sign is the bit)
 * ---------------------------------------------------------------------------------------- */
static int g_sc_stats_leaves, g_sc_stats_inner;

static int sc_node(const uint32_t* F, unsigned c, unsigned n, const float* x, uint8_t* bits)
{
    if (c == n) { /* RateZeroDecoder :247 */
        g_sc_stats_leaves++;
        memset(bits, 0, n);
        return 0;
    }
    if (c == 0) { /* RateOneDecoder :257-263 */
        g_sc_stats_leaves++;
        for (unsigned i = 0; i < n; ++i) bits[i] = (uint8_t)sb(x[i]);
        return 0;
    }
    if (c == n - 1) { /* RepetitionDecoder :273-287 */
        g_sc_stats_leaves++;
        const float s = lanesum(x, n);
        memset(bits, (int)sb(s), n);
        return 0;
    }
    if (c == 1) { /* SpcDecoder :342-373 */
        g_sc_stats_leaves++;
        unsigned p = 0;
        for (unsigned i = 0; i < n; ++i) {
            bits[i] = (uint8_t)sb(x[i]);
            p ^= bits[i];
        }
        if (p) bits[firstmin_abs(x, n)] ^= 1;
        return 0;
    }
    if (c == n - 2) { /* DoubleRepetitionDecoder :303-332 (ctor check :820-826) */
        if (!is_prefix(F, c)) return -1;
        g_sc_stats_leaves++;
        float a[8], e, o;
        fold8(x, n, a);
        if (n >= 8) {
            e = (a[0] + a[4]) + (a[2] + a[6]);
            o = (a[1] + a[5]) + (a[3] + a[7]);
        } else {
            e = ((a[0] + a[2]) + a[4]) + a[6];
            o = ((a[1] + a[3]) + a[5]) + a[7];
        }
        for (unsigned i = 0; i < n; i += 2) {
            bits[i] = (uint8_t)sb(e);
            bits[i + 1] = (uint8_t)sb(o);
        }
        return 0;
    }
    if (c == 2 && F[0] == 0 && F[1] == 1) {
        g_sc_stats_leaves++;
        for (unsigned i = 0; i < n; ++i) bits[i] = (uint8_t)sb(x[i]);
        if (n == 8) { /* DoubleSpcDecoderShort8 :473-488 : flip every class member tied for the min */
            for (unsigned cls = 0; cls < 2; ++cls) {
                float m = fabs32(x[cls]);
                unsigned p = 0;
                for (unsigned i = cls; i < 8; i += 2) {
                    m = minps(m, fabs32(x[i]));
                    p ^= sb(x[i]);
                }
                if (p)
                    for (unsigned i = cls; i < 8; i += 2)
                        if (fabs32(x[i]) == m) bits[i] ^= 1;
            }
        } else { /* DoubleSpcDecoder :425-466 (+ _mm256_argabsmin_ps avx_float.h:257-266) */
            float minv[8];
            unsigned mini[8];
            for (unsigned j = 0; j < 8; ++j) {
                minv[j] = 3.402823466e+38f; /* FLT_MAX */
                mini[j] = 0;
            }
            for (unsigned k = 0; k < n; k += 8)
                for (unsigned j = 0; j < 8; ++j) {
                    const float a = fabs32(x[k + j]);
                    if (!(a > minv[j])) { /* '<=' replaces: the LAST row with the lane minimum */
                        minv[j] = a;
                        mini[j] = k + j;
                    }
                }
            for (unsigned cls = 0; cls < 2; ++cls) {
                float m = minv[cls];
                unsigned p = 0;
                for (unsigned j = cls; j < 8; j += 2) m = minps(m, minv[j]);
                for (unsigned i = cls; i < n; i += 2) p ^= sb(x[i]);
                unsigned lane = cls;
                for (unsigned j = cls; j < 8; j += 2)
                    if (minv[j] == m) { /* tzcnt: lowest lane of the class */
                        lane = j;
                        break;
                    }
                if (p) bits[mini[lane]] ^= 1;
            }
        }
        return 0;
    }
    if (c == n - 3 && n > 8 && F[c - 1] == n - 4) { /* TripleRepetitionDecoder :572-589 */
        if (!is_prefix(F, c)) return -1;
        g_sc_stats_leaves++;
        float a[8], t[4], r[4];
        fold8(x, n, a);
        for (unsigned j = 0; j < 4; ++j) t[j] = a[j] + a[j + 4];
        spc4(t, r);
        for (unsigned i = 0; i < n; ++i) bits[i] = (uint8_t)sb(r[i & 3]);
        return 0;
    }
    if (c == n - 4 && F[c - 1] == n - 4 && F[c - 2] == n - 6) { /* TypeFiveDecoder :762-792 */
        g_sc_stats_leaves++;
        float l[8], q[4], gg[4], s[4];
        fold8(x, n, l);
        for (unsigned j = 0; j < 4; ++j) q[j] = polar_f(l[j], l[j + 4]);
        const float rho = (q[0] + q[1]) + (q[2] + q[3]); /* two _mm256_hadd_ps */
        for (unsigned j = 0; j < 4; ++j) gg[j] = polar_g(l[j], l[j + 4], sb(rho));
        spc4(gg, s);
        for (unsigned i = 0; i < n; i += 8)
            for (unsigned j = 0; j < 4; ++j) {
                bits[i + j] = (uint8_t)(sb(rho) ^ sb(s[j]));
                bits[i + 4 + j] = (uint8_t)sb(s[j]);
            }
        return 0;
    }
    if (n == 8 && c == 3 && F[0] == 0 && F[1] == 1 && F[2] == 2) {
        /* RepetitionRateOneDecoderShort8 :718-739 */
        g_sc_stats_leaves++;
        float q[4], s[4];
        for (unsigned j = 0; j < 4; ++j) q[j] = polar_f(x[j], x[j + 4]);
        const float rho = (q[0] + q[1]) + (q[2] + q[3]);
        for (unsigned j = 0; j < 4; ++j) s[j] = polar_g(x[j], x[j + 4], sb(rho));
        for (unsigned j = 0; j < 4; ++j) {
            bits[j] = (uint8_t)(sb(rho) ^ sb(s[j]));
            bits[4 + j] = (uint8_t)sb(s[j]);
        }
        return 0;
    }
    if (n == 8 && c == 5 && F[c - 1] == 4 && F[c - 2] == 3) { /* ZeroSpcDecoderShort8 :556-565 */
        g_sc_stats_leaves++;
        float t[4], r[4];
        for (unsigned j = 0; j < 4; ++j) t[j] = x[j] + x[j + 4];
        spc4(t, r);
        for (unsigned j = 0; j < 4; ++j) bits[j] = bits[4 + j] = (uint8_t)sb(r[j]);
        return 0;
    }

    /* split (:872-895) */
    const unsigned h = n / 2;
    uint32_t* FL = (uint32_t*)malloc(sizeof(uint32_t) * (c + 1) * 2);
    uint32_t* FR = FL + c + 1;
    unsigned cl, cr;
    split_frozen(F, c, h, FL, &cl, FR, &cr);
    float* y = (float*)malloc(sizeof(float) * h);
    int rc = 0;

    if (n > 8 && cl == h && cr == 1) { /* ZeroSpcDecoder :503-546 (outputs `right`) */
        g_sc_stats_leaves++;
        unsigned p = 0;
        for (unsigned i = 0; i < h; ++i) {
            y[i] = x[i] + x[i + h];
            p ^= sb(y[i]);
            bits[i] = bits[i + h] = (uint8_t)sb(x[i + h]);
        }
        if (p) {
            const unsigned m = firstmin_abs(y, h);
            bits[m] ^= 1;
            bits[m + h] ^= 1;
        }
    } else if (n > 8 && cr == 0) { /* ROneNode :198-219 */
        g_sc_stats_inner++;
        po_f(x, y, h);
        rc = sc_node(FL, cl, h, y, bits);
        if (!rc)
            for (unsigned i = 0; i < h; ++i) {
                const float lam = polar_g(x[i], x[i + h], bits[i]);
                bits[i] ^= (uint8_t)sb(lam);
                bits[i + h] = (uint8_t)sb(lam);
            }
    } else if (n > 8 && cl == h) { /* ZeroRNode :232-237 */
        g_sc_stats_inner++;
        for (unsigned i = 0; i < h; ++i) y[i] = x[i] + x[i + h];
        rc = sc_node(FR, cr, h, y, bits + h);
        if (!rc) memcpy(bits, bits + h, h);
    } else { /* RateRNode :148-155 / ShortRateRNode :178-185 */
        g_sc_stats_inner++;
        po_f(x, y, h);
        rc = sc_node(FL, cl, h, y, bits);
        if (!rc) {
            for (unsigned i = 0; i < h; ++i) y[i] = polar_g(x[i], x[i + h], bits[i]);
            rc = sc_node(FR, cr, h, y, bits + h);
        }
        if (!rc)
            for (unsigned i = 0; i < h; ++i) bits[i] ^= bits[i + h];
    }
    free(y);
    free(FL);
    return rc;
}

int po_sc_decode_bits(unsigned N, const uint32_t* frozen, unsigned nF, const float* llr,
                      uint8_t* bits_out)
{
    return sc_node(frozen, nF, N, llr, bits_out);
}

/* ------------------------------------------------------------------------------------------
 * Fast-SSCL      scl_avx_float.cpp
 * ---------------------------------------------------------------------------------------- */

/* ------------------------------------------------------------------------------------------
 * int8 ("char") decoders: FastSscFipChar (src/polarcode/decoding/fastssc_fip_char.cpp) and
 * SclFipChar (scl_fip_char.cpp), primitives include/polarcode/decoding/fip_char.h.
 * The list decoder below reuses the float routine: an int8 LLR is an exactly representable float,
 * saturating int8 arithmetic is add + clamp, and the `long` path metrics (scl_fip_char.cpp:155)
 * stay below 127 * N <= 2^22 in magnitude, i.e. exact in fp32 whatever the summation order.
 * ---------------------------------------------------------------------------------------- */
static inline int8_t sat8(int v) { return (int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v)); }
/* F_function_calc fip_char.h:45-66: magnitudes clamped to [1, 127], sign = xor of the sign bits (never zero) */
static inline int8_t f_char(int8_t a, int8_t b)
{
    const int neg = ((a ^ b) & 0x80) != 0;
    int ma = a < -127 ? -127 : a, mb = b < -127 ? -127 : b;
    ma = ma < 0 ? -ma : ma;
    mb = mb < 0 ? -mb : mb;
    if (ma < 1) ma = 1;
    if (mb < 1) mb = 1;
    const int m = ma < mb ? ma : mb;
    return (int8_t)(neg ? -m : m);
}
/* G_function_calc fip_char.h:68-74: saturating right + left, or right - left where the bit's sign is set */
static inline int8_t g_char(int8_t l, int8_t r, int bit) { return bit ? sat8((int)r - l) : sat8((int)r + l); }
static inline float f_charf(float a, float b) { return (float)f_char((int8_t)a, (int8_t)b); }
static inline float g_charf(float a, float b, unsigned u) { return (float)g_char((int8_t)a, (int8_t)b, (int)u); }
/* char negation as the reference's `x = -x` on a char: -(-128) wraps to -128, -0 stays 0 */
static inline int8_t neg8(int8_t v) { return (int8_t)(uint8_t)(0u - (uint8_t)v); }
/* reduce_adds_epi8(__m256i) avxconvenience.h:92-101: tree of SATURATING adds over the 32 byte lanes */
static int8_t reduce_adds32(const int8_t v[32])
{
    int8_t a[16], b[8], c[4], d[2];
    for (int l = 0; l < 16; ++l) a[l] = sat8((int)v[l] + v[l + 16]);
    for (int l = 0; l < 8; ++l) b[l] = sat8((int)a[l] + a[l + 8]);
    for (int l = 0; l < 4; ++l) c[l] = sat8((int)b[l] + b[l + 4]);
    for (int l = 0; l < 2; ++l) d[l] = sat8((int)c[l] + c[l + 2]);
    return sat8((int)d[0] + d[1]);
}
/* half_reduce_adds_epi8 avxconvenience.h:202-210: every byte ends as the saturating sum of the lanes of its parity */
static void half_reduce_adds32(const int8_t v[32], int8_t out[32])
{
    int8_t a[32], b[32], c[32];
    for (int l = 0; l < 32; ++l) a[l] = sat8((int)v[l] + v[(l + 16) & 31]);
    for (int l = 0; l < 32; ++l) b[l] = sat8((int)a[l] + a[(l & 16) | ((l + 8) & 15)]);
    for (int l = 0; l < 32; ++l) c[l] = sat8((int)b[l] + b[(l & ~7) | ((l + 4) & 7)]);
    for (int l = 0; l < 32; ++l) out[l] = sat8((int)c[l] + c[(l & ~3) | ((l + 2) & 3)]);
}
/* the 32 lane accumulators of the REP / DREP leaves: zero padded single vector, or vector-by-vector saturating sums */
static void lane_sums32(const int8_t* x, unsigned n, int8_t v[32])
{
    for (unsigned l = 0; l < 32; ++l) v[l] = 0;
    if (n <= 32) {
        for (unsigned l = 0; l < n; ++l) v[l] = x[l];
    } else {
        for (unsigned i = 0; i < n; i += 32)
            for (unsigned l = 0; l < 32; ++l) v[l] = sat8((int)v[l] + x[i + l]);
    }
}
/* fi_abs_epi8 read as unsigned bytes (abs(-128) = 128) + minpos_epu8 avxconvenience.cpp:13-56: first index of the minimum */
static unsigned minpos_abs32(const int8_t v[32], int8_t* val)
{
    unsigned best = 0, bv = 256;
    for (unsigned l = 0; l < 32; ++l) {
        const unsigned a = v[l] == -128 ? 128u : (unsigned)(v[l] < 0 ? -v[l] : v[l]);
        if (a < bv) {
            bv = a;
            best = l;
        }
    }
    if (val) *val = (int8_t)(uint8_t)bv;
    return best;
}

static int g_scc_leaves, g_scc_inner;
/* FastSscFip::createDecoder fastssc_fip_char.cpp:496-570 and the decode() bodies :196-494; BYTESPERVECTOR = 32 (AVX2).
 * bits: int8 "soft bits" as the reference keeps them (only their signs matter outside the leaves). */
static void scc_node(const uint32_t* F, unsigned c, unsigned n, const int8_t* x, int8_t* bits)
{
    const unsigned h = n / 2;
    int8_t v[32];
    if (c == n) { /* RateZeroDecoder :196-202 */
        g_scc_leaves++;
        memset(bits, 127, n);
        return;
    }
    if (c == 0) { /* RateOneDecoder :204-209 */
        g_scc_leaves++;
        memcpy(bits, x, n);
        return;
    }
    if (c == n - 1) { /* RepetitionDecoder :219-234 / ShortRepetitionDecoder :259-266 */
        g_scc_leaves++;
        lane_sums32(x, n, v);
        memset(bits, reduce_adds32(v), n);
        return;
    }
    if (c == 1) { /* SpcDecoder :268-297 / ShortSpcDecoder :299-313 */
        g_scc_leaves++;
        unsigned par = 0;
        memcpy(bits, x, n);
        for (unsigned i = 0; i < n; ++i) par ^= (uint8_t)x[i] >> 7;
        if (n <= 32) {
            for (unsigned l = 0; l < 32; ++l) v[l] = l < n ? x[l] : 127; /* SpcPrepare fip_char.h:231-236 */
            if (par) {
                const unsigned k = minpos_abs32(v, NULL);
                if (k < n) bits[k] = neg8(bits[k]);
            }
        } else {
            unsigned minIdx = 0;
            int8_t minAbs = 127, t;
            for (unsigned i = 0; i < n; i += 32) {
                if (minAbs > 0) {
                    const unsigned k = minpos_abs32(x + i, &t);
                    if (t < minAbs) {
                        minIdx = k + i;
                        minAbs = t;
                    }
                }
            }
            if (par) bits[minIdx] = neg8(bits[minIdx]);
        }
        return;
    }
    if (c == n - 2 && n >= 32) { /* DoubleRepetitionDecoder :242-257 (no check WHICH two bits are free) */
        g_scc_leaves++;
        int8_t r[32];
        lane_sums32(x, n, v);
        half_reduce_adds32(v, r);
        for (unsigned i = 0; i < n; ++i) bits[i] = r[i & 31];
        return;
    }
    uint32_t* FL = (uint32_t*)malloc(sizeof(uint32_t) * (c + 1) * 2);
    uint32_t* FR = FL + c + 1;
    unsigned cl, cr;
    split_frozen(F, c, h, FL, &cl, FR, &cr);
    int8_t* child = (int8_t*)malloc(n);
    if (cl == h && cr == 0 && n <= 32) { /* ShortZeroOneDecoder :386-395 */
        g_scc_leaves++;
        for (unsigned i = 0; i < h; ++i) bits[i] = bits[h + i] = sat8((int)x[i] + x[h + i]);
    } else if (cl == h && cr == 1) { /* ZeroSpcDecoder :315-353 / ShortZeroSpcDecoder :355-384 */
        g_scc_leaves++;
        unsigned par = 0;
        for (unsigned i = 0; i < h; ++i) {
            child[i] = sat8((int)x[i] + x[h + i]);
            par ^= (uint8_t)child[i] >> 7;
        }
        if (n <= 32) {
            for (unsigned l = 0; l < 32; ++l) v[l] = l < h ? child[l] : 127;
            if (par) {
                const unsigned k = minpos_abs32(v, NULL);
                if (k < h) child[k] = neg8(child[k]);
            }
            memcpy(bits, child, h);
            memcpy(bits + h, child, h);
        } else {
            unsigned minIdx = 0;
            int8_t minAbs = 127, t;
            for (unsigned i = 0; i < h; i += 32) {
                if (minAbs > 0) {
                    const unsigned k = minpos_abs32(child + i, &t);
                    if (t < minAbs) {
                        minIdx = k + i;
                        minAbs = t;
                    }
                }
            }
            memcpy(bits, child, h);
            memcpy(bits + h, child, h);
            if (par) {
                bits[minIdx] = neg8(bits[minIdx]);
                bits[minIdx + h] = neg8(bits[minIdx + h]);
            }
        }
    } else if (cr == 0) { /* ROneNode :412-434 / ShortROneNode :436-470 */
        g_scc_inner++;
        for (unsigned i = 0; i < h; ++i) child[i] = f_char(x[i], x[h + i]);
        scc_node(FL, cl, h, child, bits);
        for (unsigned i = 0; i < h; ++i) {
            const int8_t o = g_char(x[i], x[h + i], (uint8_t)bits[i] >> 7);
            bits[i] = (int8_t)(bits[i] ^ o);
            bits[h + i] = o;
        }
    } else if (cl == h) { /* ZeroRNode :473-480 / ShortZeroRNode :482-489 */
        g_scc_inner++;
        for (unsigned i = 0; i < h; ++i) child[i] = sat8((int)x[i] + x[h + i]);
        scc_node(FR, cr, h, child, bits + h);
        memcpy(bits, bits + h, h);
    } else { /* RateRNode :397-408 / ShortRateRNode :410 (CombineBitsShort clamps to -127 first: signs unchanged) */
        g_scc_inner++;
        for (unsigned i = 0; i < h; ++i) child[i] = f_char(x[i], x[h + i]);
        scc_node(FL, cl, h, child, bits);
        for (unsigned i = 0; i < h; ++i) child[i] = g_char(x[i], x[h + i], (uint8_t)bits[i] >> 7);
        scc_node(FR, cr, h, child, bits + h);
        for (unsigned i = 0; i < h; ++i) bits[i] = (int8_t)(bits[i] ^ bits[h + i]);
    }
    free(child);
    free(FL);
}

int po_scc_decode_bits(unsigned N, const uint32_t* frozen, unsigned nF, const int8_t* llr, uint8_t* bits_out)
{
    int8_t* b = (int8_t*)malloc(N);
    scc_node(frozen, nF, N, llr, b);
    for (unsigned i = 0; i < N; ++i) bits_out[i] = (uint8_t)b[i] >> 7;
    free(b);
    return 0;
}

typedef struct {
    float* llr;   /* stage s at offset (1<<s), 2N floats            (PathList::Llr)     */
    uint8_t* bit; /* stage s at offset (1<<s), 2N sign bits         (PathList::Bit)     */
    uint8_t* left; /*                                               (PathList::LeftBit) */
    float metric;
} po_path;

typedef struct {
    unsigned N, L, n_paths, stages;
    int chr; /* 1: SclFipChar (int8 LLRs held as exact small floats, see "int8 decoders" below) */
    po_path* cur;
    po_path* nxt;
    /* scratch for leaves */
    float* cand_metric;
    unsigned* cand_idx;
    unsigned (*flips)[4];
    unsigned* nflips;
    float* rep_result;
    float* tmp;
    unsigned* tmpi;
    int leaves, inner;
    float* trace; /* optional: after every leaf (1 + L) floats: path count, metrics */
    int trace_cap, trace_n;
} po_scl;

static void scl_trace(po_scl* d)
{
    if (!d->trace || d->trace_n >= d->trace_cap) return;
    float* t = d->trace + (size_t)d->trace_n * (1 + d->L);
    t[0] = (float)d->n_paths;
    for (unsigned p = 0; p < d->L; ++p) t[1 + p] = p < d->n_paths ? d->cur[p].metric : 0.0f;
    d->trace_n++;
}

/* findWeakLlrs  include/polarcode/arrayfuncs.h:209-231 (swap-mutating selection, strict '<') */
static void find_weak(unsigned* idx, float* val, unsigned size, unsigned n)
{
    for (unsigned i = 0; i < size; ++i) idx[i] = i;
    const unsigned lim = (size - 1 < n) ? size - 1 : n;
    for (unsigned i = 0; i < lim; ++i) {
        unsigned k = i;
        for (unsigned j = i + 1; j < size; ++j)
            if (val[j] < val[k]) k = j;
        float tv = val[i];
        val[i] = val[k];
        val[k] = tv;
        unsigned ti = idx[i];
        idx[i] = idx[k];
        idx[k] = ti;
    }
}
/* simplePartialSortDescending  arrayfuncs.h:161-183 (strict '>') */
static void topk_desc(unsigned* idx, float* val, unsigned n, unsigned size)
{
    for (unsigned i = 0; i < size; ++i) idx[i] = i;
    const unsigned lim = (size - 1 < n) ? size - 1 : n;
    for (unsigned i = 0; i < lim; ++i) {
        unsigned k = i;
        for (unsigned j = i + 1; j < size; ++j)
            if (val[j] > val[k]) k = j;
        float tv = val[i];
        val[i] = val[k];
        val[k] = tv;
        unsigned ti = idx[i];
        idx[i] = idx[k];
        idx[k] = ti;
    }
}

/* PathList::duplicatePath :58-66 -- stages >= stage are shared with the source (copied here) */
static void dup_path(po_scl* d, unsigned dst, unsigned src, unsigned stage)
{
    const size_t off = (size_t)1 << stage, len = ((size_t)2 * d->N) - off;
    memcpy(d->nxt[dst].llr + off, d->cur[src].llr + off, len * sizeof(float));
    memcpy(d->nxt[dst].bit + off, d->cur[src].bit + off, len);
    memcpy(d->nxt[dst].left + off, d->cur[src].left + off, len);
}
static void switch_to_next(po_scl* d, unsigned new_count) /* :94-101 */
{
    po_path* t = d->cur;
    d->cur = d->nxt;
    d->nxt = t;
    d->n_paths = new_count;
}

static void scl_node(po_scl* d, const uint32_t* F, unsigned c, unsigned n, unsigned s)
{
    const unsigned P = d->n_paths;
    const size_t o = (size_t)1 << s;

    if (c == 0) { /* RateOneDecoder::decode :353-413 */
        d->leaves++;
        for (unsigned p = 0; p < P; ++p) {
            const float m = d->cur[p].metric;
            for (unsigned i = 0; i < n; ++i) d->tmp[i] = fabs32(d->cur[p].llr[o + i]);
            if (d->chr) /* abs(max(llr, -127)) scl_fip_char.cpp:413-417 */
                for (unsigned i = 0; i < n; ++i) d->tmp[i] = minps(d->tmp[i], 127.0f);
            find_weak(d->tmpi, d->tmp, n, 2);
            d->cand_metric[p * 4 + 0] = m;
            d->cand_metric[p * 4 + 1] = m - d->tmp[0];
            d->cand_metric[p * 4 + 2] = m - d->tmp[1];
            d->cand_metric[p * 4 + 3] = (m - d->tmp[0]) - d->tmp[1];
            if (d->chr && n == 1) /* scl_fip_char.cpp:455-457 (-0x100000000000: far below every metric) */
                d->cand_metric[p * 4 + 2] = d->cand_metric[p * 4 + 3] = -17592186044416.0f;
            d->nflips[p * 4 + 0] = 0;
            d->nflips[p * 4 + 1] = 1;
            d->flips[p * 4 + 1][0] = d->tmpi[0];
            d->nflips[p * 4 + 2] = 1;
            d->flips[p * 4 + 2][0] = d->tmpi[1];
            d->nflips[p * 4 + 3] = 2;
            d->flips[p * 4 + 3][0] = d->tmpi[0];
            d->flips[p * 4 + 3][1] = d->tmpi[1];
        }
        const unsigned np = (P * 4 < d->L) ? P * 4 : d->L;
        topk_desc(d->cand_idx, d->cand_metric, np, P * 4);
        for (unsigned j = 0; j < np; ++j) {
            const unsigned src = d->cand_idx[j];
            dup_path(d, j, src / 4, s);
            d->nxt[j].metric = d->cand_metric[j];
            for (unsigned i = 0; i < n; ++i) d->nxt[j].bit[o + i] = (uint8_t)sb(d->nxt[j].llr[o + i]);
            for (unsigned k = 0; k < d->nflips[src]; ++k) d->nxt[j].bit[o + d->flips[src][k]] ^= 1;
        }
        switch_to_next(d, np);
        scl_trace(d);
        return;
    }
    if (c == n) { /* RateZeroDecoder::decode :316-337 */
        d->leaves++;
        for (unsigned p = 0; p < P; ++p) {
            for (unsigned i = 0; i < n; ++i) {
                d->tmp[i] = minps(d->cur[p].llr[o + i], 0.0f);
                d->cur[p].bit[o + i] = 0;
            }
            d->cur[p].metric = d->cur[p].metric + lanesum(d->tmp, n);
        }
        scl_trace(d);
        return;
    }
    if (c == n - 1 && (n < 8 || d->chr)) { /* RepetitionDecoder::decode :428-481; SclFip: at every size, scl_fip_char.cpp:762-764 */
        d->leaves++;
        for (unsigned p = 0; p < P; ++p) {
            const float m = d->cur[p].metric;
            const float* x = d->cur[p].llr + o;
            if (d->chr) { /* scl_fip_char.cpp:497-547: exact sums; candidate 0 is the non-negative result, 1 its complement */
                float zs = 0.0f, os = 0.0f;
                for (unsigned i = 0; i < n; ++i) {
                    zs += minps(x[i], 0.0f);
                    os += maxps(x[i], 0.0f);
                }
                d->rep_result[p * 2 + 0] = 1.0f;
                d->rep_result[p * 2 + 1] = -1.0f;
                d->cand_metric[p * 2 + 0] = m + zs;
                d->cand_metric[p * 2 + 1] = m - os;
                continue;
            }
            float z[8], on[8];
            for (unsigned i = 0; i < n; ++i) {
                z[i] = minps(x[i], 0.0f);
                on[i] = maxps(x[i], 0.0f);
            }
            const float r = fabs32(lanesum(x, n));
            d->rep_result[p * 2 + 0] = r;
            d->rep_result[p * 2 + 1] = -r;
            d->cand_metric[p * 2 + 0] = m + lanesum(z, n);
            d->cand_metric[p * 2 + 1] = m - lanesum(on, n);
        }
        const unsigned np = (P * 2 < d->L) ? P * 2 : d->L;
        topk_desc(d->cand_idx, d->cand_metric, np, P * 2);
        for (unsigned j = 0; j < np; ++j) {
            const unsigned src = d->cand_idx[j];
            dup_path(d, j, src / 2, s);
            d->nxt[j].metric = d->cand_metric[j];
            memset(d->nxt[j].bit + o, (int)sb(d->rep_result[src]), n);
        }
        switch_to_next(d, np);
        scl_trace(d);
        return;
    }
    if (c == 1) { /* SpcDecoder::decode :498-621 */
        d->leaves++;
        for (unsigned p = 0; p < P; ++p) {
            float m = d->cur[p].metric;
            const float* x = d->cur[p].llr + o;
            unsigned par = 0;
            for (unsigned i = 0; i < n; ++i) {
                par ^= sb(x[i]);
                d->tmp[i] = fabs32(x[i]);
                if (d->chr) d->tmp[i] = minps(d->tmp[i], 127.0f); /* scl_fip_char.cpp:587-590 */
            }
            find_weak(d->tmpi, d->tmp, n, 4);
            const float* v = d->tmp;
            const unsigned* ix = d->tmpi;
            unsigned(*fl)[4] = &d->flips[p * 8];
            unsigned* nf = &d->nflips[p * 8];
            float pinv;
            memset(nf, 0, 8 * sizeof(unsigned));
            if (par) {
                pinv = 0.0f;
                m = m - v[0];
                fl[0][nf[0]++] = ix[0];
                fl[4][nf[4]++] = ix[0];
                fl[5][nf[5]++] = ix[0];
                fl[6][nf[6]++] = ix[0];
            } else {
                pinv = 1.0f;
                fl[1][nf[1]++] = ix[0];
                fl[2][nf[2]++] = ix[0];
                fl[3][nf[3]++] = ix[0];
                fl[7][nf[7]++] = ix[0];
            }
            float* cm = &d->cand_metric[p * 8];
            cm[0] = m;
            cm[1] = (m - pinv * v[0]) - v[1];
            cm[2] = (m - pinv * v[0]) - v[2];
            cm[3] = (m - pinv * v[0]) - v[3];
            cm[4] = (m - v[1]) - v[2];
            cm[5] = (m - v[1]) - v[3];
            cm[6] = (m - v[2]) - v[3];
            cm[7] = (((m - pinv * v[0]) - v[1]) - v[2]) - v[3];
            fl[1][nf[1]++] = ix[1];
            fl[2][nf[2]++] = ix[2];
            fl[3][nf[3]++] = ix[3];
            fl[4][nf[4]++] = ix[1];
            fl[4][nf[4]++] = ix[2];
            fl[5][nf[5]++] = ix[1];
            fl[5][nf[5]++] = ix[3];
            fl[6][nf[6]++] = ix[2];
            fl[6][nf[6]++] = ix[3];
            fl[7][nf[7]++] = ix[1];
            fl[7][nf[7]++] = ix[2];
            fl[7][nf[7]++] = ix[3];
        }
        const unsigned np = (P * 8 < d->L) ? P * 8 : d->L;
        topk_desc(d->cand_idx, d->cand_metric, np, P * 8);
        for (unsigned j = 0; j < np; ++j) {
            const unsigned src = d->cand_idx[j];
            dup_path(d, j, src / 8, s);
            d->nxt[j].metric = d->cand_metric[j];
            for (unsigned i = 0; i < n; ++i) d->nxt[j].bit[o + i] = (uint8_t)sb(d->nxt[j].llr[o + i]);
            for (unsigned k = 0; k < d->nflips[src]; ++k) d->nxt[j].bit[o + d->flips[src][k]] ^= 1;
        }
        switch_to_next(d, np);
        scl_trace(d);
        return;
    }

    /* RateRNode::decode :229-263 / ShortRateRNode::decode :273-307 */
    d->inner++;
    const unsigned h = n / 2, t = s - 1;
    const size_t ot = (size_t)1 << t;
    uint32_t* FL = (uint32_t*)malloc(sizeof(uint32_t) * (c + 1) * 2);
    uint32_t* FR = FL + c + 1;
    unsigned cl, cr;
    split_frozen(F, c, h, FL, &cl, FR, &cr);

    for (unsigned p = 0; p < d->n_paths; ++p) {
        if (d->chr)
            for (unsigned i = 0; i < h; ++i) d->cur[p].llr[ot + i] = f_charf(d->cur[p].llr[o + i], d->cur[p].llr[o + h + i]);
        else
            po_f(d->cur[p].llr + o, d->cur[p].llr + ot, h);
    }
    scl_node(d, FL, cl, h, t);
    for (unsigned p = 0; p < d->n_paths; ++p) { /* prepareRightDecoding :145-151 + G_function */
        po_path* q = &d->cur[p];
        for (unsigned i = 0; i < h; ++i) {
            q->left[ot + i] = q->bit[ot + i];
            q->llr[ot + i] = d->chr ? g_charf(q->llr[o + i], q->llr[o + h + i], q->left[ot + i])
                                    : polar_g(q->llr[o + i], q->llr[o + h + i], q->left[ot + i]);
        }
    }
    scl_node(d, FR, cr, h, t);
    for (unsigned p = 0; p < d->n_paths; ++p) { /* CombineBitsLong/Short avx_float.h:206-236 */
        po_path* q = &d->cur[p];
        for (unsigned i = 0; i < h; ++i) {
            q->bit[o + i] = q->left[ot + i] ^ q->bit[ot + i];
            q->bit[o + h + i] = q->bit[ot + i];
        }
    }
    free(FL);
}

static po_scl* scl_new(unsigned N, unsigned L)
{
    po_scl* d = (po_scl*)calloc(1, sizeof(po_scl));
    d->N = N;
    d->L = L;
    d->cur = (po_path*)calloc(L, sizeof(po_path));
    d->nxt = (po_path*)calloc(L, sizeof(po_path));
    for (unsigned p = 0; p < L; ++p) {
        d->cur[p].llr = (float*)calloc(2 * (size_t)N, sizeof(float));
        d->nxt[p].llr = (float*)calloc(2 * (size_t)N, sizeof(float));
        d->cur[p].bit = (uint8_t*)calloc(2 * (size_t)N, 1);
        d->nxt[p].bit = (uint8_t*)calloc(2 * (size_t)N, 1);
        d->cur[p].left = (uint8_t*)calloc(2 * (size_t)N, 1);
        d->nxt[p].left = (uint8_t*)calloc(2 * (size_t)N, 1);
    }
    d->cand_metric = (float*)calloc(8 * (size_t)L, sizeof(float));
    d->cand_idx = (unsigned*)calloc(8 * (size_t)L, sizeof(unsigned));
    d->flips = (unsigned(*)[4])calloc(8 * (size_t)L, sizeof(unsigned[4]));
    d->nflips = (unsigned*)calloc(8 * (size_t)L, sizeof(unsigned));
    d->rep_result = (float*)calloc(2 * (size_t)L, sizeof(float));
    d->tmp = (float*)calloc(N < 8 ? 8 : N, sizeof(float));
    d->tmpi = (unsigned*)calloc(N < 8 ? 8 : N, sizeof(unsigned));
    return d;
}
static void scl_free(po_scl* d)
{
    for (unsigned p = 0; p < d->L; ++p) {
        free(d->cur[p].llr);
        free(d->nxt[p].llr);
        free(d->cur[p].bit);
        free(d->nxt[p].bit);
        free(d->cur[p].left);
        free(d->nxt[p].left);
    }
    free(d->cur);
    free(d->nxt);
    free(d->cand_metric);
    free(d->cand_idx);
    free(d->flips);
    free(d->nflips);
    free(d->rep_result);
    free(d->tmp);
    free(d->tmpi);
    free(d);
}

static unsigned ilog2(unsigned n)
{
    unsigned s = 0;
    while ((1u << s) < n) ++s;
    return s;
}

static int scl_run(po_scl* d, const uint32_t* frozen, unsigned nF, const float* llr)
{
    const unsigned s = ilog2(d->N);
    /* makeInitialPathList :705-709 -- ONE path, metric 0 (fresh decoder) */
    d->n_paths = 1;
    d->cur[0].metric = 0.0f;
    memcpy(d->cur[0].llr + d->N, llr, sizeof(float) * d->N);
    scl_node(d, frozen, nF, d->N, s);
    return (int)d->n_paths;
}

int po_scl_decode_paths(unsigned N, unsigned L, const uint32_t* frozen, unsigned nF,
                        const float* llr, uint8_t* bits_out, float* metrics)
{
    po_scl* d = scl_new(N, L);
    const int P = scl_run(d, frozen, nF, llr);
    for (int p = 0; p < P; ++p) {
        memcpy(bits_out + (size_t)p * N, d->cur[p].bit + N, N);
        metrics[p] = d->cur[p].metric;
    }
    scl_free(d);
    return P;
}

int po_scl_trace(unsigned N, unsigned L, const uint32_t* frozen, unsigned nF, const float* llr,
                 float* trace, int cap)
{
    po_scl* d = scl_new(N, L);
    d->trace = trace;
    d->trace_cap = cap;
    scl_run(d, frozen, nF, llr);
    const int n = d->trace_n;
    scl_free(d);
    return n;
}

int po_tree_stats(unsigned N, const uint32_t* frozen, unsigned nF, int list, unsigned counts[2])
{
    float* z = (float*)calloc(N, sizeof(float));
    int rc = 0;
    if (list) {
        po_scl* d = scl_new(N, 1);
        scl_run(d, frozen, nF, z);
        counts[0] = (unsigned)d->leaves;
        counts[1] = (unsigned)d->inner;
        scl_free(d);
    } else {
        uint8_t* b = (uint8_t*)calloc(N, 1);
        g_sc_stats_leaves = g_sc_stats_inner = 0;
        rc = sc_node(frozen, nF, N, z, b);
        counts[0] = (unsigned)g_sc_stats_leaves;
        counts[1] = (unsigned)g_sc_stats_inner;
        free(b);
    }
    free(z);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Error detection   src/polarcode/errordetection/
 * ---------------------------------------------------------------------------------------- */
/* CRC++ CalculateBits on whole bytes (CRC.h:613-645), MSB-first, init 0, no reflection, no xorout:
 * CRC_24_NRC poly 0xB2B117 (CRC.h:2155-2161), CRC_11_NR poly 0x621 (CRC.h:1513-1519) */
static uint32_t crc_msb(const uint8_t* d, int bytes, unsigned width, uint32_t poly)
{
    uint32_t r = 0;
    const uint32_t top = 1u << (width - 1), mask = (width == 32) ? 0xFFFFFFFFu : ((1u << width) - 1);
    for (int i = 0; i < bytes; ++i)
        for (int b = 7; b >= 0; --b) {
            const uint32_t in = (d[i] >> b) & 1u;
            const uint32_t fb = ((r & top) ? 1u : 0u) ^ in;
            r = (r << 1) & mask;
            if (fb) r ^= poly;
        }
    return r & mask;
}
/* CRC32::gen crc32.cpp:33-41: _mm_crc32_u32 over little-endian words = reflected CRC-32C
 * (poly 0x1EDC6F41, reflected 0x82F63B78), init 0, no final xor */
static uint32_t crc32c_words(const uint8_t* d, int words)
{
    uint32_t r = 0;
    for (int i = 0; i < words * 4; ++i) {
        r ^= d[i];
        for (int b = 0; b < 8; ++b) r = (r >> 1) ^ ((r & 1u) ? 0x82F63B78u : 0u);
    }
    return r;
}

uint32_t po_det_calc(int det, const uint8_t* data, int bytes)
{
    switch (det) {
    case PO_DET_CRC8: return crc_msb(data, bytes, 8, 0x07); /* crc8.cpp:18-47 */
    case PO_DET_CRC11NR: return crc_msb(data, bytes, 11, 0x621);
    case PO_DET_CRC24NRC: return crc_msb(data, bytes, 24, 0xB2B117);
    case PO_DET_CRC32: return crc32c_words(data, bytes / 4);
    default: return 0;
    }
}

int po_det_check(int det, const uint8_t* p, int bytes)
{
    switch (det) {
    case PO_DET_DUMMY: return 1; /* dummy.cpp:25 */
    case PO_DET_CRC8: /* crc8.cpp:49-54 */
        return po_det_calc(det, p, bytes - 1) == p[bytes - 1];
    case PO_DET_CRC11NR: { /* crc11nr.cpp:33-40: 11-bit CRC right-aligned in a 16-bit BE field */
        const uint32_t rx = ((uint32_t)p[bytes - 2] << 8) | p[bytes - 1];
        return po_det_calc(det, p, bytes - 2) == rx;
    }
    case PO_DET_CRC24NRC: { /* crc24nrc.cpp:34-42 */
        const uint32_t rx = ((uint32_t)p[bytes - 3] << 16) | ((uint32_t)p[bytes - 2] << 8) | p[bytes - 1];
        return po_det_calc(det, p, bytes - 3) == rx;
    }
    case PO_DET_CRC32: { /* crc32.cpp:43-57: native (LE) uint32 in the last word */
        const int w = (bytes >> 2) - 1;
        uint32_t rx;
        memcpy(&rx, p + 4 * w, 4);
        return crc32c_words(p, w) == rx;
    }
    default: return 0;
    }
}

void po_det_generate(int det, uint8_t* p, int bytes)
{
    uint32_t c;
    switch (det) {
    case PO_DET_CRC8:
        p[bytes - 1] = (uint8_t)po_det_calc(det, p, bytes - 1);
        break;
    case PO_DET_CRC11NR:
        c = po_det_calc(det, p, bytes - 2);
        p[bytes - 2] = (uint8_t)(c >> 8);
        p[bytes - 1] = (uint8_t)c;
        break;
    case PO_DET_CRC24NRC:
        c = po_det_calc(det, p, bytes - 3);
        p[bytes - 3] = (uint8_t)(c >> 16);
        p[bytes - 2] = (uint8_t)(c >> 8);
        p[bytes - 1] = (uint8_t)c;
        break;
    case PO_DET_CRC32: {
        const int w = (bytes / 4) - 1;
        c = crc32c_words(p, w);
        memcpy(p + 4 * w, &c, 4);
        break;
    }
    default: break;
    }
}

/* ------------------------------------------------------------------------------------------
 * output stage
 * ---------------------------------------------------------------------------------------- */
/* polar transform x = u G, natural order, parent = [a^b | b]  (butterfly_fip.cpp:15-63) */
static void transform_bits(uint8_t* b, unsigned N)
{
    for (unsigned h = 1; h < N; h <<= 1)
        for (unsigned o = 0; o < N; o += 2 * h)
            for (unsigned i = 0; i < h; ++i) b[o + i] ^= b[o + h + i];
}

/* FloatContainer::getPackedInformationBits bitcontainer.cpp:225-292 (LUT :68-84): sign bits at the
 * non-frozen indices, ascending, MSB first */
static void pack_info(const uint8_t* bits, unsigned N, const uint32_t* F, unsigned nF, uint8_t* out)
{
    const unsigned K = N - nF, nb = (K + 7) / 8;
    memset(out, 0, nb);
    unsigned fc = 0, t = 0;
    for (unsigned i = 0; i < N; ++i) {
        if (!(fc == nF || F[fc] > i)) { /* calculateLUT: frozen (or out-of-order entry) -> skip */
            fc++;
            continue;
        }
        if (bits[i]) out[t >> 3] |= (uint8_t)(0x80u >> (t & 7));
        ++t;
    }
}

static void extract(const uint8_t* bits, unsigned N, const uint32_t* F, unsigned nF, int systematic,
                    uint8_t* scratch, uint8_t* out)
{
    if (systematic) {
        pack_info(bits, N, F, nF, out);
    } else { /* re-encode: fastssc_avx_float.cpp:944-948 / scl_avx_float.cpp:731-746 */
        memcpy(scratch, bits, N);
        transform_bits(scratch, N);
        pack_info(scratch, N, F, nF, out);
    }
}

static int sc_decode_row(unsigned N, const uint32_t* F, unsigned nF, int det, int systematic,
                         const float* llr, uint8_t* bits, uint8_t* scratch, uint8_t* out, uint8_t* ok)
{
    /* FastSscAvxFloat::decode :940-955 */
    const unsigned nb = (N - nF + 7) / 8;
    if (po_sc_decode_bits(N, F, nF, llr, bits)) return -1;
    extract(bits, N, F, nF, systematic, scratch, out);
    *ok = (uint8_t)po_det_check(det, out, (int)nb);
    return 0;
}

static void scl_decode_row(po_scl* d, const uint32_t* F, unsigned nF, int det, int systematic,
                           const float* llr, uint8_t* bits, uint8_t* scratch, uint8_t* out,
                           uint8_t* ok, float* metrics)
{
    /* SclAvxFloat::decode :696-703 + extractBestPath :711-750 */
    const unsigned N = d->N, nb = (N - nF + 7) / 8;
    const int P = scl_run(d, F, nF, llr);
    int success = 0;
    for (int p = 0; p < P && !success; ++p) {
        extract(d->cur[p].bit + N, N, F, nF, systematic, scratch, out);
        if (po_det_check(det, out, (int)nb)) success = 1;
    }
    if (!success) extract(d->cur[0].bit + N, N, F, nF, systematic, scratch, out);
    *ok = (uint8_t)success;
    if (metrics) {
        for (unsigned p = 0; p < d->L; ++p) metrics[p] = ((int)p < P) ? d->cur[p].metric : 0.0f;
    }
    (void)bits;
}

int po_decode_batch(unsigned N, unsigned L, const uint32_t* frozen, unsigned nF, int det,
                    int systematic, int dectype, const float* llr, size_t rows, uint8_t* out,
                    uint8_t* ok, float* metrics)
{
    const unsigned nb = (N - nF + 7) / 8;
    uint8_t* bits = (uint8_t*)malloc(N);
    uint8_t* scratch = (uint8_t*)malloc(N);
    uint8_t* tmp = (uint8_t*)calloc(nb + 8, 1);
    po_scl* d = (L > 1) ? scl_new(N, L) : NULL;
    int8_t* row8 = (int8_t*)malloc(N);
    int rc = 0;
    if (d) d->chr = dectype == 3;
    for (size_t r = 0; r < rows && !rc; ++r) {
        const float* x = llr + r * N;
        uint8_t* o = out + r * nb;
        float* m = metrics ? metrics + r * L : NULL;
        if (m) memset(m, 0, sizeof(float) * L);
        if (dectype == 3 && L <= 1) { /* FastSscFipChar::decode fastssc_fip_char.cpp:617-631; the row holds int8 values */
            for (unsigned i = 0; i < N; ++i) row8[i] = (int8_t)x[i];
            po_scc_decode_bits(N, frozen, nF, row8, bits);
            extract(bits, N, frozen, nF, systematic, scratch, tmp);
            ok[r] = (uint8_t)po_det_check(det, tmp, (int)nb);
            memcpy(o, tmp, nb);
        } else if (L <= 1) {
            rc = sc_decode_row(N, frozen, nF, det, systematic, x, bits, scratch, tmp, &ok[r]);
            memcpy(o, tmp, nb);
        } else if (dectype == 2) { /* AdaptiveFloat::decode adaptive_float.cpp:33-45 */
            rc = sc_decode_row(N, frozen, nF, det, systematic, x, bits, scratch, tmp, &ok[r]);
            if (!rc && !ok[r]) scl_decode_row(d, frozen, nF, det, systematic, x, bits, scratch, tmp, &ok[r], m);
            memcpy(o, tmp, nb);
        } else {
            scl_decode_row(d, frozen, nF, det, systematic, x, bits, scratch, tmp, &ok[r], m);
            memcpy(o, tmp, nb);
        }
    }
    if (d) scl_free(d);
    free(row8);
    free(bits);
    free(scratch);
    free(tmp);
    return rc;
}

void po_encode(unsigned N, const uint32_t* frozen, unsigned nF, int det, int systematic,
               uint8_t* info, uint8_t* codeword)
{
    /* ButterflyFipPacked::encode butterfly_fip_packed.cpp:45-59 */
    const unsigned K = N - nF, nb = (K + 7) / 8;
    uint8_t* u = (uint8_t*)calloc(N, 1);
    po_det_generate(det, info, (int)nb);
    unsigned fc = 0, t = 0;
    for (unsigned i = 0; i < N; ++i) {
        if (fc < nF && frozen[fc] == i) {
            fc++;
            continue;
        }
        u[i] = (info[t >> 3] >> (7 - (t & 7))) & 1u;
        ++t;
    }
    transform_bits(u, N);
    if (systematic) {
        for (unsigned i = 0; i < nF; ++i) u[frozen[i]] = 0;
        transform_bits(u, N);
    }
    memset(codeword, 0, N / 8);
    for (unsigned i = 0; i < N; ++i)
        if (u[i]) codeword[i >> 3] |= (uint8_t)(0x80u >> (i & 7));
    free(u);
}

/* ------------------------------------------------------------------------------------------
 * Puncturer   src/polarcode/puncturer.cpp, include/polarcode/puncturer.h
 * ---------------------------------------------------------------------------------------- */
int po_punc_positions(unsigned M, const uint32_t* frozen, unsigned nF, uint32_t* positions, unsigned* parent)
{
    unsigned N = 1; /* round_up_power_of_two puncturer.cpp:24-34 */
    while (N < M) N <<= 1;
    if (parent) *parent = N;
    const unsigned np = N - M; /* Puncturer::Puncturer :51-65 */
    if (np > nF) return -1;
    /* inverse_set_difference :36-49: ascending positions not among the first np frozen ones (sorted input) */
    unsigned f = 0, k = 0;
    for (unsigned i = 0; i < N; ++i) {
        if (f < np && frozen[f] == i) {
            ++f;
            continue;
        }
        positions[k++] = i;
    }
    return k == M ? 0 : -1;
}

void po_puncture_f32(const uint32_t* positions, unsigned M, const float* in, float* out)
{
    for (unsigned i = 0; i < M; ++i) out[i] = in[positions[i]]; /* puncturer.h:66-74 */
}

void po_depuncture_f32(const uint32_t* positions, unsigned M, unsigned N, const float* in, float* out)
{
    for (unsigned i = 0; i < N; ++i) out[i] = 0.0f; /* puncturer.h:89-96 */
    for (unsigned i = 0; i < M; ++i) out[positions[i]] = in[i];
}

void po_puncture_packed(const uint32_t* positions, unsigned M, const uint8_t* in, uint8_t* out)
{
    for (unsigned b = 0; b < M / 8; ++b) { /* puncturer.cpp:69-89 */
        uint8_t o = 0;
        for (unsigned i = 0; i < 8; ++i) {
            const unsigned p = positions[8 * b + i];
            if ((in[p >> 3] >> (7 - (p & 7))) & 1u) o |= (uint8_t)(0x80u >> i);
        }
        out[b] = o;
    }
}
