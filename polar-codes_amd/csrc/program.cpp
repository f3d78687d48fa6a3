// Host-side compiler: frozen set -> flat decode program (see program.h), and the linear
// syndrome tables of the error detectors.
#include "program.h"

#include <algorithm>
#include <cstring>

namespace pamd {

namespace {

unsigned ilog2(unsigned n)
{
    unsigned s = 0;
    while ((1u << s) < n) ++s;
    return s;
}

// splitFrozenBits  src/polarcode/polarcode.cpp:14-34
void split(const std::vector<uint32_t>& F, unsigned h, std::vector<uint32_t>& L,
           std::vector<uint32_t>& R)
{
    L.clear();
    R.clear();
    for (uint32_t x : F) {
        if (x < h)
            L.push_back(x);
        else
            R.push_back(x - h);
    }
}

bool is_prefix(const std::vector<uint32_t>& F)
{
    for (size_t i = 0; i < F.size(); ++i)
        if (F[i] != i) return false;
    return true;
}

std::string fmt_set(const std::vector<uint32_t>& F)
{
    std::string s = "[";
    for (size_t i = 0; i < F.size(); ++i) {
        if (i) s += ", ";
        s += std::to_string(F[i]);
    }
    return s + "]";
}

void emit(Program& p, Op op, unsigned stage, unsigned off, unsigned side = 0)
{
    Ins i;
    i.op = (uint8_t)op;
    i.stage = (uint8_t)stage;
    i.side = (uint8_t)side;
    i.pad = 0;
    i.off = off;
    p.ins.push_back(i);
}

// FastSscAvx::createDecoder  fastssc_avx_float.cpp:797-896 (same order of tests)
std::string sc_build(const std::vector<uint32_t>& F, unsigned n, unsigned off, Program& p)
{
    const unsigned c = (unsigned)F.size();
    const unsigned s = ilog2(n);
    auto leaf = [&](Op op) {
        emit(p, op, s, off);
        p.leaves++;
        return std::string();
    };
    if (c == n) return leaf(SC_R0);
    if (c == 0) return leaf(SC_R1);
    if (c == n - 1) return leaf(SC_REP);
    if (c == 1) return leaf(SC_SPC);
    if (c == n - 2) {
        if (!is_prefix(F)) return fmt_set(F);  // :820-826 throws std::invalid_argument
        return leaf(SC_DREP);
    }
    if (c == 2 && F[0] == 0 && F[1] == 1) return leaf(n == 8 ? SC_DSPC8 : SC_DSPC);
    if (c == n - 3 && n > 8 && F[c - 1] == n - 4) {
        if (!is_prefix(F)) return fmt_set(F);  // :837-844
        return leaf(SC_TREP);
    }
    if (c == n - 4 && F[c - 1] == n - 4 && F[c - 2] == n - 6) return leaf(SC_TYPE5);
    if (n == 8 && c == 3 && F[0] == 0 && F[1] == 1 && F[2] == 2) return leaf(SC_REPONE8);
    if (n == 8 && c == 5 && F[c - 1] == 4 && F[c - 2] == 3) return leaf(SC_ZSPC8);
    // (n == 8 otherwise: the reference prints a warning and carries on)

    const unsigned h = n / 2;
    std::vector<uint32_t> FL, FR;
    split(F, h, FL, FR);
    std::string e;
    if (n > 8 && FL.size() == h && FR.size() == 1) return leaf(SC_ZEROSPC);
    p.inner++;
    if (n > 8 && FR.empty()) {  // ROneNode
        emit(p, SC_F, s, off);
        if (!(e = sc_build(FL, h, off, p)).empty()) return e;
        emit(p, SC_RONE, s, off);
        return e;
    }
    if (n > 8 && FL.size() == h) {  // ZeroRNode
        emit(p, SC_G0R, s, off);
        if (!(e = sc_build(FR, h, off + h, p)).empty()) return e;
        emit(p, SC_COMBINE0R, s, off);
        return e;
    }
    emit(p, SC_F, s, off);  // RateRNode / ShortRateRNode
    if (!(e = sc_build(FL, h, off, p)).empty()) return e;
    emit(p, SC_G, s, off);
    if (!(e = sc_build(FR, h, off + h, p)).empty()) return e;
    emit(p, SC_COMBINE, s, off);
    return e;
}

// FastSscFip::createDecoder  fastssc_fip_char.cpp:496-570 (same order of tests; BYTESPERVECTOR = 32).  The Short*
// node classes compute what their long twins compute, so one op serves both; ShortZeroOneDecoder (left half frozen,
// right half rate 1, n <= 32) is the ZeroR form with a rate-1 child: sat(a + b) copied to both halves.
void scc_build(const std::vector<uint32_t>& F, unsigned n, unsigned off, Program& p)
{
    const unsigned c = (unsigned)F.size();
    const unsigned s = ilog2(n);
    auto leaf = [&](Op op) {
        emit(p, op, s, off);
        p.leaves++;
    };
    if (c == n) return leaf(SC_R0);
    if (c == 0) return leaf(SC_R1);
    if (c == n - 1) return leaf(SCC_REP);
    if (c == 1) return leaf(SCC_SPC);
    if (c == n - 2 && n >= 32) return leaf(SCC_DREP);
    const unsigned h = n / 2;
    std::vector<uint32_t> FL, FR;
    split(F, h, FL, FR);
    if (FL.size() == h && FR.size() == 1) return leaf(SCC_ZEROSPC);
    p.inner++;
    if (FL.size() == h) {  // ShortZeroOneDecoder, ZeroRNode / ShortZeroRNode
        emit(p, SC_G0R, s, off);
        scc_build(FR, h, off + h, p);
        emit(p, SC_COMBINE0R, s, off);
        return;
    }
    if (FR.empty()) {  // ROneNode / ShortROneNode
        emit(p, SC_F, s, off);
        scc_build(FL, h, off, p);
        emit(p, SC_RONE, s, off);
        return;
    }
    emit(p, SC_F, s, off);  // RateRNode / ShortRateRNode
    scc_build(FL, h, off, p);
    emit(p, SC_G, s, off);
    scc_build(FR, h, off + h, p);
    emit(p, SC_COMBINE, s, off);
}

// SclAvx::createDecoder  scl_avx_float.cpp:624-651; chr: SclFip::createDecoder scl_fip_char.cpp:745-775
void scl_build(const std::vector<uint32_t>& F, unsigned n, unsigned side, Program& p, bool chr)
{
    const unsigned c = (unsigned)F.size();
    const unsigned s = ilog2(n);
    auto leaf = [&](Op op) {
        emit(p, op, s, 0, side);
        p.leaves++;
    };
    if (c == 0) return leaf(SCL_R1);
    if (c == n) return leaf(SCL_R0);
    if (c == n - 1 && (n < 8 || chr)) return leaf(SCL_REP);
    if (c == 1) return leaf(SCL_SPC);
    p.inner++;
    const unsigned h = n / 2;
    std::vector<uint32_t> FL, FR;
    split(F, h, FL, FR);
    emit(p, SCL_F, s, 0, side);
    scl_build(FL, h, 0, p, chr);
    emit(p, SCL_G, s, 0, side);
    scl_build(FR, h, 1, p, chr);
    emit(p, SCL_COMBINE, s, 0, side);
}

}  // namespace

std::string compile_sc(unsigned N, const std::vector<uint32_t>& frozen, Program& out, bool chr)
{
    out = Program();
    std::string e;
    if (chr)
        scc_build(frozen, N, 0, out);
    else
        e = sc_build(frozen, N, 0, out);
    emit(out, OP_END, 0, 0);
    return e;
}

std::string compile_scl(unsigned N, const std::vector<uint32_t>& frozen, Program& out, bool chr)
{
    out = Program();
    scl_build(frozen, N, 0, out, chr);
    emit(out, OP_END, 0, 0);
    return std::string();
}

// ---------------------------------------------------------------------------------------------
// detectors   src/polarcode/errordetection/
// ---------------------------------------------------------------------------------------------
namespace {

// CRC++ CalculateBits on whole bytes (CRC.h:613-645): MSB first, init 0, no reflection/xorout
uint32_t crc_msb(const uint8_t* d, int bytes, unsigned width, uint32_t poly)
{
    const uint32_t top = 1u << (width - 1);
    const uint32_t mask = (width == 32) ? 0xFFFFFFFFu : ((1u << width) - 1u);
    uint32_t r = 0;
    for (int i = 0; i < bytes; ++i)
        for (int b = 7; b >= 0; --b) {
            const uint32_t fb = ((r & top) ? 1u : 0u) ^ ((d[i] >> b) & 1u);
            r = (r << 1) & mask;
            if (fb) r ^= poly;
        }
    return r;
}

// CRC32::gen crc32.cpp:33-41 (_mm_crc32_u32 over LE words): reflected CRC-32C, init 0, no xorout
uint32_t crc32c_le(const uint8_t* d, int bytes)
{
    uint32_t r = 0;
    for (int i = 0; i < bytes; ++i) {
        r ^= d[i];
        for (int b = 0; b < 8; ++b) r = (r >> 1) ^ ((r & 1u) ? 0x82F63B78u : 0u);
    }
    return r;
}

}  // namespace

std::string Detector::type_name() const
{
    if (size == 0) return "DUMMY";
    return nr ? "CRCNR" : "CRC";
}

unsigned Detector::check_bytes() const
{
    switch (size) {
    case 8: return 1;
    case 11: return 2;
    case 24: return 3;
    case 32: return 4;
    default: return 0;
    }
}

bool Detector::check(const uint8_t* p, int bytes) const
{
    switch (size) {
    case 0: return true;                                            // dummy.cpp:25
    case 8: return crc_msb(p, bytes - 1, 8, 0x07) == p[bytes - 1];  // crc8.cpp:49-54
    case 11: {                                                      // crc11nr.cpp:33-40
        const uint32_t rx = ((uint32_t)p[bytes - 2] << 8) | p[bytes - 1];
        return crc_msb(p, bytes - 2, 11, 0x621) == rx;
    }
    case 24: {  // crc24nrc.cpp:34-42
        const uint32_t rx =
            ((uint32_t)p[bytes - 3] << 16) | ((uint32_t)p[bytes - 2] << 8) | p[bytes - 1];
        return crc_msb(p, bytes - 3, 24, 0xB2B117) == rx;
    }
    case 32: {  // crc32.cpp:43-57
        const int w = (bytes >> 2) - 1;
        uint32_t rx;
        std::memcpy(&rx, p + 4 * w, 4);
        return crc32c_le(p, 4 * w) == rx;
    }
    default: return false;
    }
}

std::vector<uint32_t> Detector::syndrome_table(unsigned K) const
{
    const int nb = (int)((K + 7) / 8);
    std::vector<uint32_t> tab(K, 0u);
    if (size == 0) return tab;
    std::vector<uint8_t> msg((size_t)nb + 8, 0);
    for (unsigned t = 0; t < K; ++t) {
        std::fill(msg.begin(), msg.end(), 0);
        msg[t >> 3] = (uint8_t)(0x80u >> (t & 7));
        uint32_t syn = 0;
        switch (size) {
        case 8: syn = crc_msb(msg.data(), nb - 1, 8, 0x07) ^ msg[nb - 1]; break;
        case 11:
            syn = crc_msb(msg.data(), nb - 2, 11, 0x621) ^ (((uint32_t)msg[nb - 2] << 8) | msg[nb - 1]);
            break;
        case 24:
            syn = crc_msb(msg.data(), nb - 3, 24, 0xB2B117) ^
                  (((uint32_t)msg[nb - 3] << 16) | ((uint32_t)msg[nb - 2] << 8) | msg[nb - 1]);
            break;
        case 32: {
            const int w = (nb >> 2) - 1;
            uint32_t rx;
            std::memcpy(&rx, msg.data() + 4 * w, 4);
            syn = crc32c_le(msg.data(), 4 * w) ^ rx;
            break;
        }
        default: break;
        }
        tab[t] = syn;
    }
    return tab;
}

void Detector::generate(uint8_t* p, int bytes) const
{
    switch (size) {
    case 8: p[bytes - 1] = (uint8_t)crc_msb(p, bytes - 1, 8, 0x07); break;  // crc8.cpp:40-47
    case 11: {                                                              // crc11nr.cpp:25-31
        const uint32_t c = crc_msb(p, bytes - 2, 11, 0x621);
        p[bytes - 2] = (uint8_t)(c >> 8);
        p[bytes - 1] = (uint8_t)c;
        break;
    }
    case 24: {  // crc24nrc.cpp:25-32
        const uint32_t c = crc_msb(p, bytes - 3, 24, 0xB2B117);
        p[bytes - 3] = (uint8_t)(c >> 16);
        p[bytes - 2] = (uint8_t)(c >> 8);
        p[bytes - 1] = (uint8_t)c;
        break;
    }
    case 32: {  // crc32.cpp:33-41
        const int w = (bytes >> 2) - 1;
        const uint32_t c = crc32c_le(p, 4 * w);
        std::memcpy(p + 4 * w, &c, 4);
        break;
    }
    default: break;  // dummy.cpp: nothing
    }
}

unsigned Detector::field_offset(int bytes) const
{
    if (size == 0) return (unsigned)bytes;
    if (size == 32) return 4u * (unsigned)((bytes >> 2) - 1);
    return (unsigned)bytes - check_bytes();
}

std::vector<uint32_t> Detector::generator_table(int bytes) const
{
    std::vector<uint32_t> tab;
    if (size == 0) return tab;
    const unsigned fo = field_offset(bytes), cb = check_bytes();
    tab.assign(8u * fo, 0u);
    std::vector<uint8_t> msg((size_t)bytes, 0);
    for (unsigned t = 0; t < 8u * fo; ++t) {
        std::fill(msg.begin(), msg.end(), 0);
        msg[t >> 3] = (uint8_t)(0x80u >> (t & 7));
        generate(msg.data(), bytes);
        uint32_t v = 0;
        for (unsigned i = 0; i < cb; ++i) v |= (uint32_t)msg[fo + i] << (8 * i);
        tab[t] = v;
    }
    return tab;
}

// ---------------------------------------------------------------------------------------------
// information-bit placement   PackedContainer::insertPackedInformationBits bitcontainer.cpp (packed LUT walk)
// ---------------------------------------------------------------------------------------------
std::vector<DepositWord> deposit_table(unsigned N, const std::vector<uint32_t>& frozen)
{
    const unsigned nwords = N >= 32 ? N / 32 : 1;
    std::vector<DepositWord> tab(nwords);
    std::vector<uint32_t> mask(nwords, 0u);
    for (unsigned i = 0; i < N; ++i) mask[i >> 5] |= 1u << (i & 31);
    for (uint32_t f : frozen) mask[f >> 5] &= ~(1u << (f & 31));
    uint32_t base = 0;
    for (unsigned w = 0; w < nwords; ++w) {
        DepositWord& d = tab[w];
        d.mask = mask[w];
        d.base = base;
        d.pad = 0;
        base += (uint32_t)__builtin_popcount(mask[w]);
        // Move masks of the log-step expand: step i moves the bits that must travel a distance with bit i set.
        // mk marks the zeros to the right of each mask bit; its running parity says whether the count is odd.
        uint32_t m = mask[w], mk = ~m << 1;
        for (int i = 0; i < 5; ++i) {
            uint32_t mp = mk ^ (mk << 1);
            mp ^= mp << 2;
            mp ^= mp << 4;
            mp ^= mp << 8;
            mp ^= mp << 16;
            const uint32_t mv = mp & m;
            d.mv[i] = mv;
            m = (m ^ mv) | (mv >> (1u << i));
            mk &= ~mp;
        }
    }
    return tab;
}

uint32_t deposit_apply(const DepositWord& d, uint32_t x)
{
    for (int i = 4; i >= 0; --i) {
        const uint32_t t = x << (1u << i);
        x = (x & ~d.mv[i]) | (t & d.mv[i]);
    }
    return x & d.mask;
}

int make_detector(unsigned size, const std::string& type_in, Detector& det, std::string& err)
{
    // ErrorDetection::create errordetector.cpp:27-88
    std::string type = type_in;
    std::transform(type.begin(), type.end(), type.begin(),
                   [](unsigned char c) { return (char)std::tolower(c); });
    if (type.find("crc") != std::string::npos) {
        const bool nr = type.find("nr") != std::string::npos;
        static const unsigned sizes[] = { 0, 6, 8, 11, 16, 24, 32 };
        if (std::find(std::begin(sizes), std::end(sizes), size) == std::end(sizes)) {
            err = "CRC INVALID SIZE!";
            return -5;
        }
        if (size == 6 || size == 16) {
            err = "CRC-" + std::to_string(size) + " is not on the accelerated path (SURVEY 2, row 11)";
            return -3;
        }
        det.size = size;
        // getType(): CRC8/CRC32 report "CRC", CRC11NR/CRC24NRC report "CRCNR" whatever was asked
        det.nr = (size == 11 || size == 24);
        (void)nr;
        return 0;
    }
    if (type.find("cmac") != std::string::npos) {
        err = "CMAC detector is not on the accelerated path (SURVEY 2, row 11)";
        return -3;
    }
    err = "Unknown Error detector requested!";
    return -6;
}

}  // namespace pamd
