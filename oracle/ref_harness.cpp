// TEST INFRASTRUCTURE ONLY -- never linked into the product.
//
// Harness around the *unmodified* reference library (oracle/_ref/libPolarCode_ref.so,
// compiled from /root/reference by oracle/Makefile).  It only calls the reference's
// public API:
//   PolarCode::Construction::frozen_bits        (constructor.cpp:41-63)
//   PolarCode::Encoding::ButterflyFipPacked     (butterfly_fip_packed.cpp:45-70)
//   PolarCode::ErrorDetection::create           (errordetector.cpp:27-88)
//   PolarCode::Decoding::{SclAvxFloat,FastSscAvxFloat,AdaptiveFloat}::decode_vector
//                                               (decoder.cpp:154-167)
// and peeks the final path metrics of SclAvxFloat (scl_avx_float.h: PathList::mMetric).
//
// Sub-commands
//   gen   N K L ctor dsnr crcsize crctype ebn0_db count seed mode dectype out.bin [systematic=1] [input=f32]
//         (ctor = BB | BE | 5G, or @file with one frozen index per line: hand-made sets, K is then derived)
//         systematic = 0: setSystematic(false) on encoder and decoder (fastssc_avx_float.cpp:944-948,
//         scl_avx_float.cpp:731-746); input = i8: the row (rounded to int8 values) goes through the
//         decode_vector(const char*) overload (decoder.cpp:169-181, bitcontainer.cpp:202-207)
//         writes a golden fixture (format below).  A FRESH decoder object is used for every
//         codeword (the reference carries mMetric[0] over between calls).
//   time  N K L ctor dsnr crcsize crctype ebn0_db rows threads seconds [char]
//         the reference's own CPU throughput: one decoder object per std::thread
//         (simulator.cpp:95-101), prints one JSON line.
//   enc   N K ctor dsnr crcsize crctype systematic count seed out.bin
//         golden fixture of the ENCODER (ButterflyFipPacked::encode_vector): layout
//         u32 hdr[10] = {0x31454650 'PFE1', N, K, systematic, nF, count, nb, crcsize, crcnr, 0}, u32 F[nF],
//         count x { u8 info_in[nb] | u8 info_after[nb] (generate() writes the checksum in place) | u8 cw[N/8] }
//   enctime N K ctor dsnr crcsize crctype threads seconds     the reference encoder's CPU throughput
//   punc  M N K ctor dsnr count seed out.bin
//         golden fixture of the PUNCTURER (puncturer.cpp:51-89, puncturer.h:66-97) for punctured length M <= N:
//         u32 hdr[10] = {0x31504650 'PFP1', M, N, K, nF, count, 0, 0, 0, 0}, u32 F[nF], u32 positions[M],
//         count x { f32 in[N] | f32 punctured[M] | f32 depunctured[N] (of `punctured`) | u8 packed_in[N/8] | u8 packed_out[M/8] }
//   punctime M N K ctor dsnr threads seconds     CPU throughput of depuncture<float> (rows/s)
//   nrseq out.u16                 the order in which the "5G" construction (3GPP TS 38.212 Table 5.3.1.2-1 as
//         implemented by Construction::frozen_bits(1024, K, *, "5G")) freezes the 1024 positions: entry i is the
//         position that joins the frozen set when K drops from 1024 - i to 1023 - i; 1024 little-endian u16
//   sccheck in.txt out.txt        every input line "N f0 f1 ..." is a frozen set; the output repeats the line with a leading
//         1 (FastSscAvxFloat(N, F) constructs) or 0 (it throws std::invalid_argument, fastssc_avx_float.cpp:820-844)
//   frozen N K ctor dsnr          prints the frozen set (one index per line)
//   crc   size type hexbytes      prints Detector::calculate()/check() for KAT pinning
//
// Fixture layout (little endian):
//   u32 hdr[10] = {0x31584650 'PFX1', N, K, L, nF, count, nb=(K+7)/8, crcsize, crcnr, dectype}
//   u32 F[nF]
//   count x { f32 llr[N] | u8 info[nb] | u8 out[nb] | u8 ok | u8 npaths | f32 metrics[L] }
// mode: awgn | noise | ties | quant | sat      dectype (low byte): 0 float, 2 mixed, 3 char; bit 8: non-systematic,
// bit 9: the row was fed as int8 through decode_vector(const char*)

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <random>
#include <stack>
#include <string>
#include <thread>
#include <vector>

#include <immintrin.h>

#include <polarcode/datapool.txx>
// every STL header the reference pulls in is already included above, so redefining the
// access keywords only affects the reference's own class bodies.
#define class struct
#define private public
#define protected public
#include <polarcode/decoding/adaptive_float.h>
#include <polarcode/decoding/fastssc_avx_float.h>
#include <polarcode/decoding/scl_avx_float.h>
#include <polarcode/decoding/fastssc_fip_char.h>
#include <polarcode/decoding/scl_fip_char.h>
#undef class
#undef private
#undef protected
#include <polarcode/construction/constructor.h>
#include <polarcode/encoding/butterfly_fip_packed.h>
#include <polarcode/errordetection/errordetector.h>
#include <polarcode/puncturer.h>

using namespace PolarCode;

struct Code {
    unsigned N, K, L, crcsize;
    std::string crctype, ctor;
    float dsnr;
    int dectype;
    std::vector<unsigned> F;
};

static Decoding::Decoder* make_decoder(const Code& c)
{
    Decoding::Decoder* d;
    if (c.dectype == 3)  // "char": FastSscFipChar / SclFipChar (decoder.cpp:65-67,80-82)
        d = c.L == 1 ? (Decoding::Decoder*)new Decoding::FastSscFipChar(c.N, c.F)
                     : (Decoding::Decoder*)new Decoding::SclFipChar(c.N, c.L, c.F);
    else if (c.dectype == 2)
        d = new Decoding::AdaptiveFloat(c.N, c.L, c.F);
    else if (c.L == 1)
        d = new Decoding::FastSscAvxFloat(c.N, c.F);
    else
        d = new Decoding::SclAvxFloat(c.N, c.L, c.F);
    d->setErrorDetection(ErrorDetection::create(c.crcsize, c.crctype));
    return d;
}

struct Gauss {
    std::mt19937_64 g;
    explicit Gauss(uint64_t s) : g(s) {}
    double uni() { return (double)(g() >> 11) * (1.0 / 9007199254740992.0); }
    double operator()()
    {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

static void make_row(const Code& c, Encoding::ButterflyFipPacked& enc, Gauss& rng,
                     const std::string& mode, float ebn0_db, std::vector<uint8_t>& info,
                     std::vector<float>& llr)
{
    const unsigned nb = (c.K + 7) / 8;
    info.assign(std::max(nb, 8u), 0);
    for (unsigned i = 0; i < nb; ++i) info[i] = (uint8_t)(rng.g() & 0xFF);
    if (c.K % 8) info[nb - 1] &= (uint8_t)(0xFF << (8 - c.K % 8));
    std::vector<uint8_t> cw(c.N / 8 + 32, 0);
    enc.encode_vector(info.data(), cw.data()); // writes the CRC into info[] (encoder.cpp / generate())
    llr.resize(c.N);
    const double ebn0 = std::pow(10.0, ebn0_db / 10.0);
    const double sigma = 1.0 / std::sqrt(2.0 * ebn0 * (double)c.K / (double)c.N);
    for (unsigned i = 0; i < c.N; ++i) {
        const int bit = (cw[i / 8] >> (7 - (i % 8))) & 1;
        const double x = 1.0 - 2.0 * bit;
        if (mode == "awgn") {
            llr[i] = (float)(x + sigma * rng());
        } else if (mode == "noise") {
            llr[i] = (float)rng();
        } else if (mode == "ties") {
            llr[i] = (float)((rng.g() % 16 == 0) ? -x : x);
        } else if (mode == "sat") { // int8 LLRs that saturate: round(60 y) clipped to [-128, 127]
            double v = std::nearbyint((x + sigma * rng()) * 60.0);
            v = std::max(-128.0, std::min(127.0, v));
            llr[i] = (float)v;
        } else { // quant: int8-valued LLRs (many equal magnitudes)
            double v = std::nearbyint((x + sigma * rng()) * 8.0);
            v = std::max(-127.0, std::min(127.0, v));
            llr[i] = (float)v;
        }
    }
}

static int cmd_gen(int argc, char** argv)
{
    if (argc < 15) { fprintf(stderr, "gen: bad args\n"); return 2; }
    Code c;
    c.N = atoi(argv[2]); c.K = atoi(argv[3]); c.L = atoi(argv[4]); c.ctor = argv[5];
    c.dsnr = atof(argv[6]); c.crcsize = atoi(argv[7]); c.crctype = argv[8];
    const float ebn0_db = atof(argv[9]);
    const unsigned count = atoi(argv[10]);
    const uint64_t seed = strtoull(argv[11], nullptr, 10);
    const std::string mode = argv[12];
    c.dectype = (std::string(argv[13]) == "mixed") ? 2 : (std::string(argv[13]) == "char" ? 3 : 0);
    const char* out = argv[14];
    const bool systematic = argc > 15 ? atoi(argv[15]) != 0 : true;
    // (the int8 decoders are always fed through the char overload: their float overload converts with three different
    //  rounding routines depending on the block length, bitcontainer.cpp:449-516)
    const bool in_i8 = c.dectype == 3 || (argc > 16 && std::string(argv[16]) == "i8");
    if (!c.ctor.empty() && c.ctor[0] == '@') {  // explicit frozen set: one index per line in the named file
        FILE* ff = fopen(c.ctor.c_str() + 1, "r");
        if (!ff) { perror(c.ctor.c_str() + 1); return 1; }
        unsigned v;
        while (fscanf(ff, "%u", &v) == 1) c.F.push_back(v);
        fclose(ff);
        c.K = c.N - (unsigned)c.F.size();
    } else {
        c.F = Construction::frozen_bits(c.N, c.K, c.dsnr, c.ctor);
    }
    const unsigned nb = (c.K + 7) / 8;

    Encoding::ButterflyFipPacked enc(c.N, c.F);
    enc.setErrorDetection(ErrorDetection::create(c.crcsize, c.crctype));
    enc.setSystematic(systematic);
    Gauss rng(seed);

    FILE* f = fopen(out, "wb");
    if (!f) { perror(out); return 1; }
    const bool nr = c.crctype.find("nr") != std::string::npos;
    uint32_t hdr[10] = { 0x31584650u, c.N, c.K, c.L, (uint32_t)c.F.size(), count, nb,
                         c.crcsize, (uint32_t)nr,
                         (uint32_t)c.dectype | (systematic ? 0u : 0x100u) | (in_i8 ? 0x200u : 0u) };
    fwrite(hdr, 4, 10, f);
    fwrite(c.F.data(), 4, c.F.size(), f);

    std::vector<uint8_t> info;
    std::vector<float> llr;
    unsigned n_ok = 0, n_match = 0;
    for (unsigned r = 0; r < count; ++r) {
        make_row(c, enc, rng, mode, ebn0_db, info, llr);
        std::unique_ptr<Decoding::Decoder> dec(make_decoder(c)); // fresh: metric starts at 0
        dec->setSystematic(systematic);
        std::vector<uint8_t> outb(nb + 16, 0);
        bool ok;
        if (in_i8) {  // the fixture keeps the row as (int8-valued) floats
            std::vector<char> row8(c.N);
            for (unsigned i = 0; i < c.N; ++i) {
                const float v = std::max(-128.0f, std::min(127.0f, std::nearbyint(llr[i])));
                row8[i] = (char)(int)v;
                llr[i] = (float)(int)v;
            }
            ok = dec->decode_vector(row8.data(), outb.data());
        } else {
            ok = dec->decode_vector(llr.data(), outb.data());
        }
        std::vector<float> metrics(c.L, 0.0f);
        uint8_t npaths = 1;
        Decoding::SclAvxFloat* scl = nullptr;
        if (c.dectype == 0 && c.L > 1) scl = dynamic_cast<Decoding::SclAvxFloat*>(dec.get());
        if (scl) {
            unsigned p = scl->mPathList->mNextPathCount;
            if (p < 1) p = 1;
            npaths = (uint8_t)p;
            for (unsigned i = 0; i < p && i < c.L; ++i) metrics[i] = scl->mPathList->mMetric[i];
        }
        if (c.dectype == 3 && c.L > 1) {  // exact `long` metrics (scl_fip_char.cpp:155), below 2^24 in magnitude: exact as f32
            Decoding::SclFipChar* sc = dynamic_cast<Decoding::SclFipChar*>(dec.get());
            unsigned p = sc->mPathList->mNextPathCount;
            if (p < 1) p = 1;
            npaths = (uint8_t)p;
            for (unsigned i = 0; i < p && i < c.L; ++i) metrics[i] = (float)sc->mPathList->mMetric[i];
        }
        const uint8_t okb = ok ? 1 : 0;
        fwrite(llr.data(), 4, c.N, f);
        fwrite(info.data(), 1, nb, f);
        fwrite(outb.data(), 1, nb, f);
        fwrite(&okb, 1, 1, f);
        fwrite(&npaths, 1, 1, f);
        fwrite(metrics.data(), 4, c.L, f);
        n_ok += ok;
        n_match += (memcmp(info.data(), outb.data(), nb) == 0);
    }
    fclose(f);
    fprintf(stderr, "%s: N=%u K=%u L=%u |F|=%zu rows=%u ok=%u payload_match=%u\n", out, c.N, c.K,
            c.L, c.F.size(), count, n_ok, n_match);
    return 0;
}

static int cmd_time(int argc, char** argv)
{
    if (argc < 13) { fprintf(stderr, "time: bad args\n"); return 2; }
    Code c;
    c.N = atoi(argv[2]); c.K = atoi(argv[3]); c.L = atoi(argv[4]); c.ctor = argv[5];
    c.dsnr = atof(argv[6]); c.crcsize = atoi(argv[7]); c.crctype = argv[8];
    const float ebn0_db = atof(argv[9]);
    const unsigned rows = atoi(argv[10]);
    const unsigned threads = atoi(argv[11]);
    const double seconds = atof(argv[12]);
    // optional: "char" = the int8 decoders on rows quantised as round(8 y) through decode_vector(const char*)
    const bool chr = argc > 13 && std::string(argv[13]) == "char";
    c.dectype = chr ? 3 : 0;
    c.F = Construction::frozen_bits(c.N, c.K, c.dsnr, c.ctor);
    const unsigned nb = (c.K + 7) / 8;
    Encoding::ButterflyFipPacked enc(c.N, c.F);
    enc.setErrorDetection(ErrorDetection::create(c.crcsize, c.crctype));
    Gauss rng(12345);
    std::vector<float> all((size_t)rows * c.N);
    std::vector<char> all8(chr ? (size_t)rows * c.N : 0);
    std::vector<uint8_t> info;
    std::vector<float> llr;
    for (unsigned r = 0; r < rows; ++r) {
        make_row(c, enc, rng, chr ? "quant" : "awgn", ebn0_db, info, llr);
        memcpy(&all[(size_t)r * c.N], llr.data(), 4 * c.N);
        if (chr)
            for (unsigned i = 0; i < c.N; ++i) all8[(size_t)r * c.N + i] = (char)(int)llr[i];
    }
    std::atomic<uint64_t> total(0);
    std::atomic<uint64_t> oks(0);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) {
        pool.emplace_back([&, t]() {
            std::unique_ptr<Decoding::Decoder> dec(make_decoder(c));
            std::vector<uint8_t> outb(nb + 16);
            uint64_t n = 0, ok = 0;
            for (;;) {
                for (unsigned r = t; r < rows; r += threads) {
                    ok += chr ? dec->decode_vector(&all8[(size_t)r * c.N], outb.data())
                              : dec->decode_vector(&all[(size_t)r * c.N], outb.data());
                    ++n;
                }
                const double el = std::chrono::duration<double>(
                                      std::chrono::steady_clock::now() - t0).count();
                if (el >= seconds) break;
            }
            total += n;
            oks += ok;
        });
    }
    for (auto& th : pool) th.join();
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("{\"kind\": \"reference\", \"N\": %u, \"K\": %u, \"L\": %u, \"threads\": %u, "
           "\"codewords\": %llu, \"seconds\": %.4f, \"cw_per_s\": %.2f, \"ok_frac\": %.4f}\n",
           c.N, c.K, c.L, threads, (unsigned long long)total.load(), el, total.load() / el,
           (double)oks.load() / (double)std::max<uint64_t>(1, total.load()));
    return 0;
}

static bool load_frozen(Code& c)
{
    if (!c.ctor.empty() && c.ctor[0] == '@') {
        FILE* ff = fopen(c.ctor.c_str() + 1, "r");
        if (!ff) { perror(c.ctor.c_str() + 1); return false; }
        unsigned v;
        while (fscanf(ff, "%u", &v) == 1) c.F.push_back(v);
        fclose(ff);
        c.K = c.N - (unsigned)c.F.size();
    } else {
        c.F = Construction::frozen_bits(c.N, c.K, c.dsnr, c.ctor);
    }
    return true;
}

static int cmd_enc(int argc, char** argv)
{
    if (argc < 12) { fprintf(stderr, "enc: bad args\n"); return 2; }
    Code c;
    c.N = atoi(argv[2]); c.K = atoi(argv[3]); c.L = 1; c.ctor = argv[4];
    c.dsnr = atof(argv[5]); c.crcsize = atoi(argv[6]); c.crctype = argv[7];
    const int systematic = atoi(argv[8]);
    const unsigned count = atoi(argv[9]);
    const uint64_t seed = strtoull(argv[10], nullptr, 10);
    const char* out = argv[11];
    if (!load_frozen(c)) return 1;
    const unsigned nb = (c.K + 7) / 8;
    Encoding::ButterflyFipPacked enc(c.N, c.F);
    enc.setErrorDetection(ErrorDetection::create(c.crcsize, c.crctype));
    enc.setSystematic(systematic != 0);
    std::mt19937_64 g(seed);
    FILE* f = fopen(out, "wb");
    if (!f) { perror(out); return 1; }
    const bool nr = c.crctype.find("nr") != std::string::npos;
    uint32_t hdr[10] = { 0x31454650u, c.N, c.K, (uint32_t)systematic, (uint32_t)c.F.size(), count, nb,
                         c.crcsize, (uint32_t)nr, 0 };
    fwrite(hdr, 4, 10, f);
    fwrite(c.F.data(), 4, c.F.size(), f);
    for (unsigned r = 0; r < count; ++r) {
        std::vector<uint8_t> info(std::max(nb, 8u) + 8, 0), cw(c.N / 8 + 64, 0);
        for (unsigned i = 0; i < nb; ++i) info[i] = (uint8_t)(g() & 0xFF);
        if (r == 0) std::fill(info.begin(), info.end(), 0);
        if (r == 1) std::fill(info.begin(), info.begin() + nb, 0xFF);
        if (c.K % 8) info[nb - 1] &= (uint8_t)(0xFF << (8 - c.K % 8));
        fwrite(info.data(), 1, nb, f);
        enc.encode_vector(info.data(), cw.data());
        fwrite(info.data(), 1, nb, f);
        fwrite(cw.data(), 1, c.N / 8, f);
    }
    fclose(f);
    fprintf(stderr, "%s: N=%u K=%u sys=%d |F|=%zu rows=%u\n", out, c.N, c.K, systematic, c.F.size(), count);
    return 0;
}

static int cmd_enctime(int argc, char** argv)
{
    if (argc < 10) { fprintf(stderr, "enctime: bad args\n"); return 2; }
    Code c;
    c.N = atoi(argv[2]); c.K = atoi(argv[3]); c.L = 1; c.ctor = argv[4];
    c.dsnr = atof(argv[5]); c.crcsize = atoi(argv[6]); c.crctype = argv[7];
    const unsigned threads = atoi(argv[8]);
    const double seconds = atof(argv[9]);
    if (!load_frozen(c)) return 1;
    const unsigned nb = (c.K + 7) / 8;
    std::atomic<uint64_t> total(0);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) {
        pool.emplace_back([&, t]() {
            Encoding::ButterflyFipPacked enc(c.N, c.F);
            enc.setErrorDetection(ErrorDetection::create(c.crcsize, c.crctype));
            std::mt19937_64 g(1000 + t);
            const unsigned rows = 256;
            std::vector<uint8_t> info((size_t)rows * (nb + 8), 0), cw(c.N / 8 + 64, 0);
            for (auto& b : info) b = (uint8_t)(g() & 0xFF);
            uint64_t n = 0;
            for (;;) {
                for (unsigned r = 0; r < rows; ++r) {
                    enc.encode_vector(&info[(size_t)r * (nb + 8)], cw.data());
                    ++n;
                }
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (el >= seconds) break;
            }
            total += n;
        });
    }
    for (auto& th : pool) th.join();
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("{\"kind\": \"reference\", \"op\": \"encode\", \"N\": %u, \"K\": %u, \"threads\": %u, "
           "\"codewords\": %llu, \"seconds\": %.4f, \"cw_per_s\": %.2f}\n",
           c.N, c.K, threads, (unsigned long long)total.load(), el, total.load() / el);
    return 0;
}

static int cmd_punc(int argc, char** argv)
{
    if (argc < 10) { fprintf(stderr, "punc: bad args\n"); return 2; }
    const unsigned M = atoi(argv[2]);
    Code c;
    c.N = atoi(argv[3]); c.K = atoi(argv[4]); c.L = 1; c.ctor = argv[5]; c.dsnr = atof(argv[6]);
    const unsigned count = atoi(argv[7]);
    const uint64_t seed = strtoull(argv[8], nullptr, 10);
    const char* out = argv[9];
    if (!load_frozen(c)) return 1;
    Puncturer punc(M, c.F);
    if (punc.parentBlockLength() != c.N) { fprintf(stderr, "punc: parent length %zu != N\n", punc.parentBlockLength()); return 1; }
    const std::vector<unsigned> pos = punc.blockOutputPositions();
    FILE* f = fopen(out, "wb");
    if (!f) { perror(out); return 1; }
    uint32_t hdr[10] = { 0x31504650u, M, c.N, c.K, (uint32_t)c.F.size(), count, 0, 0, 0, 0 };
    fwrite(hdr, 4, 10, f);
    fwrite(c.F.data(), 4, c.F.size(), f);
    fwrite(pos.data(), 4, pos.size(), f);
    std::mt19937_64 g(seed);
    for (unsigned r = 0; r < count; ++r) {
        std::vector<float> in(c.N), pun(M), dep(c.N);
        for (auto& v : in) v = (float)((double)(int64_t)(g() % 2001) - 1000.0) / 64.0f;
        punc.puncture<float>(pun.data(), in.data());
        punc.depuncture<float>(dep.data(), pun.data());
        std::vector<uint8_t> pin(c.N / 8), pout(M / 8 + 1, 0);
        for (auto& b : pin) b = (uint8_t)(g() & 0xFF);
        if (M % 8 == 0) punc.puncturePacked(pout.data(), pin.data());
        fwrite(in.data(), 4, c.N, f);
        fwrite(pun.data(), 4, M, f);
        fwrite(dep.data(), 4, c.N, f);
        fwrite(pin.data(), 1, c.N / 8, f);
        fwrite(pout.data(), 1, M / 8, f);
    }
    fclose(f);
    fprintf(stderr, "%s: M=%u N=%u |F|=%zu rows=%u\n", out, M, c.N, c.F.size(), count);
    return 0;
}

static int cmd_punctime(int argc, char** argv)
{
    if (argc < 9) { fprintf(stderr, "punctime: bad args\n"); return 2; }
    const unsigned M = atoi(argv[2]);
    Code c;
    c.N = atoi(argv[3]); c.K = atoi(argv[4]); c.L = 1; c.ctor = argv[5]; c.dsnr = atof(argv[6]);
    const unsigned threads = atoi(argv[7]);
    const double seconds = atof(argv[8]);
    if (!load_frozen(c)) return 1;
    std::atomic<uint64_t> total(0);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) {
        pool.emplace_back([&, t]() {
            Puncturer punc(M, c.F);
            const unsigned rows = 256;
            std::vector<float> in((size_t)rows * M, 1.0f), outv((size_t)rows * c.N);
            uint64_t n = 0;
            for (;;) {
                for (unsigned r = 0; r < rows; ++r) {
                    punc.depuncture<float>(&outv[(size_t)r * c.N], &in[(size_t)r * M]);
                    ++n;
                }
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (el >= seconds) break;
            }
            total += n;
        });
    }
    for (auto& th : pool) th.join();
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("{\"kind\": \"reference\", \"op\": \"depuncture\", \"M\": %u, \"N\": %u, \"threads\": %u, "
           "\"rows\": %llu, \"seconds\": %.4f, \"rows_per_s\": %.2f}\n",
           M, c.N, threads, (unsigned long long)total.load(), el, total.load() / el);
    return 0;
}

static int cmd_nrseq(int argc, char** argv)
{
    if (argc < 3) return 2;
    std::vector<uint16_t> seq;
    std::vector<unsigned> prev;
    for (unsigned i = 0; i < 1024; ++i) {
        auto cur = Construction::frozen_bits(1024, 1024 - (i + 1), 0.0f, "5G");
        std::vector<unsigned> diff;
        std::set_difference(cur.begin(), cur.end(), prev.begin(), prev.end(), std::back_inserter(diff));
        if (diff.size() != 1) { fprintf(stderr, "nrseq: step %u adds %zu positions\n", i, diff.size()); return 1; }
        seq.push_back((uint16_t)diff[0]);
        prev = cur;
    }
    FILE* f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    fwrite(seq.data(), 2, seq.size(), f);
    fclose(f);
    return 0;
}

static int cmd_sccheck(int argc, char** argv)
{
    if (argc < 4) return 2;
    FILE* in = fopen(argv[2], "r");
    FILE* out = fopen(argv[3], "w");
    if (!in || !out) { perror("sccheck"); return 1; }
    char* line = nullptr;
    size_t cap = 0;
    while (getline(&line, &cap, in) > 0) {
        std::vector<unsigned> v;
        char* p = line;
        for (;;) {
            char* e;
            const unsigned long x = strtoul(p, &e, 10);
            if (e == p) break;
            v.push_back((unsigned)x);
            p = e;
        }
        if (v.empty()) continue;
        const unsigned N = v[0];
        std::vector<unsigned> F(v.begin() + 1, v.end());
        int ok = 1;
        try {
            Decoding::FastSscAvxFloat dec(N, F);
        } catch (const std::invalid_argument&) {
            ok = 0;
        }
        fprintf(out, "%d %u", ok, N);
        for (unsigned f : F) fprintf(out, " %u", f);
        fprintf(out, "\n");
    }
    free(line);
    fclose(in);
    fclose(out);
    return 0;
}

static int cmd_frozen(int argc, char** argv)
{
    if (argc < 6) return 2;
    auto F = Construction::frozen_bits(atoi(argv[2]), atoi(argv[3]), atof(argv[5]), argv[4]);
    for (auto x : F) printf("%u\n", x);
    return 0;
}

static int cmd_crc(int argc, char** argv)
{
    if (argc < 5) return 2;
    std::unique_ptr<ErrorDetection::Detector> d(ErrorDetection::create(atoi(argv[2]), argv[3]));
    std::string hex = argv[4];
    std::vector<uint8_t> b(hex.size() / 2 + 16, 0);
    const unsigned n = hex.size() / 2;
    for (unsigned i = 0; i < n; ++i) b[i] = (uint8_t)strtoul(hex.substr(2 * i, 2).c_str(), nullptr, 16);
    const bool chk = d->check(b.data(), (int)n);
    d->generate(b.data(), (int)n);
    printf("%s-%u check=%d generated=", d->getType().c_str(), d->getCheckBitCount(), (int)chk);
    for (unsigned i = 0; i < n; ++i) printf("%02x", b[i]);
    printf("\n");
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_harness gen|time|enc|enctime|punc|punctime|frozen|crc ...\n"); return 2; }
    const std::string cmd = argv[1];
    if (cmd == "gen") return cmd_gen(argc, argv);
    if (cmd == "time") return cmd_time(argc, argv);
    if (cmd == "enc") return cmd_enc(argc, argv);
    if (cmd == "enctime") return cmd_enctime(argc, argv);
    if (cmd == "punc") return cmd_punc(argc, argv);
    if (cmd == "nrseq") return cmd_nrseq(argc, argv);
    if (cmd == "sccheck") return cmd_sccheck(argc, argv);
    if (cmd == "punctime") return cmd_punctime(argc, argv);
    if (cmd == "frozen") return cmd_frozen(argc, argv);
    if (cmd == "crc") return cmd_crc(argc, argv);
    fprintf(stderr, "unknown command %s\n", cmd.c_str());
    return 2;
}
