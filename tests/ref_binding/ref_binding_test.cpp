// Driver of tests/ref_binding/gpu_float.h (test infrastructure).
//   ref_binding_test cpu                 construct the subclass through the reference's base-class interface, compare
//                                        every getter with the reference's own decoder for the same code (no GPU)
//   ref_binding_test gpu <fixture.bin>   run the reference's own Decoder::decode_vector(const float*, void*)
//                                        (decoder.cpp:154-167) row by row through the subclass on the fixture's LLRs and
//                                        compare bytes and verdicts with what the reference's CPU decoder produced
#include "gpu_float.h"

#include <polarcode/construction/constructor.h>
#include <polarcode/errordetection/errordetector.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>

using namespace PolarCode;

static int fail(const std::string& m)
{
    std::fprintf(stderr, "FAIL: %s\n", m.c_str());
    return 1;
}

static int run_cpu()
{
    const size_t N = 1024, K = 512, L = 8;
    const std::vector<unsigned> frozen = Construction::frozen_bits(N, K, 0.0, "BB");  // constructor.cpp:41-63
    std::unique_ptr<Decoding::Decoder> ref(Decoding::create(N, L, frozen, "float"));
    std::unique_ptr<Decoding::Decoder> gpu(new Decoding::GpuFloat(N, L, frozen));
    if (gpu->blockLength() != ref->blockLength()) return fail("blockLength");
    if (gpu->infoLength() != ref->infoLength()) return fail("infoLength");
    if (gpu->getListSize() != ref->getListSize()) return fail("getListSize");
    if (gpu->frozenBits() != ref->frozenBits()) return fail("frozenBits");
    if (gpu->isSystematic() != ref->isSystematic()) return fail("isSystematic default");
    gpu->setSystematic(false);
    ref->setSystematic(false);
    if (gpu->isSystematic() != ref->isSystematic()) return fail("setSystematic");
    gpu->setSystematic(true);
    for (auto cfg : { std::make_pair(24u, "crcnr"), std::make_pair(11u, "crcnr"), std::make_pair(8u, "crc"), std::make_pair(32u, "crc"),
                      std::make_pair(0u, "crc") }) {
        gpu->setErrorDetection(ErrorDetection::create(cfg.first, cfg.second));
        ref->setErrorDetection(ErrorDetection::create(cfg.first, cfg.second));
        if (gpu->getErrorDetectionMode() != ref->getErrorDetectionMode()) return fail("getErrorDetectionMode");
        char buf[64];
        pcd_error_detection_mode(static_cast<Decoding::GpuFloat*>(gpu.get())->handle(), buf, sizeof buf);
        if (ref->getErrorDetectionMode() != buf) return fail(std::string("detector of the handle: ") + buf);
    }
    if (gpu->packedOutput() == nullptr) return fail("packedOutput");
    if (gpu->inputContainer() == nullptr) return fail("inputContainer: setSignal(const char*) is not virtual and writes into it");
    {   // the int8 flavour: a CharContainer behind the same interface; both setSignal overloads must land in it without a GPU
        std::unique_ptr<Decoding::Decoder> chr(new Decoding::GpuChar(N, L, frozen));
        std::unique_ptr<Decoding::Decoder> cref(Decoding::create(N, L, frozen, "char"));
        if (chr->infoLength() != cref->infoLength() || chr->getListSize() != cref->getListSize()) return fail("GpuChar getters");
        std::vector<char> r8(N, 5);
        std::vector<float> rf(N, -3.25f);
        chr->setSignal(r8.data());
        if (static_cast<CharContainer*>(chr->inputContainer())->data()[7] != 5) return fail("GpuChar setSignal(const char*)");
        chr->setSignal(rf.data());
        cref->setSignal(rf.data());
        if (static_cast<CharContainer*>(chr->inputContainer())->data()[9] != static_cast<CharContainer*>(cref->inputContainer())->data()[9])
            return fail("GpuChar setSignal(const float*) quantises differently from the reference's char decoder");
        gpu->setSignal(r8.data());  // (GpuFloat: the int8 row is widened into its FloatContainer -- used to dereference a null container)
        if (static_cast<FloatContainer*>(gpu->inputContainer())->data()[3] != 5.0f) return fail("GpuFloat setSignal(const char*)");
    }
    std::printf("cpu ok: GpuFloat constructs through the reference's Decoder interface; getters agree with Decoding::create()\n");
    return 0;
}

static int run_gpu(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return fail(std::string("cannot open ") + path);
    uint32_t h[10];
    f.read(reinterpret_cast<char*>(h), sizeof h);
    if (h[0] != 0x31584650u) return fail("bad fixture magic");
    const uint32_t N = h[1], L = h[3], nF = h[4], count = h[5], nb = h[6], crcsize = h[7], crcnr = h[8], dectype = h[9];
    const bool chr = (dectype & 0xFF) == 3;        // the int8 decoders (type "char"): GpuChar
    const bool i8rows = chr || (dectype & 0x200);  // the row went through decode_vector(const char*) in the reference
    if ((dectype & 0xFF) != 0 && !chr) return fail("float- or char-decoder fixture expected");
    std::vector<uint32_t> fz(nF);
    f.read(reinterpret_cast<char*>(fz.data()), 4 * nF);
    std::vector<unsigned> frozen(fz.begin(), fz.end());
    std::unique_ptr<Decoding::Decoder> owner(chr ? static_cast<Decoding::Decoder*>(new Decoding::GpuChar(N, L, frozen))
                                                 : static_cast<Decoding::Decoder*>(new Decoding::GpuFloat(N, L, frozen)));
    Decoding::Decoder& dec = *owner;  // everything below goes through the reference's base class
    dec.setSystematic(!(dectype & 0x100));
    dec.setErrorDetection(ErrorDetection::create(crcsize, crcnr ? "crcnr" : "crc"));
    const size_t rec = 4 * (size_t)N + nb + nb + 1 + 1 + 4 * (size_t)L;
    std::vector<char> row(rec);
    std::vector<char> row8(N);
    std::vector<unsigned char> out(nb);
    unsigned bad = 0;
    for (uint32_t r = 0; r < count; ++r) {
        f.read(row.data(), rec);
        const float* llr = reinterpret_cast<const float*>(row.data());
        const unsigned char* want = reinterpret_cast<const unsigned char*>(row.data()) + 4 * (size_t)N + nb;
        const bool want_ok = want[nb] != 0;
        bool ok;
        if (i8rows) {  // the NON-virtual overload: Decoder::decode_vector(const char*, void*) decoder.cpp:169-181 -> setSignal(const char*)
            for (uint32_t i = 0; i < N; ++i) row8[i] = (char)(int)llr[i];  // (the fixture stores the int8 values as floats)
            ok = dec.decode_vector(row8.data(), out.data());
        } else {
            ok = dec.decode_vector(llr, out.data());  // Decoder::decode_vector decoder.cpp:154-167
        }
        if (ok != want_ok || std::memcmp(out.data(), want, nb) != 0 || std::memcmp(dec.packedOutput(), want, nb) != 0) ++bad;
    }
    if (bad) return fail(std::to_string(bad) + " of " + std::to_string(count) + " rows differ from the reference's CPU decoder");
    std::printf("gpu ok: %u rows through Decoder::decode_vector on the GPU equal the reference (%s)\n", count, path);
    return 0;
}

int main(int argc, char** argv)
{
    try {
        if (argc >= 2 && std::string(argv[1]) == "cpu") return run_cpu();
        if (argc >= 3 && std::string(argv[1]) == "gpu") return run_gpu(argv[2]);
    } catch (const std::exception& e) {
        return fail(std::string("exception: ") + e.what());
    }
    std::fprintf(stderr, "usage: ref_binding_test cpu | gpu <fixture.bin>\n");
    return 2;
}
