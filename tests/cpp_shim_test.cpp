// Exercises the C++ mirror of the reference's Decoder interface (polar-codes_amd/cpp/polar_decoder.hpp).
// argv[1] = "host": only host-side behaviour (no GPU needed);  "gpu": also decodes the N = 8 known answer
// of the reference's test/polarcode/decodingtest.cpp:1143-1173.
#include <cstdio>
#include <cstring>
#include <memory>

#include "../polar-codes_amd/cpp/polar_decoder.hpp"

using namespace PolarAmd::Decoding;

int main(int argc, char** argv)
{
    const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
    std::vector<unsigned> frozen = { 0, 1, 2, 4 };
    std::unique_ptr<Decoder> d(create(8, 4, frozen, "float"));
    if (d->blockLength() != 8 || d->infoLength() != 4 || d->getListSize() != 4) return 1;
    if (!d->isSystematic() || d->getErrorDetectionMode() != "CRC-8") return 2;
    d->setErrorDetection(0, "crc");
    if (d->getErrorDetectionMode() != "DUMMY-0") return 3;
    if (d->frozenBits() != frozen) return 4;
    bool threw = false;
    try {
        create(8, 4, frozen, "nonsense");
    } catch (const std::logic_error&) {
        threw = true;
    }
    if (!threw) return 5;
    threw = false;
    try {
        d->setErrorDetection(7, "crc");
    } catch (const std::logic_error&) {
        threw = true;
    }
    if (!threw) return 6;
    PolarAmd::Encoding::Encoder enc(8, frozen);
    if (enc.blockLength() != 8 || enc.infoLength() != 4 || !enc.isSystematic()) return 11;
    if (enc.getErrorDetectionMode() != "DUMMY-0" || enc.frozenBits() != frozen) return 12;  // encoder.cpp:18
    {
        PolarAmd::Puncturer punc(6, frozen);   // parent 8, positions 0 and 1 removed
        if (punc.blockLength() != 6 || punc.parentBlockLength() != 8) return 14;
        if (punc.blockOutputPositions() != std::vector<unsigned>({ 2, 3, 4, 5, 6, 7 })) return 15;
        threw = false;
        try {
            PolarAmd::Puncturer bad(3, std::vector<unsigned>{ 0 });  // parent 4 needs 1 position: ok; 
        } catch (const std::out_of_range&) {
            threw = true;
        }
        if (threw) return 16;
        threw = false;
        try {
            PolarAmd::Puncturer bad(5, std::vector<unsigned>{ 0, 1 });  // parent 8 needs 3 positions, 2 frozen
        } catch (const std::out_of_range&) {
            threw = true;
        }
        if (!threw) return 17;
        if (gpu) {
            const float in[8] = { 0, 1, 2, 3, 4, 5, 6, 7 };
            float pun[6], dep[8];
            punc.puncture<float>(pun, in);
            punc.depuncture<float>(dep, pun);
            for (int i = 0; i < 6; ++i)
                if (pun[i] != (float)(i + 2)) return 18;
            if (dep[0] != 0.0f || dep[1] != 0.0f || dep[7] != 7.0f) return 19;
        }
    }
    if (gpu) {
        // systematic code word of info bits 1111 on positions {3,5,6,7}: x = u G with u_frozen = 0 is 0x17 -> after
        // the systematic re-encode the information positions carry the information bits
        unsigned char info[8] = { 0xF0 }, cw[8] = { 0 };
        enc.encode_vector(info, cw);
        const unsigned char infopos = 0x17;  // bits 3,5,6,7 (MSB first)
        if ((cw[0] & infopos) != infopos) return 13;
    }
    if (gpu) {
        const float llr[8] = { -5, -6, -4, 1, -4, -5, -7, 2 };
        unsigned char out[8] = { 0 };
        const bool ok = d->decode_vector(llr, out);
        if (!ok) return 7;                       // dummy detector always passes
        if ((out[0] & 0xF0) != 0xF0) return 8;   // decodingtest.cpp:1172
        std::unique_ptr<Decoder> sc(create(8, 1, frozen, "float"));
        sc->setErrorDetection(0, "crc");
        unsigned char out2[8] = { 0 };
        sc->decode_vector(llr, out2);
        if ((out2[0] & 0xF0) != 0xF0) return 9;
    } else {
        // without a GPU the product must fail loudly, never fall back
        const float llr[8] = { 1, 1, 1, 1, 1, 1, 1, 1 };
        unsigned char out[8];
        threw = false;
        try {
            d->decode_vector(llr, out);
        } catch (const std::runtime_error&) {
            threw = true;
        }
        if (!threw) return 10;
    }
    std::puts("cpp shim ok");
    return 0;
}
