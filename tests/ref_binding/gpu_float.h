// The reference-side binding of INTEGRATION.md section 1 as real code: a PolarCode::Decoding::Decoder whose decode()
// runs on the GPU through the C ABI of libpolar_amd.so.  Compiles against the reference's own headers
// (/root/reference/include/polarcode/decoding/decoder.h:40-180) -- a maintainer would drop it next to
// src/polarcode/decoding/scl_avx_float.cpp and select it in makeDecoder (decoder.cpp:54-87).
// Test infrastructure here (tests/ref_binding/): built only where /root/reference exists, never part of the product.
#pragma once
#include <polar_amd.h>
#include <polarcode/decoding/decoder.h>

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace PolarCode {
namespace Decoding {

class GpuFloat : public Decoder
{
    pcd_decoder* h_ = nullptr;
    std::vector<float> llr_;

public:
    GpuFloat(size_t N, size_t L, const std::vector<unsigned>& frozen, int device = 0)
    {
        initialize(N, frozen);  // Decoder::initialize decoder.cpp:115-119
        std::vector<uint32_t> f(frozen.begin(), frozen.end());
        if (pcd_create(N, L, f.data(), f.size(), "float", device, &h_) != PCD_OK) throw std::invalid_argument(pcd_last_error());
        mOutputContainer = new unsigned char[(N - frozen.size() + 7) / 8]();  // (released by ~Decoder)
        llr_.resize(N);
    }
    ~GpuFloat() override { pcd_destroy(h_); }
    void setSignal(const float* pLlr) override { std::copy(pLlr, pLlr + mBlockLength, llr_.begin()); }
    void setSystematic(bool s) override
    {
        Decoder::setSystematic(s);
        pcd_set_systematic(h_, s ? 1 : 0);
    }
    void setErrorDetection(ErrorDetection::Detector* d) override
    {
        Decoder::setErrorDetection(d);
        // getType() is "CRC" / "CRCNR"; the dummy detector (no check bits) is create(0, "crc") on both sides
        const unsigned bits = d->getCheckBitCount();
        if (pcd_set_error_detection(h_, bits, bits ? d->getType().c_str() : "crc") != PCD_OK) throw std::runtime_error(pcd_last_error());
    }
    bool decode() override  // the pure virtual of decoder.h:64
    {
        uint8_t ok = 0;
        if (pcd_decode_f32(h_, llr_.data(), 1, mOutputContainer, &ok, nullptr, PCD_MEM_HOST, nullptr)) throw std::runtime_error(pcd_last_error());
        return ok != 0;
    }
    size_t getListSize() override { return pcd_list_size(h_); }
    // new: whole batches without per-codeword overhead
    void decode_batch(const float* llr, size_t rows, uint8_t* out, uint8_t* ok)
    {
        if (pcd_decode_f32(h_, llr, rows, out, ok, nullptr, PCD_MEM_HOST, nullptr)) throw std::runtime_error(pcd_last_error());
    }
    pcd_decoder* handle() const { return h_; }
};

}  // namespace Decoding
}  // namespace PolarCode
