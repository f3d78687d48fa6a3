// The reference-side binding of INTEGRATION.md section 1 as real code: a PolarCode::Decoding::Decoder whose decode()
// runs on the GPU through the C ABI of libpolar_amd.so.  Compiles against the reference's own headers
// (/root/reference/include/polarcode/decoding/decoder.h:40-180) -- a maintainer would drop it next to
// src/polarcode/decoding/scl_avx_float.cpp and select it in makeDecoder (decoder.cpp:54-87).
// Test infrastructure here (tests/ref_binding/): built only where /root/reference exists, never part of the product.
#pragma once
#include <polar_amd.h>
#include <polarcode/bitcontainer.h>
#include <polarcode/decoding/decoder.h>

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace PolarCode {
namespace Decoding {

// type "float" / "mixed": float arithmetic on the GPU.  The reference's setSignal(const float*) / the NON-virtual
// setSignal(const char*) (decoder.h:155-161, decoder.cpp:140-142) both fill mLlrContainer, so the subclass owns a real
// FloatContainer (bitcontainer.h:169-200) and needs no override: int8 rows are widened there exactly as in the reference's own
// float decoders (FloatContainer::insertLlr(const char*) bitcontainer.cpp:202-207); decode() hands the container's row to the C ABI.
class GpuFloat : public Decoder
{
protected:
    pcd_decoder* h_ = nullptr;

    GpuFloat() {}
    void open(size_t N, size_t L, const std::vector<unsigned>& frozen, const char* type, int device)
    {
        initialize(N, frozen);  // Decoder::initialize decoder.cpp:115-119
        std::vector<uint32_t> f(frozen.begin(), frozen.end());
        if (pcd_create(N, L, f.data(), f.size(), type, device, &h_) != PCD_OK) throw std::invalid_argument(pcd_last_error());
        mOutputContainer = new unsigned char[(N - frozen.size() + 7) / 8]();  // (released by ~Decoder, like the container)
    }

public:
    GpuFloat(size_t N, size_t L, const std::vector<unsigned>& frozen, int device = 0, const char* type = "float")
    {
        open(N, L, frozen, type, device);
        mLlrContainer = new FloatContainer(N);
    }
    ~GpuFloat() override { pcd_destroy(h_); }
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
        const float* row = static_cast<FloatContainer*>(mLlrContainer)->data();
        if (pcd_decode_f32(h_, row, 1, mOutputContainer, &ok, nullptr, PCD_MEM_HOST, nullptr)) throw std::runtime_error(pcd_last_error());
        return ok != 0;
    }
    size_t getListSize() override { return pcd_list_size(h_); }
    // new: whole batches without per-codeword overhead
    void decode_batch(const float* llr, size_t rows, uint8_t* out, uint8_t* ok)
    {
        if (pcd_decode_f32(h_, llr, rows, out, ok, nullptr, PCD_MEM_HOST, nullptr)) throw std::runtime_error(pcd_last_error());
    }
    void decode_batch(const char* llr, size_t rows, uint8_t* out, uint8_t* ok)
    {
        if (pcd_decode_i8(h_, reinterpret_cast<const int8_t*>(llr), rows, out, ok, nullptr, PCD_MEM_HOST, nullptr)) throw std::runtime_error(pcd_last_error());
    }
    pcd_decoder* handle() const { return h_; }
};

// type "char" (the factory's default branch, decoder.cpp:65-67, 80-82: FastSscFipChar / SclFipChar): int8 arithmetic.  The
// container is a CharContainer (bitcontainer.h:211-239) -- float rows are quantised there exactly as the reference does
// (CharContainer::insertLlr(const float*) bitcontainer.cpp:505-516), int8 rows are copied --, decode() calls pcd_decode_i8.
class GpuChar : public GpuFloat
{
public:
    GpuChar(size_t N, size_t L, const std::vector<unsigned>& frozen, int device = 0)
    {
        open(N, L, frozen, "char", device);
        mLlrContainer = new CharContainer(N);
    }
    bool decode() override
    {
        uint8_t ok = 0;
        const int8_t* row = reinterpret_cast<const int8_t*>(static_cast<CharContainer*>(mLlrContainer)->data());
        if (pcd_decode_i8(h_, row, 1, mOutputContainer, &ok, nullptr, PCD_MEM_HOST, nullptr)) throw std::runtime_error(pcd_last_error());
        return ok != 0;
    }
};

}  // namespace Decoding
}  // namespace PolarCode
