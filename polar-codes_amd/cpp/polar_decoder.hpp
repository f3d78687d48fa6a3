// C++ host-side mirror of the reference's decoder and encoder interfaces over the C ABI (include/polar_amd.h).
//
// Mirrors PolarCode::Decoding::Decoder (reference: include/polarcode/decoding/decoder.h:40-180) and the factory
// PolarCode::Decoding::create (decoder.cpp:26-52) for the float / mixed decoders: same method names, argument
// meaning and exception types, plus decode_batch for B codewords per call.  Header-only; link libpolar_amd.so.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/polar_amd.h"

namespace PolarAmd {
namespace Decoding {

class Decoder {
public:
    Decoder(size_t blockLength, size_t listSize, const std::vector<unsigned>& frozenBits,
            const std::string& decoderType, int device = 0)
    {
        std::vector<uint32_t> f(frozenBits.begin(), frozenBits.end());
        check(pcd_create(blockLength, listSize, f.data(), f.size(), decoderType.c_str(), device, &h_));
    }
    ~Decoder() { pcd_destroy(h_); }
    Decoder(const Decoder&) = delete;
    Decoder& operator=(const Decoder&) = delete;

    // Decoder::decode_vector(const float*, void*) decoder.cpp:154-167 -- returns the detector verdict
    bool decode_vector(const float* pLlr, void* pData)
    {
        uint8_t ok = 0;
        check(pcd_decode_f32(h_, pLlr, 1, static_cast<uint8_t*>(pData), &ok, nullptr, PCD_MEM_HOST, nullptr));
        return ok != 0;
    }
    // Decoder::decode_vector(const char*, void*) decoder.cpp:169-181
    bool decode_vector(const char* pLlr, void* pData)
    {
        uint8_t ok = 0;
        check(pcd_decode_i8(h_, reinterpret_cast<const int8_t*>(pLlr), 1, static_cast<uint8_t*>(pData), &ok, nullptr,
                            PCD_MEM_HOST, nullptr));
        return ok != 0;
    }
    // batched form (no reference counterpart): rows x N LLRs -> rows x (K+7)/8 bytes, rows verdicts
    void decode_batch(const float* llr, size_t rows, uint8_t* out, uint8_t* ok = nullptr, float* metrics = nullptr,
                      bool device_pointers = false, void* hip_stream = nullptr)
    {
        check(pcd_decode_f32(h_, llr, rows, out, ok, metrics, device_pointers ? PCD_MEM_DEVICE : PCD_MEM_HOST, hip_stream));
    }

    size_t blockLength() const { return pcd_block_length(h_); }
    size_t infoLength() const { return pcd_info_length(h_); }
    size_t getListSize() const { return pcd_list_size(h_); }
    void setSystematic(bool sys) { check(pcd_set_systematic(h_, sys ? 1 : 0)); }
    bool isSystematic() const { return pcd_is_systematic(h_) != 0; }
    std::vector<unsigned> frozenBits() const
    {
        std::vector<uint32_t> f(pcd_frozen_bits(h_, nullptr, 0));
        pcd_frozen_bits(h_, f.data(), f.size());
        return std::vector<unsigned>(f.begin(), f.end());
    }
    // Decoder::setErrorDetection(ErrorDetection::create(size, type))
    void setErrorDetection(unsigned size = 0, const std::string& type = "crc")
    {
        check(pcd_set_error_detection(h_, size, type.c_str()));
    }
    std::string getErrorDetectionMode() const
    {
        char buf[64];
        pcd_error_detection_mode(h_, buf, sizeof buf);
        return buf;
    }
    pcd_decoder* handle() { return h_; }

private:
    static void check(int rc)
    {
        if (rc == PCD_OK) return;
        const std::string msg = pcd_last_error();
        switch (rc) {
        case PCD_ERR_UNKNOWN_TYPE:
        case PCD_ERR_CRC_SIZE: throw std::logic_error(msg);           // as the reference does
        case PCD_ERR_FROZEN_SET:
        case PCD_ERR_INVALID_ARG: throw std::invalid_argument(msg);
        default: throw std::runtime_error(msg);
        }
    }
    pcd_decoder* h_ = nullptr;
};

// PolarCode::Decoding::create(blockLength, listSize, frozenBits, decoderType) -- owning raw pointer, like the reference
inline Decoder* create(size_t blockLength, size_t listSize, const std::vector<unsigned>& frozenBits,
                       std::string decoderType)
{
    return new Decoder(blockLength, listSize, frozenBits, decoderType);
}

}  // namespace Decoding

namespace Encoding {

// Mirrors PolarCode::Encoding::Encoder / ButterflyFipPacked (reference: include/polarcode/encoding/encoder.h,
// src/polarcode/encoding/butterfly_fip_packed.cpp:27-59): same method names and argument meaning.
class Encoder {
public:
    Encoder(size_t blockLength, const std::vector<unsigned>& frozenBits, int device = 0)
    {
        std::vector<uint32_t> f(frozenBits.begin(), frozenBits.end());
        check(pce_create(blockLength, f.data(), f.size(), device, &h_));
    }
    ~Encoder() { pce_destroy(h_); }
    Encoder(const Encoder&) = delete;
    Encoder& operator=(const Encoder&) = delete;

    // Encoder::encode_vector(void* pInfo, void* pCode) encoder.cpp:79-91; the checksum is written into pInfo
    void encode_vector(void* pInfo, void* pCode)
    {
        check(pce_encode(h_, static_cast<uint8_t*>(pInfo), 1, static_cast<uint8_t*>(pCode), PCD_MEM_HOST, nullptr));
    }
    // batched form (no reference counterpart): rows x (K+7)/8 bytes -> rows x N/8 bytes
    void encode_batch(uint8_t* info, size_t rows, uint8_t* code, bool device_pointers = false, void* hip_stream = nullptr)
    {
        check(pce_encode(h_, info, rows, code, device_pointers ? PCD_MEM_DEVICE : PCD_MEM_HOST, hip_stream));
    }

    size_t blockLength() const { return pce_block_length(h_); }
    size_t infoLength() const { return pce_info_length(h_); }
    void setSystematic(bool sys) { check(pce_set_systematic(h_, sys ? 1 : 0)); }
    bool isSystematic() const { return pce_is_systematic(h_) != 0; }
    std::vector<unsigned> frozenBits() const
    {
        std::vector<uint32_t> f(pce_frozen_bits(h_, nullptr, 0));
        pce_frozen_bits(h_, f.data(), f.size());
        return std::vector<unsigned>(f.begin(), f.end());
    }
    void setErrorDetection(unsigned size = 0, const std::string& type = "crc")
    {
        check(pce_set_error_detection(h_, size, type.c_str()));
    }
    std::string getErrorDetectionMode() const
    {
        char buf[64];
        pce_error_detection_mode(h_, buf, sizeof buf);
        return buf;
    }
    pce_encoder* handle() { return h_; }

private:
    static void check(int rc)
    {
        if (rc == PCD_OK) return;
        const std::string msg = pcd_last_error();
        switch (rc) {
        case PCD_ERR_CRC_SIZE: throw std::logic_error(msg);
        case PCD_ERR_INVALID_ARG: throw std::invalid_argument(msg);
        default: throw std::runtime_error(msg);
        }
    }
    pce_encoder* h_ = nullptr;
};

}  // namespace Encoding

// Mirrors PolarCode::Puncturer (include/polarcode/puncturer.h, src/polarcode/puncturer.cpp): same method names.
class Puncturer {
public:
    Puncturer(size_t blockLength, const std::vector<unsigned>& frozenBitPositions, int device = 0)
    {
        std::vector<uint32_t> f(frozenBitPositions.begin(), frozenBitPositions.end());
        const int rc = ppu_create(blockLength, f.data(), f.size(), device, &h_);
        if (rc == PCD_ERR_PUNCTURE) throw std::out_of_range(pcd_last_error());  // as the reference does
        if (rc == PCD_ERR_INVALID_ARG) throw std::invalid_argument(pcd_last_error());
        if (rc) throw std::runtime_error(pcd_last_error());
    }
    ~Puncturer() { ppu_destroy(h_); }
    Puncturer(const Puncturer&) = delete;
    Puncturer& operator=(const Puncturer&) = delete;

    size_t blockLength() const { return ppu_block_length(h_); }
    size_t parentBlockLength() const { return ppu_parent_block_length(h_); }
    std::vector<unsigned> blockOutputPositions() const
    {
        std::vector<uint32_t> p(ppu_output_positions(h_, nullptr, 0));
        ppu_output_positions(h_, p.data(), p.size());
        return std::vector<unsigned>(p.begin(), p.end());
    }
    template <typename T>
    void puncture(T* pOutput, const T* pInput, size_t rows = 1, bool device_pointers = false, void* hip_stream = nullptr)
    {
        static_assert(sizeof(T) == 1 || sizeof(T) == 4 || sizeof(T) == 8, "uint8, float or double elements");
        check(ppu_puncture(h_, pInput, rows, pOutput, sizeof(T), device_pointers ? PCD_MEM_DEVICE : PCD_MEM_HOST, hip_stream));
    }
    template <typename T>
    void depuncture(T* pOutput, const T* pInput, size_t rows = 1, bool device_pointers = false, void* hip_stream = nullptr)
    {
        static_assert(sizeof(T) == 1 || sizeof(T) == 4 || sizeof(T) == 8, "uint8, float or double elements");
        check(ppu_depuncture(h_, pInput, rows, pOutput, sizeof(T), device_pointers ? PCD_MEM_DEVICE : PCD_MEM_HOST, hip_stream));
    }
    void puncturePacked(unsigned char* pOutput, const unsigned char* pInput, size_t rows = 1)
    {
        check(ppu_puncture_packed(h_, pInput, rows, pOutput, PCD_MEM_HOST, nullptr));
    }

private:
    static void check(int rc)
    {
        if (rc == PCD_OK) return;
        if (rc == PCD_ERR_INVALID_ARG) throw std::invalid_argument(pcd_last_error());
        throw std::runtime_error(pcd_last_error());
    }
    ppu_puncturer* h_ = nullptr;
};

}  // namespace PolarAmd
