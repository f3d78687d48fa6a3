// C ABI of the batched polar encoder (include/polar_amd.h, pce_*): the reference's
// PolarCode::Encoding::ButterflyFipPacked behind pypolar.PolarEncoder (encoder_python.cc:21-61).
// Host logic only: argument checking, bit-placement and detector tables, launch geometry.
#include "../../include/polar_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "encoder.h"
#include "internal.h"
#include "program.h"

struct pce_encoder {
    unsigned N = 0, logN = 0, K = 0, nb = 0;
    int device = 0;
    bool systematic = true;  // Encoder::Encoder encoder.cpp:17-26
    std::vector<uint32_t> frozen;
    pamd::Detector det;      // starts as the dummy detector (encoder.cpp:18), unlike the decoder's CRC-8
    std::vector<pamd::DepositWord> dep;

    bool dev_ready = false, det_dirty = true;
    int num_cu = 0;
    pamd::DepositWord* d_dep = nullptr;
    uint32_t* d_gen = nullptr;
    size_t d_gen_cap = 0;
    uint8_t* d_info = nullptr;
    size_t d_info_cap = 0;
    uint8_t* d_code = nullptr;
    size_t d_code_cap = 0;
};

namespace {

using pamd::set_error;

#define HIP_TRY(expr)                                                                                \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return set_error(PCD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

template <typename T>
int grow(T*& p, size_t& cap, size_t bytes)
{
    if (bytes <= cap && p) return PCD_OK;
    if (p) HIP_TRY(hipFree(p));
    p = nullptr;
    cap = 0;
    const size_t want = std::max<size_t>(bytes, 256);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p), want));
    cap = want;
    return PCD_OK;
}

int ensure_device(pce_encoder* e, pamd::DeviceScope& scope)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return set_error(PCD_ERR_NO_DEVICE, "no HIP device: the encoder has no CPU fallback");
    if (e->device < 0 || e->device >= count) return set_error(PCD_ERR_NO_DEVICE, "HIP device ordinal out of range");
    HIP_TRY(scope.enter(e->device));
    if (!e->dev_ready) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, e->device));
        e->num_cu = prop.multiProcessorCount;
        const size_t bytes = e->dep.size() * sizeof(pamd::DepositWord);
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_dep), bytes));
        HIP_TRY(hipMemcpy(e->d_dep, e->dep.data(), bytes, hipMemcpyHostToDevice));
        e->dev_ready = true;
    }
    if (e->det_dirty) {
        if (e->det.size != 0) {
            // generator by nibble of the little-endian information dwords: bit q of nibble j of dword k is bit
            // p = 4 (j & 1) + q of byte B = 4k + j/2, i.e. information bit t = 8B + 7 - p (packed MSB first)
            const std::vector<uint32_t> gen = e->det.generator_table((int)e->nb);
            const unsigned nbw = (e->nb + 3) / 4;
            std::vector<uint32_t> gen4((size_t)nbw * 128, 0u);
            for (unsigned k = 0; k < nbw; ++k)
                for (unsigned j = 0; j < 8; ++j)
                    for (unsigned val = 1; val < 16; ++val) {
                        uint32_t x = 0;
                        for (unsigned q = 0; q < 4; ++q) {
                            if (!((val >> q) & 1u)) continue;
                            const unsigned t = 8 * (4 * k + j / 2) + 7 - (4 * (j & 1) + q);
                            if (t < gen.size()) x ^= gen[t];
                        }
                        gen4[((size_t)k * 8 + j) * 16 + val] = x;
                    }
            int rc = grow(e->d_gen, e->d_gen_cap, gen4.size() * sizeof(uint32_t));
            if (rc) return rc;
            HIP_TRY(hipMemcpy(e->d_gen, gen4.data(), gen4.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        e->det_dirty = false;
    }
    return PCD_OK;
}

int encode_device(pce_encoder* e, uint8_t* info, size_t rows, uint8_t* code, hipStream_t st)
{
    if (rows == 0) return PCD_OK;
    if (rows > 0xFFFFFFFFull / 64) return set_error(PCD_ERR_CAPACITY, "too many rows for one call");
    pamd::EncodeParams P{};
    P.info = info;
    P.code = code;
    P.rows = (uint32_t)rows;
    P.N = e->N;
    P.logN = e->logN;
    P.K = e->K;
    P.nb = e->nb;
    P.dep = e->d_dep;
    P.gen4 = e->det.size ? e->d_gen : nullptr;
    P.field_off = e->det.size ? e->det.field_offset((int)e->nb) : 0;
    P.field_bytes = e->det.check_bytes();
    P.systematic = e->systematic ? 1u : 0u;
    P.iw_words = (e->nb + 3) / 4 + 2;
    P.info_aligned = ((reinterpret_cast<uintptr_t>(info) & 3u) == 0 && (e->nb & 3u) == 0) ? 1u : 0u;
    if ((reinterpret_cast<uintptr_t>(code) & 15u) != 0)
        return set_error(PCD_ERR_INVALID_ARG, "code buffer must be 16-byte aligned");
    const unsigned G = pamd::encode_codewords_per_wave(e->N), wpb = 4;
    const size_t groups = (rows + G - 1) / G;
    const size_t want = (groups + wpb - 1) / wpb;
    const unsigned grid = (unsigned)std::min<size_t>(want, (size_t)e->num_cu * 16);
    HIP_TRY(pamd::launch_encode(P, grid, wpb, st));
    return PCD_OK;
}

}  // namespace

extern "C" {

int pce_create(size_t block_length, const uint32_t* frozen, size_t n_frozen, int device, pce_encoder** out)
{
    if (!out) return set_error(PCD_ERR_INVALID_ARG, "null output handle");
    *out = nullptr;
    const size_t N = block_length;
    if (N < 8 || (N & (N - 1)) != 0 || N > 32768)
        return set_error(PCD_ERR_INVALID_ARG, "blockLength must be a power of two in [8, 32768]");
    if (n_frozen >= N) return set_error(PCD_ERR_INVALID_ARG, "no information bits left");
    if (n_frozen && !frozen) return set_error(PCD_ERR_INVALID_ARG, "null frozen set");
    for (size_t i = 0; i < n_frozen; ++i) {
        if (frozen[i] >= N) return set_error(PCD_ERR_INVALID_ARG, "frozen position out of range");
        if (i && frozen[i] <= frozen[i - 1])
            return set_error(PCD_ERR_INVALID_ARG, "frozen positions must be strictly ascending");
    }
    pce_encoder* e = new pce_encoder();
    e->N = (unsigned)N;
    while ((1u << e->logN) < e->N) ++e->logN;
    e->frozen.assign(frozen, frozen + n_frozen);
    e->K = (unsigned)(N - n_frozen);
    e->nb = (e->K + 7) / 8;  // Encoder::informationByteSize encoder.h
    e->device = device;
    e->det.size = 0;
    e->dep = pamd::deposit_table(e->N, e->frozen);
    // self-check of the log-step bit deposit against the bit-by-bit definition
    for (const pamd::DepositWord& d : e->dep) {
        const uint32_t pats[3] = { 0xFFFFFFFFu, 0xA5C3E169u, 0x0F1E2D3Cu };
        for (uint32_t x : pats) {
            uint32_t want = 0, src = x;
            for (int b = 0; b < 32; ++b)
                if ((d.mask >> b) & 1u) {
                    want |= (src & 1u) << b;
                    src >>= 1;
                }
            if (pamd::deposit_apply(d, x) != want) {
                delete e;
                return set_error(PCD_ERR_INVALID_ARG, "internal error: bit-deposit table self-check failed");
            }
        }
    }
    *out = e;
    return PCD_OK;
}

void pce_destroy(pce_encoder* e)
{
    if (!e) return;
    if (e->dev_ready || e->d_gen || e->d_info || e->d_code) {
        pamd::DeviceScope scope;
        (void)scope.enter(e->device);
        (void)hipFree(e->d_dep);
        (void)hipFree(e->d_gen);
        (void)hipFree(e->d_info);
        (void)hipFree(e->d_code);
    }
    delete e;
}

int pce_set_error_detection(pce_encoder* e, unsigned size, const char* type)
{
    if (!e || !type) return set_error(PCD_ERR_INVALID_ARG, "null argument");
    pamd::Detector det;
    std::string err;
    const int rc = pamd::make_detector(size, type, det, err);
    if (rc) return set_error(rc, err);
    e->det = det;
    e->det_dirty = true;
    return PCD_OK;
}

int pce_set_systematic(pce_encoder* e, int systematic)
{
    if (!e) return set_error(PCD_ERR_INVALID_ARG, "null encoder");
    e->systematic = systematic != 0;
    return PCD_OK;
}

int pce_is_systematic(const pce_encoder* e) { return e && e->systematic; }
size_t pce_block_length(const pce_encoder* e) { return e ? e->N : 0; }
size_t pce_info_length(const pce_encoder* e) { return e ? e->K : 0; }
size_t pce_info_bytes(const pce_encoder* e) { return e ? e->nb : 0; }
size_t pce_code_bytes(const pce_encoder* e) { return e ? e->N / 8 : 0; }

size_t pce_frozen_bits(const pce_encoder* e, uint32_t* out, size_t cap)
{
    if (!e) return 0;
    if (out) std::memcpy(out, e->frozen.data(), std::min(cap, e->frozen.size()) * sizeof(uint32_t));
    return e->frozen.size();
}

int pce_error_detection_mode(const pce_encoder* e, char* buf, size_t cap)
{
    if (!e || !buf || !cap) return set_error(PCD_ERR_INVALID_ARG, "null argument");
    std::snprintf(buf, cap, "%s-%u", e->det.type_name().c_str(), e->det.size);
    return PCD_OK;
}

int pce_encode(pce_encoder* e, uint8_t* info, size_t rows, uint8_t* code, int mem, void* stream)
{
    if (!e) return set_error(PCD_ERR_INVALID_ARG, "null encoder");
    if (rows && (!info || !code)) return set_error(PCD_ERR_INVALID_ARG, "null info/code pointer");
    if (e->det.size != 0 && e->nb < e->det.check_bytes())
        return set_error(PCD_ERR_INVALID_ARG, "information word shorter than the detector's checksum field");
    if (e->det.size == 32 && e->nb < 4) return set_error(PCD_ERR_INVALID_ARG, "CRC-32 needs at least 4 information bytes");
    pamd::DeviceScope scope;
    int rc = ensure_device(e, scope);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (mem == PCD_MEM_DEVICE) return encode_device(e, info, rows, code, st);
    if (mem != PCD_MEM_HOST) return set_error(PCD_ERR_INVALID_ARG, "mem must be 0 (host pointers) or 1 (device pointers)");
    const size_t in_bytes = rows * e->nb, out_bytes = rows * (e->N / 8);
    if ((rc = grow(e->d_info, e->d_info_cap, in_bytes))) return rc;
    if ((rc = grow(e->d_code, e->d_code_cap, out_bytes))) return rc;
    HIP_TRY(hipMemcpyAsync(e->d_info, info, in_bytes, hipMemcpyHostToDevice, st));
    if ((rc = encode_device(e, e->d_info, rows, e->d_code, st))) return rc;
    HIP_TRY(hipMemcpyAsync(code, e->d_code, out_bytes, hipMemcpyDeviceToHost, st));
    // Detector::generate works in place on the caller's buffer (encoder.cpp:79-83 -> butterfly_fip_packed.cpp:47-48)
    if (e->det.size != 0) HIP_TRY(hipMemcpyAsync(info, e->d_info, in_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PCD_OK;
}

}  // extern "C"
