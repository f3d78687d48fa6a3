// C ABI of the batched puncturer (include/polar_amd.h, ppu_*): the reference's PolarCode::Puncturer behind
// pypolar.Puncturer (puncturer_python.cc:21-150).  Host logic only: position tables and launch geometry.
#include "../../include/polar_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "internal.h"
#include "puncturer.h"

struct ppu_puncturer {
    unsigned M = 0, N = 0;  // punctured / parent block length
    int device = 0;
    std::vector<uint32_t> positions;  // M ascending parent positions that survive
    bool dev_ready = false;
    int num_cu = 0;
    int32_t* d_pos = nullptr;  // M
    int32_t* d_inv = nullptr;  // N: index into the punctured row, -1 = punctured position
    void* d_in = nullptr;
    size_t d_in_cap = 0;
    void* d_out = nullptr;
    size_t d_out_cap = 0;
};

namespace {

using pamd::set_error;

#define HIP_TRY(expr)                                                                                \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return set_error(PCD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

int grow(void*& p, size_t& cap, size_t bytes)
{
    if (bytes <= cap && p) return PCD_OK;
    if (p) HIP_TRY(hipFree(p));
    p = nullptr;
    cap = 0;
    const size_t want = std::max<size_t>(bytes, 256);
    HIP_TRY(hipMalloc(&p, want));
    cap = want;
    return PCD_OK;
}

int ensure_device(ppu_puncturer* u, pamd::DeviceScope& scope)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return set_error(PCD_ERR_NO_DEVICE, "no HIP device: the puncturer has no CPU fallback");
    if (u->device < 0 || u->device >= count) return set_error(PCD_ERR_NO_DEVICE, "HIP device ordinal out of range");
    HIP_TRY(scope.enter(u->device));
    if (!u->dev_ready) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, u->device));
        u->num_cu = prop.multiProcessorCount;
        std::vector<int32_t> pos(u->positions.begin(), u->positions.end()), inv(u->N, -1);
        for (unsigned i = 0; i < u->M; ++i) inv[u->positions[i]] = (int32_t)i;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&u->d_pos), std::max<size_t>(4, pos.size() * 4)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&u->d_inv), inv.size() * 4));
        HIP_TRY(hipMemcpy(u->d_pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(u->d_inv, inv.data(), inv.size() * 4, hipMemcpyHostToDevice));
        u->dev_ready = true;
    }
    return PCD_OK;
}

// op: 0 puncture (N -> M elements), 1 depuncture (M -> N), 2 puncture packed bits (N/8 -> M/8 bytes)
int run(ppu_puncturer* u, int op, const void* in, size_t rows, void* out, unsigned elem_bytes, int mem, void* stream)
{
    if (!u) return set_error(PCD_ERR_INVALID_ARG, "null puncturer");
    if (rows && (!in || !out)) return set_error(PCD_ERR_INVALID_ARG, "null input/output pointer");
    if (rows > 0xFFFFFFFFull) return set_error(PCD_ERR_INVALID_ARG, "rows must fit in 32 bits");
    if (op == 2 && (u->M % 8 || u->N % 8)) return set_error(PCD_ERR_INVALID_ARG, "packed puncturing needs block lengths that are multiples of 8");
    pamd::DeviceScope scope;
    int rc = ensure_device(u, scope);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t in_cols = op == 1 ? u->M : u->N, out_cols = op == 1 ? u->N : u->M;
    const size_t in_bytes = op == 2 ? rows * (u->N / 8) : rows * in_cols * elem_bytes;
    const size_t out_bytes = op == 2 ? rows * (u->M / 8) : rows * out_cols * elem_bytes;
    const void* din = in;
    void* dout = out;
    if (mem == PCD_MEM_HOST) {
        if ((rc = grow(u->d_in, u->d_in_cap, in_bytes))) return rc;
        if ((rc = grow(u->d_out, u->d_out_cap, out_bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(u->d_in, in, in_bytes, hipMemcpyHostToDevice, st));
        din = u->d_in;
        dout = u->d_out;
    } else if (mem != PCD_MEM_DEVICE) {
        return set_error(PCD_ERR_INVALID_ARG, "mem must be 0 (host pointers) or 1 (device pointers)");
    }
    const unsigned grid = (unsigned)std::min<size_t>(rows ? (rows + 1) / 2 : 1, (size_t)u->num_cu * 4);  // (measured: 4 workgroups per CU 64 % of the HBM roofline, 8 58 %, 32 61 %)
    if (op == 2)
        HIP_TRY(pamd::launch_puncture_packed((const uint8_t*)din, (uint8_t*)dout, u->d_pos, (uint32_t)rows, u->N / 8, u->M / 8, grid, st));
    else
        HIP_TRY(pamd::launch_gather_rows(din, dout, op == 1 ? u->d_inv : u->d_pos, (uint32_t)rows, (uint32_t)in_cols,
                                         (uint32_t)out_cols, elem_bytes, grid, st));
    if (mem == PCD_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(out, u->d_out, out_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return PCD_OK;
}

}  // namespace

extern "C" {

int ppu_create(size_t block_length, const uint32_t* frozen, size_t n_frozen, int device, ppu_puncturer** out)
{
    if (!out) return set_error(PCD_ERR_INVALID_ARG, "null output handle");
    *out = nullptr;
    const size_t M = block_length;
    if (M < 1 || M > 32768) return set_error(PCD_ERR_INVALID_ARG, "blockLength must be in [1, 32768]");
    if (n_frozen && !frozen) return set_error(PCD_ERR_INVALID_ARG, "null frozen set");
    size_t N = 1;  // round_up_power_of_two puncturer.cpp:24-34
    while (N < M) N <<= 1;
    const size_t np = N - M;  // Puncturer::Puncturer puncturer.cpp:51-65
    if (np > n_frozen)
        return set_error(PCD_ERR_PUNCTURE, "Number of required puncturing positions exceeds frozen bit positions!");
    for (size_t i = 0; i < np; ++i) {
        if (frozen[i] >= N) return set_error(PCD_ERR_INVALID_ARG, "frozen position out of range");
        if (i && frozen[i] <= frozen[i - 1]) return set_error(PCD_ERR_INVALID_ARG, "frozen positions must be strictly ascending");
    }
    ppu_puncturer* u = new ppu_puncturer();
    u->M = (unsigned)M;
    u->N = (unsigned)N;
    u->device = device;
    u->positions.reserve(M);
    size_t f = 0;
    for (size_t i = 0; i < N; ++i) {  // inverse_set_difference puncturer.cpp:36-49
        if (f < np && frozen[f] == i) {
            ++f;
            continue;
        }
        u->positions.push_back((uint32_t)i);
    }
    *out = u;
    return PCD_OK;
}

void ppu_destroy(ppu_puncturer* u)
{
    if (!u) return;
    if (u->dev_ready || u->d_in || u->d_out) {
        pamd::DeviceScope scope;
        (void)scope.enter(u->device);
        (void)hipFree(u->d_pos);
        (void)hipFree(u->d_inv);
        (void)hipFree(u->d_in);
        (void)hipFree(u->d_out);
    }
    delete u;
}

size_t ppu_block_length(const ppu_puncturer* u) { return u ? u->M : 0; }
size_t ppu_parent_block_length(const ppu_puncturer* u) { return u ? u->N : 0; }

size_t ppu_output_positions(const ppu_puncturer* u, uint32_t* out, size_t cap)
{
    if (!u) return 0;
    if (out) std::memcpy(out, u->positions.data(), std::min(cap, u->positions.size()) * sizeof(uint32_t));
    return u->positions.size();
}

int ppu_puncture(ppu_puncturer* u, const void* in, size_t rows, void* out, unsigned elem_bytes, int mem, void* stream)
{
    if (elem_bytes != 1 && elem_bytes != 4 && elem_bytes != 8) return set_error(PCD_ERR_INVALID_ARG, "element size must be 1, 4 or 8 bytes");
    return run(u, 0, in, rows, out, elem_bytes, mem, stream);
}

int ppu_depuncture(ppu_puncturer* u, const void* in, size_t rows, void* out, unsigned elem_bytes, int mem, void* stream)
{
    if (elem_bytes != 1 && elem_bytes != 4 && elem_bytes != 8) return set_error(PCD_ERR_INVALID_ARG, "element size must be 1, 4 or 8 bytes");
    return run(u, 1, in, rows, out, elem_bytes, mem, stream);
}

int ppu_puncture_packed(ppu_puncturer* u, const uint8_t* in, size_t rows, uint8_t* out, int mem, void* stream)
{
    return run(u, 2, in, rows, out, 1, mem, stream);
}

}  // extern "C"
