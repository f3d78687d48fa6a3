// Batched puncturing / depuncturing for gfx950: the reference's PolarCode::Puncturer (puncturer.h:66-97,
// puncturer.cpp:69-89) applied to B rows per launch.  Pure data movement (HBM-bound):
//   puncture    out[r, i] = in[r, pos[i]]                       (M of the N parent positions survive, ascending)
//   depuncture  out[r, p] = inv[p] >= 0 ? in[r, inv[p]] : 0     (gather form: coalesced writes)
//   packed      bit i of the output row = bit pos[i] of the input row, MSB first
#include "puncturer.h"

namespace pamd {

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                          const int32_t* __restrict__ map, uint32_t rows,
                                                          uint32_t in_cols, uint32_t out_cols)
{
    // map[c] = source column of output column c, or -1 for zero.  One workgroup walks rows; threads walk columns.
    for (uint32_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const T* src = in + (size_t)r * in_cols;
        T* dst = out + (size_t)r * out_cols;
        for (uint32_t c = threadIdx.x; c < out_cols; c += blockDim.x) {
            const int32_t m = map[c];
            dst[c] = m >= 0 ? src[m] : T(0);
        }
    }
}

// Four output columns per thread and two rows per trip: 16-byte map loads and (for 4-byte elements) 16-byte stores,
// eight independent gathers in flight per thread.  out_cols % 4 == 0 and 16-byte aligned output rows.
template <typename T>
__global__ __launch_bounds__(256) void gather_rows4_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                           const int32_t* __restrict__ map, uint32_t rows,
                                                           uint32_t in_cols, uint32_t out_cols)
{
    struct alignas(4 * sizeof(T)) Vec {
        T v[4];
    };
    const uint32_t quads = out_cols >> 2;
    for (uint32_t r = 2u * blockIdx.x; r < rows; r += 2u * gridDim.x) {
        const bool two = r + 1u < rows;
        const T* s0 = in + (size_t)r * in_cols;
        const T* s1 = in + (size_t)(two ? r + 1u : r) * in_cols;
        Vec* d0 = reinterpret_cast<Vec*>(out + (size_t)r * out_cols);
        Vec* d1 = reinterpret_cast<Vec*>(out + (size_t)(r + 1u) * out_cols);
        for (uint32_t q = threadIdx.x; q < quads; q += blockDim.x) {
            const int4 m = reinterpret_cast<const int4*>(map)[q];
            Vec a, b;
            a.v[0] = m.x >= 0 ? s0[m.x] : T(0);
            a.v[1] = m.y >= 0 ? s0[m.y] : T(0);
            a.v[2] = m.z >= 0 ? s0[m.z] : T(0);
            a.v[3] = m.w >= 0 ? s0[m.w] : T(0);
            b.v[0] = m.x >= 0 ? s1[m.x] : T(0);
            b.v[1] = m.y >= 0 ? s1[m.y] : T(0);
            b.v[2] = m.z >= 0 ? s1[m.z] : T(0);
            b.v[3] = m.w >= 0 ? s1[m.w] : T(0);
            d0[q] = a;
            if (two) d1[q] = b;
        }
    }
}

__global__ __launch_bounds__(256) void puncture_packed_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                              const int32_t* __restrict__ pos, uint32_t rows,
                                                              uint32_t in_bytes, uint32_t out_bytes)
{
    for (uint32_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const uint8_t* src = in + (size_t)r * in_bytes;
        uint8_t* dst = out + (size_t)r * out_bytes;
        for (uint32_t b = threadIdx.x; b < out_bytes; b += blockDim.x) {
            uint32_t o = 0;
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i) {
                const uint32_t p = (uint32_t)pos[8u * b + i];
                o |= ((src[p >> 3] >> (7u - (p & 7u))) & 1u) << (7u - i);
            }
            dst[b] = (uint8_t)o;
        }
    }
}

hipError_t launch_gather_rows(const void* in, void* out, const int32_t* map, uint32_t rows, uint32_t in_cols,
                              uint32_t out_cols, unsigned elem_bytes, unsigned grid, hipStream_t st)
{
    if (rows == 0) return hipSuccess;
    const dim3 g(grid), b(256);
    const bool vec = (out_cols & 3u) == 0 && (reinterpret_cast<uintptr_t>(out) & (4u * elem_bytes - 1u)) == 0 &&
                     (reinterpret_cast<uintptr_t>(map) & 15u) == 0;
    if (vec) {
        switch (elem_bytes) {
        case 1:
            hipLaunchKernelGGL(gather_rows4_kernel<uint8_t>, g, b, 0, st, (const uint8_t*)in, (uint8_t*)out, map, rows, in_cols, out_cols);
            return hipGetLastError();
        case 4:
            hipLaunchKernelGGL(gather_rows4_kernel<float>, g, b, 0, st, (const float*)in, (float*)out, map, rows, in_cols, out_cols);
            return hipGetLastError();
        case 8:
            hipLaunchKernelGGL(gather_rows4_kernel<double>, g, b, 0, st, (const double*)in, (double*)out, map, rows, in_cols, out_cols);
            return hipGetLastError();
        default: return hipErrorInvalidValue;
        }
    }
    switch (elem_bytes) {
    case 1:
        hipLaunchKernelGGL(gather_rows_kernel<uint8_t>, g, b, 0, st, (const uint8_t*)in, (uint8_t*)out, map, rows, in_cols, out_cols);
        break;
    case 4:
        hipLaunchKernelGGL(gather_rows_kernel<float>, g, b, 0, st, (const float*)in, (float*)out, map, rows, in_cols, out_cols);
        break;
    case 8:
        hipLaunchKernelGGL(gather_rows_kernel<double>, g, b, 0, st, (const double*)in, (double*)out, map, rows, in_cols, out_cols);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_puncture_packed(const uint8_t* in, uint8_t* out, const int32_t* pos, uint32_t rows, uint32_t in_bytes,
                                  uint32_t out_bytes, unsigned grid, hipStream_t st)
{
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(puncture_packed_kernel, dim3(grid), dim3(256), 0, st, in, out, pos, rows, in_bytes, out_bytes);
    return hipGetLastError();
}

}  // namespace pamd
