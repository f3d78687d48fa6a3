// What a taken / not-taken scalar branch costs a wave that is alone on its SIMD (gfx950).   hipcc --offload-arch=gfx950 -O2 -o branch_cost branch_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(64) void k(uint64_t* out, uint32_t zero, uint32_t iters)
{
    float a = threadIdx.x, b = 1.0f, c = 2.0f, d = 3.0f;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %0\n\tv_fma_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if constexpr (MODE == 1) asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" ::"s"(zero) : "scc");          // taken, short
            if constexpr (MODE == 2) asm volatile("s_cmp_eq_u32 %0, 1\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" ::"s"(zero) : "scc");          // not taken
            if constexpr (MODE == 3) asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\t.rept 64\n\ts_nop 0\n\t.endr\n1:" ::"s"(zero) : "scc");  // taken, over 256 bytes
            if constexpr (MODE == 4) asm volatile("s_cmp_eq_u32 %0, 0\n\ts_nop 0" ::"s"(zero) : "scc");                                     // the compare alone
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a + b + c + d == 12345.0f) out[0] = 0;
}
int main()
{
    uint64_t* d;
    hipMalloc(&d, 8 * 1024);
    const uint32_t iters = 2000;
    const char* names[] = { "no branch", "taken (short)", "not taken", "taken (over 256 B)", "compare + nop" };
    for (int grid : { 1, 1024 }) {
        for (int m = 0; m < 5; ++m) {
            uint64_t h[1024];
            for (int rep = 0; rep < 2; ++rep) {
                if (m == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, d, 0u, iters);
                if (m == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, d, 0u, iters);
                if (m == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, d, 0u, iters);
                if (m == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64), 0, 0, d, 0u, iters);
                if (m == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(64), 0, 0, d, 0u, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, d, 8 * grid, hipMemcpyDeviceToHost);
            double s = 0;
            for (int i = 0; i < grid; ++i) s += (double)h[i];
            // s_memtime ticks at 100 MHz: report ticks per (4 fma + branch) group and, against mode 0, per branch
            printf("grid %4d  %-20s  %.3f memtime ticks per group of 4 fma (+ branch)\n", grid, names[m], s / grid / (iters * 16.0));
        }
    }
    return 0;
}
