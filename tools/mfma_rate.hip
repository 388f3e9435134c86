// tools/mfma_rate.hip -- microbenchmark: sustained issue rate of the matrix instructions the Hamming sweep can use.
// hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o tools/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define ITER 4096

template <int MODE, int CHAINS>
__global__ void __launch_bounds__(256) k(int *out, unsigned long long *clk)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
    v8i a8 = {1, 2, 3, 4, 5, 6, 7, (int)threadIdx.x}, b8 = {1, 2, 3, 4, 5, 6, 7, 8};
    v16i acc[CHAINS];
    v16f facc[CHAINS];
    for (int c = 0; c < CHAINS; c++)
        for (int i = 0; i < 16; i++) { acc[c][i] = 0; facc[c][i] = 0.f; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            asm volatile("" : "+v"(a), "+v"(b), "+v"(a8), "+v"(b8));  // operands opaque per iteration: nothing can be hoisted
            if (MODE == 0) {  // 4 dependent i8 MFMAs starting from C = 0 (one 32x32 tile over 128 bits)
                v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                z = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, z, 0, 0, 0);
                acc[c] = z;
                asm volatile("" : "+v"(acc[c]));
            }
            if (MODE == 1) {  // fp4 block-scaled 32x32x64 (scales 1.0): 2 dependent MFMAs = 128 bits
                v16f z = {0};
                z = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, z, 4, 4, 0, 127, 0, 127);
                z = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, z, 4, 4, 0, 127, 0, 127);
                facc[c] = z;
                asm volatile("" : "+v"(facc[c]));
            }
            if (MODE == 2) {  // i8 16x16x64: 2 dependent MFMAs = 128 bits for a 16x16 tile
                typedef int v4 __attribute__((ext_vector_type(4)));
                v4 z = {0, 0, 0, 0};
                z = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, z, 0, 0, 0);
                acc[c][0] = z[0]; acc[c][1] = z[1];
                asm volatile("" : "+v"(acc[c]));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c][0] + (int)facc[c][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <class F>
static double time_ms(F launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 3; r++) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}

int main()
{
    int *out; unsigned long long *clk;
    hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&clk, 16);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; blocks_per_cu++) {
        const int blocks = 256 * blocks_per_cu;
        const double tiles = (double)blocks * 4 * ITER * 2;  // waves x iterations x CHAINS(2)
        unsigned long long h[2];
        double ms = time_ms([&] { hipLaunchKernelGGL((k<0, 2>), dim3(blocks), dim3(256), 0, 0, out, clk); });
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double ghz = (double)h[0] / h[1] * 0.1;
        std::printf("i8 32x32x32 x4  %d wave/SIMD: %.3f ms, %.2f GHz, %.1f clk per 32x32x128b tile per SIMD, %.2f Tpairs/s\n", blocks_per_cu, ms, ghz,
                    ms * 1e-3 * ghz * 1e9 * 1024 / tiles, tiles * 1024 / (ms * 1e-3) / 1e12);
        ms = time_ms([&] { hipLaunchKernelGGL((k<1, 2>), dim3(blocks), dim3(256), 0, 0, out, clk); });
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        ghz = (double)h[0] / h[1] * 0.1;
        std::printf("fp4 32x32x64 x2 %d wave/SIMD: %.3f ms, %.2f GHz, %.1f clk per 32x32x128b tile per SIMD, %.2f Tpairs/s\n", blocks_per_cu, ms, ghz,
                    ms * 1e-3 * ghz * 1e9 * 1024 / tiles, tiles * 1024 / (ms * 1e-3) / 1e12);
        ms = time_ms([&] { hipLaunchKernelGGL((k<2, 2>), dim3(blocks), dim3(256), 0, 0, out, clk); });
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        ghz = (double)h[0] / h[1] * 0.1;
        std::printf("i8 16x16x64 x2  %d wave/SIMD: %.3f ms, %.2f GHz, %.1f clk per 16x16x128b tile per SIMD, %.2f Tpairs/s\n", blocks_per_cu, ms, ghz,
                    ms * 1e-3 * ghz * 1e9 * 1024 / tiles, tiles * 256 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
