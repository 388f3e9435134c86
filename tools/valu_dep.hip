// tools/valu_dep.hip -- how fast does ONE wave issue dependent / independent VALU chains on gfx950, and how does it
// scale with the number of resident waves per SIMD (1, 2, 4, 8)?  Prints cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
template <int OP, int CHAINS>
__global__ void __launch_bounds__(64) k(float *out, float seed, unsigned long long *clk)
{
    extern __shared__ float lds[];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; i++) f[i] = seed + i + threadIdx.x;
    uint32_t s = 0x3c003c00u;
    float fs = 1.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            float &x = f[u % CHAINS];
            if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(fs));
            if (OP == 1) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(s));
            if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(fs));
            if (OP == 3) {  // scan-like: hs chain (fma_mix x2) feeding sum chain (add, sub)
                asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(f[0]) : "v"(s));
                asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(f[0]) : "v"(s));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[1]) : "v"(f[0]));
                asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[1]) : "v"(f[2]));
            }
            if (OP == 4) asm volatile("v_dot2c_f32_f16 %0, 0x60965cac, %1" : "+v"(x) : "v"(s));
            if (OP == 5) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x) : "v"(s));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc += f[i];
    out[blockIdx.x * 64 + threadIdx.x] = acc + lds[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int OP, int CHAINS>
void run(const char *name, int waves_per_simd)
{
    float *out; unsigned long long *clk;
    hipMalloc(&out, 1024 * 64 * 64 * 4); hipMalloc(&clk, 8);
    // LDS per block chosen so that exactly waves_per_simd*4 blocks fit a CU (160 KB LDS; granularity ignored)
    size_t lds = (size_t)(160 * 1024) / (waves_per_simd * 4) - 512;
    if (lds > 65536) lds = 65536;
    hipFuncSetAttribute((const void *)k<OP, CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int blocks = 256 * 4 * waves_per_simd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(64), lds, 0, out, 1.0f, clk);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(64), lds, 0, out, 1.0f, clk);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    int per_iter = OP == 3 ? 64 : 16;
    // s_memtime counts at 100 MHz on gfx950? report both wall-based and counter-based
    double inst_per_wave = (double)ITER * per_iter;
    double ns_per_inst_simd = ms * 1e6 / (inst_per_wave * waves_per_simd);
    std::printf("%-28s chains=%d waves/SIMD=%d  %.3f ms  %.2f ns per instr per SIMD (= %.2f clk at 2.4 GHz), memtime/instr/wave %.2f\n", name, CHAINS,
                waves_per_simd, ms, ns_per_inst_simd, ns_per_inst_simd * 2.4, (double)c / inst_per_wave);
    hipFree(out); hipFree(clk);
}
int main()
{
    for (int w : {1, 2, 3, 4, 6, 8}) {
        run<0, 1>("v_add_f32 dependent", w);
        run<0, 8>("v_add_f32 independent", w);
        run<1, 1>("v_fma_mix_f32 dependent", w);
        run<1, 8>("v_fma_mix_f32 independent", w);
        run<2, 1>("v_fma_f32 dependent", w);
        run<4, 1>("v_dot2c dependent", w);
        run<4, 8>("v_dot2c independent", w);
        run<5, 8>("v_pk_add_f16 independent", w);
        run<3, 1>("scan-like 2mix+add+sub", w);
    }
    return 0;
}
