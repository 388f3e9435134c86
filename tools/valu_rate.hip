// tools/valu_rate.hip -- microbenchmark: sustained wave64 VALU instruction rate on gfx950 for the opcodes the
// kernels of this repo are made of.  Prints lane-ops/clk/SIMD (256 CUs x 4 SIMDs assumed, clock from the kernel's
// own s_memtime / s_memrealtime ratio).   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 2048
#define UNROLL 16

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed, unsigned long long *clk)
{
    uint32_t a[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B1u; f[i] = (float)(a[i] & 1023); }
    uint32_t s = seed ^ 0x55AA55AAu;
    float fs = 1.0009765625f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL / 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(s));
                if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(s));
                if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(fs));
                if (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fs));
                if (OP == 4) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
                if (OP == 5) asm volatile("v_dot2_f32_f16 %0, %1, %1, %0" : "+v"(f[i]) : "v"(s));
                if (OP == 6) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (OP == 7) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(f[i]) : "v"(s));
                if (OP == 8) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[i]) : "v"(a[i]));
                if (OP == 9) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
                if (OP == 10) asm volatile("v_trunc_f32 %0, %0" : "+v"(f[i]));
                if (OP == 11) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %1" : "=v"(a[i]) : "v"(f[i]));
                if (OP == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fs));
                if (OP == 13) asm volatile("v_dot2c_f32_f16 %0, 0x60965cac, %1" : "+v"(f[i]) : "v"(s));
                if (OP == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(s));
                if (OP == 15) asm volatile("v_dot2_f32_f16 %0, %1, %2, 0" : "=v"(f[i]) : "v"(s), "s"(seed));
                if (OP == 16) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(f[i]) : "v"(fs));
                if (OP == 17) asm volatile("v_fmamk_f32 %0, %1, 0x3a83126f, %0" : "+v"(f[i]) : "v"(fs));
                if (OP == 18) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fs));
                if (OP == 19) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(s));
                if (OP == 20) asm volatile("v_dot2_f32_f16 %0, %1, %1, 0" : "=v"(f[i]) : "v"(s));
                if (OP == 21) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(f[i]) : "s"(seed));
                if (OP == 22) asm volatile("v_mov_b32 %0, 0" : "=v"(a[i]));
                if (OP == 23) asm volatile("v_pk_add_f16 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a[i]) : "s"(seed));
                if (OP == 24) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(s));
                if (OP == 25) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc += a[i] + (uint32_t)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int OP>
__global__ void __launch_bounds__(256) kpk(uint32_t *out, uint32_t seed)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 f[8];
#pragma unroll
    for (int i = 0; i < 8; i++) f[i] = f2{(float)(threadIdx.x + i), 1.0f};
    f2 fs = f2{1.0009765625f, 0.9990234375f};
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL / 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(fs));
                if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fs));
                if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fs));
            }
        }
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc += f[i].x + f[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)acc;
}

template <class F>
static double time_ms(F launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main()
{
    const int blocks = 256 * 8, threads = 256;  // 8 blocks x 4 waves = 32 waves per CU = 8 per SIMD
    uint32_t *out; unsigned long long *clk;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipMalloc(&clk, 16);
    const char *names[] = {"v_xor_b32", "v_bcnt_u32_b32", "v_fma_f32", "v_add_f32", "v_perm_b32", "v_dot2_f32_f16", "v_pk_add_f16",
                           "v_fma_mix_f32", "v_cvt_f32_ubyte0", "v_min3_u32", "v_trunc_f32", "v_cvt_pkrtz_f16_f32", "v_mul_f32",
                           "v_dot2c_f32_f16 literal", "v_mov_b32 v", "v_dot2_f32_f16 v,s,0", "v_fmac_f32", "v_fmamk_f32", "v_sub_f32", "v_add_u32",
                           "v_dot2_f32_f16 v,v,0", "v_fma_f32 v,s,0.5", "v_mov_b32 0", "v_pk_add_f16 v,s", "v_lshl_add_u32", "v_cndmask_b32"};
    const double insts = (double)blocks * (threads / 64) * ITER * UNROLL;  // wave-instructions
    double ghz = 0;
#define RUN(OP)                                                                                                      \
    {                                                                                                                \
        double ms = time_ms([&] { hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 12345u, clk); }); \
        unsigned long long h[2];                                                                                     \
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);                                                                \
        ghz = (double)h[0] / (double)h[1] * 0.1;                                                                     \
        double per_simd_clk = insts * 64 / (ms * 1e-3) / (ghz * 1e9) / 1024.0;                                       \
        std::printf("%-22s %8.3f ms  clock %.2f GHz  %6.2f lanes/clk/SIMD  (%.1f T lane-op/s)\n", names[OP], ms, ghz, per_simd_clk, insts * 64 / (ms * 1e-3) / 1e12); \
    }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23) RUN(24) RUN(25)
    const char *pk[] = {"v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32"};
#define RUNPK(OP)                                                                                                    \
    {                                                                                                                \
        double ms = time_ms([&] { hipLaunchKernelGGL(kpk<OP>, dim3(blocks), dim3(threads), 0, 0, out, 12345u); });   \
        double per_simd_clk = insts * 64 / (ms * 1e-3) / (ghz * 1e9) / 1024.0;                                       \
        std::printf("%-22s %8.3f ms  (clock of last run) %6.2f lane-instr/clk/SIMD (x2 elements)\n", pk[OP], ms, per_simd_clk); \
    }
    RUNPK(0) RUNPK(1) RUNPK(2)
    return 0;
}
