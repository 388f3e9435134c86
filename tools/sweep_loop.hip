// tools/sweep_loop.hip -- experiment bench for the inner loop of hamming_mfma_kernel<FmtFp4, 4>: how should one wave order its
// four MFMAs (two 32x32 tiles x two 64-bit slices) and the 16-instruction max tree of a tile pair, and how does that interact with
// the number of co-resident waves per SIMD?  Same instruction mix as the real kernel (A fragments in VGPRs, B fragments from LDS once
// per column block, running max per row-block pair), no global memory traffic inside the loop.
//   hipcc -O3 --offload-arch=gfx950 tools/sweep_loop.hip -o tools/sweep_loop && tools/sweep_loop
// Prints clk per tile pair per SIMD (128 = the matrix pipe's floor: 4 x 32 clk).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v16f mfma(v4i a, v4i b, v16f c)
{
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{a[0], a[1], a[2], a[3], 0, 0, 0, 0}, v8i{b[0], b[1], b[2], b[3], 0, 0, 0, 0}, c, 4, 4, 0, 127, 0,
                                                           127);
}
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f mfma16(v4i a, v4i b, v4f c)
{
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(v8i{a[0], a[1], a[2], a[3], 0, 0, 0, 0}, v8i{b[0], b[1], b[2], b[3], 0, 0, 0, 0}, c, 4, 4, 0, 127, 0,
                                                            127);
}
__device__ __forceinline__ int max3i(int a, int b, int c)
{
    const int ab = a > b ? a : b;
    return ab > c ? ab : c;
}
__device__ __forceinline__ int K(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ int tree(const v16f &x, const v16f &y, int prev)
{
    const int t0 = max3i(K(x[0]), K(x[1]), K(x[2])), t1 = max3i(K(x[3]), K(x[4]), K(x[5])), t2 = max3i(K(x[6]), K(x[7]), K(x[8]));
    const int t3 = max3i(K(x[9]), K(x[10]), K(x[11])), t4 = max3i(K(x[12]), K(x[13]), K(x[14]));
    const int t5 = max3i(K(y[0]), K(y[1]), K(y[2])), t6 = max3i(K(y[3]), K(y[4]), K(y[5])), t7 = max3i(K(y[6]), K(y[7]), K(y[8]));
    const int t8 = max3i(K(y[9]), K(y[10]), K(y[11])), t9 = max3i(K(y[12]), K(y[13]), K(y[14]));
    const int u0 = max3i(t0, t1, t2), u1 = max3i(t3, t4, K(x[15])), u2 = max3i(t5, t6, t7), u3 = max3i(t8, t9, K(y[15]));
    return max3i(max3i(u0, u1, u2), u3, prev);
}

constexpr int NCB = 4, NRP = 4, PITCH = 80, CHUNKS = 64;

// VARIANT 0: MFMAs of a tile pair, then its tree (the shipped order)
// VARIANT 1: software pipelined: the MFMAs of tile pair k + 1 are issued before the tree of tile pair k (two accumulator sets)
// VARIANT 2: MFMAs only (no tree): the matrix-pipe ceiling of this loop structure
// VARIANT 3: like 1, and the B fragments of the next column block are fetched one block ahead
template <int VARIANT, int WPS>
__global__ void __launch_bounds__(256, WPS) k(const v4i *in, int *out, int iters, unsigned long long *clk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, c32 = lane & 31, h = lane >> 5;
    v4i A[8][2];
#pragma unroll
    for (int rb = 0; rb < 8; rb++)
#pragma unroll
        for (int f = 0; f < 2; f++) A[rb][f] = in[640 + ((threadIdx.x * 16 + rb * 2 + f) % 3456)];
    for (int t = threadIdx.x; t < 128 * PITCH / 16; t += 256) reinterpret_cast<v4i *>(lds)[t] = in[t & 4095];
    __syncthreads();
    int runmax[NRP] = {(int)0x80000000, (int)0x80000000, (int)0x80000000, (int)0x80000000};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const v16f Z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        for (int chunk = 0; chunk < CHUNKS; chunk++) {
            if (VARIANT == 4 || VARIANT == 5) {
                // 16x16x128: the whole 128-bit prefix of a 16 x 16 tile in one instruction (16 clk); 8 of them = the 2048 pairs of a tile pair
                const v4f Z4 = {0, 0, 0, 0};
#pragma unroll 1
                for (int cb = 0; cb < NCB; cb++) {
                    const unsigned char *bp = lds + (cb * 32 + (lane & 15)) * PITCH + (lane >> 4) * 16;
                    const v4i B0 = *reinterpret_cast<const v4i *>(bp), B1 = *reinterpret_cast<const v4i *>(bp + 16 * PITCH);
#pragma unroll
                    for (int p = 0; p < NRP; p++) {
                        v4f r[8];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            r[2 * q] = mfma16(A[2 * p + (q >> 1)][q & 1], B0, Z4);
                            r[2 * q + 1] = mfma16(A[2 * p + (q >> 1)][q & 1], B1, Z4);
                        }
                        if (VARIANT == 4) {
                            const v16f x = {r[0][0], r[0][1], r[0][2], r[0][3], r[1][0], r[1][1], r[1][2], r[1][3], r[2][0], r[2][1], r[2][2], r[2][3], r[3][0], r[3][1], r[3][2], r[3][3]};
                            const v16f y = {r[4][0], r[4][1], r[4][2], r[4][3], r[5][0], r[5][1], r[5][2], r[5][3], r[6][0], r[6][1], r[6][2], r[6][3], r[7][0], r[7][1], r[7][2], r[7][3]};
                            runmax[p] = tree(x, y, runmax[p]);
                        } else {
                            runmax[p] += K(r[0][0]) + K(r[7][1]);
                        }
                    }
                }
            } else if (VARIANT == 0 || VARIANT == 2) {
#pragma unroll 1
                for (int cb = 0; cb < NCB; cb++) {
                    const unsigned char *bp = lds + (cb * 32 + c32) * PITCH + h * 32;
                    const v4i B0 = *reinterpret_cast<const v4i *>(bp), B1 = *reinterpret_cast<const v4i *>(bp + 16);
#pragma unroll
                    for (int p = 0; p < NRP; p++) {
                        v16f a0 = mfma(A[2 * p][0], B0, Z), a1 = mfma(A[2 * p + 1][0], B0, Z);
                        a0 = mfma(A[2 * p][1], B1, a0);
                        a1 = mfma(A[2 * p + 1][1], B1, a1);
                        if (VARIANT == 0)
                            runmax[p] = tree(a0, a1, runmax[p]);
                        else
                            runmax[p] += K(a0[0]) + K(a1[5]);
                    }
                }
            } else {
                // flattened (cb, p) sequence; accumulator sets alternate
                const unsigned char *bp = lds + c32 * PITCH + h * 32;
                v4i B0 = *reinterpret_cast<const v4i *>(bp), B1 = *reinterpret_cast<const v4i *>(bp + 16);
                v16f a0 = mfma(A[0][0], B0, Z), a1 = mfma(A[1][0], B0, Z);
                a0 = mfma(A[0][1], B1, a0);
                a1 = mfma(A[1][1], B1, a1);
#pragma unroll
                for (int s = 1; s <= NCB * NRP; s++) {
                    const int p = s % NRP, cb = s / NRP;
                    v16f n0 = Z, n1 = Z;
                    if (s < NCB * NRP) {
                        if (p == 0) {
                            const unsigned char *bq = lds + (cb * 32 + c32) * PITCH + h * 32;
                            B0 = *reinterpret_cast<const v4i *>(bq);
                            B1 = *reinterpret_cast<const v4i *>(bq + 16);
                        }
                        n0 = mfma(A[2 * p][0], B0, Z);
                        n1 = mfma(A[2 * p + 1][0], B0, Z);
                        n0 = mfma(A[2 * p][1], B1, n0);
                        n1 = mfma(A[2 * p + 1][1], B1, n1);
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep the tree of the previous pair behind the issue of this pair's MFMAs
                    runmax[(s - 1) % NRP] = tree(a0, a1, runmax[(s - 1) % NRP]);
                    __builtin_amdgcn_sched_barrier(0);
                    a0 = n0;
                    a1 = n1;
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = runmax[0] + runmax[1] + runmax[2] + runmax[3];
    if (threadIdx.x == 0 && blockIdx.x == 300) {
        clk[0] = t1 - t0;
        clk[1] = r1 - r0;
    }
}

template <int VARIANT, int WPS>
void run(const char *name, const v4i *d_in, int *d_out, unsigned long long *d_clk)
{
    const int iters = 8;
    // LDS per block chosen so that exactly WPS blocks (of 4 waves) fit a CU
    size_t lds = (size_t)(160 * 1024) / WPS - 1024;
    if (lds > 65536) lds = 65536;
    if (lds < 128 * PITCH) lds = 128 * PITCH;
    hipFuncSetAttribute((const void *)k<VARIANT, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * WPS * 2;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k<VARIANT, WPS>), dim3(blocks), dim3(256), lds, 0, d_in, d_out, iters, d_clk);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<VARIANT, WPS>), dim3(blocks), dim3(256), lds, 0, d_in, d_out, iters, d_clk);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // tile pairs per SIMD: blocks / 256 CUs blocks per CU, each block puts one wave on every SIMD
    const double pairs_per_simd = (double)(blocks / 256) * iters * CHUNKS * NCB * NRP;
    const double ns = ms * 1e6 / pairs_per_simd;
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void *)k<VARIANT, WPS>);
    unsigned long long c[2];
    hipMemcpy(c, d_clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)c[0] / (double)c[1] * 0.1;  // s_memtime ticks per 100 MHz realtime tick
    const double wave_pairs = (double)iters * CHUNKS * NCB * NRP;
    std::printf("%-36s waves/SIMD %d regs %3d  %.3f ms  in-kernel clock %.2f GHz  %.1f ns = %.0f clk per tile pair per SIMD (floor 128); one wave: %.0f clk per tile pair\n",
                name, WPS, fa.numRegs, ms, ghz, ns, ns * ghz, (double)c[0] / wave_pairs);
}

int main(int argc, char **argv)
{
    // e2m1 magnitude code of the +-x encoding: 1 = 0.5, 2 = 1.0 (shipped), 4 = 2.0, 6 = 4.0; 0 = {0, 1} encoding (code 0 / code 2)
    const int mag = argc > 1 ? atoi(argv[1]) : 2;
    const int magb = argc > 2 ? atoi(argv[2]) : mag;  // encoding of the second half of the input table (the B operands come from there)
    v4i *d_in;
    int *d_out;
    hipMalloc(&d_in, 4096 * 16);
    hipMalloc(&d_out, 256 * 8 * 2 * 256 * 4);
    unsigned long long *d_clk;
    hipMalloc(&d_clk, 16);
    v4i *h = (v4i *)malloc(4096 * 16);
    for (int i = 0; i < 4096; i++)
        for (int j = 0; j < 4; j++) {
            unsigned v = 0;
            for (int nib = 0; nib < 8; nib++) {
                const unsigned bit = rand() & 1;
                const int mm = i < 640 ? magb : mag;  // entries 0..639 feed the LDS image (B), A comes from the whole table
                const unsigned code = mm == 0 ? (bit ? 2u : 0u) : ((unsigned)mm | (bit ? 8u : 0u));
                v |= code << (4 * nib);
            }
            h[i][j] = (int)v;
        }
    std::printf("encoding: A magnitude code %d, B magnitude code %d\n", mag, magb);
    hipMemcpy(d_in, h, 4096 * 16, hipMemcpyHostToDevice);
    run<0, 2>("MFMA x4 then tree (shipped)", d_in, d_out, d_clk);
    run<0, 3>("MFMA x4 then tree (shipped)", d_in, d_out, d_clk);
    return 0;
}
