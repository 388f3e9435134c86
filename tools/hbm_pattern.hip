// tools/hbm_pattern.hip -- read-only kernels that reproduce the fused PDQ kernel's access pattern (one wave per 512x512x3 image,
// 8 bands x strips of SW pixels, lane (c, g) = 24 bytes x 8 rows per tile) for several strip widths, against a plain stream.
// Tells how much of the HBM read bandwidth the pattern itself can reach, independent of any arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SW>
__global__ void __launch_bounds__(64, 2) pattern(const uint8_t *__restrict__ px, uint32_t n, uint32_t *sink)
{
    __shared__ uint32_t lds[4480];  // 17.9 KB like the real kernel: 8 waves per CU
    const uint8_t *img = px + (size_t)blockIdx.x * 786432;
    constexpr int CL = SW / 8, NG = 64 / CL, NH = 64 / (NG * 8);
    const int lane = threadIdx.x, c = lane & (CL - 1), g = lane / CL;
    uint32_t acc = 0;
    for (int b = 0; b < 8; b++)
        for (int s = 0; s < 512 / SW; s++)
            for (int h = 0; h < NH; h++) {
                uint4 a[8];
                uint2 d[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    int y = 64 * b + 4 + NG * 8 * h + 8 * g + k;
                    y = y > 511 ? 511 : y;
                    const uint8_t *p = img + (size_t)y * 1536 + (SW * s + 8 * c) * 3;
                    a[k] = *reinterpret_cast<const uint4 *>(p);
                    d[k] = *reinterpret_cast<const uint2 *>(p + 16);
                }
#pragma unroll
                for (int k = 0; k < 8; k++) acc ^= a[k].x ^ a[k].y ^ a[k].z ^ a[k].w ^ d[k].x ^ d[k].y;
            }
    lds[lane] = acc;
    if (acc == 0x12345u) sink[0] = lds[(lane + 1) & 63];
}
int main()
{
    const uint32_t n = 80000;
    uint8_t *px; uint32_t *sink;
    if (hipMalloc(&px, (size_t)n * 786432) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 4);
    hipMemset(px, 1, (size_t)n * 786432);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(SW)                                                                                      \
    {                                                                                                \
        hipLaunchKernelGGL(pattern<SW>, dim3(n), dim3(64), 0, 0, px, n, sink);                       \
        hipDeviceSynchronize();                                                                      \
        hipEventRecord(e0);                                                                          \
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(pattern<SW>, dim3(n), dim3(64), 0, 0, px, n, sink); \
        hipEventRecord(e1); hipEventSynchronize(e1);                                                 \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                                  \
        std::printf("strips of %3d px (%4d-byte row pieces): %.2f TB/s algorithmic = %.2f M images/s\n", SW, SW * 3,  \
                    3.0 * n * 786432.0 / (ms * 1e-3) / 1e12, 3.0 * n / (ms * 1e-3) / 1e6);           \
    }
    RUN(64) RUN(128) RUN(256) RUN(512)
    return 0;
}
