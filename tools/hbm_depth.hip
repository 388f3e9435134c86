// tools/hbm_depth.hip -- how much does a deeper prefetch buy the fused PDQ kernel?  One wave per image, the kernel's access pattern
// (64-px strips, 8 rows x 24 B per lane and tile), 8 waves per CU; between consuming a tile and the next the wave sleeps for the
// time the real kernel computes (~3.8 us per tile: 8.4 M images/s compute-only), with DEPTH tiles (or half tiles) of loads in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Tile { uint4 a[8]; uint2 d[8]; };
__device__ __forceinline__ void issue(Tile &t, const uint8_t *img, int tile, int c, int g, int k0, int k1)
{
    const int b = tile >> 3, s = tile & 7;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k < k0 || k >= k1) continue;
        int y = 64 * b + 4 + 8 * g + k;
        y = y > 511 ? 511 : y;
        const uint8_t *p = img + (size_t)y * 1536 + (64 * s + 8 * c) * 3;
        t.a[k] = *reinterpret_cast<const uint4 *>(p);
        t.d[k] = *reinterpret_cast<const uint2 *>(p + 16);
    }
}
__device__ __forceinline__ uint32_t eat(const Tile &t, int k0, int k1)
{
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (k >= k0 && k < k1) acc ^= t.a[k].x ^ t.a[k].y ^ t.a[k].z ^ t.a[k].w ^ t.d[k].x ^ t.d[k].y;
    return acc;
}
// HALF = 0: depth 1 (one tile ahead, like the real kernel); 1: depth 1.5 (rows 0..3 two tiles ahead); 2: depth 2
template <int MODE>
__global__ void __launch_bounds__(64, 2) k(const uint8_t *__restrict__ px, uint32_t *sink, int sleep_units)
{
    __shared__ uint32_t lds[4480];
    const uint8_t *img = px + (size_t)blockIdx.x * 786432;
    const int lane = threadIdx.x, c = lane & 7, g = lane >> 3;
    uint32_t acc = 0;
    Tile A, B;
    issue(A, img, 0, c, g, 0, 8);
    if (MODE == 2) issue(B, img, 1, c, g, 0, 8);
    if (MODE == 1) issue(B, img, 1, c, g, 0, 4);
    for (int t = 0; t < 64; t += 2) {
        if (MODE == 3) {  // like the real kernel: the next tile's rows are issued one by one over the first 3/8 of the period (the luma phase)
            for (int half = 0; half < 2; half++) {
                acc ^= eat(A, 0, 8);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    issue(A, img, (t + 1 + half) & 63, c, g, k, k + 1);
                    for (int i = 0; i < (3 * sleep_units) / 8; i++) __builtin_amdgcn_s_sleep(1);
                }
                for (int i = 0; i < 5 * sleep_units; i++) __builtin_amdgcn_s_sleep(1);
            }
            continue;
        }
        acc ^= eat(A, 0, 8);
        if (MODE == 0) { issue(A, img, (t + 1) & 63, c, g, 0, 8); }
        if (MODE == 2) { issue(A, img, (t + 2) & 63, c, g, 0, 8); }
        if (MODE == 1) { issue(A, img, (t + 1) & 63, c, g, 4, 8); issue(A, img, (t + 2) & 63, c, g, 0, 4); }
        for (int i = 0; i < sleep_units; i++) __builtin_amdgcn_s_sleep(8);
        if (MODE == 0) { acc ^= eat(A, 0, 8); issue(A, img, (t + 2) & 63, c, g, 0, 8); }
        if (MODE == 2) { acc ^= eat(B, 0, 8); issue(B, img, (t + 3) & 63, c, g, 0, 8); }
        if (MODE == 1) { acc ^= eat(B, 0, 4) ^ eat(A, 4, 8); issue(A, img, (t + 2) & 63, c, g, 4, 8); issue(B, img, (t + 3) & 63, c, g, 0, 4); }
        for (int i = 0; i < sleep_units; i++) __builtin_amdgcn_s_sleep(8);
    }
    lds[lane] = acc;
    if (acc == 0x12345u) sink[0] = lds[(lane + 1) & 63];
}
int main()
{
    const uint32_t n = 60000;
    uint8_t *px; uint32_t *sink;
    if (hipMalloc(&px, (size_t)n * 786432) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 4);
    hipMemset(px, 1, (size_t)n * 786432);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int sl : {12, 16, 18}) {
#define RUN(MODE, NAME)                                                                             \
    {                                                                                               \
        hipLaunchKernelGGL(k<MODE>, dim3(n), dim3(64), 0, 0, px, sink, sl);                         \
        hipDeviceSynchronize();                                                                     \
        hipEventRecord(e0);                                                                         \
        for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k<MODE>, dim3(n), dim3(64), 0, 0, px, sink, sl); \
        hipEventRecord(e1); hipEventSynchronize(e1);                                                \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                                 \
        std::printf("sleep %2d x s_sleep(8) per tile, %-28s %.2f M images/s\n", sl, NAME, 2.0 * n / (ms * 1e-3) / 1e6); \
    }
        RUN(0, "depth 1, issue at once:") RUN(3, "depth 1, issue over luma:") RUN(1, "depth 1.5:") RUN(2, "depth 2:")
    }
    return 0;
}
