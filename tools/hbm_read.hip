// tools/hbm_read.hip -- achievable read-only HBM bandwidth on this GPU (grid-stride dwordx4 loads, xor-reduced), for several
// numbers of resident waves per CU.  The PDQ kernel is a pure read stream; this is its real ceiling.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) rd(const uint4 *__restrict__ p, size_t n, uint32_t *out)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    for (; i + 3 * step < n; i += 4 * step) {
        uint4 a = p[i], b = p[i + step], c = p[i + 2 * step], d = p[i + 3 * step];
        acc.x ^= a.x ^ b.x ^ c.x ^ d.x; acc.y ^= a.y ^ b.y ^ c.y ^ d.y; acc.z ^= a.z ^ b.z ^ c.z ^ d.z; acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
    }
    for (; i < n; i += step) { uint4 a = p[i]; acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
int main()
{
    const size_t bytes = (size_t)64 << 30;
    uint4 *p; uint32_t *out;
    if (hipMalloc(&p, bytes) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 4);
    hipMemset(p, 1, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks_per_cu : {1, 2, 4, 8}) {
        const int grid = 256 * blocks_per_cu;
        hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, p, bytes / 16, out);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, p, bytes / 16, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::printf("read-only stream, %d blocks x 256 threads per CU: %.2f TB/s\n", blocks_per_cu, 3.0 * bytes / (ms * 1e-3) / 1e12);
    }
    return 0;
}
