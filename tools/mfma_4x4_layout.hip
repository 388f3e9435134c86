// tools/mfma_4x4_layout.hip -- which lane and element hold what in v_mfma_f32_4x4x4_16b_f16 (16 independent 4x4x4 products per wave)?
// A[i][k] = 10 i + k + 1 in block 0 (+100 b in block b), B[k][j] = 1 if k == j (identity) in every block: D must come back as A.
// hipcc --offload-arch=gfx950 -O2 tools/mfma_4x4_layout.hip -o tools/mfma_4x4_layout && tools/mfma_4x4_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(float *out, int swap)
{
    const int lane = threadIdx.x, q = lane & 3, b = lane >> 2;
    h4 a, id;
    for (int k = 0; k < 4; k++) {
        a[k] = (_Float16)(float)(100 * b + 10 * q + k + 1);  // "row q of A", element k
        id[k] = (_Float16)(k == q ? 1.0f : 0.0f);            // "column q of B" of the identity
    }
    f4 c = {0, 0, 0, 0};
    f4 d = swap ? __builtin_amdgcn_mfma_f32_4x4x4f16(id, a, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_4x4x4f16(a, id, c, 0, 0, 0);
    for (int e = 0; e < 4; e++) out[lane * 4 + e] = d[e];
}
int main()
{
    float *d, h[256];
    hipMalloc(&d, sizeof h);
    for (int swap = 0; swap < 2; swap++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, swap);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("%s\n", swap ? "D = I x Adata (data given as B: lane = column j, element = k)" : "D = Adata x I (data given as A: lane = row i, element = k)");
        for (int lane = 0; lane < 8; lane++) printf("  lane %d: %6.0f %6.0f %6.0f %6.0f\n", lane, h[lane * 4], h[lane * 4 + 1], h[lane * 4 + 2], h[lane * 4 + 3]);
    }
    return 0;
}
