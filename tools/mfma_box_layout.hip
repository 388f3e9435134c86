// mfma_box_layout.hip -- checks the register layouts the streaming PDQ kernel (csrc/pdq_stream.hip) relies on:
//   * v_mfma_i32_32x32x32_i8 with the image bytes as A (lane = row, 16 consecutive bytes per lane half) and a 0/1 band
//     matrix as B gives the horizontal window sums with lane = column;
//   * v_permlane32_swap_b32 regroups two 32x32 accumulator blocks so that every lane holds all 32 rows of ONE column;
//   * unsigned bytes through the signed instruction: bytes ^ 0x80, and 128 * n added through spare K slots.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_box_layout.hip -o /tmp/mfma_box_layout ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// img: 32 rows x 96 bytes (pitch 96).  Output column x in [0, 64) sums img[row][16 + x - a .. 16 + x + b].
__global__ void __launch_bounds__(64) box_kernel(const uint8_t *img, int a, int b, int *out /* [64 cols][32 rows] */)
{
    const int lane = threadIdx.x, r = lane & 31, kh = lane >> 5;
    v4i chunk[3];
    for (int c = 0; c < 3; c++) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(img + r * 96 + 32 * c + 16 * kh);
        chunk[c] = v4i{(int)(p[0] ^ 0x80808080u), (int)(p[1] ^ 0x80808080u), (int)(p[2] ^ 0x80808080u), (int)(p[3] ^ 0x80808080u)};
    }
    v16i acc[2];
    for (int nb = 0; nb < 2; nb++) {
        acc[nb] = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int ks = 0; ks < 2; ks++) {
            // B: lane = output column n of the block (lane & 31), slots 32 ks + 16 kh + j  <->  source column xs = 32 nb + 32 ks + 16 kh + j (in buffer bytes)
            const int n = lane & 31, x = 16 + 32 * nb + n;  // buffer column of the output
            uint32_t bd[4];
            for (int q = 0; q < 4; q++) {
                uint32_t d = 0;
                for (int jj = 0; jj < 4; jj++) {
                    const int xs = 32 * nb + 32 * ks + 16 * kh + 4 * q + jj;
                    if (xs >= x - a && xs <= x + b) d |= 1u << (8 * jj);
                }
                bd[q] = d;
            }
            v4i av = chunk[nb + ks];
            if (ks == 1 && kh == 1) {  // slots 52..63 are spare: A = 64 there, B = 2 n in one of them -> + 128 n
                av[1] = av[2] = av[3] = 0x40404040;
                bd[1] = 2u * (uint32_t)(a + b + 1);
            }
            acc[nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, v4i{(int)bd[0], (int)bd[1], (int)bd[2], (int)bd[3]}, acc[nb], 0, 0, 0);
        }
    }
    // regroup: lane l < 32 -> column l of block 0, lane l >= 32 -> column l - 32 of block 1; rows 8 q + i from P[4 q + i], rows 8 q + 4 + i from R[4 q + i]
    int P[16], R[16];
    for (int i = 0; i < 16; i++) {
        auto sw = __builtin_amdgcn_permlane32_swap(acc[0][i], acc[1][i], false, false);
        P[i] = sw[0];
        R[i] = sw[1];
    }
    for (int q = 0; q < 4; q++)
        for (int i = 0; i < 4; i++) {
            out[lane * 32 + 8 * q + i] = P[4 * q + i];
            out[lane * 32 + 8 * q + 4 + i] = R[4 * q + i];
        }
}

int main()
{
    std::vector<uint8_t> img(32 * 96);
    srand(7);
    for (auto &v : img) v = (uint8_t)(rand() >> 5);
    uint8_t *d_img;
    int *d_out;
    hipMalloc(&d_img, img.size());
    hipMalloc(&d_out, 64 * 32 * 4);
    hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice);
    int bad = 0;
    for (int win = 1; win <= 8; win++) {
        const int half = (win + 2) / 2, a = win - half, b = half - 1;
        hipLaunchKernelGGL(box_kernel, dim3(1), dim3(64), 0, 0, d_img, a, b, d_out);
        std::vector<int> out(64 * 32);
        hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
        for (int x = 0; x < 64; x++)
            for (int r = 0; r < 32; r++) {
                int s = 0;
                for (int xs = 16 + x - a; xs <= 16 + x + b; xs++) s += img[r * 96 + xs];
                if (s != out[x * 32 + r]) {
                    if (bad < 10) printf("win %d col %d row %d: got %d want %d\n", win, x, r, out[x * 32 + r], s);
                    bad++;
                }
            }
    }
    printf(bad ? "mfma_box_layout: %d MISMATCHES\n" : "mfma_box_layout: all window sums exact (layouts as assumed)\n", bad);
    return bad != 0;
}
