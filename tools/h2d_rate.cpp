// tools/h2d_rate.cpp -- pinned host -> device transfer rate against the size of one hipMemcpyAsync (the one-image queue's transfers are
// 0.8 .. 100 MB): `reps` copies back to back on one stream, and split over two streams.
//   hipcc -O2 tools/h2d_rate.cpp -o tools/h2d_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

int main()
{
    const size_t cap = (size_t)256 << 20;
    uint8_t *h = nullptr, *d = nullptr;
    if (hipHostMalloc((void **)&h, cap) != hipSuccess || hipMalloc((void **)&d, cap) != hipSuccess) return 3;
    std::memset(h, 1, cap);
    hipStream_t s[2];
    for (auto &x : s) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    for (size_t bytes : {(size_t)786432, (size_t)3 << 20, (size_t)6 << 20, (size_t)20 << 20, (size_t)86 << 20}) {
        const int reps = (int)std::max<size_t>(4, ((size_t)1 << 30) / bytes / 4);
        for (int streams = 1; streams <= 2; streams++) {
            (void)hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; r++) {
                const size_t off = ((size_t)r * bytes) % (cap - bytes + 1) & ~(size_t)255;
                (void)hipMemcpyAsync(d + off, h + off, bytes, hipMemcpyHostToDevice, s[r % streams]);
            }
            (void)hipDeviceSynchronize();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("%8.2f MB per copy, %d stream(s): %6.1f GB/s, %7.1f us per copy\n", bytes / 1e6, streams, reps * (double)bytes / dt / 1e9, 1e6 * dt / reps);
        }
        // one copy at a time, synchronised (latency of an isolated transfer)
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 50; r++) {
            (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s[0]);
            (void)hipStreamSynchronize(s[0]);
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%8.2f MB per copy, isolated + synchronised: %6.1f GB/s, %7.1f us per copy\n", bytes / 1e6, 50 * (double)bytes / dt / 1e9, 1e6 * dt / 50);
    }
    return 0;
}
