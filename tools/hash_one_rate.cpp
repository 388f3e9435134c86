// tools/hash_one_rate.cpp -- rph_pdq_hash_one from T native threads (what the reference's rayon workers would see; no interpreter
// lock in the way).   g++ -O2 -std=c++17 -I include tools/hash_one_rate.cpp -o tools/hash_one_rate_c -L rupphash_amd -lrupphash_hip
//                         -Wl,-rpath,$PWD/rupphash_amd -Wl,-rpath,/opt/rocm/lib -pthread
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "rupphash.h"

int main(int argc, char **argv)
{
    // optional arguments: thread counts (default 1 4 8 16 32 64); RPH_RATE_ONLY512=1 skips the resized geometry
    std::vector<int> counts;
    for (int i = 1; i < argc; i++) counts.push_back(std::atoi(argv[i]));
    if (counts.empty()) counts = {1, 4, 8, 16, 32, 64};
    rph_ctx *ctx = nullptr;
    if (rph_init(0, &ctx) != RPH_OK) {
        std::printf("no device: %s\n", rph_last_error());
        return 3;
    }
    struct Geo {
        const char *name;
        uint32_t w, h;
    } geos[] = {{"512x512 RGB8", 512, 512}, {"1280x854 RGB8 (resized path)", 1280, 854}};
    for (const Geo &g : geos) {
        if (g.w != 512 && std::getenv("RPH_RATE_ONLY512")) continue;
        const size_t bytes = (size_t)g.w * g.h * 3;
        std::vector<std::vector<uint8_t>> pool(32, std::vector<uint8_t>(bytes));
        uint32_t x = 12345;
        for (auto &img : pool)
            for (auto &b : img) {
                x = x * 1664525u + 1013904223u;
                b = (uint8_t)(x >> 24);
            }
        for (int threads : counts) {
            const int per = std::max(8, (g.w == 512 ? 16384 : 4096) / threads);
            uint64_t b0, i0, b1, i1;
            rph_pdq_batcher_stats(ctx, &b0, &i0);
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < threads; t++)
                th.emplace_back([&, t] {
                    uint8_t hash[32], valid;
                    float q;
                    for (int k = 0; k < per; k++)
                        if (rph_pdq_hash_one(ctx, pool[(t + k) % pool.size()].data(), g.w, g.h, 3, (size_t)g.w * 3, hash, &q, nullptr, &valid) != RPH_OK) std::abort();
                });
            for (auto &t : th) t.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            rph_pdq_batcher_stats(ctx, &b1, &i1);
            std::printf("%-30s threads=%3d: %9.0f hashes/s  %5.1f GB/s over PCIe, %6.1f images/batch\n", g.name, threads, threads * per / dt,
                        threads * per * (double)bytes / dt / 1e9, (double)(i1 - i0) / (double)(b1 - b0));
            std::fflush(stdout);
        }
    }
    rph_shutdown(ctx);
    return 0;
}
