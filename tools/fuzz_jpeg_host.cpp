// ASan/UBSan driver for the host half of the JPEG path: every file of a directory, intact and with random damage (bytes overwritten,
// truncation, bytes inserted), through parse_frame / decode_coefficients / prepare_stream.  Results are not checked; the sanitizers are.
#include <dirent.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "jpeg_host.h"
static uint32_t intern(void *, const rphj::TableSpec &t) { rphj::DeviceLut L; return rphj::build_device_lut(t, L) == 0 ? 0u : UINT32_MAX; }
static void run(const std::vector<uint8_t> &f)
{
    rphj::Frame fr;
    if (rphj::parse_frame(f.data(), f.size(), fr) != 0) return;
    if (fr.total_blocks > (1u << 22)) return;
    std::vector<int16_t> coef((size_t)fr.total_blocks * 64);
    rphj::Frame a = fr;
    (void)rphj::decode_coefficients(f.data(), f.size(), a, coef.data());
    rphj::Frame b = fr;
    rphj::StreamPlan plan;
    std::vector<uint8_t> out(f.size() + 160);
    size_t used = 0;
    (void)rphj::prepare_stream(f.data(), f.size(), b, plan, out.data(), out.size(), &used, intern, nullptr);  // (too small for most progressive files: must be refused, not overrun)
    rphj::Frame c = fr;
    std::vector<uint8_t> out2(f.size() + 160 + 32 * (size_t)rphj::MAX_PROG_SCANS);
    (void)rphj::prepare_stream(f.data(), f.size(), c, plan, out2.data(), out2.size(), &used, intern, nullptr);
}
int main(int argc, char **argv)
{
    std::mt19937 rng(7);
    DIR *d = opendir(argv[1]);
    std::vector<std::vector<uint8_t>> files;
    while (dirent *e = readdir(d)) {
        std::string n = e->d_name;
        if (n.size() < 4 || n.substr(n.size() - 4) != ".jpg") continue;
        FILE *fp = fopen((std::string(argv[1]) + "/" + n).c_str(), "rb");
        std::vector<uint8_t> b;
        uint8_t buf[65536];
        size_t g;
        while ((g = fread(buf, 1, sizeof buf, fp)) > 0) b.insert(b.end(), buf, buf + g);
        fclose(fp);
        files.push_back(b);
    }
    closedir(d);
    const int rounds = argc > 2 ? atoi(argv[2]) : 200;
    long n = 0;
    for (auto &f : files) {
        run(f);
        for (int r = 0; r < rounds; r++) {
            std::vector<uint8_t> g = f;
            const int kind = rng() % 4;
            if (kind == 0) for (int k = 0; k < 1 + (int)(rng() % 6); k++) g[rng() % g.size()] = (uint8_t)rng();
            else if (kind == 1) g.resize(1 + rng() % g.size());
            else if (kind == 2) g.insert(g.begin() + rng() % g.size(), (uint8_t)(rng() % 3 == 0 ? 0xFF : rng()));
            else { size_t at = rng() % g.size(); g[at] = 0xFF; if (at + 1 < g.size()) g[at + 1] = (uint8_t)(0xC0 + rng() % 0x40); }
            run(g);
            n++;
        }
    }
    printf("%zu files, %ld damaged variants: no sanitizer report\n", files.size(), n);
    return 0;
}
