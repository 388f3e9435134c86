// tests/cpp/reference_tests.cpp -- the reference's own hot-path unit tests (src/pdqhash.rs:464-648,
// src/hamminghash.rs:273-412, NOTES.txt:64-67) re-run through the C++ mirror (include/rupphash.hpp) on the GPU.
// Built by __graft_entry__.build(); run by tests/test_cpp_mirror.py (-m gpu).  Exit code = number of failures.
#include <algorithm>
#include <cstdio>
#include <random>

#include <thread>

#include "rupphash.hpp"

using namespace rupphash;
using pdqhash::Hash;
using pdqhash::PdqFeatures;

static int failures = 0;
#define EXPECT(cond, ...)                                   \
    do {                                                    \
        if (!(cond)) {                                      \
            failures++;                                     \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);                       \
            std::printf("\n");                              \
        }                                                   \
    } while (0)

// ---- naive ground truth of the reference's tests (pdqhash.rs:470-535): sort-based median, explicit transposes/flips
static int32_t total_key(float f)
{
    int32_t b;
    std::memcpy(&b, &f, 4);
    return b ^ (int32_t)(((uint32_t)(b >> 31)) >> 1);
}
static Hash naive_to_hash(const PdqFeatures &f)
{
    auto buf = f.coefficients;
    std::sort(buf.begin(), buf.end(), [](float a, float b) { return total_key(a) < total_key(b); });
    const float median = buf[(buf.size() - 1) / 2];
    Hash h{};
    for (int i = 0; i < 32; i++) {
        uint8_t byte = 0;
        for (int j = 0; j < 8; j++)
            if (f.coefficients[i * 8 + j] > median) byte |= (uint8_t)(1u << j);
        h[32 - i - 1] = byte;
    }
    return h;
}
static PdqFeatures naive_transpose(const PdqFeatures &f)
{
    PdqFeatures o{};
    for (int r = 0; r < 16; r++)
        for (int c = 0; c < 16; c++) o.coefficients[c * 16 + r] = f.coefficients[r * 16 + c];
    return o;
}
static PdqFeatures naive_flip_x(PdqFeatures f)
{
    for (int r = 0; r < 16; r++)
        for (int c = 0; c < 16; c++)
            if ((c + 1) % 2 != 0) f.coefficients[r * 16 + c] = -f.coefficients[r * 16 + c];
    return f;
}
static PdqFeatures naive_flip_y(PdqFeatures f)
{
    for (int r = 0; r < 16; r++)
        if ((r + 1) % 2 != 0)
            for (int c = 0; c < 16; c++) f.coefficients[r * 16 + c] = -f.coefficients[r * 16 + c];
    return f;
}
static std::array<Hash, 8> naive_dihedral(const PdqFeatures &f)
{
    return {naive_to_hash(f),
            naive_to_hash(naive_flip_x(naive_transpose(f))),
            naive_to_hash(naive_flip_y(naive_flip_x(f))),
            naive_to_hash(naive_flip_y(naive_transpose(f))),
            naive_to_hash(naive_flip_x(f)),
            naive_to_hash(naive_flip_y(f)),
            naive_to_hash(naive_transpose(f)),
            naive_to_hash(naive_flip_y(naive_flip_x(naive_transpose(f))))};
}
static PdqFeatures pseudo_random_features(uint32_t seed)  // pdqhash.rs:537-545
{
    uint32_t state = seed;
    PdqFeatures f{};
    for (auto &c : f.coefficients) {
        state = state * 1664525u + 1013904223u;
        c = (float)(state >> 8) / 65536.0f - 128.0f;
    }
    return f;
}

static void fast_dihedral_matches_naive()  // pdqhash.rs:548-558
{
    for (uint32_t seed : {1u, 42u, 0x12345678u, 0xDEADBEEFu}) {
        const PdqFeatures f = pseudo_random_features(seed);
        EXPECT(f.to_hash() == naive_to_hash(f), "to_hash mismatch, seed %u", seed);
        EXPECT(f.generate_dihedral_hashes() == naive_dihedral(f), "dihedral mismatch, seed %u", seed);
    }
}
static void dihedral_set_is_the_full_group()  // pdqhash.rs:561-570
{
    const auto h = pseudo_random_features(7).generate_dihedral_hashes();
    for (int i = 0; i < 8; i++)
        for (int j = i + 1; j < 8; j++) EXPECT(h[i] != h[j], "variants %d and %d collided", i, j);
}
static void dihedral_hashes_match_physically_transformed_buffer()  // pdqhash.rs:583-628
{
    // PdqFeatures::new(buffer64x64) is private in the reference; a 64x64 Luma8 image has Jarosz windows of 1 and an identity
    // decimation, so generate_pdq_features(image) runs exactly the same DCT on the same buffer.
    constexpr int N = 64;
    for (uint32_t seed : {1u, 42u, 0xDEADBEEFu}) {
        uint32_t state = seed;
        std::vector<uint8_t> buf(N * N);
        for (auto &px : buf) {
            state = state * 1664525u + 1013904223u;
            px = (uint8_t)((state >> 16) & 0xFF);
        }
        auto at = [&](int x, int y) { return buf[x * N + y]; };
        const auto predicted = pdqhash::generate_pdq_features({buf.data(), N, N, 1})->first.generate_dihedral_hashes();
        for (int variant = 0; variant < 8; variant++) {
            std::vector<uint8_t> out(N * N);
            for (int x = 0; x < N; x++)
                for (int y = 0; y < N; y++) {
                    uint8_t v;
                    switch (variant) {
                        case 0: v = at(x, y); break;
                        case 1: v = at(N - 1 - y, x); break;
                        case 2: v = at(N - 1 - x, N - 1 - y); break;
                        case 3: v = at(y, N - 1 - x); break;
                        case 4: v = at(x, N - 1 - y); break;
                        case 5: v = at(N - 1 - x, y); break;
                        case 6: v = at(y, x); break;
                        default: v = at(N - 1 - y, N - 1 - x); break;
                    }
                    out[x * N + y] = v;
                }
            const Hash actual = pdqhash::generate_pdq({out.data(), N, N, 1})->first;
            const uint32_t dist = hamminghash::HammingHash<Hash>::hamming_distance(actual, predicted[variant]);
            EXPECT(dist == 0, "variant %d (seed %u) is %u bits from the real transform", variant, seed, dist);
        }
    }
}
static void quality_and_dimensions()  // pdqhash.rs:631-647, :167-169
{
    std::vector<uint8_t> flat(64 * 64, 128);
    EXPECT(pdqhash::generate_pdq_features({flat.data(), 64, 64, 1})->second == 0.0f, "flat image quality");
    EXPECT(pdqhash::calculate_target_dimensions(4000, 5, 512) == std::make_pair(512u, 1u), "dims 4000x5");
    EXPECT(pdqhash::calculate_target_dimensions(5, 4000, 512) == std::make_pair(1u, 512u), "dims 5x4000");
    EXPECT(pdqhash::calculate_target_dimensions(1024, 1024, 512) == std::make_pair(512u, 512u), "dims 1024");
    EXPECT(pdqhash::calculate_target_dimensions(1024, 512, 512) == std::make_pair(512u, 256u), "dims 1024x512");
    EXPECT(!pdqhash::generate_pdq({flat.data(), 4, 64, 1}).has_value(), "4 px wide -> None");
    EXPECT(pdqhash::generate_pdq({flat.data(), 5, 5, 1}).has_value(), "5x5 -> Some");
}
static void generate_pdq_features_from_many_threads()  // scanner.rs:1202-1205, :1410: one call per file from every worker
{
    constexpr int T = 24, PER = 8, N = 96;
    std::vector<std::vector<uint8_t>> imgs(T * PER, std::vector<uint8_t>((size_t)N * N * 3));
    uint32_t s = 99;
    for (auto &im : imgs)
        for (auto &p : im) p = (uint8_t)((s = s * 1664525u + 1013904223u) >> 24);
    std::vector<Hash> serial(imgs.size()), par(imgs.size());
    for (size_t k = 0; k < imgs.size(); k++) serial[k] = pdqhash::generate_pdq({imgs[k].data(), N, N, 3})->first;
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            for (int k = 0; k < PER; k++) par[t * PER + k] = pdqhash::generate_pdq({imgs[t * PER + k].data(), N, N, 3})->first;
        });
    for (auto &x : th) x.join();
    EXPECT(serial == par, "hashes from concurrent callers equal the serial ones");
    EXPECT(serial[0] != serial[1], "distinct images, distinct hashes");
}

static void test_high_similarity_support()  // hamminghash.rs:283-332
{
    EXPECT(hamminghash::HammingHash<uint64_t>::hamming_distance(0, 0xFFF) == 12, "u64 distance");
    {  // pHash (u64) at max_dist 12: NOTES.txt:13-14
        hamminghash::MIHIndex64 index64({0ull, 0xFFFull});
        const auto g64 = hamminghash::find_groups(index64, 12);
        EXPECT(!g64.empty() && g64[0] == (std::vector<uint32_t>{0, 1}), "pHash Group should contain both indices");
        // CSR of MIHIndex<u64>: chunk 0 of 0xFFF is 0xFF, chunk 1 is 0x0F, chunks 2..7 are 0 for both
        EXPECT(index64.bucket(0, 0xFF).second == 1 && index64.bucket(0, 0xFF).first[0] == 1, "bucket(0, 0xFF) = [1]");
        EXPECT(index64.bucket(1, 0x0F).second == 1 && index64.bucket(1, 0).second == 1, "chunk 1 buckets");
        EXPECT(index64.bucket(5, 0).second == 2 && index64.bucket(5, 0).first[0] == 0 && index64.bucket(5, 0).first[1] == 1, "bucket(5, 0) = [0, 1]");
    }
    hamminghash::Hash256 base{}, target{};
    for (int i = 0; i < 30; i++) target[i / 8] |= (uint8_t)(1u << (i % 8));
    hamminghash::MIHIndex index({base, target});
    const auto groups = hamminghash::find_groups(index, 30);
    EXPECT(!groups.empty(), "Failed to find any groups for PDQ");
    if (!groups.empty())
        EXPECT(groups[0] == (std::vector<uint32_t>{0, 1}), "PDQ Group should contain both indices");  // NOTES.txt:15
}
static void test_injected_cluster()  // hamminghash.rs:336-412 with [u8;32] hashes
{
    const size_t n = 200000;
    std::mt19937_64 rng(12345);
    std::vector<hamminghash::Hash256> hashes(n);
    for (auto &h : hashes)
        for (int w = 0; w < 4; w++) {
            const uint64_t v = rng();
            std::memcpy(h.data() + 8 * w, &v, 8);
        }
    hamminghash::Hash256 target = hashes[0];
    const uint64_t flips[5] = {0, 1, 2, 0x8000, 0x8001};
    std::vector<size_t> idx;
    while (idx.size() < 5) {
        const size_t i = rng() % n;
        if (std::find(idx.begin(), idx.end(), i) == idx.end()) idx.push_back(i);
    }
    for (int k = 0; k < 5; k++) {
        hashes[idx[k]] = target;
        uint64_t w0;
        std::memcpy(&w0, hashes[idx[k]].data(), 8);
        w0 ^= flips[k];
        std::memcpy(hashes[idx[k]].data(), &w0, 8);
    }
    hashes[0][31] ^= 0xFF;  // the donor itself moves away
    hamminghash::MIHIndex index(hashes);
    const auto groups = hamminghash::find_groups(index, 5);
    const std::vector<uint32_t> *found = nullptr;
    for (const auto &g : groups)
        if (std::find(g.begin(), g.end(), (uint32_t)idx[0]) != g.end()) found = &g;
    EXPECT(found != nullptr, "The injected images were not found in any group!");
    if (found)
        for (size_t i : idx) EXPECT(std::find(found->begin(), found->end(), (uint32_t)i) != found->end(), "Group missing injected index %zu", i);
    // SparseBitSet semantics (hamminghash.rs:152-189)
    hamminghash::SparseBitSet s(1000);
    EXPECT(!s.set(5) && s.set(5) && !s.set(999), "SparseBitSet::set returns was_set");
    s.clear();
    EXPECT(!s.set(5), "SparseBitSet::clear");
}
static void phash_known_answer()  // NOTES.txt:64-67
{
    EXPECT(phash::calculate_rotation_invariant_hash(0xDEB1E20C136F983Cull) == 0x8B1BB7A646C5CD96ull, "rotation invariant hash");
    EXPECT(phash::generate_dihedral_hashes(0xDEB1E20C136F983Cull).size() == 8, "8 variants");
}
static void grouping_rules()  // scanner.rs:1588-1594, 1640-1823
{
    EXPECT(!scanner::is_low_pdq_quality(std::nullopt) && scanner::is_low_pdq_quality(49) && !scanner::is_low_pdq_quality(50), "quality rule");
    std::vector<scanner::ScannedFile> files(4);
    hamminghash::Hash256 a{}, b{};
    b[0] = 1;
    files[0].pdqhash = a; files[0].pdq_quality = 100;
    files[1].pdqhash = b; files[1].pdq_quality = 100;   // distance 1 from file 0
    files[2].pdqhash = a; files[2].pdq_quality = 10;    // exact duplicate of file 0 but low quality: still pairs at distance 0
    /* files[3] has no hash: filtered like valid_entries does */
    auto r = scanner::group_with_pdqhash(files, 10);
    EXPECT(r.first.size() == 1 && r.first[0] == (std::vector<uint32_t>{0, 1, 2}), "groups");
    EXPECT(r.second == 2, "comparison_count = %zu (0-1 fuzzy, 0-2 exact; 1-2 blocked by low quality)", r.second);
    bool threw = false;
    try { scanner::group_with_pdqhash(files, 64); } catch (const std::runtime_error &) { threw = true; }
    EXPECT(threw, "similarity 64 must be rejected (scanner.rs:1650-1655)");
}

// Row N3: load_image_fast + generate_pdq_features on the reference's own bench image (tests/bench.jpg, hamminghash.rs:422), file by file
// and as a batch of files: the same hash either way, and the decode of a one-component re-encode is Luma8
static void jpeg_files_through_the_mirror()
{
    const char *root = std::getenv("RPH_TEST_GOLDEN");
    if (!root) {
        std::printf("jpeg_files_through_the_mirror: RPH_TEST_GOLDEN not set, skipped\n");
        return;
    }
    std::vector<uint8_t> bytes;
    {
        const std::string path = std::string(root) + "/bench.jpg";
        FILE *f = std::fopen(path.c_str(), "rb");
        EXPECT(f != nullptr, "f != nullptr");
        if (!f) return;
        uint8_t buf[65536];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + got);
        std::fclose(f);
    }
    const auto img = rupphash::scanner::load_image_fast(bytes.data(), bytes.size());
    EXPECT(img.has_value(), "img.has_value()");
    if (!img) return;
    EXPECT(img->width == 1280 && img->height == 854 && img->channels == 3, "img->width == 1280 && img->height == 854 && img->channels == 3");
    const auto one = rupphash::pdqhash::generate_pdq(img->view());
    EXPECT(one.has_value(), "one.has_value()");
    const auto batch = rupphash::scanner::hash_jpeg_files({{bytes.data(), bytes.size()}, {bytes.data(), 100}, {bytes.data(), bytes.size()}});
    EXPECT(batch.size() == 3 && batch[0].valid && !batch[1].valid && batch[1].status != 0 && batch[2].valid, "batch.size() == 3 && batch[0].valid && !batch[1].valid && batch[1].status != 0 && batch[2].valid");
    if (one) EXPECT(batch[0].hash == one->first && batch[2].hash == one->first && batch[0].quality == one->second, "batch[0].hash == one->first && batch[2].hash == one->first && batch[0].quality == one->second");
    EXPECT(!rupphash::scanner::load_image_fast(bytes.data(), 100).has_value(), "!rupphash::scanner::load_image_fast(bytes.data(), 100).has_value()");  // truncated header: the caller's next decoder takes it
}

int main()
{
    fast_dihedral_matches_naive();
    dihedral_set_is_the_full_group();
    dihedral_hashes_match_physically_transformed_buffer();
    quality_and_dimensions();
    generate_pdq_features_from_many_threads();
    test_high_similarity_support();
    test_injected_cluster();
    phash_known_answer();
    grouping_rules();
    jpeg_files_through_the_mirror();
    std::printf("%s (%d failures)\n", failures ? "FAILED" : "ok", failures);
    return failures;
}
