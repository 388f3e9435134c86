// rupphash.hpp -- header-only C++ mirror of the reference's hot-path modules on top of the C ABI
// (include/rupphash.h).  Same names, argument meaning and error behaviour as the Rust originals, so that
// host code and tests written against the reference read the same here:
//
//   rupphash::pdqhash      <->  /root/reference/src/pdqhash.rs      (PdqFeatures, generate_pdq_features, generate_pdq)
//   rupphash::hamminghash  <->  /root/reference/src/hamminghash.rs  (HammingHash, MIHIndex, SparseBitSet, find_groups)
//   rupphash::phash        <->  /root/reference/src/phash.rs:137-255 (u64 bit operations)
//   rupphash::scanner      <->  /root/reference/src/scanner.rs:1588-1832 (is_low_pdq_quality, group_with_pdqhash)
//
// Option<T> -> std::optional<T>; panics/asserts -> std::runtime_error.  All arithmetic happens in
// librupphash_hip.so on the GPU; this header only marshals.
#pragma once
#include <array>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rupphash.h"

namespace rupphash {

inline void check(int status, const char *where)
{
    if (status != RPH_OK) throw std::runtime_error(std::string(where) + ": " + rph_last_error() + " (" + rph_status_string(status) + ")");
}

// One context per process and GPU (one process per GPU); lazily created on LOCAL_RANK's device or 0.
class Context {
public:
    static rph_ctx *get()
    {
        static Context c;
        return c.ctx_;
    }

private:
    Context()
    {
        const char *lr = std::getenv("LOCAL_RANK");
        check(rph_init(lr ? std::atoi(lr) : 0, &ctx_), "rph_init");
    }
    ~Context() { rph_shutdown(ctx_); }
    rph_ctx *ctx_ = nullptr;
};

// Minimal stand-in for image::DynamicImage: interleaved u8 pixels, channels 1 (Luma8), 3 (Rgb8) or 4 (Rgba8).
struct ImageView {
    const uint8_t *data;
    uint32_t width, height, channels;
};

namespace pdqhash {
using Hash = std::array<uint8_t, 32>;

struct PdqFeatures {                      // pdqhash.rs:48-51
    std::array<float, 256> coefficients;

    // Host-scalar like the original (compare + bit operations on 256 floats: no GPU round trip per file, scanner.rs:1412, :1622)
    Hash to_hash() const                  // pdqhash.rs:59-61
    {
        Hash h{};
        rph_pdq_to_hash(coefficients.data(), h.data());
        return h;
    }
    std::array<Hash, 8> generate_dihedral_hashes() const  // pdqhash.rs:71-87
    {
        std::array<Hash, 8> d{};
        rph_pdq_dihedral_one(coefficients.data(), d[0].data());
        return d;
    }
};

// pdqhash.rs:166-196.  nullopt <=> None (width or height < 5).  Thread-safe like the original: the scanner calls it from
// every rayon worker (scanner.rs:1410); concurrent calls share one GPU batch inside the library (rph_pdq_hash_one).
inline std::optional<std::pair<PdqFeatures, float>> generate_pdq_features(const ImageView &img)
{
    PdqFeatures f{};
    Hash h{};
    float q = 0.f;
    uint8_t valid = 0;
    check(rph_pdq_hash_one(Context::get(), img.data, img.width, img.height, img.channels, (size_t)img.width * img.channels, h.data(), &q,
                           f.coefficients.data(), &valid),
          "generate_pdq_features");
    if (!valid) return std::nullopt;
    return std::make_pair(f, q);
}
// pdqhash.rs:199-201: the hash the kernel already computed beside the features is returned as is (no second pass)
inline std::optional<std::pair<Hash, float>> generate_pdq(const ImageView &img)
{
    Hash h{};
    float q = 0.f;
    uint8_t valid = 0;
    check(rph_pdq_hash_one(Context::get(), img.data, img.width, img.height, img.channels, (size_t)img.width * img.channels, h.data(), &q,
                           nullptr, &valid),
          "generate_pdq");
    if (!valid) return std::nullopt;
    return std::make_pair(h, q);
}
// pdqhash.rs:224-235
inline std::pair<uint32_t, uint32_t> calculate_target_dimensions(uint32_t w, uint32_t h, uint32_t max_dim)
{
    uint32_t nw, nh;
    rph_pdq_target_dimensions(w, h, max_dim, &nw, &nh);
    return {nw, nh};
}
}  // namespace pdqhash

namespace hamminghash {
constexpr uint32_t MAX_SIMILARITY_64 = RPH_MAX_SIMILARITY_64;    // hamminghash.rs:5
constexpr uint32_t MAX_SIMILARITY_256 = RPH_MAX_SIMILARITY_256;  // hamminghash.rs:8
using Hash256 = std::array<uint8_t, 32>;

// trait HammingHash (hamminghash.rs:11-20) as traits classes
template <class H>
struct HammingHash;
template <>
struct HammingHash<uint64_t> {  // hamminghash.rs:23-41
    static constexpr size_t NUM_CHUNKS = 8, NUM_BUCKETS = 256;
    static constexpr uint32_t MAX_DIST = MAX_SIMILARITY_64;
    static uint16_t get_chunk(const uint64_t &h, size_t k) { return rph_get_chunk64(h, (uint32_t)k); }
    static uint32_t hamming_distance(const uint64_t &a, const uint64_t &b) { return rph_hamming_distance64(a, b); }
    static size_t bit_width_per_chunk() { return 8; }
};
template <>
struct HammingHash<Hash256> {  // hamminghash.rs:44-63
    static constexpr size_t NUM_CHUNKS = 16, NUM_BUCKETS = 65536;
    static constexpr uint32_t MAX_DIST = MAX_SIMILARITY_256;
    static uint16_t get_chunk(const Hash256 &h, size_t k) { return rph_get_chunk256(h.data(), (uint32_t)k); }
    static uint32_t hamming_distance(const Hash256 &a, const Hash256 &b) { return rph_hamming_distance256(a.data(), b.data()); }
    static size_t bit_width_per_chunk() { return 16; }
};

struct DenseId {  // hamminghash.rs:67-76
    uint32_t v;
    size_t index() const { return v; }
};

// MIHIndex<[u8;32]> (hamminghash.rs:82-149): the CSR arrays are built on the GPU.
class MIHIndex {
public:
    explicit MIHIndex(std::vector<Hash256> hashes) : db_hashes_(std::move(hashes))
    {
        offsets_.resize(16 * 65536 + 1);
        values_.resize(16 * db_hashes_.size() + 1);
        check(rph_mih_build256(Context::get(), db_hashes_.empty() ? nullptr : db_hashes_[0].data(), db_hashes_.size(),
                               offsets_.data(), values_.data()),
              "MIHIndex::new");
    }
    std::pair<const uint32_t *, size_t> bucket(size_t chunk, uint16_t value) const
    {
        const size_t flat = chunk * 65536 + value;
        return {values_.data() + offsets_[flat], offsets_[flat + 1] - offsets_[flat]};
    }
    const Hash256 &hash(DenseId id) const { return db_hashes_[id.index()]; }
    size_t len() const { return db_hashes_.size(); }
    const std::vector<Hash256> &hashes() const { return db_hashes_; }

private:
    std::vector<Hash256> db_hashes_;
    std::vector<uint32_t> offsets_, values_;
};

// MIHIndex<u64> (hamminghash.rs:23-41, :82-149): 8 chunks of 8 bits, CSR built on the GPU.
class MIHIndex64 {
public:
    explicit MIHIndex64(std::vector<uint64_t> hashes) : db_hashes_(std::move(hashes))
    {
        offsets_.resize(8 * 256 + 1);
        values_.resize(8 * db_hashes_.size() + 1);
        check(rph_mih_build64(Context::get(), db_hashes_.data(), db_hashes_.size(), offsets_.data(), values_.data()), "MIHIndex::new");
    }
    std::pair<const uint32_t *, size_t> bucket(size_t chunk, uint16_t value) const
    {
        const size_t flat = chunk * 256 + value;
        return {values_.data() + offsets_[flat], offsets_[flat + 1] - offsets_[flat]};
    }
    const uint64_t &hash(DenseId id) const { return db_hashes_[id.index()]; }
    size_t len() const { return db_hashes_.size(); }
    const std::vector<uint64_t> &hashes() const { return db_hashes_; }

private:
    std::vector<uint64_t> db_hashes_;
    std::vector<uint32_t> offsets_, values_;
};

// SparseBitSet (hamminghash.rs:152-189)
class SparseBitSet {
public:
    explicit SparseBitSet(size_t size) : data_((size + 63) / 64, 0) { dirty_.reserve(512); }
    bool set(size_t idx)
    {
        uint64_t &w = data_[idx / 64];
        const uint64_t mask = 1ull << (idx % 64);
        const bool was = (w & mask) != 0;
        if (!was) {
            if (w == 0) dirty_.push_back(idx / 64);
            w |= mask;
        }
        return was;
    }
    void clear()
    {
        for (size_t i : dirty_) data_[i] = 0;
        dirty_.clear();
    }

private:
    std::vector<uint64_t> data_;
    std::vector<size_t> dirty_;
};

// find_groups::<[u8;32]> (hamminghash.rs:191-271), bit-exact including member order
inline std::vector<std::vector<uint32_t>> find_groups(const MIHIndex &index, uint32_t max_dist)
{
    const size_t n = index.len();
    std::vector<uint32_t> members(n ? n : 1), offsets(n / 2 + 2);
    uint32_t ng = 0;
    check(rph_find_groups256(Context::get(), n ? index.hashes()[0].data() : nullptr, n, max_dist, members.data(), offsets.data(), &ng),
          "find_groups");
    std::vector<std::vector<uint32_t>> out(ng);
    for (uint32_t g = 0; g < ng; g++) out[g].assign(members.begin() + offsets[g], members.begin() + offsets[g + 1]);
    return out;
}
// find_groups::<u64> (hamminghash.rs:191-271 with the u64 impl :23-41)
inline std::vector<std::vector<uint32_t>> find_groups(const MIHIndex64 &index, uint32_t max_dist)
{
    const size_t n = index.len();
    std::vector<uint32_t> members(n ? n : 1), offsets(n / 2 + 2);
    uint32_t ng = 0;
    check(rph_find_groups64(Context::get(), index.hashes().data(), n, max_dist, members.data(), offsets.data(), &ng), "find_groups");
    std::vector<std::vector<uint32_t>> out(ng);
    for (uint32_t g = 0; g < ng; g++) out[g].assign(members.begin() + offsets[g], members.begin() + offsets[g + 1]);
    return out;
}
}  // namespace hamminghash

namespace phash {  // phash.rs:137-255
inline uint64_t rotate_hash_90(uint64_t h) { return rph_phash_rotate_90(h); }
inline uint64_t rotate_hash_180(uint64_t h) { return rph_phash_rotate_180(h); }
inline uint64_t rotate_hash_270(uint64_t h) { return rph_phash_rotate_270(h); }
inline uint64_t flip_hash_horizontal(uint64_t h) { return rph_phash_flip_horizontal(h); }
inline uint64_t calculate_rotation_invariant_hash(uint64_t h) { return rph_phash_rotation_invariant(h); }
inline std::vector<uint64_t> generate_dihedral_hashes(uint64_t h)
{
    std::vector<uint64_t> v(8);
    rph_phash_dihedral(h, v.data());
    return v;
}
}  // namespace phash

namespace scanner {
constexpr int PDQ_MIN_QUALITY = RPH_PDQ_MIN_QUALITY;                                               // scanner.rs:1588

// The "jpg" | "jpeg" arm of load_image_fast (scanner.rs:461-508): the decoded image (Luma8 or Rgb8, what the reference wraps into a
// DynamicImage), or nullopt for anything the library does not take -- the caller goes on to its next decoder (scanner.rs:510-551).
struct DecodedImage {
    std::vector<uint8_t> pixels;
    uint32_t width = 0, height = 0, channels = 0;
    ImageView view() const { return ImageView{pixels.data(), width, height, channels}; }
};
inline std::optional<DecodedImage> load_image_fast(const uint8_t *bytes, size_t len, int flavour = RPH_JPEG_ZUNE)
{
    DecodedImage img;
    if (rph_jpeg_info(bytes, len, &img.width, &img.height, &img.channels) != RPH_OK) return std::nullopt;
    img.pixels.resize((size_t)img.width * img.height * img.channels);
    if (rph_jpeg_decode(Context::get(), bytes, len, flavour, img.pixels.data()) != RPH_OK) return std::nullopt;
    return img;
}
// A batch of files -> hashes (scan loop of scanner.rs:1202-1418 for JPEG files): hash, quality and validity per file; a file that
// cannot be decoded here comes back with valid = false and keeps the caller's own path.
struct FileHash {
    pdqhash::Hash hash{};
    float quality = 0.f;
    bool valid = false;
    int status = 0;
};
inline std::vector<FileHash> hash_jpeg_files(const std::vector<std::pair<const uint8_t *, size_t>> &files, int flavour = RPH_JPEG_ZUNE, uint32_t threads = 0)
{
    const uint32_t n = (uint32_t)files.size();
    std::vector<const uint8_t *> ptr(n);
    std::vector<size_t> len(n);
    for (uint32_t i = 0; i < n; i++) ptr[i] = files[i].first, len[i] = files[i].second;
    std::vector<uint8_t> hashes((size_t)n * 32), valid(n);
    std::vector<float> quality(n);
    std::vector<int32_t> status(n);
    check(rph_jpeg_pdq_hash_batch(Context::get(), ptr.data(), len.data(), n, flavour, threads, hashes.data(), quality.data(), nullptr, nullptr, valid.data(), status.data()),
          "hash_jpeg_files");
    std::vector<FileHash> out(n);
    for (uint32_t i = 0; i < n; i++) {
        std::memcpy(out[i].hash.data(), &hashes[(size_t)i * 32], 32);
        out[i].quality = quality[i];
        out[i].valid = valid[i] != 0;
        out[i].status = status[i];
    }
    return out;
}
inline bool is_low_pdq_quality(std::optional<uint16_t> q) { return rph_is_low_pdq_quality(q ? (int32_t)*q : -1) != 0; }  // :1592

struct ScannedFile {  // the fields group_files_generic reads (scanner.rs:1610-1636)
    std::optional<hamminghash::Hash256> pdqhash;
    std::optional<pdqhash::PdqFeatures> pdq_features;
    std::optional<uint16_t> pdq_quality;
};

// group_with_pdqhash / group_files_generic up to the union-find (scanner.rs:1640-1817).
// Returns (groups of indices into valid_files, comparison_count).
inline std::pair<std::vector<std::vector<uint32_t>>, size_t> group_with_pdqhash(const std::vector<ScannedFile> &valid_files,
                                                                               uint32_t similarity)
{
    if (similarity > hamminghash::MAX_SIMILARITY_256)  // scanner.rs:1650-1655 (assert!)
        throw std::runtime_error("Similarity distances above 63 require R=4 bit-flip checks, which are not implemented.");
    std::vector<uint32_t> dense_to_sparse;  // scanner.rs:1658-1669
    std::vector<uint8_t> hashes, has;
    std::vector<float> coeffs;
    std::vector<int32_t> quality;
    bool any_features = false;
    for (size_t i = 0; i < valid_files.size(); i++) {
        const auto &f = valid_files[i];
        if (!f.pdqhash) continue;
        dense_to_sparse.push_back((uint32_t)i);
        hashes.insert(hashes.end(), f.pdqhash->begin(), f.pdqhash->end());
        has.push_back(f.pdq_features ? 1 : 0);
        any_features |= f.pdq_features.has_value();
        const size_t at = coeffs.size();
        coeffs.resize(at + 256, 0.0f);
        if (f.pdq_features) std::memcpy(&coeffs[at], f.pdq_features->coefficients.data(), 1024);
        quality.push_back(f.pdq_quality ? (int32_t)*f.pdq_quality : -1);
    }
    const size_t n = dense_to_sparse.size();
    std::vector<uint32_t> members(n ? n : 1), offsets(n / 2 + 2);
    uint32_t ng = 0;
    uint64_t cmp = 0;
    check(rph_group_files_pdq(Context::get(), hashes.data(), any_features ? coeffs.data() : nullptr, any_features ? has.data() : nullptr,
                              quality.data(), n, similarity, members.data(), offsets.data(), &ng, &cmp),
          "group_with_pdqhash");
    std::vector<std::vector<uint32_t>> out(ng);
    for (uint32_t g = 0; g < ng; g++)
        for (uint32_t t = offsets[g]; t < offsets[g + 1]; t++) out[g].push_back(dense_to_sparse[members[t]]);
    return {out, (size_t)cmp};
}
}  // namespace scanner

}  // namespace rupphash
