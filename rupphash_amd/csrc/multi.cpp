// multi.cpp -- several GPUs under ONE host process (`rph_init(ngpu, ...)`, the RCCL communicator inside).
//
// The reference is a single process (/root/reference/src/scanner.rs:1146: one scan call hashes every file and then groups
// them), so a host that binds this library cannot spread itself over processes the way rupphash_amd/dist.py + torch.distributed
// do.  An rph_multi owns one rph_ctx per device and an RCCL communicator over them (ncclCommInitAll); its entry points run the
// same sharded path as dist.py:
//     images / coefficient vectors are dealt to the devices in contiguous ranges (no communication),
//     ONE ncclAllGather of the per-file hash blocks (the only exchange step of the path),
//     every device sweeps the blocks p == i (mod n_devices) of the sweep's enumeration (part / nparts),
//     the (few) edges go to the host, where the serial union-find runs as in the reference.
// RCCL is loaded at run time (dlopen): a host that never asks for more than one GPU does not need it, and a process that
// already carries an RCCL (PyTorch) shares it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "rph_internal.h"

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

int load_rccl(Rccl &r)
{
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) {
        rph_set_error("rph_multi_init: cannot load RCCL (librccl.so.1): %s", dlerror());
        return RPH_ERR_UNSUPPORTED;
    }
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
        rph_set_error("rph_multi_init: the loaded RCCL lacks a needed symbol");
        return RPH_ERR_UNSUPPORTED;
    }
    return RPH_OK;
}

#define RPH_NCCL_CHECK(m, expr)                                                                                   \
    do {                                                                                                          \
        ncclResult_t r_ = (expr);                                                                                 \
        if (r_ != ncclSuccess) {                                                                                  \
            rph_set_error("%s failed: %s (%s:%d)", #expr, (m)->rccl.GetErrorString(r_), __FILE__, __LINE__);      \
            return RPH_ERR_HIP;                                                                                   \
        }                                                                                                         \
    } while (0)

#define RPH_TRY(expr)                  \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != RPH_OK) return rc_; \
    } while (0)

struct DevMem {  // device buffer bound to a device (freed there)
    void *p = nullptr;
    int device = 0;
    ~DevMem() { reset(); }
    void reset()
    {
        if (p) {
            (void)hipSetDevice(device);
            (void)hipFree(p);
            p = nullptr;
        }
    }
    int alloc(int dev, size_t bytes)
    {
        reset();
        device = dev;
        RPH_HIP_CHECK(hipSetDevice(dev));
        RPH_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1));
        return RPH_OK;
    }
    template <class T>
    T *as() const
    {
        return reinterpret_cast<T *>(p);
    }
};

// contiguous range of device i: the same split as rupphash_amd/dist.py::shard_range
inline void shard_range(uint64_t n, int i, int world, uint64_t &lo, uint64_t &hi)
{
    const uint64_t base = n / world, rem = n % world;
    lo = (uint64_t)i * base + std::min<uint64_t>(i, rem);
    hi = lo + base + ((uint64_t)i < rem ? 1 : 0);
}

}  // namespace

struct rph_multi {
    std::vector<rph_ctx *> ctx;
    std::vector<int> devices;
    std::vector<ncclComm_t> comm;
    Rccl rccl;
    std::mutex mu;  // one multi-device operation at a time (they share the communicator)
    // A device id listed more than once: several contexts ("ranks") on one GPU.  RCCL refuses two ranks on one device, so the exchange step
    // is then carried by device-to-device copies with the collective's semantics -- everything around it (shard ranges, padding of unequal
    // shards, compaction, the part / nparts shares of the sweep, one host thread per rank) runs as on a node.  For rehearsals and tests on
    // a one-GPU box; a node lists every device once and takes the RCCL path.
    bool rehearsal = false;
};

namespace {

// The exchange step: device i holds rows [lo_i, hi_i) of `width` bytes in d_local[i]; afterwards every device holds all n rows in
// d_all[i] (n x width).  One ncclAllGather per device inside a group; unequal shards are padded to the largest for the
// collective and compacted afterwards.
int all_gather_rows(rph_multi *m, const std::vector<const void *> &d_local, uint64_t n, size_t width, std::vector<DevMem> &d_all)
{
    const int world = (int)m->ctx.size();
    uint64_t big = 0;
    std::vector<uint64_t> lo(world), hi(world);
    for (int i = 0; i < world; i++) {
        shard_range(n, i, world, lo[i], hi[i]);
        big = std::max(big, hi[i] - lo[i]);
    }
    const bool even = (n % world) == 0;
    std::vector<DevMem> send(world), recv(world);
    d_all.resize(world);
    for (int i = 0; i < world; i++) {
        RPH_TRY(d_all[i].alloc(m->devices[i], n * width));
        if (!even) {  // pad the shard to `big` rows
            RPH_TRY(send[i].alloc(m->devices[i], big * width));
            RPH_TRY(recv[i].alloc(m->devices[i], (uint64_t)world * big * width));
            RPH_HIP_CHECK(hipMemsetAsync(send[i].p, 0, big * width, m->ctx[i]->stream));
            RPH_HIP_CHECK(hipMemcpyAsync(send[i].p, d_local[i], (hi[i] - lo[i]) * width, hipMemcpyDeviceToDevice, m->ctx[i]->stream));
        }
    }
    if (m->rehearsal) {
        // ncclAllGather's contract with copies: rank i receives rank r's `big * width` bytes at offset r * big * width
        for (int i = 0; i < world; i++) {  // every rank's send buffer is complete before anybody reads it
            RPH_HIP_CHECK(hipSetDevice(m->devices[i]));
            RPH_HIP_CHECK(hipStreamSynchronize(m->ctx[i]->stream));
        }
        for (int i = 0; i < world; i++) {
            RPH_HIP_CHECK(hipSetDevice(m->devices[i]));
            for (int r = 0; r < world; r++)
                RPH_HIP_CHECK(hipMemcpyAsync((even ? d_all[i].as<uint8_t>() : recv[i].as<uint8_t>()) + (uint64_t)r * big * width, even ? d_local[r] : send[r].p, big * width,
                                             hipMemcpyDeviceToDevice, m->ctx[i]->stream));
        }
    } else {
        RPH_NCCL_CHECK(m, m->rccl.GroupStart());
        for (int i = 0; i < world; i++) {
            RPH_HIP_CHECK(hipSetDevice(m->devices[i]));
            RPH_NCCL_CHECK(m, m->rccl.AllGather(even ? d_local[i] : send[i].p, even ? d_all[i].p : recv[i].p, big * width, ncclUint8, m->comm[i], m->ctx[i]->stream));
        }
        RPH_NCCL_CHECK(m, m->rccl.GroupEnd());
    }
    if (!even)
        for (int i = 0; i < world; i++) {
            RPH_HIP_CHECK(hipSetDevice(m->devices[i]));
            for (int r = 0; r < world; r++)
                RPH_HIP_CHECK(hipMemcpyAsync(d_all[i].as<uint8_t>() + lo[r] * width, recv[i].as<uint8_t>() + (uint64_t)r * big * width, (hi[r] - lo[r]) * width,
                                             hipMemcpyDeviceToDevice, m->ctx[i]->stream));
        }
    for (int i = 0; i < world; i++) {  // the padded buffers die with this call
        RPH_HIP_CHECK(hipSetDevice(m->devices[i]));
        RPH_HIP_CHECK(hipStreamSynchronize(m->ctx[i]->stream));
    }
    return RPH_OK;
}

// Every device sweeps its share (part i of n_devices) of the pairs, edges -> host, merged.  d_rows[i] / d_hashes[i] / d_low[i] /
// d_hf[i] are the device's copies of the FULL arrays.
int sweep_all(rph_multi *m, const std::vector<const uint8_t *> &d_rows, uint32_t n_variants, const std::vector<const uint8_t *> &d_hashes,
              const std::vector<const uint8_t *> &d_low, const std::vector<const uint8_t *> &d_hf, uint64_t n, uint32_t threshold,
              std::vector<rph_edge> &edges)
{
    const int world = (int)m->ctx.size();
    std::vector<uint64_t> cap(world, std::max<uint64_t>(1u << 20, 32 * n / world));
    std::vector<DevMem> d_e(world), d_c(world);
    std::vector<unsigned long long> found(world, 0);
    std::vector<bool> done(world, false);
    for (int attempt = 0; attempt < 3; attempt++) {
        for (int i = 0; i < world; i++) {
            if (done[i]) continue;
            rph_ctx *c = m->ctx[i];
            RPH_TRY(d_e[i].alloc(m->devices[i], cap[i] * sizeof(rph_edge)));
            RPH_TRY(d_c[i].alloc(m->devices[i], 8));
            RPH_HIP_CHECK(hipMemsetAsync(d_c[i].p, 0, 8, c->stream));
            RPH_TRY(rph_launch_hamming_sweep(c, d_rows[i], n_variants, d_hashes[i], d_low.empty() ? nullptr : d_low[i], d_hf.empty() ? nullptr : d_hf[i], n,
                                             threshold, (uint32_t)i, (uint32_t)world, d_e[i].as<rph_edge>(), cap[i], d_c[i].as<unsigned long long>(), c->stream,
                                             c->hamming_kernel));
            RPH_HIP_CHECK(hipMemcpyAsync(&found[i], d_c[i].p, 8, hipMemcpyDeviceToHost, c->stream));
        }
        bool again = false;
        for (int i = 0; i < world; i++) {
            if (done[i]) continue;
            RPH_HIP_CHECK(hipSetDevice(m->devices[i]));
            RPH_HIP_CHECK(hipStreamSynchronize(m->ctx[i]->stream));
            if (found[i] <= cap[i]) {
                const size_t at = edges.size();
                edges.resize(at + found[i]);
                if (found[i]) RPH_HIP_CHECK(hipMemcpy(edges.data() + at, d_e[i].p, found[i] * sizeof(rph_edge), hipMemcpyDeviceToHost));
                done[i] = true;
            } else {
                cap[i] = found[i] + found[i] / 16 + 1024;  // rare: this device's share again, into a buffer of the size it reported
                again = true;
            }
        }
        if (!again) return RPH_OK;
    }
    rph_set_error("rph_multi: edge list kept growing");
    return RPH_ERR_CAPACITY;
}

}  // namespace

extern "C" {

int rph_multi_init(const int *devices, int n_devices, rph_multi **out)
{
    return rph_guarded("rph_multi_init", [&]() -> int {
        if (!out || n_devices <= 0) {
            rph_set_error("rph_multi_init: invalid argument");
            return RPH_ERR_INVALID_ARG;
        }
        *out = nullptr;
        std::unique_ptr<rph_multi> m(new rph_multi());
        for (int i = 0; i < n_devices; i++) m->devices.push_back(devices ? devices[i] : i);
        auto fail = [&](int rc) {
            for (rph_ctx *c : m->ctx) rph_shutdown(c);
            return rc;
        };
        for (int i = 0; i < n_devices; i++) {
            rph_ctx *c = nullptr;
            const int rc = rph_init(m->devices[i], &c);
            if (rc != RPH_OK) return fail(rc);
            m->ctx.push_back(c);
        }
        for (int i = 0; i < n_devices; i++)
            for (int j = 0; j < i; j++) m->rehearsal = m->rehearsal || m->devices[i] == m->devices[j];
        if (m->rehearsal) {  // several ranks on one GPU: no communicator (see struct rph_multi)
            *out = m.release();
            return RPH_OK;
        }
        int rc = load_rccl(m->rccl);
        if (rc != RPH_OK) return fail(rc);
        m->comm.resize(n_devices);
        const ncclResult_t r = m->rccl.CommInitAll(m->comm.data(), n_devices, m->devices.data());
        if (r != ncclSuccess) {
            rph_set_error("ncclCommInitAll over %d device(s) failed: %s", n_devices, m->rccl.GetErrorString(r));
            m->comm.clear();
            return fail(RPH_ERR_HIP);
        }
        *out = m.release();
        return RPH_OK;
    });
}

int rph_multi_shutdown(rph_multi *m)
{
    if (!m) return RPH_ERR_INVALID_ARG;
    for (size_t i = 0; i < m->comm.size(); i++) {
        (void)hipSetDevice(m->devices[i]);
        (void)hipStreamSynchronize(m->ctx[i]->stream);
        (void)m->rccl.CommDestroy(m->comm[i]);
    }
    for (rph_ctx *c : m->ctx) rph_shutdown(c);
    delete m;
    return RPH_OK;
}

int rph_multi_size(rph_multi *m) { return m ? (int)m->ctx.size() : 0; }
rph_ctx *rph_multi_ctx(rph_multi *m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }

int rph_multi_hamming_all_pairs(rph_multi *m, const uint8_t *hashes32, uint64_t n, uint32_t threshold, rph_edge *edges, uint64_t cap,
                                uint64_t *n_edges_out)
{
    return rph_guarded("rph_multi_hamming_all_pairs", [&]() -> int {
        if (!m || (!hashes32 && n) || !n_edges_out || (!edges && cap)) {
            rph_set_error("rph_multi_hamming_all_pairs: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        *n_edges_out = 0;
        if (n < 2) return RPH_OK;
        std::lock_guard<std::mutex> lock(m->mu);
        const int world = (int)m->ctx.size();
        // every device receives its shard from the host; the all-gather completes the array on all of them
        std::vector<DevMem> d_shard(world), d_all;
        std::vector<const void *> local(world);
        for (int i = 0; i < world; i++) {
            uint64_t lo, hi;
            shard_range(n, i, world, lo, hi);
            RPH_TRY(d_shard[i].alloc(m->devices[i], (hi - lo) * 32));
            RPH_HIP_CHECK(hipMemcpyAsync(d_shard[i].p, hashes32 + lo * 32, (hi - lo) * 32, hipMemcpyHostToDevice, m->ctx[i]->stream));
            local[i] = d_shard[i].p;
        }
        RPH_TRY(all_gather_rows(m, local, n, 32, d_all));
        std::vector<const uint8_t *> rows(world), none;
        for (int i = 0; i < world; i++) rows[i] = d_all[i].as<uint8_t>();
        std::vector<rph_edge> found;
        RPH_TRY(sweep_all(m, rows, 1, rows, none, none, n, threshold, found));
        *n_edges_out = found.size();
        std::memcpy(edges, found.data(), std::min<uint64_t>(found.size(), cap) * sizeof(rph_edge));
        if (found.size() > cap) {
            rph_set_error("rph_multi_hamming_all_pairs: %zu edges found, capacity %llu", found.size(), (unsigned long long)cap);
            return RPH_ERR_CAPACITY;
        }
        return RPH_OK;
    });
}

int rph_multi_hash_and_group(rph_multi *m, const uint8_t *px, uint32_t n, uint32_t w, uint32_t h, uint32_t channels, size_t row_stride,
                             size_t image_stride, uint32_t similarity, uint8_t *hash32_out, float *quality_out, float *coeffs_out,
                             uint8_t *valid_out, uint32_t *members, uint32_t *offsets, uint32_t *n_groups_out, uint64_t *comparison_count_out)
{
    return rph_guarded("rph_multi_hash_and_group", [&]() -> int {
        if (!m || (!px && n) || !hash32_out || !members || !offsets || !n_groups_out) {
            rph_set_error("rph_multi_hash_and_group: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        if (similarity > RPH_MAX_SIMILARITY_256) {
            rph_set_error("Similarity distances above %u require R=4 bit-flip checks, which are not implemented.", RPH_MAX_SIMILARITY_256);
            return RPH_ERR_INVALID_ARG;
        }
        *n_groups_out = 0;
        offsets[0] = 0;
        if (comparison_count_out) *comparison_count_out = 0;
        if (n == 0) return RPH_OK;
        std::lock_guard<std::mutex> lock(m->mu);
        const int world = (int)m->ctx.size();
        const size_t per = n > 1 ? image_stride : 0;
        // ---- phase 1, no communication: device i hashes images [lo_i, hi_i) (hash, quality, 8 dihedral hashes), one host thread
        //      per device drives its double-buffered upload; the per-file blocks stay on the device for the exchange
        std::vector<DevMem> d_q(world), d_dih(world), d_low(world), d_valid(world);
        std::vector<int> rcs(world, RPH_OK);
        std::vector<std::string> errs(world);
        std::vector<std::thread> th;
        std::vector<uint64_t> lo(world), hi(world);
        for (int i = 0; i < world; i++) {
            shard_range(n, i, world, lo[i], hi[i]);
            const uint64_t cnt = hi[i] - lo[i];
            RPH_TRY(d_q[i].alloc(m->devices[i], cnt * 4));
            RPH_TRY(d_dih[i].alloc(m->devices[i], cnt * 256));
            RPH_TRY(d_low[i].alloc(m->devices[i], cnt));
            RPH_TRY(d_valid[i].alloc(m->devices[i], cnt));
        }
        for (int i = 0; i < world; i++)
            th.emplace_back([&, i] {
                const uint64_t cnt = hi[i] - lo[i];
                if (!cnt) return;
                rcs[i] = rph_pdq_hash_batch_keep(m->ctx[i], px + lo[i] * per, (uint32_t)cnt, w, h, channels, row_stride, image_stride, hash32_out + lo[i] * 32,
                                                 quality_out ? quality_out + lo[i] : nullptr, coeffs_out ? coeffs_out + lo[i] * 256 : nullptr, nullptr,
                                                 valid_out ? valid_out + lo[i] : nullptr, nullptr, d_q[i].p, d_dih[i].p);
                if (rcs[i] != RPH_OK) errs[i] = rph_last_error();
            });
        for (auto &t : th) t.join();
        for (int i = 0; i < world; i++)
            if (rcs[i] != RPH_OK) {
                rph_set_error("device %d: %s", m->devices[i], errs[i].c_str());
                return rcs[i];
            }
        if (w < RPH_PDQ_MIN_DIM || h < RPH_PDQ_MIN_DIM || n < 2) return RPH_OK;  // nothing is hashable (pdqhash.rs:167-169) / nothing to pair
        for (int i = 0; i < world; i++) {
            RPH_HIP_CHECK(hipSetDevice(m->devices[i]));
            RPH_TRY(rph_launch_lowconf_from_quality(d_q[i].as<float>(), nullptr, hi[i] - lo[i], d_low[i].as<uint8_t>(), m->ctx[i]->stream));
        }
        // ---- phase 2, the one exchange: all-gather of the dihedral blocks (8 x 32 B per file; slot 0 is the hash) and of the flags
        std::vector<const void *> local(world);
        std::vector<DevMem> all_dih, all_low, all_hash(world);
        for (int i = 0; i < world; i++) local[i] = d_dih[i].p;
        RPH_TRY(all_gather_rows(m, local, n, 256, all_dih));
        for (int i = 0; i < world; i++) local[i] = d_low[i].p;
        RPH_TRY(all_gather_rows(m, local, n, 1, all_low));
        std::vector<const uint8_t *> rows(world), hashes(world), low(world), none;
        for (int i = 0; i < world; i++) {
            RPH_TRY(all_hash[i].alloc(m->devices[i], (uint64_t)n * 32));
            RPH_HIP_CHECK(hipMemcpy2DAsync(all_hash[i].p, 32, all_dih[i].p, 256, 32, n, hipMemcpyDeviceToDevice, m->ctx[i]->stream));
            rows[i] = all_dih[i].as<uint8_t>();
            hashes[i] = all_hash[i].as<uint8_t>();
            low[i] = all_low[i].as<uint8_t>();
        }
        // ---- phase 3, no communication: variant sweep shares; phase 4: edges to the host, union-find (serial in the reference too)
        std::vector<rph_edge> edges;
        RPH_TRY(sweep_all(m, rows, 8, hashes, low, none, n, similarity, edges));
        if (comparison_count_out) *comparison_count_out = edges.size();
        return rph_host_union_find(edges.data(), edges.size(), n, members, offsets, n_groups_out);
    });
}

// The scan-then-group call from the FILES: every device decodes and hashes a contiguous range of the JPEG files (rph_jpeg_pdq_hash_batch:
// no communication), then the grouping of rph_multi_group_files_pdq over the files that produced a hash (the reference groups
// `valid_entries`, scanner.rs:1658-1662), whose one exchange is the all-gather of the dihedral blocks.
int rph_multi_jpeg_hash_and_group(rph_multi *m, const uint8_t *const *data, const size_t *len, uint32_t n, int flavour, uint32_t n_threads, uint32_t similarity,
                                  uint8_t *hash32_out, float *quality_out, float *coeffs_out, uint8_t *valid_out, int32_t *status_out, uint32_t *members,
                                  uint32_t *offsets, uint32_t *n_groups_out, uint64_t *comparison_count_out)
{
    return rph_guarded("rph_multi_jpeg_hash_and_group", [&]() -> int {
        if (!m || (n && (!data || !len)) || !hash32_out || !members || !offsets || !n_groups_out) {
            rph_set_error("rph_multi_jpeg_hash_and_group: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        if (similarity > RPH_MAX_SIMILARITY_256) {
            rph_set_error("Similarity distances above %u require R=4 bit-flip checks, which are not implemented.", RPH_MAX_SIMILARITY_256);
            return RPH_ERR_INVALID_ARG;
        }
        *n_groups_out = 0;
        offsets[0] = 0;
        if (comparison_count_out) *comparison_count_out = 0;
        if (n == 0) return RPH_OK;
        std::vector<float> q_tmp, c_tmp;
        std::vector<uint8_t> v_tmp;
        float *quality = quality_out, *coeffs = coeffs_out;
        uint8_t *valid = valid_out;
        if (!quality) q_tmp.resize(n), quality = q_tmp.data();
        if (!coeffs) c_tmp.resize((size_t)n * 256), coeffs = c_tmp.data();
        if (!valid) v_tmp.resize(n), valid = v_tmp.data();
        {
            std::lock_guard<std::mutex> lock(m->mu);
            const int world = (int)m->ctx.size();
            const uint32_t threads_each = n_threads ? std::max(1u, n_threads / (uint32_t)world) : 0;
            std::vector<int> rcs(world, RPH_OK);
            std::vector<std::string> errs(world);
            std::vector<std::thread> th;
            for (int i = 0; i < world; i++)
                th.emplace_back([&, i] {
                    uint64_t lo, hi;
                    shard_range(n, i, world, lo, hi);
                    if (hi == lo) return;
                    rcs[i] = rph_jpeg_pdq_hash_batch(m->ctx[i], data + lo, len + lo, (uint32_t)(hi - lo), flavour, threads_each, hash32_out + lo * 32, quality + lo, coeffs + lo * 256,
                                                     nullptr, valid + lo, status_out ? status_out + lo : nullptr);
                    if (rcs[i] != RPH_OK) errs[i] = rph_last_error();
                });
            for (auto &t : th) t.join();
            for (int i = 0; i < world; i++)
                if (rcs[i] != RPH_OK) {
                    rph_set_error("device %d: %s", m->devices[i], errs[i].c_str());
                    return rcs[i];
                }
        }
        // dense ids over the files that have a hash; stored quality = (q * 100).round().clamp(0, 100) (scanner.rs:1416-1417)
        std::vector<uint32_t> dense_to_file;
        for (uint32_t i = 0; i < n; i++)
            if (valid[i]) dense_to_file.push_back(i);
        const uint64_t nd = dense_to_file.size();
        if (nd < 2) return RPH_OK;
        std::vector<uint8_t> h((size_t)nd * 32);
        std::vector<float> c((size_t)nd * 256);
        std::vector<int32_t> q(nd);
        for (uint64_t d = 0; d < nd; d++) {
            const uint32_t f = dense_to_file[d];
            memcpy(&h[d * 32], hash32_out + (size_t)f * 32, 32);
            memcpy(&c[d * 256], coeffs + (size_t)f * 256, 1024);
            const float s = quality[f] * 100.0f;
            q[d] = (int32_t)std::min(100.0f, std::max(0.0f, floorf(s + 0.5f)));
        }
        std::vector<uint32_t> mem(nd), off(nd / 2 + 2);
        uint32_t ng = 0;
        RPH_TRY(rph_multi_group_files_pdq(m, h.data(), c.data(), nullptr, q.data(), nd, similarity, mem.data(), off.data(), &ng, comparison_count_out));
        for (uint32_t g = 0; g <= ng; g++) offsets[g] = off[g];
        for (uint32_t t = 0; t < off[ng]; t++) members[t] = dense_to_file[mem[t]];
        *n_groups_out = ng;
        return RPH_OK;
    });
}

int rph_multi_group_files_pdq(rph_multi *m, const uint8_t *hashes32, const float *coeffs, const uint8_t *has_features, const int32_t *quality,
                              uint64_t n, uint32_t similarity, uint32_t *members, uint32_t *offsets, uint32_t *n_groups_out,
                              uint64_t *comparison_count_out)
{
    return rph_guarded("rph_multi_group_files_pdq", [&]() -> int {
        if (!m || (!hashes32 && n) || !members || !offsets || !n_groups_out) {
            rph_set_error("rph_multi_group_files_pdq: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        if (similarity > RPH_MAX_SIMILARITY_256) {
            rph_set_error("Similarity distances above %u require R=4 bit-flip checks, which are not implemented.", RPH_MAX_SIMILARITY_256);
            return RPH_ERR_INVALID_ARG;
        }
        *n_groups_out = 0;
        offsets[0] = 0;
        if (comparison_count_out) *comparison_count_out = 0;
        if (n < 2) return RPH_OK;
        std::lock_guard<std::mutex> lock(m->mu);
        const int world = (int)m->ctx.size();
        const bool use_hf = coeffs && has_features;
        // device i turns the coefficient vectors of files [lo_i, hi_i) into their 8 dihedral hashes; hashes, flags and feature marks
        // are small enough to go to every device whole
        std::vector<DevMem> d_var(world), d_h(world), d_low(world), d_hf(world), d_stage(world), all_var;
        std::vector<uint8_t> low_host;
        if (quality) {
            low_host.resize(n);
            for (uint64_t i = 0; i < n; i++) low_host[i] = (uint8_t)rph_is_low_pdq_quality(quality[i]);
        }
        for (int i = 0; i < world; i++) {
            rph_ctx *c = m->ctx[i];
            uint64_t lo, hi;
            shard_range(n, i, world, lo, hi);
            RPH_TRY(d_h[i].alloc(m->devices[i], n * 32));
            RPH_HIP_CHECK(hipMemcpyAsync(d_h[i].p, hashes32, n * 32, hipMemcpyHostToDevice, c->stream));
            if (quality) {
                RPH_TRY(d_low[i].alloc(m->devices[i], n));
                RPH_HIP_CHECK(hipMemcpyAsync(d_low[i].p, low_host.data(), n, hipMemcpyHostToDevice, c->stream));
            }
            if (use_hf) {
                RPH_TRY(d_hf[i].alloc(m->devices[i], n));
                RPH_HIP_CHECK(hipMemcpyAsync(d_hf[i].p, has_features, n, hipMemcpyHostToDevice, c->stream));
            }
            if (coeffs) {
                RPH_TRY(d_var[i].alloc(m->devices[i], (hi - lo) * 256));
                const uint64_t step = 1u << 18;
                RPH_TRY(d_stage[i].alloc(m->devices[i], std::min<uint64_t>(step, std::max<uint64_t>(hi - lo, 1)) * 1024));
                for (uint64_t first = lo; first < hi; first += step) {
                    const uint32_t cnt = (uint32_t)std::min<uint64_t>(step, hi - first);
                    RPH_HIP_CHECK(hipMemcpyAsync(d_stage[i].p, coeffs + first * 256, (size_t)cnt * 1024, hipMemcpyHostToDevice, c->stream));
                    RPH_TRY(rph_launch_pdq_from_coeffs(d_stage[i].as<float>(), cnt, nullptr, d_var[i].as<uint8_t>() + (first - lo) * 256, c->stream));
                }
            }
        }
        std::vector<const uint8_t *> rows(world), hashes(world), low, hf;
        if (coeffs) {
            std::vector<const void *> local(world);
            for (int i = 0; i < world; i++) local[i] = d_var[i].p;
            RPH_TRY(all_gather_rows(m, local, n, 256, all_var));  // the exchange step
            if (use_hf)
                for (int i = 0; i < world; i++) {
                    RPH_HIP_CHECK(hipSetDevice(m->devices[i]));
                    RPH_TRY(rph_launch_featureless_variants(d_h[i].as<uint8_t>(), d_hf[i].as<uint8_t>(), n, all_var[i].as<uint8_t>(), m->ctx[i]->stream));
                }
        }
        for (int i = 0; i < world; i++) {
            rows[i] = coeffs ? all_var[i].as<uint8_t>() : d_h[i].as<uint8_t>();
            hashes[i] = d_h[i].as<uint8_t>();
        }
        if (quality)
            for (int i = 0; i < world; i++) low.push_back(d_low[i].as<uint8_t>());
        if (use_hf)
            for (int i = 0; i < world; i++) hf.push_back(d_hf[i].as<uint8_t>());
        std::vector<rph_edge> edges;
        RPH_TRY(sweep_all(m, rows, coeffs ? 8 : 1, hashes, low, hf, n, similarity, edges));
        if (comparison_count_out) *comparison_count_out = edges.size();
        return rph_host_union_find(edges.data(), edges.size(), n, members, offsets, n_groups_out);
    });
}

}  // extern "C"
