// mih_build.hip -- MIHIndex::new (/root/reference/src/hamminghash.rs:89-130) on the device, for both implementations of
// HammingHash: [u8;32] (16 chunks of 16 bits, 65536 buckets each, :43-63) and u64 (8 chunks of 8 bits, 256 buckets, :23-41).
//
// The reference builds the CSR serially: count, prefix sum, fill in ascending id order per bucket.
// Here: one histogram kernel, one exclusive scan, and per chunk k one stable 16-bit radix sort of
// (chunk value, id) pairs -- stable, so ids stay ascending inside a bucket exactly like the fill
// loop of the reference.  values of chunk k occupy [k*n, (k+1)*n).
#include <hipcub/hipcub.hpp>

#include "rph_internal.h"

namespace {
// CT = chunk type: uint16_t for [u8;32] (chunk k = the little-endian u16 at bytes 2k, 2k+1, hamminghash.rs:50-53),
// uint8_t for u64 (chunk k = (h >> 8k) & 0xFF = byte k of the little-endian u64, :29-31)
template <class CT, uint32_t NUM_CHUNKS, uint32_t NUM_BUCKETS>
__global__ void __launch_bounds__(256) mih_count_kernel(const CT *__restrict__ hashes, uint64_t n, uint32_t *counts)
{
    // thread = (hash i, chunk k)
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * NUM_CHUNKS; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)(t % NUM_CHUNKS);
        atomicAdd(&counts[k * NUM_BUCKETS + hashes[t]], 1u);
    }
}

template <class CT, uint32_t NUM_CHUNKS>
__global__ void __launch_bounds__(256) mih_keys_kernel(const CT *__restrict__ hashes, uint64_t n, uint32_t k, uint16_t *keys, uint32_t *ids)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        keys[i] = hashes[i * NUM_CHUNKS + k];
        ids[i] = (uint32_t)i;
    }
}

template <class CT, uint32_t NUM_CHUNKS, uint32_t NUM_BUCKETS, int KEY_BITS>
int mih_build(const void *d_hashes, uint64_t n, uint32_t *d_offsets, uint32_t *d_values, hipStream_t stream)
{
    const size_t n_flat = (size_t)NUM_CHUNKS * NUM_BUCKETS;
    uint32_t *d_counts = nullptr;
    uint16_t *d_keys = nullptr, *d_keys_out = nullptr;
    uint32_t *d_ids = nullptr;
    void *d_temp = nullptr;
    int rc = RPH_OK;
#define MIH_CHECK(expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            rph_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            rc = RPH_ERR_HIP;                                                                   \
            goto done;                                                                          \
        }                                                                                       \
    } while (0)
    {
        size_t scan_bytes = 0, sort_bytes = 0;
        MIH_CHECK(hipMalloc((void **)&d_counts, (n_flat + 1) * 4));
        MIH_CHECK(hipMemsetAsync(d_counts, 0, (n_flat + 1) * 4, stream));
        const CT *h16 = reinterpret_cast<const CT *>(d_hashes);
        if (n) {
            const uint64_t want = (n * NUM_CHUNKS + 255) / 256;
            hipLaunchKernelGGL((mih_count_kernel<CT, NUM_CHUNKS, NUM_BUCKETS>), dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, stream, h16, n,
                               d_counts);
        }
        MIH_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_counts, d_offsets, (int)(n_flat + 1), stream));
        if (n) {
            MIH_CHECK(hipMalloc((void **)&d_keys, n * 2));
            MIH_CHECK(hipMalloc((void **)&d_keys_out, n * 2));
            MIH_CHECK(hipMalloc((void **)&d_ids, n * 4));
            MIH_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, d_keys, d_keys_out, d_ids, d_values, (int)n, 0, KEY_BITS, stream));
        }
        MIH_CHECK(hipMalloc(&d_temp, std::max(scan_bytes, sort_bytes) + 16));
        MIH_CHECK(hipcub::DeviceScan::ExclusiveSum(d_temp, scan_bytes, d_counts, d_offsets, (int)(n_flat + 1), stream));
        for (uint32_t k = 0; k < NUM_CHUNKS && n; k++) {
            const uint64_t want = (n + 255) / 256;
            hipLaunchKernelGGL((mih_keys_kernel<CT, NUM_CHUNKS>), dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, stream, h16, n, k, d_keys,
                               d_ids);
            MIH_CHECK(hipcub::DeviceRadixSort::SortPairs(d_temp, sort_bytes, d_keys, d_keys_out, d_ids, d_values + (size_t)k * n, (int)n, 0, KEY_BITS,
                                                         stream));
        }
        MIH_CHECK(hipGetLastError());
        MIH_CHECK(hipStreamSynchronize(stream));
    }
done:
    if (d_counts) (void)hipFree(d_counts);
    if (d_keys) (void)hipFree(d_keys);
    if (d_keys_out) (void)hipFree(d_keys_out);
    if (d_ids) (void)hipFree(d_ids);
    if (d_temp) (void)hipFree(d_temp);
    return rc;
#undef MIH_CHECK
}
}  // namespace

int rph_launch_mih_build256(rph_ctx *, const uint8_t *d_hashes, uint64_t n, uint32_t *d_offsets, uint32_t *d_values, hipStream_t stream)
{
    return mih_build<uint16_t, 16, 65536, 16>(d_hashes, n, d_offsets, d_values, stream);
}

int rph_launch_mih_build64(rph_ctx *, const uint64_t *d_hashes, uint64_t n, uint32_t *d_offsets, uint32_t *d_values, hipStream_t stream)
{
    return mih_build<uint8_t, 8, 256, 8>(d_hashes, n, d_offsets, d_values, stream);
}
