// host_grouping.cpp -- host-side halves of the reference's groupers (the parts that are serial in
// the reference too) and its scalar integer helpers.
//   union-find + group listing      /root/reference/src/scanner.rs:1781-1817
//   greedy star clustering          /root/reference/src/hamminghash.rs:245-268
//   HammingHash scalar functions    /root/reference/src/hamminghash.rs:23-63
//   64-bit pHash bit operations     /root/reference/src/phash.rs:137-255
#include <algorithm>
#include <cstring>
#include <vector>

#include "rph_internal.h"

// ---------------------------------------------------------------------------------------------
// scalar helpers
// ---------------------------------------------------------------------------------------------
extern "C" uint32_t rph_hamming_distance256(const uint8_t *a, const uint8_t *b)
{
    uint64_t x[4], y[4];
    std::memcpy(x, a, 32);
    std::memcpy(y, b, 32);
    return (uint32_t)(__builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
                      __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]));
}
extern "C" uint32_t rph_hamming_distance64(uint64_t a, uint64_t b) { return (uint32_t)__builtin_popcountll(a ^ b); }
extern "C" uint16_t rph_get_chunk256(const uint8_t *h, uint32_t k)
{
    return (uint16_t)(h[2 * k] | ((uint16_t)h[2 * k + 1] << 8));  // u16::from_le_bytes, hamminghash.rs:50-53
}
extern "C" uint16_t rph_get_chunk64(uint64_t h, uint32_t k) { return (uint16_t)((h >> (k * 8)) & 0xFF); }
extern "C" int rph_is_low_pdq_quality(int32_t q) { return q >= 0 && q < RPH_PDQ_MIN_QUALITY; }

extern "C" void rph_pdq_target_dimensions(uint32_t w, uint32_t h, uint32_t max_dim, uint32_t *nw, uint32_t *nh)
{
    // calculate_target_dimensions, pdqhash.rs:224-235
    if (w == 0 || h == 0) {
        *nw = std::max(w, 1u);
        *nh = std::max(h, 1u);
    } else if (w > h) {
        *nw = max_dim;
        *nh = (uint32_t)std::max<uint64_t>((uint64_t)h * max_dim / w, 1);
    } else {
        *nw = (uint32_t)std::max<uint64_t>((uint64_t)w * max_dim / h, 1);
        *nh = max_dim;
    }
}

// ---- pHash: bit i (from the MSB) is DCT cell (y = i / 8, x = i % 8); phash.rs:150-230 ----
static inline uint64_t remap(uint64_t hash, bool transpose, int flip_rule)
{
    uint64_t out = 0;
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            const int dx = transpose ? y : x, dy = transpose ? x : y;
            uint64_t bit = (hash >> (63 - (8 * y + x))) & 1;
            bool flip = false;
            switch (flip_rule) {
                case 0: flip = (dx & 1) != 0; break;         // odd destination column
                case 1: flip = (dy & 1) != 0; break;         // odd destination row
                case 2: flip = ((dx + dy) & 1) != 0; break;  // odd x + y
            }
            out |= (bit ^ (flip ? 1u : 0u)) << (63 - (8 * dy + dx));
        }
    return out;
}
extern "C" uint64_t rph_phash_rotate_90(uint64_t h) { return remap(h, true, 0); }
extern "C" uint64_t rph_phash_rotate_180(uint64_t h) { return remap(h, false, 2); }
extern "C" uint64_t rph_phash_rotate_270(uint64_t h) { return remap(h, true, 1); }
extern "C" uint64_t rph_phash_flip_horizontal(uint64_t h) { return remap(h, false, 0); }
extern "C" uint64_t rph_phash_rotation_invariant(uint64_t h)
{
    return std::min(std::min(h, rph_phash_rotate_90(h)), std::min(rph_phash_rotate_180(h), rph_phash_rotate_270(h)));
}
extern "C" void rph_phash_dihedral(uint64_t h, uint64_t out[8])
{
    const uint64_t f = rph_phash_flip_horizontal(h);
    out[0] = h;
    out[1] = rph_phash_rotate_90(h);
    out[2] = rph_phash_rotate_180(h);
    out[3] = rph_phash_rotate_270(h);
    out[4] = f;
    out[5] = rph_phash_rotate_90(f);
    out[6] = rph_phash_rotate_180(f);
    out[7] = rph_phash_rotate_270(f);
}

// ---------------------------------------------------------------------------------------------
// union-find (scanner.rs:1781-1817)
// ---------------------------------------------------------------------------------------------
namespace {
inline uint32_t uf_find(std::vector<uint32_t> &parent, uint32_t i)
{
    uint32_t root = i;
    while (root != parent[root]) root = parent[root];
    while (i != root) {
        const uint32_t next = parent[i];
        parent[i] = root;
        i = next;
    }
    return root;
}
}  // namespace

int rph_host_union_find(const rph_edge *edges, uint64_t n_edges, uint64_t n, uint32_t *members, uint32_t *offsets,
                        uint32_t *n_groups_out)
{
    std::vector<uint32_t> parent(n);
    for (uint64_t i = 0; i < n; i++) parent[i] = (uint32_t)i;
    for (uint64_t e = 0; e < n_edges; e++) {
        if (edges[e].i >= n || edges[e].j >= n) {
            rph_set_error("union-find: edge %llu references file %u/%u outside n=%llu", (unsigned long long)e, edges[e].i,
                          edges[e].j, (unsigned long long)n);
            return RPH_ERR_INVALID_ARG;
        }
        const uint32_t ri = uf_find(parent, edges[e].i), rj = uf_find(parent, edges[e].j);
        if (ri != rj) parent[ri] = rj;  // scanner.rs:1797-1803
    }
    // Components with more than one member; members ascending (the reference pushes i in 0..n order,
    // scanner.rs:1810-1814), groups ordered by their first member.
    std::vector<uint32_t> root(n), count(n, 0), slot(n, UINT32_MAX);
    for (uint64_t i = 0; i < n; i++) {
        root[i] = uf_find(parent, (uint32_t)i);
        count[root[i]]++;
    }
    uint32_t ng = 0, total = 0;
    offsets[0] = 0;
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t r = root[i];
        if (count[r] > 1 && slot[r] == UINT32_MAX) {
            slot[r] = ng++;
            total += count[r];
            offsets[ng] = total;
        }
    }
    std::vector<uint32_t> fill(ng, 0);
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t r = root[i];
        if (count[r] > 1) {
            const uint32_t g = slot[r];
            members[offsets[g] + fill[g]++] = (uint32_t)i;
        }
    }
    *n_groups_out = ng;
    return RPH_OK;
}

// ---------------------------------------------------------------------------------------------
// find_groups from an exhaustive edge list (hamminghash.rs:191-271)
//
// The reference's adjacency of query i is the list of candidates in first-seen order of its probe
// loop: chunk k ascending; within a chunk the exact bucket, then the flips of bit 0..15; within a
// bucket ascending id (CSR fill order).  A candidate is first seen at the smallest k whose 16-bit
// difference has popcount <= tolerance; the sweep kernel stored (k, slot) in edge.flags, and the
// key is symmetric in (i, j).  Pairs that no probe reaches are not adjacent.
// ---------------------------------------------------------------------------------------------
int rph_host_find_groups(const rph_edge *edges, uint64_t n_edges, uint64_t n, uint32_t *members, uint32_t *offsets,
                         uint32_t *n_groups_out)
{
    std::vector<uint64_t> deg(n + 1, 0);
    for (uint64_t e = 0; e < n_edges; e++) {
        if (!(edges[e].flags & RPH_EDGE_MIH_R1)) continue;
        if (edges[e].i >= n || edges[e].j >= n) {
            rph_set_error("find_groups: edge %llu outside n", (unsigned long long)e);
            return RPH_ERR_INVALID_ARG;
        }
        deg[edges[e].i + 1]++;
        deg[edges[e].j + 1]++;
    }
    for (uint64_t i = 0; i < n; i++) deg[i + 1] += deg[i];
    struct Nb {
        uint32_t key;  // (k << 5 | slot)
        uint32_t id;
    };
    std::vector<Nb> adj(deg[n]);
    std::vector<uint64_t> cur(deg.begin(), deg.end() - 1);
    for (uint64_t e = 0; e < n_edges; e++) {
        if (!(edges[e].flags & RPH_EDGE_MIH_R1)) continue;
        const uint32_t key = edges[e].flags & RPH_EDGE_PROBE_MASK;
        adj[cur[edges[e].i]++] = {key, edges[e].j};
        adj[cur[edges[e].j]++] = {key, edges[e].i};
    }
    for (uint64_t i = 0; i < n; i++)
        std::sort(adj.begin() + deg[i], adj.begin() + deg[i + 1],
                  [](const Nb &a, const Nb &b) { return a.key != b.key ? a.key < b.key : a.id < b.id; });

    // greedy clustering, hamminghash.rs:245-268
    std::vector<uint8_t> visited(n, 0);
    uint32_t ng = 0, total = 0;
    offsets[0] = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (visited[i] || deg[i + 1] == deg[i]) continue;
        const uint32_t start = total;
        members[total++] = (uint32_t)i;
        visited[i] = 1;
        for (uint64_t t = deg[i]; t < deg[i + 1]; t++) {
            const uint32_t nb = adj[t].id;
            if (!visited[nb]) {
                visited[nb] = 1;
                members[total++] = nb;
            }
        }
        if (total - start > 1)
            offsets[++ng] = total;
        else
            total = start;
    }
    *n_groups_out = ng;
    return RPH_OK;
}
