// pack_host.cpp -- native table packing: what Interpolator.process_mesh / load_data /
// compute_diffusion_magnitude do in Python loops in the reference (interpolator.pyx:255-451, 501-509; 7.1 s at 1 M
// cells there), as OpenMP loops behind the C ABI.  Built with -ffp-contract=off: diff_mag must be the reference's
// value bit for bit.
#include <omp.h>

#include <cstdint>
#include <algorithm>
#include <vector>
#include <cstring>

#include "../../include/ninpol_amd.h"
#include "host_threads.hpp"

// (every parallel region below names its team, nin::host_team(): the CPUs this process may use -- host_threads.hpp; the
// process-wide OpenMP default is left alone)

namespace nin {
// Node flags of the kernels (bit 0: boundary point, bit 1: Neumann flag) from the points_data row: `.astype(int)`
// (idw.pyx:28, gls.pyx:55) truncates toward zero, anything non-zero after that counts (idw.pyx:62 `== 0`).
void pack_node_flags(const double *neumann_flag, const uint8_t *boundary_points, int64_t n, uint8_t *out) {
#pragma omp parallel for schedule(static) num_threads(nin::host_team())
    for (int64_t p = 0; p < n; ++p) {
        const long long as_int = (long long)neumann_flag[p];
        out[p] = (uint8_t)((boundary_points[p] ? 1 : 0) | (as_int != 0 ? 2 : 0));
    }
}
}  // namespace nin

extern "C" {

// interpolator.pyx:333-361: fixed-width -1 padded connectivity [n_elems][8] + element_types [n_elems] from per-type
// blocks (block b: rows[b] cells of cols[b] nodes each, row-major int64, element type id type_id[b]), in block order.
int nin_pack_connectivity(int32_t n_blocks, const int64_t *const *block_data, const int64_t *rows, const int64_t *cols,
                          const int64_t *type_id, int64_t *connectivity, int64_t *element_types) {
    if (n_blocks < 0 || (n_blocks && (!block_data || !rows || !cols || !type_id)) || !connectivity || !element_types) return NIN_EINVAL;
    int64_t at = 0;
    for (int32_t b = 0; b < n_blocks; ++b) {
        const int64_t n = rows[b], w = cols[b], t = type_id[b];
        if (n < 0 || w < 1 || w > 8 || (n && !block_data[b])) return NIN_EINVAL;
        const int64_t *src = block_data[b];
        int64_t *dst = connectivity + at * 8;
        int64_t *ty = element_types + at;
#pragma omp parallel for schedule(static) num_threads(nin::host_team())
        for (int64_t i = 0; i < n; ++i) {
            for (int64_t k = 0; k < w; ++k) dst[i * 8 + k] = src[i * w + k];
            for (int64_t k = w; k < 8; ++k) dst[i * 8 + k] = -1;
            ty[i] = t;
        }
        at += n;
    }
    return NIN_OK;
}

// interpolator.pyx:397-419: one row of a data table -- the first `take` columns of a row-major (n, src_cols) array,
// flattened into dst[0 : n * take].
int nin_pack_table_row(const double *src, int64_t n, int64_t src_cols, int64_t take, double *dst) {
    if (!src || !dst || n < 0 || src_cols < 1 || take < 1 || take > src_cols) return NIN_EINVAL;
    if (take == src_cols) {
#pragma omp parallel for schedule(static) num_threads(nin::host_team())
        for (int64_t i = 0; i < n; ++i) std::memcpy(dst + i * take, src + i * take, (size_t)take * 8);
        return NIN_OK;
    }
#pragma omp parallel for schedule(static) num_threads(nin::host_team())
    for (int64_t i = 0; i < n; ++i)
        for (int64_t k = 0; k < take; ++k) dst[i * take + k] = src[i * src_cols + k];
    return NIN_OK;
}

// interpolator.pyx:501-509 AS COMPILED (cdivision: `detKs ** (1 / 3)` is `** 0`): (1 - 3 * 1.0 / tr K)^2 with the
// trace summed in np.trace's order.
int nin_diff_mag(const double *permeability, int64_t n_elems, double *diff_mag) {
    if (!permeability || !diff_mag || n_elems < 0) return NIN_EINVAL;
#pragma omp parallel for schedule(static) num_threads(nin::host_team())
    for (int64_t i = 0; i < n_elems; ++i) {
        const double *K = permeability + 9 * i;
        const double tr = (K[0] + K[4]) + K[8];
        const double x = 1 - (3 * 1.0 / tr);
        diff_mag[i] = x * x;
    }
    return NIN_OK;
}

// Hash of a byte range, in parallel: 1 MiB chunks, each run through four multiply-rotate lanes over its 8-byte words
// (the structure of xxHash64's stripe loop, own constants and finish), the chunk digests folded in order.  Used by
// ninpol_amd.Interpolator to tell whether the permeability table on the device still is the caller's (the reference re-reads
// its tables on every call, interpolator.pyx:583-600): ALL bytes are hashed -- a sample would miss an in-place edit.
int nin_hash64(const void *data, size_t bytes, uint64_t *out) {
    if ((!data && bytes) || !out) return NIN_EINVAL;
    constexpr uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull;
    auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto mix = [&](uint64_t acc, uint64_t w) { return rotl(acc + w * P2, 31) * P1; };
    auto avalanche = [&](uint64_t h) { h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32; return h; };
    constexpr size_t CH = (size_t)1 << 20;
    const size_t n_chunks = (bytes + CH - 1) / CH;
    std::vector<uint64_t> part(n_chunks ? n_chunks : 1, 0);
    const unsigned char *base = static_cast<const unsigned char *>(data);
    // memory-bound: 16 threads saturate it; never more than the CPUs the process may use (host_threads.hpp)
    const int nt = std::min(nin::host_team(), 16);
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int64_t c = 0; c < (int64_t)n_chunks; ++c) {
        const unsigned char *q = base + (size_t)c * CH;
        const size_t len = std::min(CH, bytes - (size_t)c * CH);
        uint64_t a0 = P1 + (uint64_t)c, a1 = P2, a2 = P3, a3 = P1 ^ P3;
        size_t i = 0;
        for (; i + 32 <= len; i += 32) {
            uint64_t w[4];
            std::memcpy(w, q + i, 32);
            a0 = mix(a0, w[0]); a1 = mix(a1, w[1]); a2 = mix(a2, w[2]); a3 = mix(a3, w[3]);
        }
        uint64_t h = rotl(a0, 1) + rotl(a1, 7) + rotl(a2, 12) + rotl(a3, 18) + (uint64_t)len;
        for (; i < len; ++i) h = rotl(h ^ (q[i] * P3), 11) * P1;
        part[(size_t)c] = avalanche(h);
    }
    uint64_t h = P3 + (uint64_t)bytes;
    for (size_t c = 0; c < n_chunks; ++c) h = mix(h, part[c]) ^ (h >> 29);
    *out = avalanche(h);
    return NIN_OK;
}

}  // extern "C"
