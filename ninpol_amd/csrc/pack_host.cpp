// pack_host.cpp -- native table packing: what Interpolator.process_mesh / load_data /
// compute_diffusion_magnitude do in Python loops in the reference (interpolator.pyx:255-451, 501-509; 7.1 s at 1 M
// cells there), as OpenMP loops behind the C ABI.  Built with -ffp-contract=off: diff_mag must be the reference's
// value bit for bit.
#include <cstdint>
#include <cstring>

#include "../../include/ninpol_amd.h"

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
#pragma omp parallel for schedule(static)
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
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) std::memcpy(dst + i * take, src + i * take, (size_t)take * 8);
        return NIN_OK;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
        for (int64_t k = 0; k < take; ++k) dst[i * take + k] = src[i * src_cols + k];
    return NIN_OK;
}

// interpolator.pyx:501-509 AS COMPILED (cdivision: `detKs ** (1 / 3)` is `** 0`): (1 - 3 * 1.0 / tr K)^2 with the
// trace summed in np.trace's order.
int nin_diff_mag(const double *permeability, int64_t n_elems, double *diff_mag) {
    if (!permeability || !diff_mag || n_elems < 0) return NIN_EINVAL;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_elems; ++i) {
        const double *K = permeability + 9 * i;
        const double tr = (K[0] + K[4]) + K[8];
        const double x = 1 - (3 * 1.0 / tr);
        diff_mag[i] = x * x;
    }
    return NIN_OK;
}

}  // extern "C"
