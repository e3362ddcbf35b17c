// kernels_csr.hip -- device-side finish of interpolator.pyx:622-624.
//
// The reference turns the dense weight table into COO triplets in a serial Python-level loop, builds a
// scipy csr_matrix from them and calls eliminate_zeros(); at 1 M cells that tail costs 1.5-3 s, more
// than the IDW / LS kernels themselves.  Here the weights are already written in CSR
// position (indptr = esup_ptr, indices = esup), so the finish is: count the entries of each row that
// are != 0 (what eliminate_zeros keeps: NaNs stay, +-0 go), exclusive-scan, and copy the survivors.
// Pure streaming, HBM-bound.
#include <hip/hip_runtime.h>

#include "device_grid.hpp"
#include "launch.hpp"

namespace nin {

namespace {

// A wavefront owns 64 consecutive nodes; their rows are ONE contiguous run of `data`.  The lanes walk the run 64 entries at a time
// (whole lines in), a ballot marks the entries that stay, and every lane counts the marks that fall into its own row.  (The first
// version gave a lane its node's row: 64-byte lane strides on a hexahedron mesh, 200 bytes on tetrahedra -- 1.70 ms at 10 M cells.)
__global__ __launch_bounds__(256) void nin_row_nnz_kernel(GridView g, const double *__restrict__ data,
                                                          int32_t *__restrict__ row_nnz, int32_t tile_begin, int32_t tile_end, int32_t p_end) {
    const int lane = threadIdx.x & 63;
    const int32_t wpb = blockDim.x >> 6;
    for (int32_t tile = tile_begin + blockIdx.x * wpb + (threadIdx.x >> 6); tile < tile_end; tile += gridDim.x * wpb) {
        const int32_t p0 = tile * 64, p = p0 + lane, pe = p0 + 64 < p_end ? p0 + 64 : p_end;
        const int32_t run_b = g.esup_ptr[p0], run_e = g.esup_ptr[pe];
        const bool live = p < p_end;
        const int32_t b = live ? g.esup_ptr[p] : run_e, e = live ? g.esup_ptr[p + 1] : run_e;
        int32_t c = 0;
        for (int32_t i0 = run_b; i0 < run_e; i0 += 64) {
            const int32_t i = i0 + lane;
            const unsigned long long m = __ballot(i < run_e && data[i] != 0.0);   // what eliminate_zeros keeps: NaNs stay, +-0 go
            const int32_t lo = (b > i0 ? b : i0) - i0, hi = (e < i0 + 64 ? e : i0 + 64) - i0;
            if (hi > lo) {
                const unsigned long long below_hi = hi >= 64 ? ~0ull : (1ull << hi) - 1ull;
                c += __popcll(m & below_hi & ~((1ull << lo) - 1ull));
            }
        }
        if (live) row_nnz[p] = c;
    }
}

// A wavefront owns 64 consecutive nodes: their rows are ONE contiguous run of data / esup and their surviving entries one
// contiguous run of the output.  The lanes walk the run 64 entries at a time (whole lines in), a ballot + popcount gives every
// surviving entry its place (whole lines out).  The first version gave a lane its node's row: 64-byte lane strides both ways,
// rocprofv3 WRITE_SIZE 4.97 GB for 0.95 GB of output, 2.34 ms at 10 M cells (profiles/r03, before) -- now ~0.5 ms.
__global__ __launch_bounds__(256) void nin_compact_kernel(GridView g, const double *__restrict__ data,
                                                          const int32_t *__restrict__ new_ptr,
                                                          int32_t *__restrict__ indices, double *__restrict__ vals,
                                                          int32_t tile_begin, int32_t tile_end) {
    const int lane = threadIdx.x & 63;
    const int32_t wpb = blockDim.x >> 6;
    for (int32_t tile = tile_begin + blockIdx.x * wpb + (threadIdx.x >> 6); tile < tile_end; tile += gridDim.x * wpb) {
        const int32_t p0 = tile * 64, pe = p0 + 64 < g.n_points ? p0 + 64 : g.n_points;
        const int32_t run_b = g.esup_ptr[p0], run_e = g.esup_ptr[pe];
        int32_t at = new_ptr[p0];
        for (int32_t i0 = run_b; i0 < run_e; i0 += 64) {
            const int32_t i = i0 + lane;
            const bool in = i < run_e;
            const double d = in ? data[i] : 0.0;
            const int32_t c = in ? g.esup[i] : 0;
            const bool keep = in && d != 0.0;                    // what eliminate_zeros keeps: NaNs stay, +-0 go
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int32_t pos = at + __popcll(m & ((1ull << lane) - 1ull));
                indices[pos] = c;
                vals[pos] = d;
            }
            at += __popcll(m);
        }
    }
}

// values[p] = sum_j data[esup_ptr[p] + j] * u[esup[esup_ptr[p] + j]]: what every caller of interpolate() does next
// (`weights.dot(u)`, tests/utils/analytical.py:236), without the matrix leaving the device.  One lane per node, the sum in
// row order; the 64 rows of a wavefront's nodes are one contiguous run of data / esup, copied HBM -> LDS cooperatively (whole
// lines; a lane reading its own row straight from HBM strides 64 bytes: FETCH_SIZE 2.65 GB raw for 0.97 GB of rows, 0.84 ms at
// 10 M cells), u is gathered (L2).  NF fields at once (u: [k][n_elems], values: [k][n_points]): the run is staged once.
constexpr int kApplyCap = 1024;   // most entries of LDS a wavefront may own (12 KiB) on a mesh whose 64 longest rows fit that; else
constexpr int kApplyCapLong = 1536;   // ... 18 KiB and tiles of 64, 32 or 16 nodes (kernels_idw_ls.hip's rule); a longer run goes straight from HBM
// `cap`: the entries of LDS each wavefront owns in THIS launch (64 x the longest row, at most kApplyCap: 6 KiB a wave on a
// hexahedron mesh -- six workgroups per CU instead of three; the kernel lives on loads in flight: 0.56 -> 0.33 ms at 10 M cells)
template <int NF>
__global__ __launch_bounds__(256) void nin_apply_kernel(GridView g, const double *__restrict__ data,
                                                        const double *__restrict__ u, int32_t k0, int32_t k,
                                                        double *__restrict__ values, int32_t cap, int32_t tn) {
    extern __shared__ double apply_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *const wl = apply_lds + (size_t)wave * cap;
    int32_t *const cl = reinterpret_cast<int32_t *>(apply_lds + 4 * (size_t)cap) + (size_t)wave * cap;
    const size_t E = (size_t)g.n_elems, P = (size_t)g.n_points;
    const int32_t n_tiles = (g.n_points + tn - 1) / tn;
    for (int32_t tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        const int32_t p0 = tile * tn, p = p0 + lane, pe = p0 + tn < g.n_points ? p0 + tn : g.n_points;
        const int32_t run_b = g.esup_ptr[p0], run_e = g.esup_ptr[pe], len = run_e - run_b;
        const bool staged = len <= cap, live = lane < tn && p < g.n_points;
        if (staged) {
            for (int32_t i = lane; i < len; i += 64) { wl[i] = data[run_b + i]; cl[i] = g.esup[run_b + i]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (live) {
            const int32_t b = g.esup_ptr[p], e = g.esup_ptr[p + 1];
            double acc[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[f] = 0.0;
            // eight entries at a time: their gathers of u are issued together, the sums stay in row order
            int32_t q = b;
            for (; q + 8 <= e; q += 8) {
                double w8[8], u8[8][NF];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    w8[j] = staged ? wl[q + j - run_b] : data[q + j];
                    const size_t c = (size_t)(staged ? cl[q + j - run_b] : g.esup[q + j]);
#pragma unroll
                    for (int f = 0; f < NF; ++f) u8[j][f] = k0 + f < k ? u[(size_t)(k0 + f) * E + c] : 0.0;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#pragma unroll
                    for (int f = 0; f < NF; ++f) acc[f] += w8[j] * u8[j][f];
                }
            }
            for (; q < e; ++q) {
                const double w = staged ? wl[q - run_b] : data[q];
                const size_t c = (size_t)(staged ? cl[q - run_b] : g.esup[q]);
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    if (k0 + f < k) acc[f] += w * u[(size_t)(k0 + f) * E + c];
            }
#pragma unroll
            for (int f = 0; f < NF; ++f)
                if (k0 + f < k) values[(size_t)(k0 + f) * P + p] = acc[f];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// the listed nodes only: one lane per node, its row straight from HBM, every field in turn (sums in row order)
__global__ __launch_bounds__(256) void nin_apply_list_kernel(GridView g, const double *__restrict__ data, const double *__restrict__ u,
                                                             int32_t k, double *__restrict__ values, const int32_t *__restrict__ list,
                                                             int32_t count) {
    const size_t E = (size_t)g.n_elems, P = (size_t)g.n_points;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t p = list[i];
        const int32_t b = g.esup_ptr[p], e = g.esup_ptr[p + 1];
        for (int32_t f = 0; f < k; ++f) {
            double acc = 0.0;
            for (int32_t q = b; q < e; ++q) acc += data[q] * u[(size_t)f * E + (size_t)g.esup[q]];
            values[(size_t)f * P + p] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void nin_pad_centroids_kernel(const double *__restrict__ src, int64_t n, double *__restrict__ dst) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        dst[4 * e + 0] = src[3 * e + 0]; dst[4 * e + 1] = src[3 * e + 1]; dst[4 * e + 2] = src[3 * e + 2]; dst[4 * e + 3] = 0.0;
    }
}

int grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

// rows [p_begin, p_end) (p_end < 0: all; p_begin a multiple of 64)
int launch_row_nnz(const GridView &g, const double *data, int32_t *row_nnz, hipStream_t stream, int32_t p_begin, int32_t p_end) {
    if (p_end < 0) p_end = g.n_points;
    if (p_end <= p_begin) return 0;
    hipLaunchKernelGGL(nin_row_nnz_kernel, dim3(grid_for(p_end - p_begin)), dim3(256), 0, stream, g, data, row_nnz, p_begin / 64,
                       (int32_t)(((int64_t)p_end + 63) / 64), p_end);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_compact(const GridView &g, const double *data, const int32_t *new_ptr, int32_t *indices,
                   double *vals, hipStream_t stream, int32_t p_begin, int32_t p_end) {
    if (p_end < 0) p_end = g.n_points;
    if (p_end <= p_begin) return 0;
    hipLaunchKernelGGL(nin_compact_kernel, dim3(grid_for(p_end - p_begin)), dim3(256), 0, stream, g, data, new_ptr, indices, vals,
                       p_begin / 64, (int32_t)(((int64_t)p_end + 63) / 64));
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_pad_centroids(const double *src, int64_t n_elems, double *dst, hipStream_t stream) {
    if (n_elems <= 0) return 0;
    hipLaunchKernelGGL(nin_pad_centroids_kernel, dim3(grid_for(n_elems)), dim3(256), 0, stream, src, n_elems, dst);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_apply_list(const GridView &g, const double *data, const double *u, int32_t k, double *values, const int32_t *list,
                      int32_t count, hipStream_t stream) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(nin_apply_list_kernel, dim3(grid_for(count)), dim3(256), 0, stream, g, data, u, k, values, list, count);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// LDS entries per wavefront and nodes per tile for rows of at most `mx_row` entries, `nnz` in all: 64 rows in steps of 64, at most
// kApplyCap; past that kApplyCapLong and as many nodes a tile as fit it at 1.25 x the mean row length
static int32_t apply_cap(int32_t mx_row, int64_t nnz, int32_t n_points, int32_t *tn) {
    int64_t c = 64 * (int64_t)(mx_row > 0 ? mx_row : 1);
    *tn = 64;
    if (c <= kApplyCap) return (int32_t)c;
    const double mean_row = nnz > 0 && n_points > 0 ? (double)nnz / (double)n_points : (double)mx_row;
    while (*tn > 16 && *tn * mean_row * 1.25 > (double)kApplyCapLong) *tn >>= 1;
    return kApplyCapLong;
}
static int apply_grid(const GridView &g, int32_t tn) {
    const int64_t blocks = ((int64_t)(g.n_points + tn - 1) / tn + 3) / 4;
    return (int)(blocks < 1 ? 1 : blocks > 256 * 32 ? 256 * 32 : blocks);
}
int launch_apply(const GridView &g, const double *data, const double *u, double *values, int32_t mx_row, int64_t nnz, hipStream_t stream) {
    int32_t tn;
    const int32_t cap = apply_cap(mx_row, nnz, g.n_points, &tn);
    if (allow_dynamic_lds<nin_apply_kernel<1>>((size_t)cap * 4 * 12)) return -3;
    hipLaunchKernelGGL(nin_apply_kernel<1>, dim3(apply_grid(g, tn)), dim3(256), (size_t)cap * 4 * 12, stream, g, data, u, 0, 1, values, cap, tn);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// k fields: four per pass over the weights (the rows of a wavefront are staged once per pass)
int launch_apply_fields(const GridView &g, const double *data, const double *u, int32_t k, double *values, int32_t mx_row, int64_t nnz,
                        hipStream_t stream) {
    int32_t tn;
    const int32_t cap = apply_cap(mx_row, nnz, g.n_points, &tn);
    const size_t lds = (size_t)cap * 4 * 12;
    const dim3 grid(apply_grid(g, tn));
    if (allow_dynamic_lds<nin_apply_kernel<1>>(lds) || allow_dynamic_lds<nin_apply_kernel<2>>(lds) || allow_dynamic_lds<nin_apply_kernel<4>>(lds)) return -3;
    for (int32_t k0 = 0; k0 < k; k0 += 4) {
        if (k - k0 >= 3) hipLaunchKernelGGL(nin_apply_kernel<4>, grid, dim3(256), lds, stream, g, data, u, k0, k, values, cap, tn);
        else if (k - k0 == 2) hipLaunchKernelGGL(nin_apply_kernel<2>, grid, dim3(256), lds, stream, g, data, u, k0, k, values, cap, tn);
        else hipLaunchKernelGGL(nin_apply_kernel<1>, grid, dim3(256), lds, stream, g, data, u, k0, k, values, cap, tn);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace nin
