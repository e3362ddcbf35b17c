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

__global__ __launch_bounds__(256) void nin_row_nnz_kernel(GridView g, const double *__restrict__ data,
                                                          int32_t *__restrict__ row_nnz) {
    for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < g.n_points; p += gridDim.x * blockDim.x) {
        int32_t c = 0;
        for (int32_t q = g.esup_ptr[p]; q < g.esup_ptr[p + 1]; ++q) c += (data[q] != 0.0);
        row_nnz[p] = c;
    }
}

__global__ __launch_bounds__(256) void nin_compact_kernel(GridView g, const double *__restrict__ data,
                                                          const int32_t *__restrict__ new_ptr,
                                                          int32_t *__restrict__ indices, double *__restrict__ vals) {
    for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < g.n_points; p += gridDim.x * blockDim.x) {
        int32_t at = new_ptr[p];
        for (int32_t q = g.esup_ptr[p]; q < g.esup_ptr[p + 1]; ++q) {
            const double d = data[q];
            if (d != 0.0) {
                indices[at] = g.esup[q];
                vals[at] = d;
                ++at;
            }
        }
    }
}

// values[p] = sum_j data[esup_ptr[p] + j] * u[esup[esup_ptr[p] + j]]: what every caller of interpolate() does next
// (`weights.dot(u)`, tests/utils/analytical.py:236), without the matrix leaving the device.  One lane per node;
// rows are contiguous so a wavefront streams a contiguous run of data / esup; u is gathered (L2).
__global__ __launch_bounds__(256) void nin_apply_kernel(GridView g, const double *__restrict__ data,
                                                        const double *__restrict__ u, double *__restrict__ values) {
    for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < g.n_points; p += gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int32_t q = g.esup_ptr[p]; q < g.esup_ptr[p + 1]; ++q) acc += data[q] * u[g.esup[q]];
        values[p] = acc;
    }
}

// The same for k cell fields at once (u: [k][n_elems], values: [k][n_points]): the weights of a row are read once per
// group of four fields instead of once per field.
__global__ __launch_bounds__(256) void nin_apply_fields_kernel(GridView g, const double *__restrict__ data,
                                                               const double *__restrict__ u, int32_t k,
                                                               double *__restrict__ values) {
    const size_t E = (size_t)g.n_elems, P = (size_t)g.n_points;
    for (int32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < g.n_points; p += gridDim.x * blockDim.x) {
        const int32_t b = g.esup_ptr[p], e = g.esup_ptr[p + 1];
        for (int32_t f0 = 0; f0 < k; f0 += 4) {
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            for (int32_t q = b; q < e; ++q) {
                const double w = data[q];
                const size_t c = (size_t)g.esup[q];
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    if (f0 + f < k) acc[f] += w * u[(size_t)(f0 + f) * E + c];
            }
#pragma unroll
            for (int f = 0; f < 4; ++f)
                if (f0 + f < k) values[(size_t)(f0 + f) * P + p] = acc[f];
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

int launch_row_nnz(const GridView &g, const double *data, int32_t *row_nnz, hipStream_t stream) {
    hipLaunchKernelGGL(nin_row_nnz_kernel, dim3(grid_for(g.n_points)), dim3(256), 0, stream, g, data, row_nnz);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_compact(const GridView &g, const double *data, const int32_t *new_ptr, int32_t *indices,
                   double *vals, hipStream_t stream) {
    hipLaunchKernelGGL(nin_compact_kernel, dim3(grid_for(g.n_points)), dim3(256), 0, stream, g, data, new_ptr, indices, vals);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_pad_centroids(const double *src, int64_t n_elems, double *dst, hipStream_t stream) {
    if (n_elems <= 0) return 0;
    hipLaunchKernelGGL(nin_pad_centroids_kernel, dim3(grid_for(n_elems)), dim3(256), 0, stream, src, n_elems, dst);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_apply(const GridView &g, const double *data, const double *u, double *values, hipStream_t stream) {
    hipLaunchKernelGGL(nin_apply_kernel, dim3(grid_for(g.n_points)), dim3(256), 0, stream, g, data, u, values);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_apply_fields(const GridView &g, const double *data, const double *u, int32_t k, double *values,
                        hipStream_t stream) {
    hipLaunchKernelGGL(nin_apply_fields_kernel, dim3(grid_for(g.n_points)), dim3(256), 0, stream, g, data, u, k, values);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace nin
