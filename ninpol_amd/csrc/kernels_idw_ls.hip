// kernels_idw_ls.hip -- IDW and LS weights, gfx950.
//
// Both methods are a gather of the <= MX_ELEMENTS_PER_POINT centroids around a node and ~100 flops:
// HBM-bound (190 algorithmic bytes per node on structured hexahedra), no reuse worth
// staging beyond what L2 gives (neighbouring nodes share 4 of their 8 cells).  One lane owns one
// node: its esup row is a contiguous 32-byte run, the wave's rows are contiguous in HBM, and the
// output row csr_data[esup_ptr[p] ..] is written contiguously by the same lane.
//
// Compiled with -ffp-contract=off and written in the reference's operation order so that results
// are the reference's bit for bit (the reference is built without FMA): the D == 0.0 branch of LS
// (ls.pyx:88) and the 0/0 = NaN rows LS produces for one-sided nodes must land on the same nodes.
#include <hip/hip_runtime.h>

#include "device_grid.hpp"
#include "launch.hpp"

namespace nin {

namespace {

// idw.pyx:35-84.  `machine_epsilon` is the C float (float)1e-15 compared against the SQUARED distance
// (idw.pyx:53,67-69); distances use the first `dim` coordinates (idw.pyx:66).
__global__ __launch_bounds__(256) void nin_idw_kernel(GridView g, const int32_t *__restrict__ targets,
                                                      int32_t n_targets, double *__restrict__ out,
                                                      double *__restrict__ nws) {
    const float machine_epsilon = 1e-15f;
    for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_targets; t += gridDim.x * blockDim.x) {
        const int32_t p = targets ? targets[t] : t;
        const int32_t b = g.esup_ptr[p], e = g.esup_ptr[p + 1];
        const uint8_t fl = g.flags[p];
        double *w = out + b;
        nws[p] = 0.0;
        if ((fl & 1) && !(fl & 2)) {  // Dirichlet boundary node: skipped (idw.pyx:62-63)
            for (int32_t q = b; q < e; ++q) w[q - b] = 0.0;
            continue;
        }
        const double x0 = g.coords[p * 3 + 0], x1 = g.coords[p * 3 + 1], x2 = g.coords[p * 3 + 2];
        double total = 0.0;
        int32_t n_source = 0, zero_at = -1;
        for (int32_t q = b; q < e; ++q) {
            const int32_t s = g.esup[q];
            double d0 = x0 - g.centroids[s * 3 + 0];
            double dist = 0.0 + d0 * d0;
            if (g.dim > 1) { double d1 = x1 - g.centroids[s * 3 + 1]; dist = dist + d1 * d1; }
            if (g.dim > 2) { double d2 = x2 - g.centroids[s * 3 + 2]; dist = dist + d2 * d2; }
            if (dist <= (double)machine_epsilon) { zero_at = q - b; break; }
            dist = sqrt(dist);
            const double inv = 1 / dist;
            w[q - b] = inv;
            total += inv;
            n_source += 1;
        }
        if (zero_at >= 0) {  // node sits on a centroid: row = e_j (idw.pyx:69-74)
            for (int32_t q = b; q < e; ++q) w[q - b] = (q - b == zero_at) ? 1.0 : 0.0;
        } else {
            for (int32_t k = 0; k < n_source; ++k) w[k] = w[k] / total;
        }
    }
}

// ls.pyx:33-135.  Always three coordinates.
__global__ __launch_bounds__(256) void nin_ls_kernel(GridView g, const int32_t *__restrict__ targets,
                                                     int32_t n_targets, double *__restrict__ out,
                                                     double *__restrict__ nws) {
    for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_targets; t += gridDim.x * blockDim.x) {
        const int32_t p = targets ? targets[t] : t;
        const int32_t b = g.esup_ptr[p], e = g.esup_ptr[p + 1];
        const uint8_t fl = g.flags[p];
        double *w = out + b;
        nws[p] = 0.0;
        if ((fl & 1) && !(fl & 2)) {
            for (int32_t q = b; q < e; ++q) w[q - b] = 0.0;
            continue;
        }
        const double x0 = g.coords[p * 3 + 0], x1 = g.coords[p * 3 + 1], x2 = g.coords[p * 3 + 2];
        double Ix = 0, Iy = 0, Iz = 0, Ixx = 0, Ixy = 0, Ixz = 0, Iyy = 0, Iyz = 0, Izz = 0;
        for (int32_t q = b; q < e; ++q) {
            const int32_t s = g.esup[q];
            const double vx = g.centroids[s * 3 + 0] - x0, vy = g.centroids[s * 3 + 1] - x1,
                         vz = g.centroids[s * 3 + 2] - x2;
            Ix = Ix + vx; Iy = Iy + vy; Iz = Iz + vz;
            Ixx = Ixx + vx * vx; Ixy = Ixy + vx * vy; Ixz = Ixz + vx * vz;
            Iyy = Iyy + vy * vy; Iyz = Iyz + vy * vz; Izz = Izz + vz * vz;
        }
        const bool planar = (Iz == 0.0 && Izz == 0.0 && Ixz == 0.0 && Iyz == 0.0);
        if (planar) Izz = 1.0;
        const double D = (Ixx * (Iyy * Izz - Iyz * Iyz) + Ixy * (Iyz * Ixz - Ixy * Izz) + Ixz * (Ixy * Iyz - Iyy * Ixz));
        if (D == 0.0) {  // IDW fallback (ls.pyx:88-102)
            double total = 0.0;
            for (int32_t q = b; q < e; ++q) {
                const int32_t s = g.esup[q];
                const double vx = g.centroids[s * 3 + 0] - x0, vy = g.centroids[s * 3 + 1] - x1,
                             vz = g.centroids[s * 3 + 2] - x2;
                const double inv = 1.0 / sqrt(vx * vx + vy * vy + vz * vz);
                w[q - b] = inv;
                total = total + inv;
            }
            for (int32_t q = b; q < e; ++q) w[q - b] = w[q - b] / total;
            continue;
        }
        // ls.pyx:104-105 re-tests the planar condition AFTER Izz was set to 1.0, so it never fires.
        const double lx = (Ix * (Iyz * Iyz - Iyy * Izz) + Iy * (Ixy * Izz - Iyz * Ixz) + Iz * (Iyy * Ixz - Ixy * Iyz)) / D;
        const double ly = (Ix * (Ixy * Izz - Iyz * Ixz) + Iy * (Ixz * Ixz - Ixx * Izz) + Iz * (Ixx * Iyz - Ixy * Ixz)) / D;
        const double lz = (Ix * (Iyy * Ixz - Ixy * Iyz) + Iy * (Ixx * Iyz - Ixy * Ixz) + Iz * (Ixy * Ixy - Ixx * Iyy)) / D;
        const double denom = (double)(e - b) + lx * Ix + ly * Iy + lz * Iz;
        for (int32_t q = b; q < e; ++q) {
            const int32_t s = g.esup[q];
            const double vx = g.centroids[s * 3 + 0] - x0, vy = g.centroids[s * 3 + 1] - x1,
                         vz = g.centroids[s * 3 + 2] - x2;
            double wi = (1. + lx * vx + ly * vy + lz * vz);
            w[q - b] = wi / denom;
        }
    }
}

int grid_for(int64_t n, int block) {
    int64_t blocks = (n + block - 1) / block;
    const int64_t cap = 256 * 8;  // CUs x blocks/CU; grid-stride the rest
    return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

}  // namespace

int launch_idw(const GridView &g, const int32_t *targets, int32_t n_targets, double *out, double *nws,
               hipStream_t stream) {
    hipLaunchKernelGGL(nin_idw_kernel, dim3(grid_for(n_targets, 256)), dim3(256), 0, stream, g, targets, n_targets, out, nws);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_ls(const GridView &g, const int32_t *targets, int32_t n_targets, double *out, double *nws,
              hipStream_t stream) {
    hipLaunchKernelGGL(nin_ls_kernel, dim3(grid_for(n_targets, 256)), dim3(256), 0, stream, g, targets, n_targets, out, nws);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace nin
