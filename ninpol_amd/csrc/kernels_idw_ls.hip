// kernels_idw_ls.hip -- IDW and LS weights, gfx950.
//
// Both methods are a gather of the <= MX_ELEMENTS_PER_POINT centroids around a node and ~100 flops:
// HBM-bound (190 algorithmic bytes per node on structured hexahedra).  One lane owns one node
// and walks its esup row in order, so the arithmetic is the reference's, operation for operation; what
// the kernel is about is how the bytes move:
//   * a wavefront owns 64 consecutive nodes (32 or 16 where the rows are long: unstructured tetrahedra have ~25 cells around a
//     node, and a tile's run must fit the wave's LDS -- the lane-by-lane path below it writes 8 bytes a lane at a row's stride:
//     1.1 TB/s on a Delaunay mesh against 4.3 on hexahedra), whose esup rows are ONE contiguous run of the CSR arrays.
//     The run of cell ids is copied HBM -> LDS cooperatively (lane l takes entries l, l + 64, ...:
//     whole 256-byte lines), each lane then reads its own row from LDS;
//   * the centroids are gathered per lane (irregular by nature; neighbouring nodes share most of their
//     cells, so L2 serves them);
//   * the weights go lane -> LDS -> HBM the same cooperative way, so the csr_data stream is written as
//     full lines.  (The first version stored 8 bytes per lane with a 64-byte lane stride: rocprofv3
//     WRITE_SIZE was 4.7 GB for 0.73 GB of output, profiles/r01.)
//
// Compiled with -ffp-contract=off and written in the reference's operation order so that results are
// the reference's bit for bit (the reference is built without FMA): the D == 0.0 branch of LS
// (ls.pyx:88) and the 0/0 = NaN rows LS produces for one-sided nodes land on the same nodes.
#include <hip/hip_runtime.h>

#include "device_grid.hpp"
#include "launch.hpp"

namespace nin {

namespace {

constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a cell's centroid: from the padded [E][4] copy when the grid carries one (two 16-byte loads, one 64-byte line), else
// from the packed [E][3] array (three 8-byte loads, 37 % of the records straddle two lines)
__device__ __forceinline__ void load_centroid(const GridView &g, size_t s, double (&c)[3]) {
    if (g.centroids4) {
        const double2 xy = *reinterpret_cast<const double2 *>(g.centroids4 + 4 * s);
        c[0] = xy.x; c[1] = xy.y;
        c[2] = g.centroids4[4 * s + 2];
    } else {
        c[0] = g.centroids[3 * s + 0]; c[1] = g.centroids[3 * s + 1]; c[2] = g.centroids[3 * s + 2];
    }
}

// idw.pyx:35-84 for one node.  cells / w: the node's row (LDS).  `machine_epsilon` is the C float
// (float)1e-15 compared against the SQUARED distance (idw.pyx:53,67-69); distances use the first `dim`
// coordinates (idw.pyx:66).
__device__ __forceinline__ void idw_row(const GridView &g, int32_t p, const int32_t *cells, double *w, int n) {
    const float machine_epsilon = 1e-15f;
    const double x0 = g.coords[3 * (size_t)p + 0], x1 = g.coords[3 * (size_t)p + 1], x2 = g.coords[3 * (size_t)p + 2];
    double total = 0.0;
    int n_source = 0, zero_at = -1;
    // the centroids of 8 neighbours are requested before any of them is used (one round trip per chunk
    // instead of one per neighbour); the arithmetic below is still the reference's sequential loop
    for (int j0 = 0; j0 < n && zero_at < 0; j0 += 8) {
        double c[8][3];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            load_centroid(g, (size_t)cells[j0 + u < n ? j0 + u : n - 1], c[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u;
            if (j < n && zero_at < 0) {
                double d0 = x0 - c[u][0];
                double dist = 0.0 + d0 * d0;
                if (g.dim > 1) { double d1 = x1 - c[u][1]; dist = dist + d1 * d1; }
                if (g.dim > 2) { double d2 = x2 - c[u][2]; dist = dist + d2 * d2; }
                if (dist <= (double)machine_epsilon) {
                    zero_at = j;
                } else {
                    dist = sqrt(dist);
                    const double inv = 1 / dist;
                    w[j] = inv;
                    total += inv;
                    n_source += 1;
                }
            }
        }
    }
    if (zero_at >= 0) {  // node sits on a centroid: row = e_j (idw.pyx:69-74)
        for (int j = 0; j < n; ++j) w[j] = (j == zero_at) ? 1.0 : 0.0;
    } else {
        for (int k = 0; k < n_source; ++k) w[k] = w[k] / total;
    }
}

// ls.pyx:33-135 for one node.  Always three coordinates.
__device__ __forceinline__ void ls_row(const GridView &g, int32_t p, const int32_t *cells, double *w, int n) {
    const double x0 = g.coords[3 * (size_t)p + 0], x1 = g.coords[3 * (size_t)p + 1], x2 = g.coords[3 * (size_t)p + 2];
    double Ix = 0, Iy = 0, Iz = 0, Ixx = 0, Ixy = 0, Ixz = 0, Iyy = 0, Iyz = 0, Izz = 0;
    double v8[8][3];   // (x_K - x_v) of the first 8 neighbours, reused by the weight loop below
    for (int j0 = 0; j0 < n; j0 += 8) {
        double c[8][3];
#pragma unroll
        for (int u = 0; u < 8; ++u) {   // 8 centroid gathers in flight, then the reference's sequential sums
            load_centroid(g, (size_t)cells[j0 + u < n ? j0 + u : n - 1], c[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (j0 + u < n) {
                const double vx = c[u][0] - x0, vy = c[u][1] - x1, vz = c[u][2] - x2;
                if (j0 == 0) { v8[u][0] = vx; v8[u][1] = vy; v8[u][2] = vz; }
                Ix = Ix + vx; Iy = Iy + vy; Iz = Iz + vz;
                Ixx = Ixx + vx * vx; Ixy = Ixy + vx * vy; Ixz = Ixz + vx * vz;
                Iyy = Iyy + vy * vy; Iyz = Iyz + vy * vz; Izz = Izz + vz * vz;
            }
        }
    }
    const bool planar = (Iz == 0.0 && Izz == 0.0 && Ixz == 0.0 && Iyz == 0.0);
    if (planar) Izz = 1.0;
    const double D = (Ixx * (Iyy * Izz - Iyz * Iyz) + Ixy * (Iyz * Ixz - Ixy * Izz) + Ixz * (Ixy * Iyz - Iyy * Ixz));
    if (D == 0.0) {  // IDW fallback (ls.pyx:88-102)
        double total = 0.0;
        for (int j = 0; j < n; ++j) {
            const size_t s = (size_t)cells[j];
            const double vx = g.centroids[3 * s + 0] - x0, vy = g.centroids[3 * s + 1] - x1, vz = g.centroids[3 * s + 2] - x2;
            const double inv = 1.0 / sqrt(vx * vx + vy * vy + vz * vz);
            w[j] = inv;
            total = total + inv;
        }
        for (int j = 0; j < n; ++j) w[j] = w[j] / total;
        return;
    }
    // ls.pyx:104-105 re-tests the planar condition AFTER Izz was set to 1.0, so it never fires.
    const double lx = (Ix * (Iyz * Iyz - Iyy * Izz) + Iy * (Ixy * Izz - Iyz * Ixz) + Iz * (Iyy * Ixz - Ixy * Iyz)) / D;
    const double ly = (Ix * (Ixy * Izz - Iyz * Ixz) + Iy * (Ixz * Ixz - Ixx * Izz) + Iz * (Ixx * Iyz - Ixy * Ixz)) / D;
    const double lz = (Ix * (Iyy * Ixz - Ixy * Iyz) + Iy * (Ixx * Iyz - Ixy * Ixz) + Iz * (Ixy * Ixy - Ixx * Iyy)) / D;
    const double denom = (double)n + lx * Ix + ly * Iy + lz * Iz;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (u < n) {
            double wi = (1. + lx * v8[u][0] + ly * v8[u][1] + lz * v8[u][2]);
            w[u] = wi / denom;
        }
    }
    for (int j = 8; j < n; ++j) {
        const size_t s = (size_t)cells[j];
        const double vx = g.centroids[3 * s + 0] - x0, vy = g.centroids[3 * s + 1] - x1, vz = g.centroids[3 * s + 2] - x2;
        double wi = (1. + lx * vx + ly * vy + lz * vz);
        w[j] = wi / denom;
    }
}

// METHOD 0: IDW, 1: LS.  Full node range [0, n): wave-cooperative staging as described in the header.
// cap = entries of LDS each wave owns; a 64-node run longer than that is handled straight from HBM.
template <int METHOD>
__global__ __launch_bounds__(64 * kWavesPerBlock) void nin_rows_kernel(GridView g, int32_t n, int32_t cap,
                                                                      double *__restrict__ out, double *__restrict__ nws,
                                                                      int32_t tile_begin, int32_t tile_end, int32_t tn) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *wbuf = smem + (size_t)wave * cap + (size_t)wave * ((cap + 1) >> 1);   // [cap] weights
    int32_t *cbuf = reinterpret_cast<int32_t *>(wbuf + cap);                        // [cap] cell ids
    // tiles [tile_begin, tile_end) of tn = 64 / 32 / 16 nodes (all of them, or one chunk of the pipelined interpolate())
    for (int32_t tile = tile_begin + blockIdx.x * kWavesPerBlock + wave; tile < tile_end; tile += gridDim.x * kWavesPerBlock) {
        const int32_t p0 = tile * tn, p = p0 + lane;
        const int32_t pe = p0 + tn < n ? p0 + tn : n;
        const int32_t run_b = g.esup_ptr[p0], run_e = g.esup_ptr[pe];   // wave-uniform loads
        const int32_t len = run_e - run_b;
        const bool staged = len <= cap;
        const bool live = lane < tn && p < n;
        int32_t b = 0, e = 0;
        bool skip = true;
        if (live) {
            b = g.esup_ptr[p];
            e = g.esup_ptr[p + 1];
            const uint8_t fl = g.flags[p];
            skip = (fl & 1) && !(fl & 2);   // Dirichlet boundary node: skipped (idw.pyx:62-63, ls.pyx:58-59)
            nws[p] = 0.0;
        }
        if (staged) {
            for (int32_t i = lane; i < len; i += 64) cbuf[i] = g.esup[run_b + i];
            wave_lds_sync();
            if (live) {
                double *w = wbuf + (b - run_b);
                if (skip) {
                    for (int32_t j = 0; j < e - b; ++j) w[j] = 0.0;
                } else if (METHOD == 0) {
                    idw_row(g, p, cbuf + (b - run_b), w, e - b);
                } else {
                    ls_row(g, p, cbuf + (b - run_b), w, e - b);
                }
            }
            wave_lds_sync();
            for (int32_t i = lane; i < len; i += 64) out[run_b + i] = wbuf[i];
            wave_lds_sync();
        } else if (live) {   // very long rows: straight from / to HBM
            double *w = out + b;
            if (skip) {
                for (int32_t j = 0; j < e - b; ++j) w[j] = 0.0;
            } else if (METHOD == 0) {
                idw_row(g, p, g.esup + b, w, e - b);
            } else {
                ls_row(g, p, g.esup + b, w, e - b);
            }
        }
    }
}

// Explicit target list (rare path): one lane per target, rows straight from / to HBM.
template <int METHOD>
__global__ __launch_bounds__(256) void nin_rows_targets_kernel(GridView g, const int32_t *__restrict__ targets,
                                                               int32_t n_targets, double *__restrict__ out,
                                                               double *__restrict__ nws) {
    for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_targets; t += gridDim.x * blockDim.x) {
        const int32_t p = targets[t];
        const int32_t b = g.esup_ptr[p], e = g.esup_ptr[p + 1];
        const uint8_t fl = g.flags[p];
        double *w = out + b;
        nws[p] = 0.0;
        if ((fl & 1) && !(fl & 2)) {
            for (int32_t j = 0; j < e - b; ++j) w[j] = 0.0;
        } else if (METHOD == 0) {
            idw_row(g, p, g.esup + b, w, e - b);
        } else {
            ls_row(g, p, g.esup + b, w, e - b);
        }
    }
}

template <int METHOD>
int launch_rows(const GridView &g, const int32_t *targets, int32_t n_targets, int32_t mx_row, int64_t nnz, double *out, double *nws,
                hipStream_t stream, int32_t p_begin = 0, int32_t p_end = -1) {
    if (n_targets <= 0) return 0;
    if (targets) {
        int64_t blocks = ((int64_t)n_targets + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL((nin_rows_targets_kernel<METHOD>), dim3((unsigned)blocks), dim3(256), 0, stream, g, targets, n_targets, out, nws);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    // LDS per wave: room for 64 rows of the longest length if that is at most 12 KiB of weights + ids (structured meshes); else 18 KiB
    // and as many nodes a tile -- 64, 32 or 16 -- as fit it at 1.25 x the mesh's mean row length (a tile that is longer still takes
    // the lane-by-lane path)
    int64_t cap = (int64_t)64 * (mx_row > 0 ? mx_row : 8);
    int32_t tn = 64;
    if (cap > 1024) {
        cap = 1536;
        const double mean_row = nnz > 0 && n_targets > 0 ? (double)nnz / (double)n_targets : (double)mx_row;
        while (tn > 16 && tn * mean_row * 1.25 > (double)cap) tn >>= 1;
    }
    cap = (cap + 1) & ~(int64_t)1;
    const size_t dyn = (size_t)kWavesPerBlock * (cap * 8 + ((cap + 1) / 2) * 8);
    if (p_end < 0 || p_end > n_targets) p_end = n_targets;
    const int32_t tile_begin = p_begin / tn, tile_end = (int32_t)(((int64_t)p_end + tn - 1) / tn);   // (p_begin: a multiple of 64)
    const int64_t tiles = tile_end - tile_begin;
    if (tiles <= 0) return 0;
    int64_t blocks = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t cap_blocks = 256 * 8;
    if (blocks > cap_blocks) blocks = cap_blocks;
    if (allow_dynamic_lds<nin_rows_kernel<METHOD>>(dyn)) return -3;
    hipLaunchKernelGGL((nin_rows_kernel<METHOD>), dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), dyn, stream, g, n_targets,
                       (int32_t)cap, out, nws, tile_begin, tile_end, tn);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

int launch_idw(const GridView &g, const int32_t *targets, int32_t n_targets, int32_t mx_row, int64_t nnz, double *out, double *nws,
               hipStream_t stream) {
    return launch_rows<0>(g, targets, n_targets, mx_row, nnz, out, nws, stream);
}

int launch_ls(const GridView &g, const int32_t *targets, int32_t n_targets, int32_t mx_row, int64_t nnz, double *out, double *nws,
              hipStream_t stream) {
    return launch_rows<1>(g, targets, n_targets, mx_row, nnz, out, nws, stream);
}

int launch_rows_range(const GridView &g, int method_ls, int32_t n_points, int32_t p_begin, int32_t p_end, int32_t mx_row, int64_t nnz,
                      double *out, double *nws, hipStream_t stream) {
    return method_ls ? launch_rows<1>(g, nullptr, n_points, mx_row, nnz, out, nws, stream, p_begin, p_end)
                     : launch_rows<0>(g, nullptr, n_points, mx_row, nnz, out, nws, stream, p_begin, p_end);
}

}  // namespace nin
