// grid_device.hip -- the Grid connectivity + geometry built ON the device.
//
// Same arrays, bit for bit, as grid_host.cpp (and therefore as the reference's Grid.build() +
// calculate_centroids() + calculate_normal_faces(), grid.pyx:142-231, 669-809), from the same data-parallel
// formulations whose output order is fixed by construction:
//   esup   count (atomics) -> exclusive scan -> atomic fill -> sort each short row ascending
//   esuel  one thread per (element, local face): the reference's search through the face point with the fewest
//          cells (grid.pyx:479-525)
//   faces  id = rank of (creator element, local face) among owned pairs, creator = lower element id or the only
//          one (what the first-sight sweep of grid.pyx:315-334 produces) -> flag, scan, fill, mirror
//   fsup   per point, the faces OWNED by the cells around it that contain the point, cells ascending, local
//          faces ascending: every face around the point is listed exactly once and already in ascending id order
//   esuf   (creator, neighbour) read off the owner table; boundary flags from it
//   geometry  one thread per cell / face, the reference's operation order; float32 normals (grid.pyx:732-767)
// Compiled with -ffp-contract=off (build.py): the reference is built without FMA; HIP's float divide and sqrt
// are correctly rounded by default.  The results stay in HBM for the weight kernels (no second upload) and are
// copied back into the HostGrid so that every host-side consumer sees what the host builder would have made.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "device_grid.hpp"
#include "grid_host.hpp"
#include "hex8_desc.hpp"
#include "quad4_desc.hpp"
#include "mfw_desc.hpp"
#include "mfg_desc.hpp"
#include "mfx_desc.hpp"
#include "launch.hpp"

namespace nin {

namespace {

struct Topo {   // the element tables (utils/point_ordering.yaml via the Grid ctor arguments), by value to kernels
    int8_t npoel[kNumElementTypes], nfael[kNumElementTypes];
    int8_t lnofa[kNumElementTypes][kMaxFacesPerElement];
    int8_t lpofa[kNumElementTypes][kMaxFacesPerElement][kMaxPointsPerFace];
};

constexpr int TPB = 256;
inline unsigned blocks_for(int64_t n) { return (unsigned)((n + TPB - 1) / TPB); }

__global__ void k_ingest_elems(int64_t E, int64_t P, Topo t, const int64_t *__restrict__ conn,
                               const int64_t *__restrict__ types, int32_t *__restrict__ inpoel,
                               int8_t *__restrict__ etype, int *__restrict__ bad) {
    const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= E) return;
    int64_t ty = types[e];
    bool b = false;
    if (ty < 0 || ty >= kNumElementTypes) { b = true; ty = 0; }
    etype[e] = (int8_t)ty;
    for (int j = 0; j < kMaxPointsPerElement; ++j) {
        const int64_t p = conn[e * kMaxPointsPerElement + j];
        if (j < t.npoel[ty] && (p < 0 || p >= P)) b = true;
        inpoel[e * kMaxPointsPerElement + j] = (int32_t)p;
    }
    if (b) atomicOr(bad, 1);
}

__global__ void k_ingest_coords(int64_t P, int cd, const double *__restrict__ xyz, double *__restrict__ coords) {
    const int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (p >= P) return;
    for (int k = 0; k < 3; ++k) coords[p * 3 + k] = k < cd ? xyz[p * cd + k] : 0.0;
}

__global__ void k_esup_count(int64_t E, Topo t, const int32_t *__restrict__ inpoel, const int8_t *__restrict__ etype,
                             int32_t *__restrict__ cnt) {
    const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= E) return;
    const int n = t.npoel[etype[e]];
    for (int j = 0; j < n; ++j) atomicAdd(&cnt[inpoel[e * 8 + j]], 1);
}

__global__ void k_esup_fill(int64_t E, Topo t, const int32_t *__restrict__ inpoel, const int8_t *__restrict__ etype,
                            const int32_t *__restrict__ ptr, int32_t *__restrict__ cur, int32_t *__restrict__ esup) {
    const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= E) return;
    const int n = t.npoel[etype[e]];
    for (int j = 0; j < n; ++j) {
        const int32_t p = inpoel[e * 8 + j];
        esup[ptr[p] + atomicAdd(&cur[p], 1)] = (int32_t)e;
    }
}

__global__ void k_sort_rows(int64_t P, const int32_t *__restrict__ ptr, int32_t *__restrict__ data, int32_t *__restrict__ mx) {
    const int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (p >= P) return;
    const int32_t b = ptr[p], n = ptr[p + 1] - b;
    for (int i = 1; i < n; ++i) {   // rows are short (8 on hex meshes, ~24 on tets): insertion sort
        const int32_t v = data[b + i];
        int j = i - 1;
        while (j >= 0 && data[b + j] > v) { data[b + j + 1] = data[b + j]; --j; }
        data[b + j + 1] = v;
    }
    atomicMax(mx, n);
}

__device__ inline bool elem_has_point(const int32_t *el, int n, int32_t p) {
    for (int i = 0; i < n; ++i)
        if (el[i] == p) return true;
    return false;
}

__global__ void k_esuel(int64_t E, Topo t, const int32_t *__restrict__ inpoel, const int8_t *__restrict__ etype,
                        const int32_t *__restrict__ esup_ptr, const int32_t *__restrict__ esup,
                        int32_t *__restrict__ esuel) {
    const int64_t id = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (id >= E * kMaxFacesPerElement) return;
    const int64_t ie = id / kMaxFacesPerElement;
    const int j = (int)(id % kMaxFacesPerElement);
    const int it = etype[ie];
    if (j >= t.nfael[it]) { esuel[id] = -1; return; }
    const int32_t *el = inpoel + ie * 8;
    const int nf = t.lnofa[it][j];
    int32_t fp[kMaxPointsPerFace];
    for (int k = 0; k < kMaxPointsPerFace; ++k) fp[k] = k < nf ? el[t.lpofa[it][j][k]] : -1;
    int32_t point = fp[0];
    int32_t nmin = esup_ptr[point + 1] - esup_ptr[point];
    for (int k = 0; k < nf; ++k) {   // first minimum (grid.pyx:479-488)
        const int32_t n = esup_ptr[fp[k] + 1] - esup_ptr[fp[k]];
        if (n < nmin) { point = fp[k]; nmin = n; }
    }
    int32_t found = -1;
    for (int32_t q = esup_ptr[point]; q < esup_ptr[point + 1] && found < 0; ++q) {
        const int32_t je = esup[q];
        if (je == ie) continue;
        const int jt = etype[je];
        const int32_t *jl = inpoel + (int64_t)je * 8;
        bool all = true;
        for (int k = 0; k < nf && all; ++k) all = elem_has_point(jl, t.npoel[jt], fp[k]);
        if (!all) continue;
        for (int l = 0; l < t.nfael[jt]; ++l) {   // the reference's own face test (:503-512)
            int is_equal = 0;
            for (int m = 0; m < t.lnofa[jt][l]; ++m) {
                const int32_t jp = jl[t.lpofa[jt][l][m]];
                for (int o = 0; o < nf; ++o)
                    if (jp == fp[o]) { ++is_equal; break; }
            }
            if (is_equal == nf) { found = je; break; }
        }
    }
    esuel[id] = found;
}

__global__ void k_own_count(int64_t E, Topo t, const int8_t *__restrict__ etype, const int32_t *__restrict__ esuel,
                            int32_t *__restrict__ own_cnt) {
    const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= E) return;
    int c = 0;
    for (int j = 0; j < t.nfael[etype[e]]; ++j) {
        const int32_t k = esuel[e * 6 + j];
        c += (k == -1 || k > e);
    }
    own_cnt[e] = c;
}

__global__ void k_faces_fill(int64_t E, Topo t, const int32_t *__restrict__ inpoel, const int8_t *__restrict__ etype,
                             const int32_t *__restrict__ esuel, const int32_t *__restrict__ own_start,
                             int32_t *__restrict__ infael, int32_t *__restrict__ inpofa, int32_t *__restrict__ face_cells) {
    const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= E) return;
    const int ty = etype[e];
    int32_t f = own_start[e];
    for (int j = 0; j < kMaxFacesPerElement; ++j) {
        if (j >= t.nfael[ty]) { infael[e * 6 + j] = -1; continue; }
        const int32_t k = esuel[e * 6 + j];
        if (k == -1 || k > e) {
            infael[e * 6 + j] = f;
            for (int q = 0; q < kMaxPointsPerFace; ++q)
                inpofa[(int64_t)f * 4 + q] = q < t.lnofa[ty][j] ? inpoel[e * 8 + t.lpofa[ty][j][q]] : -1;
            face_cells[2 * (int64_t)f] = (int32_t)e;
            face_cells[2 * (int64_t)f + 1] = k;
            ++f;
        } else {
            infael[e * 6 + j] = -1;   // the mirror pass fills it
        }
    }
}

__global__ void k_faces_mirror(int64_t E, Topo t, const int8_t *__restrict__ etype, const int32_t *__restrict__ esuel,
                               int32_t *__restrict__ infael) {
    const int64_t id = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (id >= E * kMaxFacesPerElement) return;
    const int64_t e = id / kMaxFacesPerElement;
    const int j = (int)(id % kMaxFacesPerElement);
    if (j >= t.nfael[etype[e]]) return;
    const int32_t k = esuel[id];
    if (k != -1 && k < e) {   // created by the lower element: its id (first l with esuel[k, l] == e)
        for (int l = 0; l < t.nfael[etype[k]]; ++l)
            if (esuel[(int64_t)k * 6 + l] == e) { infael[id] = infael[(int64_t)k * 6 + l]; break; }
    }
}

// FILL = false: count the faces around each point; FILL = true: write them (ascending by construction)
template <bool FILL>
__global__ void k_fsup(int64_t P, Topo t, const int32_t *__restrict__ inpoel, const int8_t *__restrict__ etype,
                       const int32_t *__restrict__ esup_ptr, const int32_t *__restrict__ esup,
                       const int32_t *__restrict__ esuel, const int32_t *__restrict__ infael,
                       int32_t *__restrict__ cnt, const int32_t *__restrict__ fsup_ptr, int32_t *__restrict__ fsup,
                       int32_t *__restrict__ mx) {
    const int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (p >= P) return;
    int n = 0;
    int32_t at = FILL ? fsup_ptr[p] : 0;
    for (int32_t q = esup_ptr[p]; q < esup_ptr[p + 1]; ++q) {
        const int64_t e = esup[q];
        const int ty = etype[e];
        for (int j = 0; j < t.nfael[ty]; ++j) {
            const int32_t k = esuel[e * 6 + j];
            if (!(k == -1 || k > e)) continue;   // listed from the creator's side only
            bool has = false;
            for (int c = 0; c < t.lnofa[ty][j]; ++c) has |= (inpoel[e * 8 + t.lpofa[ty][j][c]] == p);
            if (has) {
                if (FILL) fsup[at++] = infael[e * 6 + j];
                ++n;
            }
        }
    }
    if (!FILL) {
        cnt[p] = n;
        atomicMax(mx, n);
    }
}

__global__ void k_boundary(int64_t F, const int32_t *__restrict__ face_cells, const int32_t *__restrict__ inpofa,
                           uint8_t *__restrict__ bfaces, uint8_t *__restrict__ bpoints, int32_t *__restrict__ any_internal) {
    const int64_t f = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (f >= F) return;
    const bool b = face_cells[2 * f + 1] == -1;
    bfaces[f] = b ? 1 : 0;
    if (b) {
        for (int k = 0; k < kMaxPointsPerFace && inpofa[f * 4 + k] != -1; ++k) bpoints[inpofa[f * 4 + k]] = 1;
    } else {
        *any_internal = 1;   // benign race: every writer stores 1
    }
}

__global__ void k_centroids(int64_t E, int d, Topo t, const int32_t *__restrict__ inpoel, const int8_t *__restrict__ etype,
                            const double *__restrict__ X, double *__restrict__ cen) {
    const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= E) return;
    const int n = t.npoel[etype[e]];
    double c[3] = {0.0, 0.0, 0.0};
    for (int j = 0; j < n; ++j)   // divide-then-add, vertex order (grid.pyx:699-704)
        for (int k = 0; k < d; ++k) c[k] += X[(int64_t)inpoel[e * 8 + j] * 3 + k] / (double)n;
    for (int k = 0; k < 3; ++k) cen[e * 3 + k] = c[k];
}

__global__ void k_faces_geometry(int64_t F, int d, const int32_t *__restrict__ inpofa, const double *__restrict__ X,
                                 double *__restrict__ fc, float *__restrict__ fn, double *__restrict__ fa) {
    const int64_t f = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (f >= F) return;
    int npofa = 0;
    double c[3] = {0.0, 0.0, 0.0};
    for (int j = 0; j < kMaxPointsPerFace && inpofa[f * 4 + j] != -1; ++j) {
        ++npofa;
        for (int k = 0; k < d; ++k) c[k] += X[(int64_t)inpofa[f * 4 + j] * 3 + k];
    }
    for (int k = 0; k < d; ++k) c[k] /= (double)npofa;
    for (int k = 0; k < 3; ++k) fc[f * 3 + k] = c[k];
    const int64_t p1 = inpofa[f * 4 + 0], p2 = inpofa[f * 4 + 1];
    if (d == 3) {
        // float locals exactly as grid.pyx:732-736; the reference module is C++, so sqrt(float) is sqrtf
        const int64_t p3 = inpofa[f * 4 + 2];
        float v1x = (float)(X[p1 * 3 + 0] - X[p2 * 3 + 0]), v1y = (float)(X[p1 * 3 + 1] - X[p2 * 3 + 1]),
              v1z = (float)(X[p1 * 3 + 2] - X[p2 * 3 + 2]);
        float v2x = (float)(X[p3 * 3 + 0] - X[p2 * 3 + 0]), v2y = (float)(X[p3 * 3 + 1] - X[p2 * 3 + 1]),
              v2z = (float)(X[p3 * 3 + 2] - X[p2 * 3 + 2]);
        float nx = v1y * v2z - v1z * v2y, ny = v1z * v2x - v1x * v2z, nz = v1x * v2y - v1y * v2x;
        const float norm = fabsf(sqrtf(nx * nx + ny * ny + nz * nz));
        fn[f * 3 + 0] = nx / norm;
        fn[f * 3 + 1] = ny / norm;
        fn[f * 3 + 2] = nz / norm;
        if (inpofa[f * 4 + 3] == -1) {
            fa[f] = (double)norm / 2.0;
        } else {
            const int64_t p4 = inpofa[f * 4 + 3];
            v1x = (float)(X[p1 * 3 + 0] - X[p4 * 3 + 0]); v1y = (float)(X[p1 * 3 + 1] - X[p4 * 3 + 1]);
            v1z = (float)(X[p1 * 3 + 2] - X[p4 * 3 + 2]);
            v2x = (float)(X[p3 * 3 + 0] - X[p4 * 3 + 0]); v2y = (float)(X[p3 * 3 + 1] - X[p4 * 3 + 1]);
            v2z = (float)(X[p3 * 3 + 2] - X[p4 * 3 + 2]);
            nx = v1y * v2z - v1z * v2y; ny = v1z * v2x - v1x * v2z; nz = v1x * v2y - v1y * v2x;
            fa[f] = (double)(norm + sqrtf(nx * nx + ny * ny + nz * nz)) / 2.0;
        }
    } else {
        const float v1x = (float)(X[p1 * 3 + 0] - X[p2 * 3 + 0]), v1y = (float)(X[p1 * 3 + 1] - X[p2 * 3 + 1]);
        const float nx = -v1y, ny = v1x;
        const float norm = fabsf(sqrtf(nx * nx + ny * ny));
        fn[f * 3 + 0] = nx / norm;
        fn[f * 3 + 1] = ny / norm;
        fn[f * 3 + 2] = 0.0f;
        fa[f] = (double)norm;
    }
}

struct Scope {   // temporaries of the build, freed on every exit path
    std::vector<void *> tmp;
    ~Scope() { for (void *p : tmp) if (p) (void)hipFree(p); }
};

#define GB_TRY(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) { *err = std::string(#expr) + ": " + hipGetErrorString(e_); return -3; } \
    } while (0)

template <class T>
int tmp_alloc(Scope &s, T **p, size_t n, std::string *err) {
    void *q = nullptr;
    const hipError_t e = hipMalloc(&q, (n ? n : 1) * sizeof(T));
    if (e != hipSuccess) { *err = std::string("hipMalloc: ") + hipGetErrorString(e); return -2; }
    s.tmp.push_back(q);
    *p = static_cast<T *>(q);
    return 0;
}

template <class T>
int keep_alloc(DeviceGrid &d, T **p, size_t n, std::string *err) {
    void *q = nullptr;
    const hipError_t e = hipMalloc(&q, (n ? n : 1) * sizeof(T));
    if (e != hipSuccess) { *err = std::string("hipMalloc: ") + hipGetErrorString(e); return -2; }
    d.allocs.push_back(q);
    *p = static_cast<T *>(q);
    return 0;
}

template <class T>
int fetch_array(std::vector<T> &dst, const T *src, size_t n, std::string *err) {
    dst.resize(n);
    if (n) GB_TRY(hipMemcpy(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}

int scan_exclusive(Scope &s, const int32_t *in, int32_t *out, int64_t n, std::string *err) {
    size_t bytes = 0;
    GB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)n));
    char *ws = nullptr;
    int rc = tmp_alloc(s, &ws, bytes, err);
    if (rc) return rc;
    GB_TRY(hipcub::DeviceScan::ExclusiveSum(ws, bytes, in, out, (int)n));
    return 0;
}

// The device copies of everything a HostGrid holds.  A grid built here brings an array to the host only when
// somebody asks for it (nin_grid_array_*, the psup / edge builders, an upload to ANOTHER device): the weight
// kernels never do.  Owns the arrays no kernel needs afterwards; the others belong to the DeviceGrid, which
// calls ensure(A_ALL) before it frees them.
struct DeviceMirror : LazyArrays {
    int device = 0;
    int64_t E = 0, P = 0, F = 0, nnz_e = 0, nnz_f = 0;
    int32_t *inpoel = nullptr, *esuel = nullptr, *infael = nullptr, *inpofa = nullptr;
    int8_t *etype = nullptr;
    uint8_t *bfaces = nullptr, *bpoints = nullptr;
    double *fa = nullptr;
    const int32_t *esup_ptr = nullptr, *esup = nullptr, *fsup_ptr = nullptr, *fsup = nullptr, *face_cells = nullptr;
    const double *coords = nullptr, *cen = nullptr, *fc = nullptr;
    const float *fn = nullptr;
    std::vector<void *> owned;

    ~DeviceMirror() override {
        (void)hipSetDevice(device);
        for (void *p : owned) (void)hipFree(p);
    }
    int fetch(HostGrid &h, unsigned which, std::string *err) override {
        GB_TRY(hipSetDevice(device));
        int rc = 0;
        std::vector<int32_t> tmp;
        if (!rc && (which & A_INPOEL)) rc = fetch_array(h.inpoel, inpoel, (size_t)E * 8, err);
        if (!rc && (which & A_ETYPE)) rc = fetch_array(h.etype, etype, (size_t)E, err);
        if (!rc && (which & A_ESUP_PTR)) { rc = fetch_array(tmp, esup_ptr, (size_t)P + 1, err); if (!rc) HostGrid::widen(tmp, h.esup_ptr); }
        if (!rc && (which & A_ESUP)) rc = fetch_array(h.esup, esup, (size_t)nnz_e, err);
        if (!rc && (which & A_FSUP_PTR)) { rc = fetch_array(tmp, fsup_ptr, (size_t)P + 1, err); if (!rc) HostGrid::widen(tmp, h.fsup_ptr); }
        if (!rc && (which & A_FSUP)) rc = fetch_array(h.fsup, fsup, (size_t)nnz_f, err);
        if (!rc && (which & A_ESUF)) { rc = fetch_array(tmp, face_cells, (size_t)F * 2, err); if (!rc) h.esuf_from_pairs(tmp); }
        if (!rc && (which & A_ESUEL)) rc = fetch_array(h.esuel, esuel, (size_t)E * 6, err);
        if (!rc && (which & A_INFAEL)) rc = fetch_array(h.infael, infael, (size_t)E * 6, err);
        if (!rc && (which & A_INPOFA)) rc = fetch_array(h.inpofa, inpofa, (size_t)F * 4, err);
        if (!rc && (which & A_BFACES)) rc = fetch_array(h.boundary_faces, bfaces, (size_t)F, err);
        if (!rc && (which & A_BPOINTS)) rc = fetch_array(h.boundary_points, bpoints, (size_t)P, err);
        if (!rc && (which & A_COORDS)) rc = fetch_array(h.coords, coords, (size_t)P * 3, err);
        if (!rc && (which & A_CENTROIDS)) rc = fetch_array(h.centroids, cen, (size_t)E * 3, err);
        if (!rc && (which & A_FCENTERS)) rc = fetch_array(h.faces_centers, fc, (size_t)F * 3, err);
        if (!rc && (which & A_NORMALS)) rc = fetch_array(h.normal_faces, fn, (size_t)F * 3, err);
        if (!rc && (which & A_AREAS)) rc = fetch_array(h.faces_areas, fa, (size_t)F, err);
        return rc;
    }
};

}  // namespace

// Returns 0, -1 (bad connectivity), -2 (memory), -3 (HIP), -5 (a count does not fit int32).
int build_grid_on_device(HostGrid &h, DeviceGrid &d, int device, const int64_t *connectivity,
                         const int64_t *element_types, const double *xyz, int coords_dim, std::string *err) {
    const int64_t E = h.n_elems, P = h.n_points;
    const bool timing = getenv("NIN_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipDeviceSynchronize();
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[nin_grid/device] %-10s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    GB_TRY(hipSetDevice(device));
    d.device = device;
    Topo t;
    for (int ty = 0; ty < kNumElementTypes; ++ty) {
        t.npoel[ty] = (int8_t)h.npoel[ty];
        t.nfael[ty] = (int8_t)h.nfael[ty];
        for (int f = 0; f < kMaxFacesPerElement; ++f) {
            t.lnofa[ty][f] = (int8_t)h.lnofa[ty][f];
            for (int k = 0; k < kMaxPointsPerFace; ++k) t.lpofa[ty][f][k] = (int8_t)h.lpofa[ty][f][k];
        }
    }
    Scope s;
    int rc;
    GridView &v = d.v;
    v.n_points = (int32_t)P; v.n_elems = (int32_t)E; v.dim = (int32_t)h.dim;

    // ---- ingest ------------------------------------------------------------------------------------------
    int32_t *inpoel = nullptr; int8_t *etype = nullptr; double *coords = nullptr;
    int32_t *scal = nullptr;   // [0] bad, [1] max esup row, [2] max fsup row, [3] any internal face
    if ((rc = tmp_alloc(s, &inpoel, (size_t)E * 8, err)) || (rc = tmp_alloc(s, &etype, (size_t)E, err)) ||
        (rc = keep_alloc(d, &coords, (size_t)P * 3, err)) || (rc = tmp_alloc(s, &scal, 4, err)))
        return rc;
    GB_TRY(hipMemset(scal, 0, 16));
    {
        int64_t *conn = nullptr, *types = nullptr; double *x = nullptr;
        if ((rc = tmp_alloc(s, &conn, (size_t)E * 8, err)) || (rc = tmp_alloc(s, &types, (size_t)E, err)) ||
            (rc = tmp_alloc(s, &x, (size_t)P * coords_dim, err)))
            return rc;
        GB_TRY(hipMemcpy(conn, connectivity, (size_t)E * 64, hipMemcpyHostToDevice));
        GB_TRY(hipMemcpy(types, element_types, (size_t)E * 8, hipMemcpyHostToDevice));
        GB_TRY(hipMemcpy(x, xyz, (size_t)P * coords_dim * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_ingest_elems, dim3(blocks_for(E)), dim3(TPB), 0, 0, E, P, t, conn, types, inpoel, etype, scal);
        hipLaunchKernelGGL(k_ingest_coords, dim3(blocks_for(P)), dim3(TPB), 0, 0, P, coords_dim, x, coords);
        int32_t hs[4];
        GB_TRY(hipMemcpy(hs, scal, 16, hipMemcpyDeviceToHost));
        if (hs[0]) return -1;
    }
    v.coords = coords;
    lap("ingest");

    // ---- esup ------------------------------------------------------------------------------------------------
    int32_t *cnt = nullptr, *esup_ptr = nullptr, *esup = nullptr;
    if ((rc = tmp_alloc(s, &cnt, (size_t)P + 1, err)) || (rc = keep_alloc(d, &esup_ptr, (size_t)P + 1, err))) return rc;
    GB_TRY(hipMemset(cnt, 0, ((size_t)P + 1) * 4));
    hipLaunchKernelGGL(k_esup_count, dim3(blocks_for(E)), dim3(TPB), 0, 0, E, t, inpoel, etype, cnt);
    if ((rc = scan_exclusive(s, cnt, esup_ptr, P + 1, err))) return rc;
    int32_t nnz_e = 0;
    GB_TRY(hipMemcpy(&nnz_e, esup_ptr + P, 4, hipMemcpyDeviceToHost));
    if ((rc = keep_alloc(d, &esup, (size_t)nnz_e, err))) return rc;
    GB_TRY(hipMemset(cnt, 0, ((size_t)P + 1) * 4));
    hipLaunchKernelGGL(k_esup_fill, dim3(blocks_for(E)), dim3(TPB), 0, 0, E, t, inpoel, etype, esup_ptr, cnt, esup);
    hipLaunchKernelGGL(k_sort_rows, dim3(blocks_for(P)), dim3(TPB), 0, 0, P, esup_ptr, esup, scal + 1);
    v.esup_ptr = esup_ptr; v.esup = esup;
    d.nnz_e = nnz_e;
    lap("esup");

    // ---- esuel -----------------------------------------------------------------------------------------------
    int32_t *esuel = nullptr;
    if ((rc = tmp_alloc(s, &esuel, (size_t)E * 6, err))) return rc;
    hipLaunchKernelGGL(k_esuel, dim3(blocks_for(E * 6)), dim3(TPB), 0, 0, E, t, inpoel, etype, esup_ptr, esup, esuel);
    lap("esuel");

    // ---- faces: numbering, inpofa, owner pairs -----------------------------------------------------------
    int32_t *own_cnt = nullptr, *own_start = nullptr, *infael = nullptr, *inpofa = nullptr, *face_cells = nullptr;
    if ((rc = tmp_alloc(s, &own_cnt, (size_t)E + 1, err)) || (rc = tmp_alloc(s, &own_start, (size_t)E + 1, err)) ||
        (rc = tmp_alloc(s, &infael, (size_t)E * 6, err)))
        return rc;
    GB_TRY(hipMemset(own_cnt, 0, ((size_t)E + 1) * 4));
    hipLaunchKernelGGL(k_own_count, dim3(blocks_for(E)), dim3(TPB), 0, 0, E, t, etype, esuel, own_cnt);
    if ((rc = scan_exclusive(s, own_cnt, own_start, E + 1, err))) return rc;
    int32_t F32 = 0;
    GB_TRY(hipMemcpy(&F32, own_start + E, 4, hipMemcpyDeviceToHost));
    const int64_t F = F32;
    if (F * 4 >= INT32_MAX) return -5;
    if ((rc = tmp_alloc(s, &inpofa, (size_t)F * 4, err)) || (rc = keep_alloc(d, &face_cells, (size_t)F * 2, err))) return rc;
    hipLaunchKernelGGL(k_faces_fill, dim3(blocks_for(E)), dim3(TPB), 0, 0, E, t, inpoel, etype, esuel, own_start, infael, inpofa, face_cells);
    hipLaunchKernelGGL(k_faces_mirror, dim3(blocks_for(E * 6)), dim3(TPB), 0, 0, E, t, etype, esuel, infael);
    v.n_faces = (int32_t)F; v.face_cells = face_cells;
    h.n_faces = F;
    lap("infael");

    // ---- fsup ------------------------------------------------------------------------------------------------
    int32_t *fsup_ptr = nullptr, *fsup = nullptr;
    if ((rc = keep_alloc(d, &fsup_ptr, (size_t)P + 1, err))) return rc;
    GB_TRY(hipMemset(cnt, 0, ((size_t)P + 1) * 4));
    hipLaunchKernelGGL((k_fsup<false>), dim3(blocks_for(P)), dim3(TPB), 0, 0, P, t, inpoel, etype, esup_ptr, esup, esuel, infael, cnt,
                       (const int32_t *)nullptr, (int32_t *)nullptr, scal + 2);
    if ((rc = scan_exclusive(s, cnt, fsup_ptr, P + 1, err))) return rc;
    int32_t nnz_f = 0;
    GB_TRY(hipMemcpy(&nnz_f, fsup_ptr + P, 4, hipMemcpyDeviceToHost));
    if ((rc = keep_alloc(d, &fsup, (size_t)nnz_f, err))) return rc;
    hipLaunchKernelGGL((k_fsup<true>), dim3(blocks_for(P)), dim3(TPB), 0, 0, P, t, inpoel, etype, esup_ptr, esup, esuel, infael, cnt,
                       fsup_ptr, fsup, scal + 2);
    v.fsup_ptr = fsup_ptr; v.fsup = fsup;
    d.nnz_f = nnz_f;
    lap("fsup");

    // ---- boundary flags ----------------------------------------------------------------------------------
    uint8_t *bfaces = nullptr, *bpoints = nullptr;
    if ((rc = tmp_alloc(s, &bfaces, (size_t)F, err)) || (rc = tmp_alloc(s, &bpoints, (size_t)P, err))) return rc;
    GB_TRY(hipMemset(bpoints, 0, (size_t)P));
    hipLaunchKernelGGL(k_boundary, dim3(blocks_for(F)), dim3(TPB), 0, 0, F, face_cells, inpofa, bfaces, bpoints, scal + 3);
    lap("esuf");

    // ---- geometry ----------------------------------------------------------------------------------------
    double *cen = nullptr, *fc = nullptr, *fa = nullptr; float *fn = nullptr;
    if ((rc = keep_alloc(d, &cen, (size_t)E * 3, err)) || (rc = keep_alloc(d, &fc, (size_t)F * 3, err)) ||
        (rc = keep_alloc(d, &fn, (size_t)F * 3, err)) || (rc = tmp_alloc(s, &fa, (size_t)F, err)))
        return rc;
    hipLaunchKernelGGL(k_centroids, dim3(blocks_for(E)), dim3(TPB), 0, 0, E, (int)h.dim, t, inpoel, etype, coords, cen);
    hipLaunchKernelGGL(k_faces_geometry, dim3(blocks_for(F)), dim3(TPB), 0, 0, F, (int)h.dim, inpofa, coords, fc, fn, fa);
    v.centroids = cen; v.face_center = fc; v.face_normal = fn;
    GB_TRY(hipDeviceSynchronize());
    lap("geometry");

    // ---- the host side: scalars now, arrays on first use -------------------------------------------------
    int32_t hs[4];
    GB_TRY(hipMemcpy(hs, scal, 16, hipMemcpyDeviceToHost));
    h.mx_elems_per_point = hs[1];
    h.mx_faces_per_point = hs[2];
    h.mx_elems_per_face = hs[3] ? 2 : 1;
    h.nnz_esup = nnz_e;
    h.nnz_fsup = nnz_f;
    auto *mir = new DeviceMirror();
    mir->device = device; mir->E = E; mir->P = P; mir->F = F; mir->nnz_e = nnz_e; mir->nnz_f = nnz_f;
    mir->inpoel = inpoel; mir->etype = etype; mir->esuel = esuel; mir->infael = infael; mir->inpofa = inpofa;
    mir->bfaces = bfaces; mir->bpoints = bpoints; mir->fa = fa;
    mir->esup_ptr = esup_ptr; mir->esup = esup; mir->fsup_ptr = fsup_ptr; mir->fsup = fsup; mir->face_cells = face_cells;
    mir->coords = coords; mir->cen = cen; mir->fc = fc; mir->fn = fn;
    for (void *q : {(void *)inpoel, (void *)etype, (void *)esuel, (void *)infael, (void *)inpofa, (void *)bfaces,
                    (void *)bpoints, (void *)fa}) {   // these outlive the build: the mirror owns them now
        for (auto &t_ : s.tmp)
            if (t_ == q) t_ = nullptr;
        mir->owned.push_back(q);
    }
    h.lazy.reset(mir);
    h.have = 0;
    d.prebuilt = true;
    if (getenv("NIN_GRID_EAGER_MIRROR")) {   // diagnostic: the cost of bringing every array to the host at once
        if ((rc = h.ensure(A_ALL, err))) return rc;
        lap("to host");
    }
    if (h.build_edges) h.build_inedel();
    return 0;
}

// ---- GLS launch plan on the device ---------------------------------------------------------------------------
namespace {

__global__ void k_classify(GridView g, int use_group, int force_global, uint8_t *__restrict__ node_class,
                           unsigned long long *__restrict__ class_max) {
    const int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (p >= g.n_points) return;
    const int64_t ne = g.esup_ptr[p + 1] - g.esup_ptr[p], nf = g.fsup_ptr[p + 1] - g.fsup_ptr[p];
    int64_t nbf = 0;
    for (int32_t q = g.fsup_ptr[p]; q < g.fsup_ptr[p + 1]; ++q) nbf += g.face_cells[2 * (int64_t)g.fsup[q] + 1] == -1;
    if ((use_group & 1) && ne == 8 && nf == 12 && nbf == 0 && g.dim == 3) {
        int32_t d[4];   // the hex8 kernel needs the cells to form the cube graph (hex8_desc.hpp)
        if (hex8_descriptor(g, (int32_t)p, d)) { node_class[p] = 255; return; }
    }
    // (general-kind nodes that fit the small-node kernel -- pyramid apexes: 7 cells, 43 rows -- are cheaper there: 26.64 -> 26.54 ms
    //  on BASELINE config [3])
    const bool small_fits = (use_group & 8) && !force_global && g.dim == 3 && ne <= 12 && nf <= 48 && ne + 3 * nf <= 64;
    if ((use_group & 2) && nbf == 0 && ne <= kMfwMaxCells) {
        uint32_t w[kMfwDescWords];   // fronts of 3-face cells that share no face + dense cells (mfw_desc.hpp)
        const int kind = mfw_descriptor(g, (int32_t)p, w);
        if (kind == 1) {
            const int F = w[24] & 255, D = (w[24] >> 8) & 255;
            node_class[p] = (F <= kMfwSmallFronts && D <= kMfwSmallDense) ? 253 : 254;
            return;
        }
        // (bit 6: the wide kernel takes the general kind's nodes too -- the default; NIN_GLS_MFW_GENERAL=1 clears it)
        if (kind == 2 && (use_group & 4) && !small_fits && !(use_group & 64)) { node_class[p] = 252; return; }
    }
    // interior nodes of unstructured meshes: more cells than the kinds above hold, no two-colouring (kernels_gls_mfx.hip, mfx_desc.hpp)
    // (round 4: boundary nodes too -- computed only when the variable flags them Neumann; their boundary faces are one row each)
    const bool small_fits_b = (use_group & 8) && !force_global && g.dim == 3 && ne <= 12 && nf <= 48 && ne + 3 * (nf - nbf) + nbf <= 64;
    // (an interior node the small-node kernel could take -- at most 12 cells, 64 rows -- comes here too if it has at least 9 cells: with fronts
    //  its dense problem is 43 x 19 where the small-node kernel sweeps 55 x 31: the wide kernel's small class, bit 9: NIN_GLS_NO_MFX_SMALL)
    const bool small_class_candidate = (use_group & 512) && nbf == 0 && small_fits && ne >= 9;
    if ((use_group & 32) && !force_global && ne <= kMfxMaxCells && (!(nbf == 0 ? small_fits : small_fits_b) || small_class_candidate) &&
        (nbf == 0 || !(use_group & 128))) {
        uint32_t w[kMfxDescWords];
        const int k = mfx_descriptor(g, (int32_t)p, w);   // 1 + the size class of its dense problem, or kMfxSmallCode
        if (k == kMfxSmallCode && (use_group & 512)) { node_class[p] = 240; return; }
        if (k == kMfxMidCode && (use_group & 1024)) { node_class[p] = 239; return; }            // (bit 10: NIN_GLS_NO_MFX_7X12 clears it: class (8, 13))
        if (k > 0 && nbf == 0 && !small_fits) { node_class[p] = (uint8_t)(243 + (k == kMfxSmallCode ? 1 : k == kMfxMidCode ? 3 : k) - 1); return; }
        if (k > 0 && k <= 2 && nbf > 0) { node_class[p] = 242; return; }   // a boundary node that fits 7 x 11 tiles: the boundary instantiation's list
    }
    // interior nodes beyond the wide kernel's 16 fronts + 21 dense cells (a random point cloud's Delaunay mesh: 6 % of its nodes): the
    // dense problem in global-memory tiles (kernels_gls_mfg.hip, mfg_desc.hpp; bit 8: NIN_GLS_NO_MFG clears it)
    if ((use_group & 256) && (use_group & 32) && !force_global && nbf == 0 && ne > 12 && ne <= kMfgMaxCells) {
        uint32_t w[kMfgDescWords];
        if (mfg_descriptor(g, (int32_t)p, w)) { node_class[p] = 241; return; }
    }
    // nodes inside a boundary face of a hexahedron mesh: two lanes per node (kernels_gls_quad4.hip)
    if ((use_group & 16) && !force_global && ne == 4 && nf == 8 && nbf == 4 && g.dim == 3) {
        int32_t d2[2];
        if (quad4_descriptor(g, (int32_t)p, d2)) { node_class[p] = 248; return; }
    }
    // small nodes (in practice: boundary nodes): the one-wavefront dense kernel, lane = row (kernels_gls_mfw.hip, nin_gls_small_kernel)
    if ((use_group & 8) && !force_global && g.dim == 3 && ne <= 12 && nf <= 48 && ne + 3 * (nf - nbf) + nbf <= 64) {
        node_class[p] = ne <= 4 ? 249 : ne <= 8 ? 250 : 251;
        return;
    }
    int64_t bytes, rows, cols;
    const int c = gls_node_class(ne, nf, nbf, force_global != 0, &bytes, &rows, &cols);
    node_class[p] = (uint8_t)c;
    atomicMax(&class_max[3 * c + 0], (unsigned long long)bytes);
    atomicMax(&class_max[3 * c + 1], (unsigned long long)rows);
    atomicMax(&class_max[3 * c + 2], (unsigned long long)cols);
}

}  // namespace

int launch_classify(const GridView &g, int use_group, int force_global, uint8_t *node_class,
                    unsigned long long *class_max, hipStream_t stream) {
    hipLaunchKernelGGL(k_classify, dim3(blocks_for(g.n_points)), dim3(TPB), 0, stream, g, use_group, force_global, node_class,
                       class_max);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace nin
