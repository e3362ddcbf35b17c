// abi.hip -- the extern "C" surface declared in include/ninpol_amd.h.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cmath>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/ninpol_amd.h"
#include "device_grid.hpp"
#include "grid_host.hpp"
#include "launch.hpp"
#include "mfw_desc.hpp"
#include "mfg_desc.hpp"
#include "mfx_desc.hpp"

using namespace nin;

struct nin_grid {
    HostGrid h;
    DeviceGrid d;
    std::vector<uint8_t> node_class;  // GLS size class of every node (host copy, for target subsets)
    int coords_dim = 3;
};

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace

namespace nin {
// the same for the other translation units of the C ABI (exchange.hip): sets what nin_last_error() returns
int abi_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
}  // namespace nin

namespace {

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(NIN_EHIP, "%s: %s", #expr, hipGetErrorString(e_));       \
    } while (0)

template <class T>
int dev_alloc(DeviceGrid &d, T **ptr, size_t count) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) return fail(NIN_ENOMEM, "hipMalloc(%zu bytes): %s", count * sizeof(T), hipGetErrorString(e));
    d.allocs.push_back(p);
    *ptr = static_cast<T *>(p);
    return 0;
}

template <class T>
int dev_upload(DeviceGrid &d, const T **ptr, const std::vector<T> &src) {
    T *p = nullptr;
    int rc = dev_alloc(d, &p, src.size());
    if (rc) return rc;
    if (!src.empty()) HIP_TRY(hipMemcpy(p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    *ptr = p;
    return 0;
}

void dev_free_all(DeviceGrid &d) {
    if (d.device >= 0) (void)hipSetDevice(d.device);
    if (d.flag_staging) (void)hipHostFree(d.flag_staging);
    if (d.ev_weights) (void)hipEventDestroy(static_cast<hipEvent_t>(d.ev_weights));
    if (d.ev_scan) (void)hipEventDestroy(static_cast<hipEvent_t>(d.ev_scan));
    if (d.copy_stream) (void)hipStreamDestroy(static_cast<hipStream_t>(d.copy_stream));
    if (d.copy_stream2) (void)hipStreamDestroy(static_cast<hipStream_t>(d.copy_stream2));
    if (d.ev_fork) (void)hipEventDestroy(static_cast<hipEvent_t>(d.ev_fork));
    if (d.ev_join) (void)hipEventDestroy(static_cast<hipEvent_t>(d.ev_join));
    if (d.side_stream) (void)hipStreamDestroy(static_cast<hipStream_t>(d.side_stream));
    for (void *p : d.allocs) (void)hipFree(p);
    d.allocs.clear();
    d = DeviceGrid{};
}

struct ArrayRef {
    int dtype;      // NIN_I64 / NIN_F64
    int64_t count;
    int kind;       // source element type: 0 int32, 1 int64, 2 uint8, 3 int8, 4 double, 5 float
    const void *ptr;
};

unsigned array_bits(const std::string &name) {
    if (name == "esup") return A_ESUP;
    if (name == "esup_ptr") return A_ESUP_PTR;
    if (name == "fsup") return A_FSUP;
    if (name == "fsup_ptr") return A_FSUP_PTR;
    if (name == "esuf" || name == "esuf_ptr") return A_ESUF;
    if (name == "esuel") return A_ESUEL;
    if (name == "infael") return A_INFAEL;
    if (name == "inpofa") return A_INPOFA;
    if (name == "inpoel") return A_INPOEL;
    if (name == "boundary_faces") return A_BFACES;
    if (name == "boundary_points") return A_BPOINTS;
    if (name == "element_types") return A_ETYPE;
    if (name == "point_coords") return A_COORDS;
    if (name == "centroids") return A_CENTROIDS;
    if (name == "faces_centers") return A_FCENTERS;
    if (name == "faces_areas") return A_AREAS;
    if (name == "normal_faces") return A_NORMALS;
    return 0;   // psup / edges: their builders ask for what they read
}

bool lookup_array(nin_grid *g, const std::string &name, ArrayRef *r) {
    HostGrid &h = g->h;
    if (h.ensure(array_bits(name))) return false;   // a grid built on the device: bring the array over first
    auto I32 = [&](const std::vector<int32_t> &v) { *r = {NIN_I64, (int64_t)v.size(), 0, v.data()}; return true; };
    auto I64 = [&](const std::vector<int64_t> &v) { *r = {NIN_I64, (int64_t)v.size(), 1, v.data()}; return true; };
    auto U8 = [&](const std::vector<uint8_t> &v) { *r = {NIN_I64, (int64_t)v.size(), 2, v.data()}; return true; };
    auto F64 = [&](const std::vector<double> &v) { *r = {NIN_F64, (int64_t)v.size(), 4, v.data()}; return true; };
    if (name == "esup") return I32(h.esup);
    if (name == "esup_ptr") return I64(h.esup_ptr);
    if (name == "fsup") return I32(h.fsup);
    if (name == "fsup_ptr") return I64(h.fsup_ptr);
    if (name == "esuf") return I32(h.esuf);
    if (name == "esuf_ptr") return I64(h.esuf_ptr);
    if (name == "esuel") return I32(h.esuel);
    if (name == "infael") return I32(h.infael);
    if (name == "inpofa") return I32(h.inpofa);
    if (name == "inpoel") return I32(h.inpoel);
    if (name == "boundary_faces") return U8(h.boundary_faces);
    if (name == "boundary_points") return U8(h.boundary_points);
    if (name == "element_types") { *r = {NIN_I64, (int64_t)h.etype.size(), 3, h.etype.data()}; return true; }
    if (name == "psup" || name == "psup_ptr") {
        h.build_psup();
        return name == "psup" ? I32(h.psup) : I64(h.psup_ptr);
    }
    if (name == "inedel" || name == "inpoed") {
        if (!h.build_edges) { *r = {NIN_I64, 0, 0, nullptr}; return true; }  // empty, like the reference's (0,0) arrays
        h.build_inedel();
        return name == "inedel" ? I32(h.inedel) : I32(h.inpoed);
    }
    if (name == "point_coords") return F64(h.coords);
    if (name == "centroids") return F64(h.centroids);
    if (name == "faces_centers") return F64(h.faces_centers);
    if (name == "faces_areas") return F64(h.faces_areas);
    if (name == "normal_faces") { *r = {NIN_F64, (int64_t)h.normal_faces.size(), 5, h.normal_faces.data()}; return true; }
    return false;
}


// ---- locality order of a kernel's node list --------------------------------------------------------------------------------
// A list in node order walks the mesh the way its nodes are numbered (a structured mesh: along x, row by row, plane by plane),
// and a node shares its cells and faces with neighbours that are a whole plane of the numbering away: with 2 x 256 groups of
// the cube-node kernel in flight per XCD the data comes in again for every plane (FETCH_SIZE 3.58 GiB per launch at 216^3
// against 2.6 with half as many waves).  So the list is reordered by where the nodes ARE -- inside each piece of interpolate()'s
// pipeline, whose pieces stay sub-ranges of the list:
//   default   strips of 16 mesh rows (axis 1 quantised to ~cbrt(P) levels) walked plane by plane (axis 2), node order inside:
//             the rows a plane apart meet in the 4 MB L2 and the requests stay as coalesced as the numbering makes them --
//             FETCH_SIZE 3.58 -> 1.97 GiB (traffic = the algorithmic bytes), 4.90 -> 4.78 ms in one session (s4 / s8 / s32: 4.84 /
//             4.85 / 4.82);
//   "m"       Morton order of the coordinates (10 bits an axis over the bounding cube): 3.05 GiB but 1.6 % SLOWER than node order
//             -- uncoalesced requests are instructions too, and the kernel is bound by what it issues;
//   "off"     node order.
// Any order gives the same weights bit for bit (a node's arithmetic does not depend on its neighbours in the list; tested).
// Runs of 16 consecutive entries (one pass of a 16-nodes-per-wavefront kernel) move as one: inside a run the rows of the CSR
// tables and of the output stay next to each other -- single nodes in Morton order cost more in uncoalesced requests than
// the order saves (5.09 against 4.83 ms at 216^3, measured; runs of 64 / 256 are slower still).
#ifndef NIN_LOCALITY_RUN
#define NIN_LOCALITY_RUN 16
#endif
constexpr int kLocalityRun = NIN_LOCALITY_RUN;
__device__ __forceinline__ uint32_t spread10(uint32_t v) {   // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// strip > 0: instead, strips of `strip` mesh rows (axis 1 at ~cbrt(P) levels) walked plane by plane (axis 2), node order inside:
// the rows a plane apart meet in the L2 while the requests stay as coalesced as the numbering makes them
__global__ void k_locality_keys(const double *__restrict__ coords, const int32_t *__restrict__ nodes, int32_t count, double lo,
                                double scale, int32_t c1, int32_t c2, int32_t c3, int32_t strip, double lscale,
                                uint32_t *__restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int32_t p = nodes[i & ~(int64_t)(kLocalityRun - 1)];   // the key of the run's first entry: the (stable) sort moves whole runs
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double t = (coords[3 * (size_t)p + k] - lo) * scale;
        q[k] = t > 0.0 ? (t < 1023.0 ? (uint32_t)t : 1023u) : 0u;
    }
    const int32_t self = nodes[i];
    const uint32_t piece = (uint32_t)(self >= c1) + (uint32_t)(self >= c2) + (uint32_t)(self >= c3);
    if (strip > 0) {
        const double ty = (coords[3 * (size_t)p + 1] - lo) * lscale, tz = (coords[3 * (size_t)p + 2] - lo) * lscale;
        const uint32_t yq = ty > 0.0 ? (ty < 32767.0 ? (uint32_t)(ty + 0.5) : 32767u) : 0u;
        const uint32_t zq = tz > 0.0 ? (tz < 32767.0 ? (uint32_t)(tz + 0.5) : 32767u) : 0u;
        keys[i] = (piece << 30) | ((yq / (uint32_t)strip) << 15) | zq;
        return;
    }
    keys[i] = (piece << 30) | spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
}
// `list` (device, count entries, ascending node ids) -> the same entries in locality order; pieces = the pipeline's node boundaries
int locality_order(const DeviceGrid &d, int32_t *list, int32_t count, const int32_t *chunk_node) {
    if (count < 2048) return NIN_OK;
    double lo = 0.0, hi = 0.0;
    {   // the bounding cube: smallest and largest coordinate of any axis
        double *mm = nullptr;
        void *rt = nullptr;
        size_t rb = 0, rb2 = 0;
        const int n3 = 3 * d.v.n_points;
        bool ok = hipMalloc((void **)&mm, 16) == hipSuccess &&
                  hipcub::DeviceReduce::Min(nullptr, rb, d.v.coords, mm, n3) == hipSuccess &&
                  hipcub::DeviceReduce::Max(nullptr, rb2, d.v.coords, mm + 1, n3) == hipSuccess &&
                  hipMalloc(&rt, rb > rb2 ? rb : rb2) == hipSuccess;
        rb = rb > rb2 ? rb : rb2;
        ok = ok && hipcub::DeviceReduce::Min(rt, rb, d.v.coords, mm, n3) == hipSuccess &&
             hipcub::DeviceReduce::Max(rt, rb, d.v.coords, mm + 1, n3) == hipSuccess;
        double h2[2] = {0.0, 0.0};
        ok = ok && hipMemcpy(h2, mm, 16, hipMemcpyDeviceToHost) == hipSuccess;
        (void)hipFree(mm); (void)hipFree(rt);
        if (!ok) return fail(NIN_EHIP, "locality order: bounding cube: %s", hipGetErrorString(hipGetLastError()));
        lo = h2[0]; hi = h2[1];
    }
    if (!(hi > lo)) return NIN_OK;
    uint32_t *keys = nullptr, *keys_out = nullptr;
    int32_t *list_out = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    int rc = NIN_OK;
    auto step = [&](hipError_t e) { if (e != hipSuccess && rc == NIN_OK) rc = fail(NIN_EHIP, "locality order: %s", hipGetErrorString(e)); return e == hipSuccess; };
    if (step(hipMalloc((void **)&keys, (size_t)count * 4)) && step(hipMalloc((void **)&keys_out, (size_t)count * 4)) &&
        step(hipMalloc((void **)&list_out, (size_t)count * 4))) {
        const char *mode = getenv("NIN_GLS_LOCALITY_ORDER");
        const int32_t strip = !mode ? 16 : mode[0] == 's' ? (mode[1] ? atoi(mode + 1) : 16) : 0;   // default: strips of 16 rows; "m": Morton
        const double levels = std::cbrt((double)d.v.n_points) - 1.0;                          // mesh intervals per axis of a cube of this many nodes
        hipLaunchKernelGGL(k_locality_keys, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, nullptr, d.v.coords, list, count, lo,
                           1024.0 / (hi - lo), chunk_node[1], chunk_node[2], chunk_node[3], strip, (levels > 1.0 ? levels : 1.0) / (hi - lo), keys);
        if (step(hipGetLastError()) && step(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys_out, list, list_out, count)) &&
            step(hipMalloc(&tmp, tmp_bytes)) &&
            step(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys_out, list, list_out, count)))
            (void)step(hipMemcpy(list, list_out, (size_t)count * 4, hipMemcpyDeviceToDevice));
    }
    (void)hipFree(keys); (void)hipFree(keys_out); (void)hipFree(list_out); (void)hipFree(tmp);
    return rc;
}
}  // namespace

extern "C" {

const char *nin_last_error(void) { return g_err.c_str(); }
const char *nin_version(void) { return "ninpol_amd 0.1 (gfx950)"; }

static int grid_create_common(int64_t dim, int64_t n_elems, int64_t n_points, const int64_t *npoel, const int64_t *nfael,
                              const int64_t *lnofa, const int64_t *lpofa, const int64_t *nedel, const int64_t *lpoed,
                              const int64_t *connectivity, const int64_t *element_types, const double *coords,
                              int coords_dim, int build_edges, int num_threads, int device, nin_grid **out) {
    if (!out) return fail(NIN_EINVAL, "out is NULL");
    *out = nullptr;
    // the three checks of grid.pyx:55-60 (the Python layer turns them into the reference's ValueErrors)
    if (dim < 1) return fail(NIN_EINVAL, "The number of dimensions must be greater than 0.");
    if (n_elems < 1) return fail(NIN_EINVAL, "The number of elements must be greater than 0.");
    if (n_points < 1) return fail(NIN_EINVAL, "The number of points must be greater than 0.");
    if (!npoel || !nfael || !lnofa || !lpofa || !nedel || !lpoed || !connectivity || !element_types || !coords)
        return fail(NIN_EINVAL, "NULL table or array");
    if (coords_dim < 1 || coords_dim > 3) return fail(NIN_EINVAL, "coords_dim must be 1..3");
    if (n_elems * 8 >= INT32_MAX || n_points >= INT32_MAX) return fail(NIN_ERANGE, "mesh too large for the int32 layout");
    if (device >= 0) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            return fail(NIN_ENODEVICE, "no HIP device visible: the device grid build has no CPU fallback (use nin_grid_create)");
        if (device >= ndev) return fail(NIN_EINVAL, "device %d out of range (%d visible)", device, ndev);
    }
    nin_grid *g = new (std::nothrow) nin_grid();
    if (!g) return fail(NIN_ENOMEM, "out of host memory");
    HostGrid &h = g->h;
    h.dim = dim; h.n_elems = n_elems; h.n_points = n_points;
    h.build_edges = build_edges; h.num_threads = num_threads;
    g->coords_dim = coords_dim;
    for (int t = 0; t < kNumElementTypes; ++t) {
        h.npoel[t] = (int32_t)npoel[t]; h.nfael[t] = (int32_t)nfael[t]; h.nedel[t] = (int32_t)nedel[t];
        for (int f = 0; f < kMaxFacesPerElement; ++f) {
            h.lnofa[t][f] = (int32_t)lnofa[t * kMaxFacesPerElement + f];
            for (int k = 0; k < kMaxPointsPerFace; ++k)
                h.lpofa[t][f][k] = (int32_t)lpofa[(t * kMaxFacesPerElement + f) * kMaxPointsPerFace + k];
        }
        for (int e = 0; e < kMaxEdgesPerElement; ++e)
            for (int k = 0; k < 2; ++k) h.lpoed[t][e][k] = (int32_t)lpoed[(t * kMaxEdgesPerElement + e) * 2 + k];
    }
    int rc;
    std::string err;
    try {
        if (device >= 0) rc = build_grid_on_device(h, g->d, device, connectivity, element_types, coords, coords_dim, &err);
        else rc = h.build(connectivity, element_types, coords, coords_dim);
    } catch (const std::bad_alloc &) {
        dev_free_all(g->d);
        delete g;
        return fail(NIN_ENOMEM, "out of host memory while building the grid");
    }
    if (rc) { dev_free_all(g->d); delete g; }
    if (rc == -5) return fail(NIN_ERANGE, "a connectivity count does not fit int32");
    if (rc == -2) return fail(NIN_ENOMEM, "device grid build: %s", err.c_str());
    if (rc == -3) return fail(NIN_EHIP, "device grid build: %s", err.c_str());
    if (rc) return fail(NIN_EINVAL, "connectivity references a point outside [0, n_points) or an unknown element type");
    *out = g;
    return NIN_OK;
}

int nin_grid_create(int64_t dim, int64_t n_elems, int64_t n_points, const int64_t *npoel, const int64_t *nfael,
                    const int64_t *lnofa, const int64_t *lpofa, const int64_t *nedel, const int64_t *lpoed,
                    const int64_t *connectivity, const int64_t *element_types, const double *coords, int coords_dim,
                    int build_edges, int num_threads, nin_grid **out) {
    return grid_create_common(dim, n_elems, n_points, npoel, nfael, lnofa, lpofa, nedel, lpoed, connectivity, element_types,
                              coords, coords_dim, build_edges, num_threads, -1, out);
}

int nin_grid_create_on_device(int64_t dim, int64_t n_elems, int64_t n_points, const int64_t *npoel, const int64_t *nfael,
                              const int64_t *lnofa, const int64_t *lpofa, const int64_t *nedel, const int64_t *lpoed,
                              const int64_t *connectivity, const int64_t *element_types, const double *coords,
                              int coords_dim, int build_edges, int device, nin_grid **out) {
    if (device < 0) return fail(NIN_EINVAL, "device must be >= 0");
    return grid_create_common(dim, n_elems, n_points, npoel, nfael, lnofa, lpofa, nedel, lpoed, connectivity, element_types,
                              coords, coords_dim, build_edges, 0, device, out);
}

void nin_grid_destroy(nin_grid *g) {
    if (!g) return;
    dev_free_all(g->d);
    delete g;
}

int64_t nin_grid_scalar(const nin_grid *g, const char *name) {
    if (!g || !name) return -1;
    const HostGrid &h = g->h;
    const std::string n(name);
    if (n == "dim") return h.dim;
    if (n == "n_elems") return h.n_elems;
    if (n == "n_points") return h.n_points;
    if (n == "n_faces") return h.n_faces;
    if (n == "n_edges") { if (h.build_edges) const_cast<HostGrid &>(h).build_inedel(); return h.n_edges; }
    if (n == "MX_ELEMENTS_PER_POINT") return h.mx_elems_per_point;
    if (n == "MX_POINTS_PER_POINT") { const_cast<HostGrid &>(h).build_psup(); return h.mx_points_per_point; }
    if (n == "MX_ELEMENTS_PER_FACE") return h.mx_elems_per_face;
    if (n == "MX_FACES_PER_POINT") return h.mx_faces_per_point;
    if (n == "coords_dim") return g->coords_dim;
    if (n == "nnz_esup") return h.nnz_esup;
    if (n == "nnz_fsup") return h.nnz_fsup;
    return -1;
}

int nin_grid_array_info(nin_grid *g, const char *name, int64_t *count, int *dtype) {
    if (!g || !name || !count || !dtype) return fail(NIN_EINVAL, "NULL argument");
    ArrayRef r;
    if (!lookup_array(g, name, &r)) return fail(NIN_EINVAL, "unknown grid array '%s'", name);
    *count = r.count;
    *dtype = r.dtype;
    return NIN_OK;
}

int nin_grid_array_copy(nin_grid *g, const char *name, void *dst, int64_t count) {
    if (!g || !name || (!dst && count)) return fail(NIN_EINVAL, "NULL argument");
    ArrayRef r;
    if (!lookup_array(g, name, &r)) return fail(NIN_EINVAL, "unknown grid array '%s'", name);
    if (count != r.count) return fail(NIN_EINVAL, "array '%s' has %lld elements, caller gave room for %lld", name, (long long)r.count, (long long)count);
    const int64_t n = r.count;
    switch (r.kind) {
        case 0: { auto s = (const int32_t *)r.ptr; auto d = (int64_t *)dst;
#pragma omp parallel for schedule(static)
                  for (int64_t i = 0; i < n; ++i) d[i] = s[i]; } break;
        case 1: if (n) memcpy(dst, r.ptr, (size_t)n * 8); break;
        case 2: { auto s = (const uint8_t *)r.ptr; auto d = (int64_t *)dst; for (int64_t i = 0; i < n; ++i) d[i] = s[i]; } break;
        case 3: { auto s = (const int8_t *)r.ptr; auto d = (int64_t *)dst; for (int64_t i = 0; i < n; ++i) d[i] = s[i]; } break;
        case 4: if (n) memcpy(dst, r.ptr, (size_t)n * 8); break;
        case 5: { auto s = (const float *)r.ptr; auto d = (double *)dst; for (int64_t i = 0; i < n; ++i) d[i] = (double)s[i]; } break;
    }
    return NIN_OK;
}

int nin_device_count(int *count) {
    if (!count) return fail(NIN_EINVAL, "NULL argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; return fail(NIN_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = c;
    return NIN_OK;
}

// a grid built on the device holds its arrays there but has no launch plan yet: not "on the device" until to_device
int nin_grid_device(const nin_grid *g) { return (g && !g->d.prebuilt) ? g->d.device : -1; }

int nin_grid_to_device(nin_grid *g, int device) {
    if (!g) return fail(NIN_EINVAL, "NULL grid");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(NIN_ENODEVICE, "no HIP device visible: libninpol_amd has no CPU fallback for the weight kernels");
    if (device < 0 || device >= ndev) return fail(NIN_EINVAL, "device %d out of range (%d visible)", device, ndev);
    const bool adopt = g->d.prebuilt && g->d.device == device;   // built on this device: the arrays are already there
    HostGrid &h = g->h;
    if (!adopt) {
        std::string err;   // a device-built grid moving elsewhere: everything comes to the host before its HBM copy goes
        if (h.ensure(A_ALL, &err)) return fail(NIN_EHIP, "mirroring the device-built grid: %s", err.c_str());
        dev_free_all(g->d);
    }
    HIP_TRY(hipSetDevice(device));
    DeviceGrid &d = g->d;
    d.device = device;
    d.prebuilt = false;
    const int64_t P = h.n_points, E = h.n_elems, F = h.n_faces;
    d.nnz_e = h.nnz_esup;
    d.nnz_f = h.nnz_fsup;
    GridView &v = d.v;
    v.n_points = (int32_t)P; v.n_elems = (int32_t)E; v.n_faces = (int32_t)F; v.dim = (int32_t)h.dim;
    int rc;
    if (!adopt) {
    {
        std::vector<int32_t> p32((size_t)P + 1);
        for (int64_t i = 0; i <= P; ++i) p32[i] = (int32_t)h.esup_ptr[i];
        if ((rc = dev_upload(d, &v.esup_ptr, p32))) return rc;
        for (int64_t i = 0; i <= P; ++i) p32[i] = (int32_t)h.fsup_ptr[i];
        if ((rc = dev_upload(d, &v.fsup_ptr, p32))) return rc;
    }
    if ((rc = dev_upload(d, &v.esup, h.esup))) return rc;
    if ((rc = dev_upload(d, &v.fsup, h.fsup))) return rc;
    if ((rc = dev_upload(d, &v.coords, h.coords))) return rc;
    if ((rc = dev_upload(d, &v.centroids, h.centroids))) return rc;
    if ((rc = dev_upload(d, &v.face_center, h.faces_centers))) return rc;
    if ((rc = dev_upload(d, &v.face_normal, h.normal_faces))) return rc;
    {
        std::vector<int32_t> fc((size_t)F * 2);
#pragma omp parallel for schedule(static)
        for (int64_t f = 0; f < F; ++f) {
            const int64_t b = h.esuf_ptr[f], n = h.esuf_ptr[f + 1] - b;
            fc[2 * f] = h.esuf[b];
            fc[2 * f + 1] = n > 1 ? h.esuf[b + 1] : -1;
        }
        if ((rc = dev_upload(d, &v.face_cells, fc))) return rc;
    }
    }
    {   // flags start as "boundary only"; nin_fields_set adds the Neumann bit
        uint8_t *fl = nullptr;
        if ((rc = dev_alloc(d, &fl, (size_t)P))) return rc;
        if (h.ensure(A_BPOINTS)) return fail(NIN_EHIP, "mirroring boundary_points failed");
        HIP_TRY(hipMemcpy(fl, h.boundary_points.data(), (size_t)P, hipMemcpyHostToDevice));
        v.flags = fl;
        double *perm = nullptr, *dm = nullptr;
        if ((rc = dev_alloc(d, &perm, (size_t)E * 9))) return rc;
        if ((rc = dev_alloc(d, &dm, (size_t)E))) return rc;
        v.perm = perm; v.diff_mag = dm;
    }
    v.centroids4 = nullptr;
    if (getenv("NIN_ROWS_PAD4") != nullptr) {   // experiment: (x, y, z, 0) per cell, a centroid in two 16-byte loads -- 3 % SLOWER than the packed array
        double *c4 = nullptr;
        if ((rc = dev_alloc(d, &c4, (size_t)E * 4))) return rc;
        if (launch_pad_centroids(v.centroids, E, c4, nullptr)) return fail(NIN_EHIP, "centroid padding kernel");
        v.centroids4 = c4;
    }
    // ---- GLS launch plan: bin nodes by the size of their least-squares system (classified on the device) ----
    g->node_class.assign((size_t)P, 0);
    std::vector<std::vector<int32_t>> lists(kGlsClasses);
    std::vector<int32_t> hex8_list, mfw_list[3], small_list[3], quad4_list, mfx_list[DeviceGrid::kMfxLists], mfg_list;
    // debugging switches: keep nodes away from the hex8 kernel (bit 0) / the one-wavefront multifrontal kernel (bit 1)
    const int use_group = (getenv("NIN_GLS_NO_GROUP") == nullptr ? 1 : 0) | (getenv("NIN_GLS_NO_MFW") == nullptr ? 2 : 0) |
                          (getenv("NIN_GLS_NO_MFW_GENERAL") == nullptr ? 4 : 0) |   // (bit 2: the multifrontal kernel's general kind)
                          (getenv("NIN_GLS_NO_SMALL") == nullptr ? 8 : 0) |         // (bit 3: the one-wavefront dense kernel for small nodes)
                          (getenv("NIN_GLS_NO_QUAD4") == nullptr ? 16 : 0) |        // (bit 4: the two-lanes-per-node kernel for quad nodes)
                          (getenv("NIN_GLS_NO_MFX") == nullptr ? 32 : 0) |          // (bit 5: the wide multifrontal kernel: unstructured meshes)
                          (getenv("NIN_GLS_NO_MFX_7X12") == nullptr ? 1024 : 0) |    // (bit 10: ... its class (7, 12))
                          (getenv("NIN_GLS_NO_MFX_SMALL") == nullptr ? 512 : 0) |    // (bit 9: ... its small class (4, 7) for interior nodes of 9 .. 14 cells)
                          (getenv("NIN_GLS_NO_MFG") == nullptr ? 256 : 0) |          // (bit 8: the multifrontal kernel on global-memory tiles: nodes beyond the wide kernel)
                          (getenv("NIN_GLS_MFX_NO_BOUNDARY") != nullptr ? 128 : 0) | // (bit 7: ... leaves the boundary nodes to the block kernel: round 3's route)
                          (getenv("NIN_GLS_MFW_GENERAL") == nullptr ? 64 : 0);      // (bit 6: ... takes the general kind's nodes too -- the default
                                                                                    //  since its dense phase runs straight-line per size class: 37 against
                                                                                    //  38 ns a node on a Delaunay mesh, equal on the mixed mesh; NIN_GLS_MFW_GENERAL=1
                                                                                    //  gives the nodes that fit back to kernels_gls_mfw.hip's general kind)
    const bool force_global = getenv("NIN_GLS_FORCE_GLOBAL") != nullptr;   // testing switch: systems in global scratch
    int64_t need_max[kGlsClasses] = {0}, rows_max[kGlsClasses] = {0}, cols_max[kGlsClasses] = {0};
    {
        uint8_t *dcls = nullptr;
        unsigned long long *dmax = nullptr, hmax[3 * kGlsClasses];
        HIP_TRY(hipMalloc((void **)&dcls, (size_t)P));
        if (hipMalloc((void **)&dmax, sizeof hmax) != hipSuccess) { (void)hipFree(dcls); return fail(NIN_ENOMEM, "hipMalloc"); }
        hipError_t e1 = hipMemset(dmax, 0, sizeof hmax);
        const int lrc = launch_classify(d.v, use_group, force_global, dcls, dmax, nullptr);
        if (e1 == hipSuccess) e1 = hipMemcpy(g->node_class.data(), dcls, (size_t)P, hipMemcpyDeviceToHost);
        if (e1 == hipSuccess) e1 = hipMemcpy(hmax, dmax, sizeof hmax, hipMemcpyDeviceToHost);
        (void)hipFree(dcls);
        (void)hipFree(dmax);
        if (lrc || e1 != hipSuccess) return fail(NIN_EHIP, "node classification: %s", hipGetErrorString(e1));
        for (int c = 0; c < kGlsClasses; ++c) {
            need_max[c] = (int64_t)hmax[3 * c]; rows_max[c] = (int64_t)hmax[3 * c + 1]; cols_max[c] = (int64_t)hmax[3 * c + 2];
        }
    }
    for (int64_t p = 0; p < P; ++p) {
        const uint8_t c = g->node_class[p];
        if (c == 255) hex8_list.push_back((int32_t)p);
        else if (c >= 252 && c <= 254) mfw_list[254 - c].push_back((int32_t)p);
        else if (c >= 249 && c <= 251) small_list[c - 249].push_back((int32_t)p);
        else if (c == 248) quad4_list.push_back((int32_t)p);
        else if (c >= 243 && c <= 247) mfx_list[c - 243].push_back((int32_t)p);
        else if (c == 242) mfx_list[5].push_back((int32_t)p);
        else if (c == 240) mfx_list[6].push_back((int32_t)p);
        else if (c == 239) mfx_list[7].push_back((int32_t)p);
        else if (c == 241) mfg_list.push_back((int32_t)p);
        else lists[c].push_back((int32_t)p);
    }
    for (int c = 0; c < kGlsClasses; ++c) {
        auto &k = d.gls[c];
        k.count = (int32_t)lists[c].size();
        k.lds_bytes = c == kGlsClasses - 1 ? 0 : (int32_t)need_max[c];
        k.rows_per_lane = (int32_t)std::max<int64_t>(1, (rows_max[c] + 63) / 64);
        k.max_rows = (int32_t)rows_max[c];
        k.max_cols = (int32_t)cols_max[c];
        k.waves = gls_class_waves(c);
        k.col_slots = (int32_t)std::max<int64_t>(1, (cols_max[c] + 63) / 64);
        const int32_t *lp = nullptr;
        if (k.count && (rc = dev_upload(d, &lp, lists[c]))) return rc;
        k.nodes = const_cast<int32_t *>(lp);
    }
    {   // every node the cube-node kernel does not take, ascending: what the fused apply leaves to the list kernel
        std::vector<int32_t> rest;
        rest.reserve((size_t)P - hex8_list.size());
        for (int64_t p = 0; p < P; ++p)
            if (g->node_class[p] != 255) rest.push_back((int32_t)p);
        d.noncube_count = (int32_t)rest.size();
        const int32_t *lp = nullptr;
        if (d.noncube_count && (rc = dev_upload(d, &lp, rest))) return rc;
        d.noncube_nodes = lp;
        d.noncube_nodes_ready = true;
    }
    // interpolate()'s pipeline: its pieces' node boundaries (multiples of 64 nodes)
    {
        constexpr int K = DeviceGrid::kE2eChunks;
        for (int k = 0; k <= K; ++k) d.chunk_node[k] = k == K ? (int32_t)P : (int32_t)((P * k / K) & ~(int64_t)63);
    }
    // (NIN_GLS_LOCALITY_ORDER: "s<rows>" = strips of that many mesh rows, the default s16; "m" = Morton order; "off" = node order)
    const char *lo_env = getenv("NIN_GLS_LOCALITY_ORDER");
    const bool locality = !(lo_env && (lo_env[0] == 'o' || lo_env[0] == '0'));
    {
        d.hex8.count = (int32_t)hex8_list.size();
        const int32_t *lp = nullptr;
        if (d.hex8.count && (rc = dev_upload(d, &lp, hex8_list))) return rc;
        d.hex8.nodes = const_cast<int32_t *>(lp);
        if (d.hex8.count && locality && (rc = locality_order(d, d.hex8.nodes, d.hex8.count, d.chunk_node))) return rc;
        if (d.hex8.count) {   // lane descriptors of the multifrontal kernel, one 16-byte record per list entry
            if ((rc = dev_alloc(d, &d.hex8_desc, (size_t)d.hex8.count * 4))) return rc;
            if (launch_hex8_desc(d.v, d.hex8.nodes, d.hex8.count, d.hex8_desc, nullptr)) return fail(NIN_EHIP, "hex8 descriptor kernel");
        }
    }
    for (int i = 0; i < 3; ++i) {
        d.mfw[i].count = (int32_t)mfw_list[i].size();
        const int32_t *lp = nullptr;
        if (d.mfw[i].count && (rc = dev_upload(d, &lp, mfw_list[i]))) return rc;
        d.mfw[i].nodes = const_cast<int32_t *>(lp);
        if (d.mfw[i].count) {   // descriptors of the one-wavefront multifrontal kernel, kMfwDescWords (40) words per list entry
            if ((rc = dev_alloc(d, &d.mfw_desc[i], (size_t)d.mfw[i].count * kMfwDescWords))) return rc;
            if (launch_mfw_desc(d.v, d.mfw[i].nodes, d.mfw[i].count, d.mfw_desc[i], nullptr)) return fail(NIN_EHIP, "mfw descriptor kernel");
        }
    }
    for (int i = 0; i < DeviceGrid::kMfxLists; ++i) {
        d.mfx[i].count = (int32_t)mfx_list[i].size();
        const int32_t *lp = nullptr;
        if (d.mfx[i].count && (rc = dev_upload(d, &lp, mfx_list[i]))) return rc;
        d.mfx[i].nodes = const_cast<int32_t *>(lp);
        if (d.mfx[i].count) {   // descriptors of the wide multifrontal kernel, kMfxDescWords (56) words per list entry
            if ((rc = dev_alloc(d, &d.mfx_desc[i], (size_t)d.mfx[i].count * kMfxDescWords))) return rc;
            if (launch_mfx_desc(d.v, d.mfx[i].nodes, d.mfx[i].count, d.mfx_desc[i], nullptr)) return fail(NIN_EHIP, "mfx descriptor kernel");
        }
    }
    {
        d.mfg.count = (int32_t)mfg_list.size();
        const int32_t *lp = nullptr;
        if (d.mfg.count && (rc = dev_upload(d, &lp, mfg_list))) return rc;
        d.mfg.nodes = const_cast<int32_t *>(lp);
        if (d.mfg.count) {   // descriptors, kMfgDescWords (124) words per list entry, and one slot of tiles per resident wavefront
            if ((rc = dev_alloc(d, &d.mfg_desc, (size_t)d.mfg.count * kMfgDescWords))) return rc;
            if (launch_mfg_desc(d.v, d.mfg.nodes, d.mfg.count, d.mfg_desc, nullptr)) return fail(NIN_EHIP, "mfg descriptor kernel");
            d.mfg_slots = std::min<int32_t>(d.mfg.count, kMfgResidentWaves);
            if ((rc = dev_alloc(d, &d.mfg_tiles, (size_t)d.mfg_slots * kMfgSlotDoubles))) return rc;
        }
    }
    for (int i = 0; i < 3; ++i) {
        d.small[i].count = (int32_t)small_list[i].size();
        const int32_t *lp = nullptr;
        if (d.small[i].count && (rc = dev_upload(d, &lp, small_list[i]))) return rc;
        d.small[i].nodes = const_cast<int32_t *>(lp);
    }
    {
        d.quad4.count = (int32_t)quad4_list.size();
        const int32_t *lp = nullptr;
        if (d.quad4.count && (rc = dev_upload(d, &lp, quad4_list))) return rc;
        d.quad4.nodes = const_cast<int32_t *>(lp);
        if (d.quad4.count) {
            if ((rc = dev_alloc(d, &d.quad4_desc, (size_t)d.quad4.count * 2))) return rc;
            if (launch_quad4_desc(d.v, d.quad4.nodes, d.quad4.count, d.quad4_desc, nullptr)) return fail(NIN_EHIP, "quad4 descriptor kernel");
        }
    }
    d.gls_too_large = rows_max[kGlsClasses - 1] > 1024;
    if ((rc = dev_alloc(d, &d.gls_queue, (size_t)kGlsQueueInts))) return rc;
    if (d.gls[kGlsClasses - 1].count) {
        d.gls_scratch_slots = std::min<int32_t>(d.gls[kGlsClasses - 1].count, 512);   // one per resident team (kernels_gls.hip)
        d.gls_scratch_stride = need_max[kGlsClasses - 1] / 8;
        if ((rc = dev_alloc(d, &d.gls_scratch, (size_t)d.gls_scratch_slots * d.gls_scratch_stride))) return rc;
    }
    // interpolate()'s pipeline: where the pieces' boundaries fall in every list (ascending here on the host; a list in locality
    // order on the device is permuted inside the pieces only)
    {
        constexpr int K = DeviceGrid::kE2eChunks;
        auto cut = [&](int li, const std::vector<int32_t> &v) {
            for (int k = 0; k <= K; ++k)
                d.chunk_off[li][k] = (int32_t)(std::lower_bound(v.begin(), v.end(), d.chunk_node[k]) - v.begin());
        };
        for (int c = 0; c < kGlsClasses; ++c) cut(c, lists[c]);
        cut(kGlsClasses, hex8_list);
        for (int i = 0; i < 3; ++i) cut(kGlsClasses + 1 + i, mfw_list[i]);
        for (int i = 0; i < 3; ++i) cut(kGlsClasses + 4 + i, small_list[i]);
        cut(kGlsClasses + 7, quad4_list);
        for (int i = 0; i < DeviceGrid::kMfxLists; ++i) cut(kGlsClasses + 8 + i, mfx_list[i]);
        cut(kGlsClasses + 8 + DeviceGrid::kMfxLists, mfg_list);
        const char *mn = getenv("NIN_E2E_MIN_NODES");                                // (tests: the pipeline on small meshes too)
        d.chunkable = P >= (mn ? atoll(mn) : 64 * 1024) && P >= 64 * K && getenv("NIN_E2E_NO_PIPELINE") == nullptr;   // small meshes: one piece
    }
    // a single class holding every node in order needs no list: the kernel walks 0..P-1 directly
    for (int c = 0; c < kGlsClasses; ++c)
        if (d.gls[c].count == P) { d.gls[c].nodes = nullptr; d.chunkable = false; }
    HIP_TRY(hipDeviceSynchronize());
    return NIN_OK;
}

int nin_fields_set(nin_grid *g, const double *permeability, const double *diff_mag, const double *neumann_flag,
                   const double *neumann_val) {
    (void)neumann_val;  // only feeds the Neumann RHS column, which gls.pyx:464-472 never reads back
    if (!g) return fail(NIN_EINVAL, "NULL grid");
    if (g->d.device < 0 || g->d.prebuilt) return fail(NIN_ENODEVICE, "grid is not on a device (call nin_grid_to_device first)");
    if (g->h.ensure(A_BPOINTS)) return fail(NIN_EHIP, "mirroring boundary_points failed");
    if (!neumann_flag) return fail(NIN_EINVAL, "neumann_flag is required by every method");
    HIP_TRY(hipSetDevice(g->d.device));
    HostGrid &h = g->h;
    DeviceGrid &d = g->d;
    const int64_t P = h.n_points, E = h.n_elems;
    // packed by the host's OpenMP team (pack_host.cpp) into a page-locked staging buffer the grid keeps: 82 MB of float64
    // flags in, 10 MB out at 10 M nodes -- the serial loop + pageable copy this replaces took ~15 ms of every interpolate()
    if (!d.flag_staging) HIP_TRY(hipHostMalloc((void **)&d.flag_staging, (size_t)std::max<int64_t>(P, 1), hipHostMallocDefault));
    pack_node_flags(neumann_flag, h.boundary_points.data(), P, d.flag_staging);
    HIP_TRY(hipMemcpy(const_cast<uint8_t *>(d.v.flags), d.flag_staging, (size_t)P, hipMemcpyHostToDevice));
    if (permeability && diff_mag) {   // NULL keeps what is resident (0.8 GB at 10 M cells: callers upload it once per mesh)
        HIP_TRY(hipMemcpy(const_cast<double *>(d.v.perm), permeability, (size_t)E * 9 * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(const_cast<double *>(d.v.diff_mag), diff_mag, (size_t)E * 8, hipMemcpyHostToDevice));
        d.have_perm = true;
    }
    d.fields_set = true;
    return NIN_OK;
}

// cube nodes: the multifrontal kernel
static int launch_hex8(DeviceGrid &d, const int32_t *nodes, const int32_t *desc, int32_t count, int add_neumann,
                       double *out, double *nws, hipStream_t stream) {
    return launch_gls_hex8mf(d.v, nodes, desc, count, add_neumann, out, nws, d.gls_queue, stream);
}

// the one-wavefront multifrontal kernel, kind 0 / 1 / 2 (work counters: ints 5, 6, 7 of the queue block)
static int launch_mfw(DeviceGrid &d, const int32_t *nodes, const uint32_t *desc, int32_t count, int kind, int add_neumann,
                      double *out, double *nws, hipStream_t stream) {
    return launch_gls_mfw(d.v, nodes, desc, count, kind, add_neumann, out, nws, d.gls_queue + 5 + kind, stream);
}

// one GLS size class: the block kernel with the system in LDS, or the wave kernel on global scratch
static int launch_class(DeviceGrid &d, int c, const int32_t *nodes, int32_t count, int add_neumann, double *out,
                        double *nws, hipStream_t stream) {
    const auto &k = d.gls[c];
    if (c < kGlsClasses - 1)   // work counter: ints 1..4 of the queue block (the hex8 kernel uses 0, 16, 32, ...)
        return launch_gls_block(d.v, nodes, count, k.waves, k.col_slots, k.lds_bytes, add_neumann, out, nws,
                                d.gls_queue + 1 + c, stream);
    return launch_gls_class(d.v, nodes, count, 0, k.rows_per_lane, add_neumann, out, nws, d.gls_scratch,
                            d.gls_scratch_stride, d.gls_scratch_slots, stream);
}

// The long poles first, on the side stream: entries [b, b + n) of the global-scratch class's list and [bg, bg + ng) of kernels_gls_mfg.hip's
// (a node on global-memory tiles takes ~0.5 ms on a wavefront of its own).  Ordered behind everything enqueued on `stream` so far (the
// zeroed work counters, the caller's buffers) and joined by gls_side_end.  NIN_GLS_NO_SIDE_STREAM=1: off.
static int gls_side_begin(DeviceGrid &d, int add_neumann, double *out, double *nws, hipStream_t stream, int32_t b, int32_t n, int32_t bg, int32_t ng) {
    d.side_pending = false;
    if ((n <= 0 && ng <= 0) || getenv("NIN_GLS_NO_SIDE_STREAM") != nullptr) return 0;
    if (!d.side_stream) {
        hipStream_t s = nullptr;
        hipEvent_t a = nullptr, e = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
            return -3;
        d.side_stream = s; d.ev_fork = a; d.ev_join = e;
    }
    hipStream_t side = static_cast<hipStream_t>(d.side_stream);
    if (hipEventRecord(static_cast<hipEvent_t>(d.ev_fork), stream) != hipSuccess || hipStreamWaitEvent(side, static_cast<hipEvent_t>(d.ev_fork), 0) != hipSuccess)
        return -3;
    const auto &k = d.gls[kGlsClasses - 1];
    int rc = 0;
    if (ng > 0)
        rc = launch_gls_mfg(d.v, d.mfg.nodes + bg, d.mfg_desc + (size_t)kMfgDescWords * bg, ng, add_neumann, out, nws, d.gls_queue + 16, d.mfg_tiles, d.mfg_slots, side);
    if (!rc && n > 0)
        rc = launch_gls_class(d.v, k.nodes ? k.nodes + b : nullptr, n, 0, k.rows_per_lane, add_neumann, out, nws, d.gls_scratch, d.gls_scratch_stride,
                              d.gls_scratch_slots, side);
    if (!rc && hipEventRecord(static_cast<hipEvent_t>(d.ev_join), side) != hipSuccess) rc = -3;
    d.side_pending = rc == 0;
    return rc;
}
static int gls_side_end(DeviceGrid &d, hipStream_t stream) {
    if (!d.side_pending) return 0;
    d.side_pending = false;
    return hipStreamWaitEvent(stream, static_cast<hipEvent_t>(d.ev_join), 0) == hipSuccess ? 0 : -3;
}

// NIN_GLS_ONLY=<k>: launch only kernel k of the plan, numbered as nin_gls_plan counts them (measurement: bench.py times the
// kernels of a plan one by one; read at every launch)
static int gls_only() {
    const char *e = getenv("NIN_GLS_ONLY");
    return e && *e ? atoi(e) : -1;
}

// every GLS kernel of the launch plan but the cube-node kernel, all their nodes (the work counters are zeroed by the caller; the
// global-scratch class is left out if gls_side_begin took it)
static int launch_gls_but_cube(DeviceGrid &d, int add_neumann, double *out, double *nws, hipStream_t stream) {
    int rc = 0;
    const int only = gls_only();
    auto on = [&](int k) { return only < 0 || only == k; };
    for (int i = 0; i < 3 && !rc; ++i)
        if (on(6 + i)) rc = launch_mfw(d, d.mfw[i].nodes, d.mfw_desc[i], d.mfw[i].count, i, add_neumann, out, nws, stream);
    for (int i = 0; i < 3 && !rc; ++i)
        if (on(9 + i)) rc = launch_gls_small(d.v, d.small[i].nodes, d.small[i].count, i, add_neumann, out, nws, stream);
    if (!rc && on(12)) rc = launch_gls_quad4(d.v, d.quad4.nodes, d.quad4_desc, d.quad4.count, add_neumann, out, nws, stream);
    for (int i = 0; i < DeviceGrid::kMfxLists && !rc; ++i)   // (work counters: ints 8 .. 15; lists 6 and 7 -- the small class, (7, 12) -- are kernels 20 and 21 of the plan)
        if (on(i < 6 ? 13 + i : 14 + i)) rc = launch_gls_mfx(d.v, d.mfx[i].nodes, d.mfx_desc[i], d.mfx[i].count, i, add_neumann, out, nws, d.gls_queue + 8 + i, stream);
    if (!rc && on(19) && !d.side_pending)   // (work counter: int 16)
        rc = launch_gls_mfg(d.v, d.mfg.nodes, d.mfg_desc, d.mfg.count, add_neumann, out, nws, d.gls_queue + 16, d.mfg_tiles, d.mfg_slots, stream);
    for (int c = 0; c < kGlsClasses && !rc; ++c) {
        if ((c == kGlsClasses - 1 && d.side_pending) || !on(c)) continue;
        rc = launch_class(d, c, d.gls[c].nodes, d.gls[c].count, add_neumann, out, nws, stream);
    }
    if (!rc) rc = gls_side_end(d, stream);
    return rc;
}

int nin_weights_device(nin_grid *g, int method, const int64_t *targets, int64_t n_targets, int add_neumann,
                       double *dev_csr_data, double *dev_neumann_ws, void *stream_) {
    if (!g || !dev_csr_data || !dev_neumann_ws) return fail(NIN_EINVAL, "NULL argument");
    DeviceGrid &d = g->d;
    if (d.device < 0 || d.prebuilt) return fail(NIN_ENODEVICE, "grid is not on a device: the weight kernels are HIP only");
    if (!d.fields_set) return fail(NIN_ESTATE, "nin_fields_set has not been called");
    if (method != NIN_METHOD_GLS && method != NIN_METHOD_IDW && method != NIN_METHOD_LS)
        return fail(NIN_EINVAL, "unknown method %d", method);
    if (method == NIN_METHOD_GLS && !d.have_perm) return fail(NIN_ESTATE, "GLS needs permeability and diff_mag");
    if (method == NIN_METHOD_GLS && d.gls_too_large)
        return fail(NIN_ERANGE, "a node's GLS system has more than 1024 rows (more than ~100 cells around one node): beyond the fallback kernel");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(d.device));
    const int64_t P = g->h.n_points;
    const bool all = targets == nullptr;
    if (!all && n_targets < 0) return fail(NIN_EINVAL, "negative n_targets");
    int rc = 0;
    if (all) {
        if (method == NIN_METHOD_IDW) rc = launch_idw(d.v, nullptr, (int32_t)P, (int32_t)g->h.mx_elems_per_point, d.nnz_e, dev_csr_data, dev_neumann_ws, stream);
        else if (method == NIN_METHOD_LS) rc = launch_ls(d.v, nullptr, (int32_t)P, (int32_t)g->h.mx_elems_per_point, d.nnz_e, dev_csr_data, dev_neumann_ws, stream);
        else {
            HIP_TRY(hipMemsetAsync(d.gls_queue, 0, kGlsQueueInts * sizeof(int32_t), stream));   // the launches' work counters
            const int only = gls_only();
            if (only < 0) rc = gls_side_begin(d, add_neumann, dev_csr_data, dev_neumann_ws, stream, 0, d.gls[kGlsClasses - 1].count, 0, d.mfg.count);
            if (!rc && (only < 0 || only == 5))
                rc = launch_hex8(d, d.hex8.nodes, d.hex8_desc, d.hex8.count, add_neumann, dev_csr_data, dev_neumann_ws, stream);
            if (!rc) rc = launch_gls_but_cube(d, add_neumann, dev_csr_data, dev_neumann_ws, stream);
        }
        if (rc) return fail(rc, "kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
        return NIN_OK;
    }
    // explicit target list: outputs of every other node are zero
    for (int64_t i = 0; i < n_targets; ++i)
        if (targets[i] < 0 || targets[i] >= P) return fail(NIN_EINVAL, "target %lld out of range", (long long)targets[i]);
    HIP_TRY(hipMemsetAsync(dev_csr_data, 0, (size_t)d.nnz_e * 8, stream));
    HIP_TRY(hipMemsetAsync(dev_neumann_ws, 0, (size_t)P * 8, stream));
    if (n_targets == 0) return NIN_OK;
    std::vector<std::vector<int32_t>> lists(method == NIN_METHOD_GLS ? kGlsClasses + 8 + DeviceGrid::kMfxLists + 1 : 1);
    for (int64_t i = 0; i < n_targets; ++i) {
        int c = method == NIN_METHOD_GLS ? g->node_class[targets[i]] : 0;
        if (c == 255) c = kGlsClasses;   // the hex8 kernel's class
        else if (c >= 252 && c <= 254) c = kGlsClasses + 1 + (254 - c);   // the one-wavefront multifrontal kernel, kind 0 / 1 / 2
        else if (c >= 249 && c <= 251) c = kGlsClasses + 4 + (c - 249);   // the small-node kernel, kind 0 / 1 / 2
        else if (c == 248) c = kGlsClasses + 7;                           // the quad-node kernel
        else if (c >= 243 && c <= 247) c = kGlsClasses + 8 + (c - 243);   // the wide multifrontal kernel, by size class
        else if (c == 242) c = kGlsClasses + 8 + 5;                       // ... its boundary nodes
        else if (c == 240) c = kGlsClasses + 8 + 6;                       // ... its small interior class
        else if (c == 239) c = kGlsClasses + 8 + 7;                       // ... its class (7, 12)
        else if (c == 241) c = kGlsClasses + 8 + DeviceGrid::kMfxLists;   // the multifrontal kernel on global-memory tiles
        lists[c].push_back((int32_t)targets[i]);
    }
    // one device buffer for all class lists, filled before the first launch: a per-class allocate / copy / free
    // cycle lets the allocator hand the same memory to the next list while the previous kernel still reads it
    // (the pageable copy is not ordered behind that kernel)
    std::vector<int32_t> flat;
    std::vector<size_t> first(lists.size() + 1, 0);
    for (size_t c = 0; c < lists.size(); ++c) {
        first[c] = flat.size();
        flat.insert(flat.end(), lists[c].begin(), lists[c].end());
    }
    first[lists.size()] = flat.size();
    int32_t *dl0 = nullptr;
    const size_t n_hex8 = method == NIN_METHOD_GLS ? lists[kGlsClasses].size() : 0;   // + 4 descriptor words per cube node
    const size_t n_mfw = method == NIN_METHOD_GLS ? lists[kGlsClasses + 1].size() + lists[kGlsClasses + 2].size() + lists[kGlsClasses + 3].size() : 0;   // + kMfwDescWords per node of the multifrontal kernel (the three lists are adjacent)
    const size_t n_quad4 = method == NIN_METHOD_GLS ? lists[kGlsClasses + 7].size() : 0;   // + 2 descriptor words per quad node
    size_t n_mfx = 0;                                                                      // + kMfxDescWords per node of the wide multifrontal kernel (its lists are adjacent)
    if (method == NIN_METHOD_GLS)
        for (int i = 0; i < DeviceGrid::kMfxLists; ++i) n_mfx += lists[kGlsClasses + 8 + i].size();
    const size_t n_mfg = method == NIN_METHOD_GLS ? lists[kGlsClasses + 8 + DeviceGrid::kMfxLists].size() : 0;   // + kMfgDescWords per node of kernels_gls_mfg.hip
    HIP_TRY(hipMalloc((void **)&dl0, (flat.size() + 4 * n_hex8 + kMfwDescWords * n_mfw + 2 * n_quad4 + kMfxDescWords * n_mfx + kMfgDescWords * n_mfg) * 4));
    int32_t *ddesc = dl0 + flat.size();
    uint32_t *dmfw = reinterpret_cast<uint32_t *>(ddesc + 4 * n_hex8);
    int32_t *dquad = reinterpret_cast<int32_t *>(dmfw + kMfwDescWords * n_mfw);
    uint32_t *dmfx = reinterpret_cast<uint32_t *>(dquad + 2 * n_quad4);
    uint32_t *dmfg = dmfx + kMfxDescWords * n_mfx;
    const hipError_t cp = hipMemcpy(dl0, flat.data(), flat.size() * 4, hipMemcpyHostToDevice);
    if (cp != hipSuccess) { (void)hipFree(dl0); return fail(NIN_EHIP, "hipMemcpy: %s", hipGetErrorString(cp)); }
    if (n_hex8 && launch_hex8_desc(d.v, dl0 + first[kGlsClasses], (int32_t)n_hex8, ddesc, stream)) {
        (void)hipFree(dl0);
        return fail(NIN_EHIP, "hex8 descriptor kernel");
    }
    if (n_mfw && launch_mfw_desc(d.v, dl0 + first[kGlsClasses + 1], (int32_t)n_mfw, dmfw, stream)) {
        (void)hipFree(dl0);
        return fail(NIN_EHIP, "mfw descriptor kernel");
    }
    if (n_quad4 && launch_quad4_desc(d.v, dl0 + first[kGlsClasses + 7], (int32_t)n_quad4, dquad, stream)) {
        (void)hipFree(dl0);
        return fail(NIN_EHIP, "quad4 descriptor kernel");
    }
    if (n_mfx && launch_mfx_desc(d.v, dl0 + first[kGlsClasses + 8], (int32_t)n_mfx, dmfx, stream)) {
        (void)hipFree(dl0);
        return fail(NIN_EHIP, "mfx descriptor kernel");
    }
    if (n_mfg && launch_mfg_desc(d.v, dl0 + first[kGlsClasses + 8 + DeviceGrid::kMfxLists], (int32_t)n_mfg, dmfg, stream)) {
        (void)hipFree(dl0);
        return fail(NIN_EHIP, "mfg descriptor kernel");
    }
    if (method == NIN_METHOD_GLS) {
        const hipError_t qe = hipMemsetAsync(d.gls_queue, 0, kGlsQueueInts * sizeof(int32_t), stream);
        if (qe != hipSuccess) { (void)hipFree(dl0); return fail(NIN_EHIP, "hipMemsetAsync: %s", hipGetErrorString(qe)); }
    }
    for (size_t c = 0; c < lists.size() && !rc; ++c) {
        if (lists[c].empty()) continue;
        const int32_t *dl = dl0 + first[c];
        const int32_t cnt = (int32_t)lists[c].size();
        if (method == NIN_METHOD_IDW) rc = launch_idw(d.v, dl, cnt, 0, 0, dev_csr_data, dev_neumann_ws, stream);
        else if (method == NIN_METHOD_LS) rc = launch_ls(d.v, dl, cnt, 0, 0, dev_csr_data, dev_neumann_ws, stream);
        else if ((int)c == kGlsClasses) rc = launch_hex8(d, dl, ddesc, cnt, add_neumann, dev_csr_data, dev_neumann_ws, stream);
        else if ((int)c == kGlsClasses + 8 + DeviceGrid::kMfxLists)
            rc = launch_gls_mfg(d.v, dl, dmfg, cnt, add_neumann, dev_csr_data, dev_neumann_ws, d.gls_queue + 16, d.mfg_tiles, d.mfg_slots, stream);
        else if ((int)c >= kGlsClasses + 8)
            rc = launch_gls_mfx(d.v, dl, dmfx + kMfxDescWords * (first[c] - first[kGlsClasses + 8]), cnt, (int)c - kGlsClasses - 8, add_neumann, dev_csr_data,
                                dev_neumann_ws, d.gls_queue + 8 + ((int)c - kGlsClasses - 8), stream);
        else if ((int)c == kGlsClasses + 7) rc = launch_gls_quad4(d.v, dl, dquad, cnt, add_neumann, dev_csr_data, dev_neumann_ws, stream);
        else if ((int)c >= kGlsClasses + 4) rc = launch_gls_small(d.v, dl, cnt, (int)c - kGlsClasses - 4, add_neumann, dev_csr_data, dev_neumann_ws, stream);
        else if ((int)c > kGlsClasses)
            rc = launch_mfw(d, dl, dmfw + kMfwDescWords * (first[c] - first[kGlsClasses + 1]), cnt, (int)c - kGlsClasses - 1, add_neumann,
                            dev_csr_data, dev_neumann_ws, stream);
        else rc = launch_class(d, (int)c, dl, cnt, add_neumann, dev_csr_data, dev_neumann_ws, stream);
    }
    const hipError_t sy = hipStreamSynchronize(stream);   // the lists must outlive the kernels
    (void)hipFree(dl0);
    if (sy != hipSuccess) return fail(NIN_EHIP, "target kernels: %s", hipGetErrorString(sy));
    if (rc) return fail(rc, "kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    return NIN_OK;
}

int nin_weights_host(nin_grid *g, int method, const int64_t *targets, int64_t n_targets, int add_neumann,
                     double *csr_data, double *neumann_ws) {
    if (!g || !csr_data || !neumann_ws) return fail(NIN_EINVAL, "NULL argument");
    DeviceGrid &d = g->d;
    if (d.device < 0) return fail(NIN_ENODEVICE, "grid is not on a device: the weight kernels are HIP only");
    HIP_TRY(hipSetDevice(d.device));
    const size_t nb = (size_t)std::max<int64_t>(d.nnz_e, 1) * 8, pb = (size_t)g->h.n_points * 8;
    double *dd = nullptr, *dn = nullptr;
    HIP_TRY(hipMalloc((void **)&dd, nb));
    hipError_t e = hipMalloc((void **)&dn, pb);
    if (e != hipSuccess) { (void)hipFree(dd); return fail(NIN_ENOMEM, "hipMalloc: %s", hipGetErrorString(e)); }
    int rc = nin_weights_device(g, method, targets, n_targets, add_neumann, dd, dn, nullptr);
    if (!rc) {
        e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(csr_data, dd, (size_t)d.nnz_e * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(neumann_ws, dn, pb, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(NIN_EHIP, "weights kernel / copy back: %s", hipGetErrorString(e));
    }
    (void)hipFree(dd);
    (void)hipFree(dn);
    return rc;
}

namespace {

struct Laps {   // NIN_TIMING=1: host-side laps of one call on stderr
    bool on = getenv("NIN_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[nin_e2e] %-22s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

int e2e_streams(DeviceGrid &d) {
    if (!d.copy_stream) {
        hipStream_t s;
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        d.copy_stream = s;
        hipEvent_t a, b;
        HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        d.ev_weights = a;
        d.ev_scan = b;
    }
    return 0;
}

// The finish of interpolator.pyx:622-624 with the transfers overlapped: `weights_ready` (an event on `stream`, or null)
// marks the weights + neumann_ws written; dev_nws / neumann_ws (optional) ride the copy stream under the count / scan /
// compaction kernels, as does indptr; indices and data follow in pieces, each as soon as the copy engine is free.
int csr_compact_pipelined(nin_grid *g, const double *dev_csr_data, const double *dev_nws, int32_t *indptr, int32_t *indices,
                          double *data, int64_t *nnz_out, double *neumann_ws, hipStream_t stream) {
    DeviceGrid &d = g->d;
    const int64_t P = g->h.n_points;
    size_t tmp_bytes = 0;
    int rc = NIN_OK;
    Laps L;
    if ((rc = e2e_streams(d))) return rc;
    hipStream_t cs = static_cast<hipStream_t>(d.copy_stream);
    hipEvent_t ev_w = static_cast<hipEvent_t>(d.ev_weights), ev_s = static_cast<hipEvent_t>(d.ev_scan);
#define TRY_C(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(NIN_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)
    // scratch of the compaction: owned by the grid, allocated on first use (sized by the mesh: P + 1 counters, nnz_e entries)
    if (!d.e2e_cnt && (rc = dev_alloc(d, &d.e2e_cnt, (size_t)(P + 1)))) return rc;
    if (!d.e2e_ptr && (rc = dev_alloc(d, &d.e2e_ptr, (size_t)(P + 1)))) return rc;
    int32_t *cnt = d.e2e_cnt, *ptr = d.e2e_ptr;
    if (dev_nws && neumann_ws) {   // 8 B per node: leaves as soon as the weight kernels are done
        TRY_C(hipEventRecord(ev_w, stream));
        TRY_C(hipStreamWaitEvent(cs, ev_w, 0));
        TRY_C(hipMemcpyAsync(neumann_ws, dev_nws, (size_t)P * 8, hipMemcpyDeviceToHost, cs));
    }
    TRY_C(hipMemsetAsync(cnt, 0, (size_t)(P + 1) * 4, stream));
    if ((rc = launch_row_nnz(d.v, dev_csr_data, cnt, stream))) return fail(rc, "launch failed");
    TRY_C(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, cnt, ptr, (int)(P + 1), stream));
    if (tmp_bytes > d.e2e_tmp_bytes) {
        char *t = nullptr;
        if ((rc = dev_alloc(d, &t, std::max<size_t>(tmp_bytes, 16)))) return rc;   // (a smaller one stays in d.allocs until the grid goes)
        d.e2e_tmp = t;
        d.e2e_tmp_bytes = tmp_bytes;
    }
    TRY_C(hipcub::DeviceScan::ExclusiveSum(d.e2e_tmp, tmp_bytes, cnt, ptr, (int)(P + 1), stream));
    TRY_C(hipEventRecord(ev_s, stream));
    const size_t cap = (size_t)std::max<int64_t>(d.nnz_e, 1);
    const bool want_entries = indices && data;
    if (want_entries) {   // the compaction does not need the count on the host: it starts right behind the scan
        if (!d.e2e_indices && (rc = dev_alloc(d, &d.e2e_indices, cap))) return rc;
        if (!d.e2e_data && (rc = dev_alloc(d, &d.e2e_data, cap))) return rc;
        if ((rc = launch_compact(d.v, dev_csr_data, ptr, d.e2e_indices, d.e2e_data, stream))) return fail(rc, "launch failed");
    }
    TRY_C(hipStreamWaitEvent(cs, ev_s, 0));
    TRY_C(hipMemcpyAsync(indptr, ptr, (size_t)(P + 1) * 4, hipMemcpyDeviceToHost, cs));   // under the compaction kernel
    TRY_C(hipStreamSynchronize(cs));
    L.lap("weights+count+scan");
    const int64_t nnz = indptr[P];
    *nnz_out = nnz;
    if (nnz > 0 && want_entries) {
        TRY_C(hipStreamSynchronize(stream));
        L.lap("compaction");
        // two copy queues: the index and the value stream each keep a DMA engine busy
        TRY_C(hipMemcpyAsync(indices, d.e2e_indices, (size_t)nnz * 4, hipMemcpyDeviceToHost, stream));
        TRY_C(hipMemcpyAsync(data, d.e2e_data, (size_t)nnz * 8, hipMemcpyDeviceToHost, cs));
        TRY_C(hipStreamSynchronize(cs));
        TRY_C(hipStreamSynchronize(stream));
        L.lap("D2H indices+data");
    }
#undef TRY_C
    return NIN_OK;
}

}  // namespace

namespace {

// The weight kernels for the nodes of chunk k only (all nodes of the chunk, add_neumann fused): sub-ranges of the launch plan's lists.
int weights_chunk(nin_grid *g, int method, int k, double *out, double *nws, hipStream_t stream) {
    DeviceGrid &d = g->d;
    const int32_t P = (int32_t)g->h.n_points;
    if (method != NIN_METHOD_GLS)
        return launch_rows_range(d.v, method == NIN_METHOD_LS ? 1 : 0, P, d.chunk_node[k], d.chunk_node[k + 1],
                                 (int32_t)g->h.mx_elems_per_point, d.nnz_e, out, nws, stream);
    HIP_TRY(hipMemsetAsync(d.gls_queue, 0, kGlsQueueInts * sizeof(int32_t), stream));   // the launches' work counters
    constexpr int lg = kGlsClasses + 8 + DeviceGrid::kMfxLists;   // kernels_gls_mfg.hip's list
    int rc = gls_side_begin(d, 1, out, nws, stream, d.chunk_off[kGlsClasses - 1][k],
                            d.chunk_off[kGlsClasses - 1][k + 1] - d.chunk_off[kGlsClasses - 1][k], d.chunk_off[lg][k], d.chunk_off[lg][k + 1] - d.chunk_off[lg][k]);
    {
        const int32_t b = d.chunk_off[kGlsClasses][k], n = d.chunk_off[kGlsClasses][k + 1] - b;
        if (!rc && n > 0) rc = launch_gls_hex8mf(d.v, d.hex8.nodes + b, d.hex8_desc + 4 * (size_t)b, n, 1, out, nws, d.gls_queue, stream);
    }
    for (int i = 0; i < 3 && !rc; ++i) {
        const int32_t b = d.chunk_off[kGlsClasses + 1 + i][k], n = d.chunk_off[kGlsClasses + 1 + i][k + 1] - b;
        if (n > 0) rc = launch_gls_mfw(d.v, d.mfw[i].nodes + b, d.mfw_desc[i] + (size_t)kMfwDescWords * b, n, i, 1, out, nws,
                                       d.gls_queue + 5 + i, stream);
    }
    for (int i = 0; i < 3 && !rc; ++i) {
        const int32_t b = d.chunk_off[kGlsClasses + 4 + i][k], n = d.chunk_off[kGlsClasses + 4 + i][k + 1] - b;
        if (n > 0) rc = launch_gls_small(d.v, d.small[i].nodes + b, n, i, 1, out, nws, stream);
    }
    if (!rc) {
        const int32_t b = d.chunk_off[kGlsClasses + 7][k], n = d.chunk_off[kGlsClasses + 7][k + 1] - b;
        if (n > 0) rc = launch_gls_quad4(d.v, d.quad4.nodes + b, d.quad4_desc + 2 * (size_t)b, n, 1, out, nws, stream);
    }
    for (int i = 0; i < DeviceGrid::kMfxLists && !rc; ++i) {
        const int32_t b = d.chunk_off[kGlsClasses + 8 + i][k], n = d.chunk_off[kGlsClasses + 8 + i][k + 1] - b;
        if (n > 0) rc = launch_gls_mfx(d.v, d.mfx[i].nodes + b, d.mfx_desc[i] + (size_t)kMfxDescWords * b, n, i, 1, out, nws, d.gls_queue + 8 + i, stream);
    }
    if (!rc && !d.side_pending) {
        constexpr int li = kGlsClasses + 8 + DeviceGrid::kMfxLists;
        const int32_t b = d.chunk_off[li][k], n = d.chunk_off[li][k + 1] - b;
        if (n > 0) rc = launch_gls_mfg(d.v, d.mfg.nodes + b, d.mfg_desc + (size_t)kMfgDescWords * b, n, 1, out, nws, d.gls_queue + 16, d.mfg_tiles, d.mfg_slots, stream);
    }
    for (int c = 0; c < kGlsClasses && !rc; ++c) {
        const int32_t b = d.chunk_off[c][k], n = d.chunk_off[c][k + 1] - b;
        if (n <= 0) continue;
        const auto &kc = d.gls[c];
        if (c < kGlsClasses - 1)
            rc = launch_gls_block(d.v, kc.nodes + b, n, kc.waves, kc.col_slots, kc.lds_bytes, 1, out, nws, d.gls_queue + 1 + c, stream);
        else if (!d.side_pending)
            rc = launch_gls_class(d.v, kc.nodes + b, n, 0, kc.rows_per_lane, 1, out, nws, d.gls_scratch, d.gls_scratch_stride,
                                  d.gls_scratch_slots, stream);
    }
    if (!rc) rc = gls_side_end(d, stream);
    return rc ? fail(rc, "kernel launch failed: %s", hipGetErrorString(hipGetLastError())) : NIN_OK;
}

// interpolate() as a pipeline over kE2eChunks pieces of the node range: while piece k's surviving entries cross PCIe (the floor
// of this path: 0.95 GB at 57 GB/s = 16.7 ms at 10 M cells), piece k + 1 is computed, counted, scanned and compacted.  Per piece:
// weight kernels -> non-zeros per row -> exclusive scan seeded with the entries of the pieces before -> (4 bytes back: the
// running total) -> compaction -> its slices of indptr / indices / data leave on the copy queues.
int interpolate_chunked(nin_grid *g, int method, int32_t *indptr, int32_t *indices, double *data, int64_t *nnz_out,
                        double *neumann_ws) {
    DeviceGrid &d = g->d;
    constexpr int K = DeviceGrid::kE2eChunks;
    const int64_t P = g->h.n_points;
    Laps L;
    int rc = NIN_OK;
    if ((rc = e2e_streams(d))) return rc;
    hipStream_t cs = static_cast<hipStream_t>(d.copy_stream), stream = nullptr;
    hipEvent_t ev = static_cast<hipEvent_t>(d.ev_scan);
    // an error in the middle must not leave copies into the caller's buffers in flight
    auto drain = [&]() {
        (void)hipStreamSynchronize(cs);
        if (d.copy_stream2) (void)hipStreamSynchronize(static_cast<hipStream_t>(d.copy_stream2));
        (void)hipStreamSynchronize(stream);
    };
#define TRY_C(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { drain(); return fail(NIN_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)
    if (!d.e2e_cnt && (rc = dev_alloc(d, &d.e2e_cnt, (size_t)(P + 1)))) return rc;
    if (!d.e2e_ptr && (rc = dev_alloc(d, &d.e2e_ptr, (size_t)(P + 1)))) return rc;
    const size_t cap = (size_t)std::max<int64_t>(d.nnz_e, 1);
    if (!d.e2e_indices && (rc = dev_alloc(d, &d.e2e_indices, cap))) return rc;
    if (!d.e2e_data && (rc = dev_alloc(d, &d.e2e_data, cap))) return rc;
    size_t tmp_bytes = 0;
    TRY_C(hipcub::DeviceScan::ExclusiveScan(nullptr, tmp_bytes, d.e2e_cnt, d.e2e_ptr, hipcub::Sum(), (int32_t)0, (int)(P + 1), stream));
    if (tmp_bytes > d.e2e_tmp_bytes) {
        char *t = nullptr;
        if ((rc = dev_alloc(d, &t, std::max<size_t>(tmp_bytes, 16)))) return rc;
        d.e2e_tmp = t;
        d.e2e_tmp_bytes = tmp_bytes;
    }
    TRY_C(hipMemsetAsync(d.e2e_cnt, 0, (size_t)(P + 1) * 4, stream));
    int32_t base = 0;
    for (int k = 0; k < K; ++k) {
        const int32_t pb = d.chunk_node[k], pe = d.chunk_node[k + 1];
        if (pe <= pb) continue;
        if ((rc = weights_chunk(g, method, k, d.e2e_weights, d.e2e_nws, stream))) { drain(); return rc; }
        if ((rc = launch_row_nnz(d.v, d.e2e_weights, d.e2e_cnt, stream, pb, pe))) { drain(); return fail(rc, "launch failed"); }
        size_t tb = d.e2e_tmp_bytes;
        // ptr[pb .. pe] (one past the piece: the next piece's seed and, in the end, nnz); cnt[pe] is not part of that sum
        TRY_C(hipcub::DeviceScan::ExclusiveScan(d.e2e_tmp, tb, d.e2e_cnt + pb, d.e2e_ptr + pb, hipcub::Sum(), base, (int)(pe - pb + 1), stream));
        if ((rc = launch_compact(d.v, d.e2e_weights, d.e2e_ptr, d.e2e_indices, d.e2e_data, stream, pb, pe))) { drain(); return fail(rc, "launch failed"); }
        int32_t next_base = 0;
        TRY_C(hipMemcpyAsync(&next_base, d.e2e_ptr + pe, 4, hipMemcpyDeviceToHost, stream));
        TRY_C(hipEventRecord(ev, stream));
        TRY_C(hipStreamSynchronize(stream));             // (the piece is compacted; the copies of the piece before still run)
        const int64_t n_k = (int64_t)next_base - base;
        // this piece's slices: rows on the copy stream, entries split over two queues
        TRY_C(hipMemcpyAsync(indptr + pb, d.e2e_ptr + pb, (size_t)(pe - pb + (k == K - 1 ? 1 : 0)) * 4, hipMemcpyDeviceToHost, cs));
        TRY_C(hipMemcpyAsync(neumann_ws + pb, d.e2e_nws + pb, (size_t)(pe - pb) * 8, hipMemcpyDeviceToHost, cs));
        if (n_k > 0) {
            TRY_C(hipMemcpyAsync(data + base, d.e2e_data + base, (size_t)n_k * 8, hipMemcpyDeviceToHost, cs));
            if (!d.copy_stream2) {
                hipStream_t s2;
                TRY_C(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
                d.copy_stream2 = s2;
            }
            TRY_C(hipMemcpyAsync(indices + base, d.e2e_indices + base, (size_t)n_k * 4, hipMemcpyDeviceToHost,
                                 static_cast<hipStream_t>(d.copy_stream2)));
        }
        base = next_base;
        if (L.on) { char nm[32]; snprintf(nm, sizeof nm, "piece %d computed", k); L.lap(nm); }
    }
    TRY_C(hipStreamSynchronize(cs));
    if (d.copy_stream2) TRY_C(hipStreamSynchronize(static_cast<hipStream_t>(d.copy_stream2)));
    L.lap("copies drained");
    *nnz_out = base;
#undef TRY_C
    return NIN_OK;
}

}  // namespace

int nin_csr_compact_host(nin_grid *g, const double *dev_csr_data, int32_t *indptr, int32_t *indices, double *data,
                         int64_t *nnz_out, void *stream_) {
    if (!g || !dev_csr_data || !indptr || !nnz_out) return fail(NIN_EINVAL, "NULL argument");
    DeviceGrid &d = g->d;
    if (d.device < 0) return fail(NIN_ENODEVICE, "grid is not on a device");
    HIP_TRY(hipSetDevice(d.device));
    return csr_compact_pipelined(g, dev_csr_data, nullptr, indptr, indices, data, nnz_out, nullptr,
                                 static_cast<hipStream_t>(stream_));
}

int nin_interpolate_csr_host(nin_grid *g, int method, int32_t *indptr, int32_t *indices, double *data,
                             int64_t *nnz_out, double *neumann_ws) {
    if (!g || !indptr || !indices || !data || !nnz_out || !neumann_ws) return fail(NIN_EINVAL, "NULL argument");
    DeviceGrid &d = g->d;
    if (d.device < 0) return fail(NIN_ENODEVICE, "grid is not on a device: the weight kernels are HIP only");
    HIP_TRY(hipSetDevice(d.device));
    int rc = NIN_OK;
    if (!d.e2e_weights && (rc = dev_alloc(d, &d.e2e_weights, (size_t)std::max<int64_t>(d.nnz_e, 1)))) return rc;
    if (!d.e2e_nws && (rc = dev_alloc(d, &d.e2e_nws, (size_t)std::max<int64_t>(g->h.n_points, 1)))) return rc;
    if (d.chunkable) {
        if (!d.fields_set) return fail(NIN_ESTATE, "nin_fields_set has not been called");
        if (method != NIN_METHOD_GLS && method != NIN_METHOD_IDW && method != NIN_METHOD_LS) return fail(NIN_EINVAL, "unknown method %d", method);
        if (method == NIN_METHOD_GLS && !d.have_perm) return fail(NIN_ESTATE, "GLS needs permeability and diff_mag");
        if (method == NIN_METHOD_GLS && d.gls_too_large)
            return fail(NIN_ERANGE, "a node's GLS system has more than 1024 rows (more than ~100 cells around one node): beyond the fallback kernel");
        return interpolate_chunked(g, method, indptr, indices, data, nnz_out, neumann_ws);
    }
    rc = nin_weights_device(g, method, nullptr, 0, 1, d.e2e_weights, d.e2e_nws, nullptr);
    if (!rc) rc = csr_compact_pipelined(g, d.e2e_weights, d.e2e_nws, indptr, indices, data, nnz_out, neumann_ws, nullptr);
    return rc;
}

// Give the grid's call scratch back (the buffers nin_interpolate_csr_host / nin_csr_compact_host / nin_apply_* allocate
// on first use and keep: ~2.3 GB of HBM at 10 M cells, + 10 MB of page-locked host memory); the next call allocates again.
int nin_grid_release_scratch(nin_grid *g) {
    if (!g) return fail(NIN_EINVAL, "NULL grid");
    DeviceGrid &d = g->d;
    if (d.device < 0) return NIN_OK;
    HIP_TRY(hipSetDevice(d.device));
    HIP_TRY(hipDeviceSynchronize());
    void *scratch[] = {d.e2e_weights, d.e2e_nws, d.e2e_data, d.e2e_cnt, d.e2e_ptr, d.e2e_indices, d.e2e_tmp, d.apply_weights};
    for (void *p : scratch) {
        if (!p) continue;
        auto it = std::find(d.allocs.begin(), d.allocs.end(), p);
        if (it != d.allocs.end()) d.allocs.erase(it);
        (void)hipFree(p);
    }
    d.e2e_weights = d.e2e_nws = d.e2e_data = nullptr;
    d.e2e_cnt = d.e2e_ptr = d.e2e_indices = nullptr;
    d.e2e_tmp = nullptr;
    d.e2e_tmp_bytes = 0;
    d.apply_weights = nullptr;
    if (d.flag_staging) { (void)hipHostFree(d.flag_staging); d.flag_staging = nullptr; }
    return NIN_OK;
}

int nin_apply_device(nin_grid *g, int method, const double *dev_u_cells, int32_t n_fields, double *dev_node_values,
                     double *dev_neumann_ws, void *stream_) {
    if (!g || !dev_u_cells || !dev_node_values || !dev_neumann_ws) return fail(NIN_EINVAL, "NULL argument");
    if (n_fields < 1) return fail(NIN_EINVAL, "n_fields must be >= 1");
    DeviceGrid &d = g->d;
    if (d.device < 0 || d.prebuilt) return fail(NIN_ENODEVICE, "grid is not on a device: the weight kernels are HIP only");
    HIP_TRY(hipSetDevice(d.device));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!d.apply_weights) {   // the weights of the last apply: one buffer per grid, allocated on first use
        int rc = dev_alloc(d, &d.apply_weights, (size_t)std::max<int64_t>(d.nnz_e, 1));
        if (rc) return rc;
    }
    // GLS on a mesh with cube nodes: the cube-node kernel forms W . u itself (the 64 bytes of a node's row are neither written
    // nor read again); the other kernels write their rows as always and a list kernel applies those (NIN_APPLY_NO_FUSION: off)
    if (method == NIN_METHOD_GLS && d.hex8.count > 0 && d.noncube_nodes_ready && getenv("NIN_APPLY_NO_FUSION") == nullptr) {
        if (!d.fields_set) return fail(NIN_ESTATE, "nin_fields_set has not been called");
        if (!d.have_perm) return fail(NIN_ESTATE, "GLS needs permeability and diff_mag");
        if (d.gls_too_large) return fail(NIN_ERANGE, "a node's GLS system has more than 1024 rows: beyond the fallback kernel");
        HIP_TRY(hipMemsetAsync(d.gls_queue, 0, kGlsQueueInts * sizeof(int32_t), stream));
        int rc = gls_side_begin(d, 1, d.apply_weights, dev_neumann_ws, stream, 0, d.gls[kGlsClasses - 1].count, 0, d.mfg.count);
        if (!rc) rc = launch_gls_hex8mf_apply(d.v, d.hex8.nodes, d.hex8_desc, d.hex8.count, 1, dev_u_cells, n_fields, dev_node_values,
                                              dev_neumann_ws, d.gls_queue, stream);
        if (!rc) rc = launch_gls_but_cube(d, 1, d.apply_weights, dev_neumann_ws, stream);
        if (!rc) rc = launch_apply_list(d.v, d.apply_weights, dev_u_cells, n_fields, dev_node_values, d.noncube_nodes, d.noncube_count, stream);
        if (rc) return fail(rc, "launch failed: %s", hipGetErrorString(hipGetLastError()));
        return NIN_OK;
    }
    // the weights are computed ONCE, whatever the number of fields (they depend on the mesh, the permeability and the
    // Neumann flags only: the reference's callers do `weights.dot(u)` per field with the same matrix)
    int rc = nin_weights_device(g, method, nullptr, 0, 1, d.apply_weights, dev_neumann_ws, stream_);
    if (rc) return rc;
    rc = n_fields == 1 ? launch_apply(d.v, d.apply_weights, dev_u_cells, dev_node_values, (int32_t)g->h.mx_elems_per_point, d.nnz_e, stream)
                       : launch_apply_fields(d.v, d.apply_weights, dev_u_cells, n_fields, dev_node_values, (int32_t)g->h.mx_elems_per_point, d.nnz_e, stream);
    if (rc) return fail(rc, "launch failed");
    return NIN_OK;
}

int nin_apply_fields_host(nin_grid *g, int method, const double *u_cells, int32_t n_fields, double *node_values,
                          double *neumann_ws) {
    if (!g || !u_cells || !node_values || !neumann_ws) return fail(NIN_EINVAL, "NULL argument");
    if (n_fields < 1) return fail(NIN_EINVAL, "n_fields must be >= 1");
    DeviceGrid &d = g->d;
    if (d.device < 0) return fail(NIN_ENODEVICE, "grid is not on a device: the weight kernels are HIP only");
    HIP_TRY(hipSetDevice(d.device));
    const size_t pb = (size_t)g->h.n_points * 8, eb = (size_t)g->h.n_elems * 8;
    double *dn = nullptr, *du = nullptr, *dv = nullptr;
    auto cleanup = [&]() { (void)hipFree(dn); (void)hipFree(du); (void)hipFree(dv); };
#define TRY_A(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(NIN_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)
    TRY_A(hipMalloc((void **)&dn, pb));
    TRY_A(hipMalloc((void **)&du, eb * n_fields));
    TRY_A(hipMalloc((void **)&dv, pb * n_fields));
    TRY_A(hipMemcpy(du, u_cells, eb * n_fields, hipMemcpyHostToDevice));
    const int rc = nin_apply_device(g, method, du, n_fields, dv, dn, nullptr);
    if (rc) { cleanup(); return rc; }
    TRY_A(hipMemcpy(node_values, dv, pb * n_fields, hipMemcpyDeviceToHost));
    TRY_A(hipMemcpy(neumann_ws, dn, pb, hipMemcpyDeviceToHost));
#undef TRY_A
    cleanup();
    return NIN_OK;
}

int nin_apply_host(nin_grid *g, int method, const double *u_cells, double *node_values, double *neumann_ws) {
    return nin_apply_fields_host(g, method, u_cells, 1, node_values, neumann_ws);
}

int64_t nin_algorithmic_bytes(const nin_grid *g, int method) {
    if (!g) return -1;
    // SURVEY 8(d), canonical device layout s_i = 4:
    //   B_idw/ls = 4(P+1) + 4 nnz_esup + 24P + 24E + 2P + (8+4) nnz_out + 8P          (nnz_out = nnz_esup)
    //   B_gls    = B_idw/ls + 4(P+1) + 4 nnz_fsup + F (2*4 + 1 + 24 + 24) + E (72+8) + 8P
    const int64_t P = g->h.n_points, E = g->h.n_elems, F = g->h.n_faces;
    const int64_t nze = g->h.nnz_esup, nzf = g->h.nnz_fsup;
    int64_t b = 4 * (P + 1) + 4 * nze + 24 * P + 24 * E + 2 * P + 12 * nze + 8 * P;
    if (method == NIN_METHOD_GLS) b += 4 * (P + 1) + 4 * nzf + F * 57 + E * 80 + 8 * P;
    return b;
}

const char *nin_kernel_name(int method) {
    if (method == NIN_METHOD_IDW) return "nin_rows_kernel<0>";
    if (method == NIN_METHOD_LS) return "nin_rows_kernel<1>";
    return kernel_name_gls_hex8mf();   // dominant on hexahedron meshes; kernel_name_gls_block() covers the other nodes
}

int nin_host_alloc(size_t bytes, void **ptr) {
    if (!ptr) return fail(NIN_EINVAL, "NULL argument");
    *ptr = nullptr;
    const hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return fail(NIN_ENOMEM, "hipHostMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    return NIN_OK;
}

int nin_host_free(void *ptr) {
    if (!ptr) return NIN_OK;
    const hipError_t e = hipHostFree(ptr);
    if (e != hipSuccess) return fail(NIN_EHIP, "hipHostFree: %s", hipGetErrorString(e));
    return NIN_OK;
}

// ---- algorithmic flops of a launch plan, kernel by kernel (tools/count_algorithmic_flops.py holds the same formulas) ----
namespace {
// n_pivot Householder steps on an m-row block with n_cols columns in all (pivot columns included): 2 m for the norm, 8 for the
// scalar chain, 4 m per trailing column (dot + update)
double hh_flops(int64_t m, int64_t n_pivot, int64_t n_cols) {
    double f = 0;
    for (int64_t k = 0; k < n_pivot; ++k) {
        const double rows = (double)(m - k);
        f += 2 * rows + 8 + 4 * rows * (double)(n_cols - k - 1);
    }
    return f;
}
// fronts of 3-face cells + a dense rest (kernels_gls_hex8mf / mfw / mfx): F fronts, D dense cells, `faces` internal faces of which
// `free_faces` join two dense cells
double multifrontal_flops(int64_t F, int64_t D, int64_t faces, int64_t free_faces, int64_t neumann_rows = 0) {
    const double face = 52.0 * faces + 15.0 * neumann_rows;
    const double p1 = F * (hh_flops(10, 3, 3 + 9 + 1) + 9 + 54 + 6);
    const int64_t m2 = 7 * F + D + 3 * free_faces + neumann_rows, n2 = 3 * D;
    const double p2 = hh_flops(m2, n2, n2 + 1);
    const double tail = (double)n2 * n2 + F * (2 * 9 + 2) + D * (2 * 3 + 1) + 2 * (m2 - n2) + (F + D) + 1;
    return face + p1 + p2 + tail;
}
// the dense m x (n + 1) system with the last-row identity (one Householder QR of the first n columns, no right-hand sides):
// the small-node, block (one wavefront) and global-scratch kernels; n_if internal faces, n_nb Neumann rows
double dense_flops(int64_t ne, int64_t n_if, int64_t n_nb) {
    const int64_t m = ne + 3 * n_if + n_nb, n = 3 * ne;
    return 52.0 * n_if + 15.0 * n_nb + hh_flops(m, n, n + 1) + (double)n * n + 7.0 * ne + 2.0 * (m - n) + ne + 1;
}
// dgels on the reference's dense m x n system with nrhs right-hand sides
double dgels_flops(double m, double n, double nrhs) { return 2 * m * n * n - 2 * n * n * n / 3 + nrhs * (4 * m * n - 2 * n * n) + nrhs * n * n; }
}  // namespace

int nin_gls_plan_flops(nin_grid *g, double alg[22], double ref[22], int64_t computed[22]) {
    if (!g || !alg || !ref || !computed) return fail(NIN_EINVAL, "NULL argument");
    DeviceGrid &d = g->d;
    HostGrid &h = g->h;
    if (d.device < 0 || d.prebuilt) return fail(NIN_ENODEVICE, "grid is not on a device (call nin_grid_to_device first)");
    if (!d.fields_set || !d.flag_staging) return fail(NIN_ESTATE, "nin_fields_set has not been called (the Neumann flags decide which boundary nodes are computed)");
    if (h.ensure(A_ESUP_PTR | A_ESUP | A_FSUP_PTR | A_FSUP | A_ESUF)) return fail(NIN_EHIP, "mirroring the connectivity failed");
    HIP_TRY(hipSetDevice(d.device));
    for (int k = 0; k < 22; ++k) { alg[k] = ref[k] = 0.0; computed[k] = 0; }
    const int64_t P = h.n_points;
    // (F, D, free faces) of the nodes of the multifrontal kernels: from their descriptors
    std::vector<uint32_t> fdq((size_t)P, 0u);
    auto read_desc = [&](const int32_t *nodes, const uint32_t *desc, int32_t count, int words, int word) -> int {
        if (count <= 0) return 0;
        std::vector<int32_t> hn((size_t)count);
        std::vector<uint32_t> hd((size_t)count * words);
        if (hipMemcpy(hn.data(), nodes, (size_t)count * 4, hipMemcpyDeviceToHost) != hipSuccess) return -3;
        if (hipMemcpy(hd.data(), desc, hd.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return -3;
        for (int32_t i = 0; i < count; ++i) fdq[hn[i]] = hd[(size_t)i * words + word] & 0xFFFFFFu;   // (F, D, free faces: the low 24 bits of either kind's word)
        return 0;
    };
    for (int i = 0; i < 3; ++i)
        if (read_desc(d.mfw[i].nodes, d.mfw_desc[i], d.mfw[i].count, kMfwDescWords, 24)) return fail(NIN_EHIP, "reading the descriptors back failed");
    for (int i = 0; i < DeviceGrid::kMfxLists; ++i)
        if (read_desc(d.mfx[i].nodes, d.mfx_desc[i], d.mfx[i].count, kMfxDescWords, 0)) return fail(NIN_EHIP, "reading the descriptors back failed");
    if (read_desc(d.mfg.nodes, d.mfg_desc, d.mfg.count, kMfgDescWords, 0)) return fail(NIN_EHIP, "reading the descriptors back failed");
    for (int64_t p = 0; p < P; ++p) {
        const int c = g->node_class[p];
        const int k = c == 255 ? 5 : (c >= 252 && c <= 254) ? 6 + (254 - c) : (c >= 249 && c <= 251) ? 9 + (c - 249) : c == 248 ? 12 : (c >= 243 && c <= 247) ? 13 + (c - 243) : c == 242 ? 18 : c == 241 ? 19 : c == 240 ? 20 : c == 239 ? 21 : c;
        const int fl = d.flag_staging[p];
        if ((fl & 1) && !(fl & 2)) continue;                   // a Dirichlet boundary node: the zero row, nothing computed (gls.pyx:165-166)
        const int64_t eb = h.esup_ptr[p], ne = h.esup_ptr[p + 1] - eb, fb = h.fsup_ptr[p], nf = h.fsup_ptr[p + 1] - fb;
        int64_t n_if = 0;
        for (int64_t q = fb; q < fb + nf; ++q) {
            const int64_t f = h.fsup[q];
            n_if += h.esuf_ptr[f + 1] - h.esuf_ptr[f] > 1;
        }
        const int64_t n_bf = nf - n_if, n_nb = (fl & 2) ? n_bf : 0;
        const int64_t m = ne + 3 * n_if + n_nb;
        if (n_if == 0 || m < 3 * ne) continue;                 // outside the parity set: the zero row
        ++computed[k];
        ref[k] += dgels_flops((double)(ne + 3 * nf + n_nb), (double)(3 * ne + 1), (double)(ne + (n_nb ? 1 : 0)));
        if (k == 5) alg[k] += multifrontal_flops(4, 4, 12, 0);
        else if ((k >= 6 && k <= 8) || k >= 13) {
            const uint32_t w = fdq[p];
            alg[k] += multifrontal_flops(w & 255u, (w >> 8) & 255u, n_if, (w >> 16) & 255u, k >= 13 ? n_nb : 0);
        } else if (k == 12) {
            // two fronts of 8 rows (cell row, two internal faces, the Neumann row) x (3 | 6 | c), then 14 x 6 over the pair
            alg[k] += 52.0 * 4 + 15.0 * 4 + 2 * (hh_flops(8, 3, 3 + 6 + 1) + 9 + 36 + 6) + hh_flops(14, 6, 7) + 36 + 2 * 20 + 2 * 7 + 2 * 8 + 5;
        } else if (k >= 1 && k <= 3) {
            // the block kernel on more than one wavefront: fronts = the greedy independent set (esup order) of the cells with 1 .. 4
            // internal faces at the node that own no Neumann row; a front of a cell with f faces is (1 + 3 f) x (3 + 3 f + 1)
            const int n_c = (int)std::min<int64_t>(ne, 64);
            uint64_t adj[64] = {0};
            int nfi[64] = {0};
            uint64_t blocked = 0;
            for (int64_t q = fb; q < fb + nf; ++q) {
                const int64_t f = h.fsup[q], e0 = h.esuf_ptr[f];
                const bool internal = h.esuf_ptr[f + 1] - e0 > 1;
                int ia = -1, ib = -1;
                for (int i = 0; i < n_c; ++i) {
                    if (h.esup[eb + i] == h.esuf[e0]) ia = i;
                    if (internal && h.esup[eb + i] == h.esuf[e0 + 1]) ib = i;
                }
                if (internal && ia >= 0 && ib >= 0) { adj[ia] |= 1ull << ib; adj[ib] |= 1ull << ia; ++nfi[ia]; ++nfi[ib]; }
                else if (!internal && ia >= 0 && n_nb) blocked |= 1ull << ia;
            }
            uint64_t chosen = 0;
            double p1 = 0;
            int64_t rows_gone = 0, Fb = 0;
            for (int i = 0; i < n_c; ++i)
                if (nfi[i] >= 1 && nfi[i] <= 4 && !((blocked >> i) & 1ull) && !(adj[i] & chosen)) {
                    chosen |= 1ull << i;
                    p1 += hh_flops(1 + 3 * nfi[i], 3, 3 + 3 * nfi[i] + 1);
                    rows_gone += 3;
                    ++Fb;
                }
            const int64_t m2 = m - rows_gone, n2 = 3 * (ne - Fb);
            alg[k] += 52.0 * n_if + 15.0 * n_nb + p1 + hh_flops(m2, n2, n2 + 1) + (double)(3 * ne) * (3 * ne) + 7.0 * ne + 2.0 * (m - 3 * ne) + ne + 1;
        } else {
            alg[k] += dense_flops(ne, n_if, n_nb);             // small-node kernel, block kernel on one wavefront, global scratch
        }
    }
    return NIN_OK;
}

int nin_gls_plan(const nin_grid *g, int64_t counts[22]) {
    if (!g || !counts) return fail(NIN_EINVAL, "NULL argument");
    if (g->d.device < 0 || g->d.prebuilt) return fail(NIN_ENODEVICE, "grid is not on a device (call nin_grid_to_device first)");
    for (int c = 0; c < kGlsClasses; ++c) counts[c] = g->d.gls[c].count;
    counts[5] = g->d.hex8.count;
    counts[6] = g->d.mfw[0].count;
    counts[7] = g->d.mfw[1].count;
    counts[8] = g->d.mfw[2].count;
    for (int i = 0; i < 3; ++i) counts[9 + i] = g->d.small[i].count;
    counts[12] = g->d.quad4.count;
    for (int i = 0; i < 6; ++i) counts[13 + i] = g->d.mfx[i].count;
    counts[19] = g->d.mfg.count;
    counts[20] = g->d.mfx[6].count;
    counts[21] = g->d.mfx[7].count;
    return NIN_OK;
}

}  // extern "C"
