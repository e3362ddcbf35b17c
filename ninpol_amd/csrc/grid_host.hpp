// grid_host.hpp -- host-side Grid of libninpol_amd (internal; the public surface is include/ninpol_amd.h)
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace nin {

constexpr int kMaxPointsPerElement = 8;  // ninpol_defines.pxd:2
constexpr int kMaxFacesPerElement = 6;   // :3
constexpr int kMaxPointsPerFace = 4;     // :4
constexpr int kNumElementTypes = 8;      // :5
constexpr int kMaxEdgesPerElement = 12;  // :6

struct DeviceGrid;  // device mirror, defined in device_grid.hpp

// the arrays of a HostGrid, as bits: a grid built on the device brings them to the host on first use
enum : unsigned {
    A_INPOEL = 1u << 0, A_ETYPE = 1u << 1, A_ESUP_PTR = 1u << 2, A_ESUP = 1u << 3, A_FSUP_PTR = 1u << 4, A_FSUP = 1u << 5,
    A_ESUF = 1u << 6, A_ESUEL = 1u << 7, A_INFAEL = 1u << 8, A_INPOFA = 1u << 9, A_BFACES = 1u << 10, A_BPOINTS = 1u << 11,
    A_COORDS = 1u << 12, A_CENTROIDS = 1u << 13, A_FCENTERS = 1u << 14, A_NORMALS = 1u << 15, A_AREAS = 1u << 16,
    A_ALL = (1u << 17) - 1
};
struct HostGrid;
struct LazyArrays {   // implemented by the device builder (grid_device.hip)
    virtual ~LazyArrays() {}
    virtual int fetch(HostGrid &h, unsigned which, std::string *err) = 0;   // 0 or a negative code
};

struct HostGrid {
    int64_t dim = 0, n_elems = 0, n_points = 0, n_faces = 0, n_edges = 0;
    int64_t mx_elems_per_point = 0, mx_points_per_point = 0, mx_elems_per_face = 0, mx_faces_per_point = 0;
    int build_edges = 0;
    int num_threads = 0;

    // topology tables (Grid ctor arguments, grid.pyx:47-53)
    int32_t npoel[kNumElementTypes], nfael[kNumElementTypes], nedel[kNumElementTypes];
    int32_t lnofa[kNumElementTypes][kMaxFacesPerElement];
    int32_t lpofa[kNumElementTypes][kMaxFacesPerElement][kMaxPointsPerFace];
    int32_t lpoed[kNumElementTypes][kMaxEdgesPerElement][2];

    std::vector<int32_t> inpoel;   // [E][8], -1 padded
    std::vector<int8_t> etype;     // [E]
    std::vector<int64_t> esup_ptr; // [P+1]
    std::vector<int32_t> esup;
    std::vector<int64_t> fsup_ptr; // [P+1]
    std::vector<int32_t> fsup;
    std::vector<int64_t> esuf_ptr; // [F+1]
    std::vector<int32_t> esuf;
    std::vector<int32_t> esuel;    // [E][6]
    std::vector<int32_t> infael;   // [E][6]
    std::vector<int32_t> inpofa;   // [F][4]
    std::vector<uint8_t> boundary_faces, boundary_points;
    std::vector<double> coords;        // [P][3]
    std::vector<double> centroids;     // [E][3]
    std::vector<double> faces_centers; // [F][3]
    std::vector<float> normal_faces;   // [F][3]  (the reference's values are float32-exact)
    std::vector<double> faces_areas;   // [F]

    // built on first request (psup is read by no method; edges only with build_edges)
    bool psup_built = false;
    std::vector<int64_t> psup_ptr;
    std::vector<int32_t> psup;
    bool edges_built = false;
    std::vector<int32_t> inedel; // [E][12]
    std::vector<int32_t> inpoed; // [n_edges][2]

    DeviceGrid *dev = nullptr;

    // arrays present on the host (everything, unless the grid was built on the device)
    unsigned have = A_ALL;
    std::unique_ptr<LazyArrays> lazy;
    int64_t nnz_esup = 0, nnz_fsup = 0;
    int ensure(unsigned which, std::string *err = nullptr) {
        const unsigned need = which & ~have;
        if (!need || !lazy) return 0;
        std::string local;
        const int rc = lazy->fetch(*this, need, err ? err : &local);
        if (!rc) have |= need;
        if (have == A_ALL) lazy.reset();
        return rc;
    }

    int build(const int64_t *connectivity, const int64_t *element_types, const double *xyz, int coords_dim);
    // used by the device builder (grid_device.hip) when it mirrors its int32 arrays into this object
    static void widen(const std::vector<int32_t> &src, std::vector<int64_t> &dst);
    void esuf_from_pairs(const std::vector<int32_t> &pairs);   // [F][2] (creator, neighbour or -1) -> esuf_ptr, esuf
    void build_psup();
    void build_inedel();
};

// pack_host.cpp (g++ -fopenmp): flags[p] = boundary bit | Neumann bit, the row cast like `.astype(int)`
void pack_node_flags(const double *neumann_flag, const uint8_t *boundary_points, int64_t n, uint8_t *out);

}  // namespace nin
