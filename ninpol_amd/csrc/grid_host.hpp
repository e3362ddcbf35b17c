// grid_host.hpp -- host-side Grid of libninpol_amd (internal; the public surface is include/ninpol_amd.h)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace nin {

constexpr int kMaxPointsPerElement = 8;  // ninpol_defines.pxd:2
constexpr int kMaxFacesPerElement = 6;   // :3
constexpr int kMaxPointsPerFace = 4;     // :4
constexpr int kNumElementTypes = 8;      // :5
constexpr int kMaxEdgesPerElement = 12;  // :6

struct DeviceGrid;  // device mirror, defined in device_grid.hpp

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

    int build(const int64_t *connectivity, const int64_t *element_types, const double *xyz, int coords_dim);
    void build_psup();
    void build_inedel();
};

}  // namespace nin
