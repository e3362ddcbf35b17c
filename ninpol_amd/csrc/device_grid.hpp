// device_grid.hpp -- the grid as it lives in HBM (internal).
//
// Canonical device layout (DESIGN.md "Data layout in HBM"): all indices int32 (every count of an
// 80 M-cell mesh is < 2^31), flags one byte per node, reals float64 except the face normals, which
// the reference computes in float32 (grid.pyx:732-767) and which are therefore stored as the float32
// values they are.  Everything is structure-of-arrays; rows of esup / fsup are contiguous.
#pragma once
#include <cstdint>
#include <vector>

namespace nin {

struct GridView {  // passed to kernels by value
    int32_t n_points, n_elems, n_faces, dim;
    const int32_t *esup_ptr;   // [P+1]
    const int32_t *esup;       // [nnz_e]   cells around node, ascending
    const int32_t *fsup_ptr;   // [P+1]
    const int32_t *fsup;       // [nnz_f]   faces around node, ascending
    const double *coords;      // [P][3]
    const double *centroids;   // [E][3]
    const int32_t *face_cells; // [F][2]    esuf pair, second = -1 on a boundary face
    const double *face_center; // [F][3]
    const float *face_normal;  // [F][3]
    const double *perm;        // [E][9]    row-major 3x3 (may be null until nin_fields_set)
    const double *diff_mag;    // [E]
    const uint8_t *flags;      // [P]       bit0 boundary_points, bit1 neumann flag
};

constexpr int kGlsClasses = 5;  // four LDS budget classes (1 / 2 / 4 / 8 waves per node) + one global-scratch class

struct DeviceGrid {
    int device = -1;
    GridView v{};
    int64_t nnz_e = 0, nnz_f = 0;
    bool fields_set = false, have_perm = false;
    std::vector<void *> allocs;  // everything hipMalloc'd, freed together

    // GLS launch plan: nodes binned by the LDS bytes their least-squares system needs
    struct GlsClass {
        int32_t count = 0;
        int32_t *nodes = nullptr;  // device list (ascending node ids)
        int32_t lds_bytes = 0;     // per node (workgroup)
        int32_t waves = 1;         // wavefronts per node
        int32_t col_slots = 1;     // ceil(max columns / 64)
        int32_t rows_per_lane = 1; // ceil(max rows / 64): the wave kernel of the scratch class
        int32_t max_cells = 0, max_cols = 0, max_rows = 0;
    } gls[kGlsClasses];
    GlsClass hex8;  // nodes with exactly 8 cells and 12 faces, all internal: kernels_gls_group.hip
    double *gls_scratch = nullptr;  // global-memory systems for the oversize class
    int64_t gls_scratch_stride = 0; // doubles per wave slot
    int32_t gls_scratch_slots = 0;
};

}  // namespace nin
