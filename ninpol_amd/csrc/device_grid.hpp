// device_grid.hpp -- the grid as it lives in HBM (internal).
//
// Canonical device layout (DESIGN.md "Data layout in HBM"): all indices int32 (every count of an
// 80 M-cell mesh is < 2^31), flags one byte per node, reals float64 except the face normals, which
// the reference computes in float32 (grid.pyx:732-767) and which are therefore stored as the float32
// values they are.  Everything is structure-of-arrays; rows of esup / fsup are contiguous.
#pragma once
#include <cstdint>
#include <string>
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
    const double *centroids4;  // [E][4] = (x, y, z, 0): one 32-byte record per cell for the IDW / LS gathers (null unless NIN_ROWS_PAD4: measured slower)
    const int32_t *face_cells; // [F][2]    esuf pair, second = -1 on a boundary face
    const double *face_center; // [F][3]
    const float *face_normal;  // [F][3]
    const double *perm;        // [E][9]    row-major 3x3 (may be null until nin_fields_set)
    const double *diff_mag;    // [E]
    const uint8_t *flags;      // [P]       bit0 boundary_points, bit1 neumann flag
};

#ifdef __HIPCC__
#define NIN_HD __host__ __device__
#else
#define NIN_HD
#endif

constexpr int kGlsQueueInts = 8 * 16;   // hex8 kernel: one work counter per XCD, each on its own 64-byte line
constexpr int kGlsClasses = 5;  // four LDS budget classes (1 / 2 / 4 / 8 waves per node) + one global-scratch class
// LDS bytes a node's system may take in class c and the waves per node the block kernel runs it with
// (16 / 5 / 2 / 1 workgroups per CU); the last class keeps its systems in global-memory scratch.
NIN_HD inline int32_t gls_class_budget(int c) { return c == 0 ? 10240 : c == 1 ? 32768 : c == 2 ? 81920 : c == 3 ? 159744 : 0; }
NIN_HD inline int32_t gls_class_waves(int c) { return c == 0 ? 1 : c == 1 ? 2 : c == 2 ? 4 : c == 3 ? 8 : 1; }
// LDS bytes of one node's system in the block kernel (kernels_gls_block.hip): the work-queue word, (n + 1) columns of odd pitch m | 1,
// the partial-dot buffers (later y and the weight row), the staged cell ids and their column blocks
NIN_HD inline int64_t gls_block_lds_bytes(int64_t ne, int64_t m, int64_t n, int waves) {
    const int64_t doubles = 2 + (n + 1) * (m | 1) + (waves == 1 ? 2 : (waves != 4 ? 2 : 1) * waves) * n + ((ne + 1) >> 1) +
                            ((ne + 7) >> 3);   // + the cells' column blocks (uint8)
    return ((doubles * 8 + 15) / 16) * 16;
}
// Size class of a node with ne cells, nf faces of which nbf on the boundary; *bytes = what it needs there.
NIN_HD inline int gls_node_class(int64_t ne, int64_t nf, int64_t nbf, bool force_global, int64_t *bytes, int64_t *rows,
                                 int64_t *cols) {
    const int64_t m = ne + 3 * (nf - nbf) + nbf, n = 3 * ne + 1;
    int c = kGlsClasses - 1;
    int64_t b = ((((ne + 1) >> 1) + n + m * n) * 8 + 15) / 16 * 16;   // the scratch slot of the wave kernel
    for (int k = 0; k < kGlsClasses - 1 && !force_global && n <= 256; ++k) {
        const int64_t need = gls_block_lds_bytes(ne, m, n, gls_class_waves(k));
        if (need <= gls_class_budget(k)) { c = k; b = need; break; }
    }
    *bytes = b; *rows = m; *cols = n;
    return c;
}

struct DeviceGrid {
    int device = -1;
    GridView v{};
    int64_t nnz_e = 0, nnz_f = 0;
    bool fields_set = false, have_perm = false;
    bool prebuilt = false;  // arrays came from build_grid_on_device(): nin_grid_to_device adopts them, no upload
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
    GlsClass hex8;  // cube nodes (8 cells, 12 internal faces, cube cell graph): kernels_gls_hex8mf.hip
    int32_t *hex8_desc = nullptr;   // [4 * hex8.count] lane descriptors (hex8_desc.hpp)
    GlsClass mfw[3];   // kernels_gls_mfw.hip (mfw_desc.hpp): two-coloured nodes large (Kuhn tetrahedra) / small (wedges), general kind
    uint32_t *mfw_desc[3] = {nullptr, nullptr, nullptr};   // [kMfwDescWords * mfw[i].count] descriptor words
    // kernels_gls_mfx.hip (mfx_desc.hpp): interior nodes of unstructured meshes, up to 16 fronts + 21 dense cells, one list per size
    // class of the dense problem (6 x 10, 7 x 11, 8 x 13, 9 x 15, 10 x 16 tiles)
    static constexpr int kMfxLists = 8;   // (the sixth: boundary nodes, kernels_gls_mfx.hip's BND instantiation; the seventh: the small interior class (4, 7); the eighth: (7, 12))
    GlsClass mfx[kMfxLists];
    uint32_t *mfx_desc[kMfxLists] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [kMfxDescWords * mfx[c].count] descriptor words
    // kernels_gls_mfg.hip (mfg_desc.hpp): interior nodes beyond the wide kernel's registers (up to 32 fronts + 40 dense cells): the tiles
    // of the dense problem in a global-memory slot per resident wavefront
    GlsClass mfg;
    uint32_t *mfg_desc = nullptr;   // [kMfgDescWords * mfg.count] descriptor words
    double *mfg_tiles = nullptr;    // [mfg_slots][kMfgSlotDoubles]
    int32_t mfg_slots = 0;
    const int32_t *noncube_nodes = nullptr;   // every node the cube-node kernel does not take (the fused apply's list kernel)
    int32_t noncube_count = 0;
    bool noncube_nodes_ready = false;
    GlsClass quad4;      // kernels_gls_quad4.hip: nodes inside a boundary face of a hexahedron mesh (4 cells, 4 + 4 faces)
    int32_t *quad4_desc = nullptr;   // [2 * quad4.count] descriptor words
    GlsClass small[3];   // kernels_gls_mfw.hip, nin_gls_small_kernel: nodes with at most 4 / 8 / 12 cells and 64 rows that no kernel above takes
    double *gls_scratch = nullptr;  // global-memory systems for the oversize class
    int64_t gls_scratch_stride = 0; // doubles per wave slot
    int32_t gls_scratch_slots = 0;
    int32_t *gls_queue = nullptr;   // [kGlsQueueInts]
    // buffers of nin_interpolate_csr_host / nin_csr_compact_host, allocated on first use and kept (0.65 + 0.98 GB at
    // 10 M cells; allocating and freeing them cost ~10 ms of every call)
    double *e2e_weights = nullptr, *e2e_nws = nullptr, *e2e_data = nullptr;
    int32_t *e2e_cnt = nullptr, *e2e_ptr = nullptr, *e2e_indices = nullptr;
    void *e2e_tmp = nullptr;
    size_t e2e_tmp_bytes = 0;
    double *apply_weights = nullptr;   // [nnz_e] weights of the last nin_apply_device (allocated on first use)
    uint8_t *flag_staging = nullptr;   // page-locked [n_points]: nin_fields_set packs the node flags here and uploads from it
    void *copy_stream = nullptr, *copy_stream2 = nullptr;   // hipStream_t of the device-to-host copies that run under the kernels
    void *ev_weights = nullptr, *ev_scan = nullptr;   // hipEvent_t: weights written / row pointers scanned
    // The long pole of a GLS launch plan: the few nodes of the global-scratch class (more cells than any in-register / in-LDS kernel
    // holds: ~0.03 % of a Delaunay mesh) take ~2 ms EACH on a wavefront of their own.  They start first, on a stream of their own,
    // and run under the other kernels (abi.hip: gls_side_begin / gls_side_end).
    void *side_stream = nullptr, *ev_fork = nullptr, *ev_join = nullptr;
    bool side_pending = false;
    // interpolate()'s pipeline (abi.hip, interpolate_chunked): the node range is cut into kE2eChunks pieces at multiples of 64 nodes;
    // every GLS list is ascending, so a piece is a sub-range of each: chunk_off[list][k] .. chunk_off[list][k + 1]
    // (lists 0 .. kGlsClasses - 1: the block kernel's classes, then the cube-node kernel, the three mfw kinds, the three small kinds, the quad nodes, the wide multifrontal kernel's five size classes)
    static constexpr int kE2eChunks = 4;
    int32_t chunk_node[kE2eChunks + 1] = {0, 0, 0, 0, 0};
    int32_t chunk_off[kGlsClasses + 8 + kMfxLists + 1][kE2eChunks + 1] = {};   // (the last: kernels_gls_mfg.hip's list)
    bool chunkable = false;
    bool gls_too_large = false;     // some node's system has more rows than the scratch kernel handles (1024)
};

struct HostGrid;
// grid_device.hip: connectivity + geometry built on `device`, left in `d` and mirrored into `h`.
// 0, -1 bad connectivity, -2 memory, -3 HIP (text in *err), -5 a count does not fit int32.
int build_grid_on_device(HostGrid &h, DeviceGrid &d, int device, const int64_t *connectivity,
                         const int64_t *element_types, const double *xyz, int coords_dim, std::string *err);

}  // namespace nin
