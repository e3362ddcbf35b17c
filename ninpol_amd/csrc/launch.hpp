// launch.hpp -- kernel launchers shared between the .hip translation units and the C ABI (internal)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_grid.hpp"

namespace nin {

#ifdef __HIPCC__
// dynamic LDS beyond the default limit: the attribute belongs to the (kernel, device) pair; raised once per new maximum
template <auto KERN>
inline int allow_dynamic_lds(size_t bytes) {
    if (bytes <= 48 * 1024) return 0;
    static size_t allowed[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t &a = allowed[dev & 63];
    if (bytes > a) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return -3;
        a = bytes;
    }
    return 0;
}
#endif

// all return 0 or a negative NIN_E* code; launches are asynchronous on `stream`
// targets == nullptr: all nodes 0 .. n_targets-1 (the wave-cooperative kernel); mx_row = MX_ELEMENTS_PER_POINT, nnz = the length of esup
// (the mean row length decides how many nodes make a tile)
int launch_idw(const GridView &g, const int32_t *targets, int32_t n_targets, int32_t mx_row, int64_t nnz, double *out, double *nws,
               hipStream_t stream);
int launch_ls(const GridView &g, const int32_t *targets, int32_t n_targets, int32_t mx_row, int64_t nnz, double *out, double *nws,
              hipStream_t stream);
// one GLS size class: `nodes` lists the class members (device), lds_bytes is per wave
int launch_gls_class(const GridView &g, const int32_t *nodes, int32_t count, int32_t lds_bytes,
                     int32_t rows_per_lane, int add_neumann, double *out, double *nws,
                     double *scratch, int64_t scratch_stride, int32_t scratch_slots, hipStream_t stream);
// the block kernel (kernels_gls_block.hip): one node per workgroup of `waves` wavefronts, system in LDS
// `queue`: one zeroed device int (the launch's work counter)
int launch_gls_block(const GridView &g, const int32_t *nodes, int32_t count, int32_t waves, int32_t col_slots,
                     int32_t lds_bytes, int add_neumann, double *out, double *nws, int32_t *queue, hipStream_t stream);
const char *kernel_name_gls_block();
// the multifrontal kernel for cube nodes (kernels_gls_hex8mf.hip): 4 lanes per node; `desc` = 4 descriptor words per
// list entry (hex8_desc.hpp, filled by launch_hex8_desc);
// `queue`: kGlsQueueInts zeroed device ints (one work counter per XCD, each on its own cache line)
int launch_hex8_desc(const GridView &g, const int32_t *nodes, int32_t count, int32_t *desc, hipStream_t stream);
int launch_gls_hex8mf(const GridView &g, const int32_t *nodes, const int32_t *desc, int32_t count, int add_neumann,
                      double *out, double *nws, int32_t *queue, hipStream_t stream);
int launch_gls_hex8mf_apply(const GridView &g, const int32_t *nodes, const int32_t *desc, int32_t count, int add_neumann,
                            const double *u_cells, int32_t n_fields, double *values, double *nws, int32_t *queue, hipStream_t stream);
const char *kernel_name_gls_hex8mf();
// the one-wavefront multifrontal kernel for two-coloured nodes (kernels_gls_mfw.hip); `desc` = kMfwDescWords (40) descriptor words per
// list entry (mfw_desc.hpp, filled by launch_mfw_desc); `queue`: one zeroed device int (the work counter)
int launch_mfw_desc(const GridView &g, const int32_t *nodes, int32_t count, uint32_t *desc, hipStream_t stream);
// `kind`: 0 = two-coloured nodes, 1 = those among them with at most kMfwSmallFronts fronts and kMfwSmallDense dense cells,
// 2 = the general kind (free faces, up to kMfwWideDense dense cells)
int launch_gls_mfw(const GridView &g, const int32_t *nodes, const uint32_t *desc, int32_t count, int kind, int add_neumann,
                   double *out, double *nws, int32_t *queue, hipStream_t stream);
const char *kernel_name_gls_mfw();
// the WIDE one-wavefront multifrontal kernel for interior nodes of unstructured meshes (kernels_gls_mfx.hip: up to 16 fronts + 21 dense
// cells, one wavefront per SIMD, the tiles in the accumulation registers); `desc` = kMfxDescWords (56) words per list entry
// (mfx_desc.hpp, filled by launch_mfx_desc); `queue`: one zeroed device int (the work counter)
int launch_mfx_desc(const GridView &g, const int32_t *nodes, int32_t count, uint32_t *desc, hipStream_t stream);
// `cls`: the size class of every node of the list (mfx_desc.hpp: mfx_size_class -- one kernel instantiation per class)
int launch_gls_mfx(const GridView &g, const int32_t *nodes, const uint32_t *desc, int32_t count, int cls, int add_neumann, double *out,
                   double *nws, int32_t *queue, hipStream_t stream);
const char *kernel_name_gls_mfx();
// interior nodes beyond the wide kernel's limits (kernels_gls_mfg.hip: up to 32 fronts + 40 dense cells; the tiles of the dense problem in a
// global-memory slot per resident wavefront); `desc` = kMfgDescWords (124) words per list entry (mfg_desc.hpp, filled by launch_mfg_desc);
// `tiles` = n_slots slots of kMfgSlotDoubles doubles; `queue`: one zeroed device int
int launch_mfg_desc(const GridView &g, const int32_t *nodes, int32_t count, uint32_t *desc, hipStream_t stream);
int launch_gls_mfg(const GridView &g, const int32_t *nodes, const uint32_t *desc, int32_t count, int add_neumann, double *out, double *nws,
                   int32_t *queue, double *tiles, int32_t n_slots, hipStream_t stream);
const char *kernel_name_gls_mfg();
// quad nodes (kernels_gls_quad4.hip: 4 cells, 4 internal + 4 boundary faces -- the nodes inside a boundary face of a hexahedron
// mesh): 2 lanes per node; `desc` = 2 descriptor words per list entry (quad4_desc.hpp), filled by launch_quad4_desc
int launch_quad4_desc(const GridView &g, const int32_t *nodes, int32_t count, int32_t *desc, hipStream_t stream);
int launch_gls_quad4(const GridView &g, const int32_t *nodes, const int32_t *desc, int32_t count, int add_neumann, double *out,
                     double *nws, hipStream_t stream);
// the one-wavefront dense kernel for small nodes (kernels_gls_mfw.hip): kind 0 / 1 / 2 = at most 4 / 8 / 12 cells, at most 64 rows
int launch_gls_small(const GridView &g, const int32_t *nodes, int32_t count, int kind, int add_neumann, double *out, double *nws,
                     hipStream_t stream);
// out[j] += nws[row(j)] for IDW / LS is a no-op (their neumann_ws is 0): nothing to launch.

// CSR finish (interpolator.pyx:622-624): count non-zeros per row, scan, compact
// rows [p_begin, p_end) only (p_end < 0: to the last node; p_begin: a multiple of 64)
int launch_row_nnz(const GridView &g, const double *data, int32_t *row_nnz, hipStream_t stream, int32_t p_begin = 0, int32_t p_end = -1);
int launch_compact(const GridView &g, const double *data, const int32_t *new_ptr, int32_t *indices,
                   double *vals, hipStream_t stream, int32_t p_begin = 0, int32_t p_end = -1);
// IDW (method_ls = 0) / LS (1) weights of the nodes [p_begin, p_end) (p_begin: a multiple of 64)
int launch_rows_range(const GridView &g, int method_ls, int32_t n_points, int32_t p_begin, int32_t p_end, int32_t mx_row, int64_t nnz,
                      double *out, double *nws, hipStream_t stream);

int launch_apply(const GridView &g, const double *data, const double *u, double *values, int32_t mx_row, int64_t nnz, hipStream_t stream);
// the same for the listed nodes only (one lane per node, rows straight from HBM)
int launch_apply_list(const GridView &g, const double *data, const double *u, int32_t k, double *values, const int32_t *list,
                      int32_t count, hipStream_t stream);
// k fields at once: u [k][n_elems], values [k][n_points]
int launch_apply_fields(const GridView &g, const double *data, const double *u, int32_t k, double *values, int32_t mx_row, int64_t nnz,
                        hipStream_t stream);

// GLS launch plan (grid_device.hip): size class of every node (255 = the hex8 kernel, 254 / 253 / 252 = the one-wavefront multifrontal kernel: two-coloured nodes large / small, general kind) and, per class, the
// maxima of (bytes, rows, columns) as 3 * kGlsClasses unsigned 64-bit values; all DEVICE pointers
// kernels_csr.hip: dst[e] = (src[3 e], src[3 e + 1], src[3 e + 2], 0)
int launch_pad_centroids(const double *src, int64_t n_elems, double *dst, hipStream_t stream);
int launch_classify(const GridView &g, int use_group, int force_global, uint8_t *node_class,
                    unsigned long long *class_max, hipStream_t stream);

const char *kernel_name_idw();
const char *kernel_name_ls();
const char *kernel_name_gls();

}  // namespace nin
