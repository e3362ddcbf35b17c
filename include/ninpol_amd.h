/*
 * ninpol_amd.h -- C ABI of libninpol_amd.so: the MI355X-native nodal-interpolation hot path.
 *
 * The reference (daviyan5/ninpol) has no C ABI: its extension points are the Python class
 * `ninpol.Interpolator` and the in-process Cython method-plugin convention
 *     prepare(Grid grid, cells_data, points_data, faces_data, variable_to_index, variable,
 *             target_points, weights[out], neumann_ws[out])
 * (ninpol/_methods/idw.pxd:19-24, ls.pxd:20-25, gls.pxd:22-27; called at
 * ninpol/_interpolator/interpolator.pyx:657-665).  The entry points below are what a ctypes / cgo /
 * Cython binding for that path binds instead; each cites the reference interface it replaces.
 * Plain pointers and sizes only: no Python, numpy or torch types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success or a negative NIN_E* code; nin_last_error() gives text.
 *     Nothing throws, nothing aborts (the reference's inner helpers are `noexcept nogil` too).
 *   - index arrays crossing the boundary are int64 (`ctypedef long long DTYPE_I_t`, grid.pxd:13),
 *     reals are float64 (grid.pxd:14); inside the library and on the device indices are int32.
 *   - "host" pointers are ordinary memory owned by the caller; "dev" pointers are HIP device
 *     pointers (e.g. torch tensor .data_ptr()) owned by the caller; `stream` is a hipStream_t passed
 *     as void* (NULL = the null stream).
 *   - a nin_grid is NOT thread-safe (like one reference Interpolator instance); distinct handles are
 *     independent.
 */
#ifndef NINPOL_AMD_H
#define NINPOL_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nin_grid nin_grid; /* host connectivity + geometry (+ device mirror once uploaded) */

enum {
    NIN_OK = 0,
    NIN_EINVAL = -1,   /* bad argument (NULL, negative size, unknown name / method)            */
    NIN_ENOMEM = -2,   /* host or device allocation failed                                       */
    NIN_EHIP = -3,     /* a HIP runtime call failed (text in nin_last_error)                     */
    NIN_ENODEVICE = -4,/* no usable GPU / grid not uploaded: the product path has no CPU fallback */
    NIN_ERANGE = -5,   /* a count does not fit the int32 device layout, or a node is too large   */
    NIN_ESTATE = -6    /* call order (fields not set, grid not built ...)                        */
};

enum { NIN_METHOD_GLS = 0, NIN_METHOD_IDW = 1, NIN_METHOD_LS = 2 }; /* interpolator.pyx:60-64 order */

/* dtype codes reported by nin_grid_array */
enum { NIN_I64 = 0, NIN_F64 = 1 };

const char *nin_last_error(void);
const char *nin_version(void);

/* ---- Grid: connectivity + geometry, built once on the host ---------------------------------
 * Replaces Grid.__cinit__ (grid.pyx:47-140) + Grid.build() (:142-231) + load_point_coords (:661)
 * + calculate_centroids (:669) + calculate_normal_faces (:721), i.e. interpolator.pyx:194,204-207.
 * Arguments are the reference's Grid ctor arguments (same meaning, same -1 padding):
 *   npoel[8], nfael[8], lnofa[8][6], lpofa[8][6][4], nedel[8], lpoed[8][12][2],
 *   connectivity[n_elems][8], element_types[n_elems], coords[n_points][coords_dim].
 * Results are bit-identical to the reference's for conforming meshes (integers, float64 centroids
 * and face centres, float32-valued normals and areas).  num_threads <= 0 picks the OpenMP default. */
int nin_grid_create(int64_t dim, int64_t n_elems, int64_t n_points,
                    const int64_t *npoel, const int64_t *nfael, const int64_t *lnofa,
                    const int64_t *lpofa, const int64_t *nedel, const int64_t *lpoed,
                    const int64_t *connectivity, const int64_t *element_types,
                    const double *coords, int coords_dim, int build_edges, int num_threads,
                    nin_grid **out);
void nin_grid_destroy(nin_grid *g);

/* The same Grid, built ON `device` (grid.pyx:142-231, 233-525, 669-809 as data-parallel HIP kernels,
 * csrc/grid_device.hip): same arguments, same arrays bit for bit.  The arrays the weight kernels read stay in
 * HBM -- a later nin_grid_to_device(g, device) adopts them instead of uploading -- and all of them are mirrored
 * to the host for nin_grid_array_*.  No CPU fallback: NIN_ENODEVICE without a GPU. */
int nin_grid_create_on_device(int64_t dim, int64_t n_elems, int64_t n_points,
                              const int64_t *npoel, const int64_t *nfael, const int64_t *lnofa,
                              const int64_t *lpofa, const int64_t *nedel, const int64_t *lpoed,
                              const int64_t *connectivity, const int64_t *element_types,
                              const double *coords, int coords_dim, int build_edges, int device,
                              nin_grid **out);

/* Readonly attributes of Grid (grid.pxd:128-187).  Scalars: dim n_elems n_points n_faces n_edges
 * MX_ELEMENTS_PER_POINT MX_POINTS_PER_POINT MX_ELEMENTS_PER_FACE MX_FACES_PER_POINT.  Unknown -> -1. */
int64_t nin_grid_scalar(const nin_grid *g, const char *name);

/* Number of elements of array `name` and its dtype code; arrays: esup esup_ptr psup psup_ptr fsup
 * fsup_ptr esuf esuf_ptr esuel infael inpofa inpoel inpoed inedel boundary_faces boundary_points
 * point_coords centroids faces_centers normal_faces faces_areas element_types. */
int nin_grid_array_info(nin_grid *g, const char *name, int64_t *count, int *dtype);
/* Copy array `name` into caller memory of `count` elements (int64 or float64 as reported). */
int nin_grid_array_copy(nin_grid *g, const char *name, void *dst, int64_t count);

/* ---- device residency --------------------------------------------------------------------- */
int nin_device_count(int *count);
/* Push the CSR connectivity (esup, fsup), the per-face pair/centre/normal and the geometry to the
 * HBM of `device` in the canonical int32 / SoA layout.  north_star: "built once as CSR
 * on host and pushed to HBM". */
int nin_grid_to_device(nin_grid *g, int device);
int nin_grid_device(const nin_grid *g); /* device id or -1 */

/* Per-call fields the plugins read from the data tables (idw.pyx:27-28, ls.pyx:27-28,
 * gls.pyx:47-59): permeability[E][3][3] row-major and diff_mag[E] (may be NULL for IDW / LS),
 * neumann_flag[P] (the points_data row, cast to integer like `.astype(int)`), neumann_val[P]
 * (may be NULL for IDW / LS).  Host pointers; uploaded to the grid's device.  permeability / diff_mag NULL leaves
 * the copies already on the device in place (they belong to the mesh, not to the variable). */
int nin_fields_set(nin_grid *g, const double *permeability, const double *diff_mag,
                   const double *neumann_flag, const double *neumann_val);

/* ---- the hot path -------------------------------------------------------------------------
 * Replaces supported_methods[method](grid, ..., target_points, weights, neumann_ws)
 * (interpolator.pyx:657-665 -> idw.pyx:14-84, ls.pyx:21-135, gls.pyx:38-474).
 *
 * Output layout: instead of the dense weights[n_target][MX_ELEMENTS_PER_POINT] table, weight j of
 * node p is written to csr_data[esup_ptr[p] + j] -- exactly the position the reference's COO fill
 * reads it into (interpolator.pyx:612-618), with column esup[esup_ptr[p] + j].  neumann_ws has
 * n_points entries.  Entries of nodes that the method skips (Dirichlet boundary nodes,
 * idw.pyx:62-63) and of nodes outside `targets` are 0.
 *
 * targets: int64 node ids (host pointer) or NULL for all nodes.  add_neumann != 0 applies
 * `data[j] = weights + neumann_ws[row]` of interpolator.pyx:618 in the same kernel.
 * dev_csr_data [nnz_esup] and dev_neumann_ws [n_points] are DEVICE pointers; with targets == NULL the
 * launch is asynchronous on `stream`; with a target list the call returns once the kernels have run
 * (the device copy of the list is owned by the call).
 * Threading: as in the reference, one Interpolator / grid is not re-entrant (interpolator.pxd: load_mesh mutates
 * self.grid): the GLS launches of a grid share its work counters, so keep ONE nin_weights_* / nin_apply_* call in
 * flight per grid (any number of grids may run concurrently, on the same or on different streams and devices). */
int nin_weights_device(nin_grid *g, int method, const int64_t *targets, int64_t n_targets,
                       int add_neumann, double *dev_csr_data, double *dev_neumann_ws, void *stream);

/* Same, with host outputs (allocates scratch on the device, synchronises, copies back). */
int nin_weights_host(nin_grid *g, int method, const int64_t *targets, int64_t n_targets,
                     int add_neumann, double *csr_data, double *neumann_ws);

/* Device-side finish of interpolator.pyx:622-624 (csr_matrix + eliminate_zeros): compacts the
 * esup-shaped (indptr = esup_ptr, indices = esup, data) triplet, dropping exact zeros.
 * Host outputs: indptr[n_points+1] (int32, what scipy picks for these sizes), indices / data sized
 * by the caller to nnz_esup; *nnz_out receives the surviving count. */
int nin_csr_compact_host(nin_grid *g, const double *dev_csr_data, int32_t *indptr, int32_t *indices,
                         double *data, int64_t *nnz_out, void *stream);

/* interpolate() in one call, host outputs: weights (with `+ neumann_ws[row]`, interpolator.pyx:618) for ALL nodes,
 * then the device-side csr_matrix + eliminate_zeros of interpolator.pyx:622-624.  indptr [n_points+1], indices and
 * data sized by the caller to nnz_esup (upper bound), neumann_ws [n_points]; *nnz_out = surviving entries. */
int nin_interpolate_csr_host(nin_grid *g, int method, int32_t *indptr, int32_t *indices, double *data,
                             int64_t *nnz_out, double *neumann_ws);

/* Interpolate a cell field to the nodes without materialising the matrix on the host: weights for all nodes (with the
 * `+ neumann_ws[row]` of interpolator.pyx:618), then node_values = W . u_cells on the device -- what the reference's
 * callers compute next as `weights.dot(u)` (tests/utils/analytical.py:236).  Host pointers: u_cells [n_elems],
 * node_values [n_points] (0 on the empty rows of Dirichlet nodes), neumann_ws [n_points]. */
int nin_apply_host(nin_grid *g, int method, const double *u_cells, double *node_values, double *neumann_ws);

/* The same for n_fields cell fields at once -- the weights are computed ONCE and applied to every field (the
 * reference's callers loop `weights.dot(u)` over their variables with one matrix, tests/utils/analytical.py:236):
 * u_cells [n_fields][n_elems] -> node_values [n_fields][n_points], row-major.  _device: DEVICE pointers, asynchronous
 * on `stream` (hipStream_t, NULL = default); _fields_host: host pointers, synchronous. */
int nin_apply_device(nin_grid *g, int method, const double *dev_u_cells, int32_t n_fields, double *dev_node_values,
                     double *dev_neumann_ws, void *stream);
int nin_apply_fields_host(nin_grid *g, int method, const double *u_cells, int32_t n_fields, double *node_values,
                          double *neumann_ws);

/* ---- native table packing (replaces the Python loops of interpolator.pyx:255-451, 501-509) ---------------------
 * nin_pack_connectivity: interpolator.pyx:333-361 -- per-type cell blocks (block b: rows[b] x cols[b] int64 node ids,
 *   element type type_id[b]) -> fixed-width, -1 padded connectivity [n_elems][8] + element_types [n_elems].
 * nin_pack_table_row: interpolator.pyx:397-419 -- the first `take` columns of a row-major (n, src_cols) float64 array,
 *   flattened into one row of a (n_vars, n * max_shape) data table.
 * nin_diff_mag: interpolator.pyx:501-509 as compiled (`** (1 / 3)` is `** 0` under cdivision): (1 - 3 / tr K)^2. */
int nin_pack_connectivity(int32_t n_blocks, const int64_t *const *block_data, const int64_t *rows, const int64_t *cols,
                          const int64_t *type_id, int64_t *connectivity, int64_t *element_types);
int nin_pack_table_row(const double *src, int64_t n, int64_t src_cols, int64_t take, double *dst);
int nin_diff_mag(const double *permeability, int64_t n_elems, double *diff_mag);
/* 64-bit hash of ALL bytes of a table, OpenMP-parallel (is the permeability resident on the device still the caller's?
 * the reference re-reads its tables on every call, interpolator.pyx:583-600). */
int nin_hash64(const void *data, size_t bytes, uint64_t *out);

/* Page-locked host memory for the arrays that come back over PCIe (nin_interpolate_csr_host's outputs: 57 GB/s into
 * pinned memory against 10-15 GB/s into pageable memory on the MI355X box).  The reference returns ordinary numpy
 * arrays (interpolator.pyx:622-628); ninpol_amd.Interpolator wraps these buffers as numpy arrays and recycles them. */
int nin_host_alloc(size_t bytes, void **ptr);
int nin_host_free(void *ptr);

/* Give back the scratch a grid keeps between calls: the device buffers nin_interpolate_csr_host / nin_csr_compact_host /
 * nin_apply_* allocate on first use (weights, compacted triplets, counters: ~2.3 GB of HBM at 10 M cells, 8 x that at
 * 80 M) and the page-locked flag staging buffer.  The next call allocates them again.  (The reference frees its dense
 * weight table when interpolate() returns, interpolator.pyx:650-651.) */
int nin_grid_release_scratch(nin_grid *g);

/* Algorithmic HBM bytes one nin_weights call moves for `method` over all nodes (DESIGN.md formula,
 * SURVEY 8d): used by bench.py for the roofline line. */
int64_t nin_algorithmic_bytes(const nin_grid *g, int method);

/* Name of the dominant kernel of `method` as it appears in rocprofv3 traces. */
const char *nin_kernel_name(int method);

/* The GLS launch plan of a grid on the device: how many nodes each kernel takes.  counts[0..4]: the block kernel's
 * size classes (1 / 2 / 4 / 8 wavefronts per node, then the global-scratch class), counts[5]: the cube-node kernel,
 * counts[6], counts[7], counts[8]: the one-wavefront multifrontal kernel -- two-coloured nodes (large / small
 * instantiation) and the general kind, counts[9], counts[10], counts[11]: the one-wavefront dense kernel for small nodes
 * (at most 4 / 8 / 12 cells; in practice the boundary nodes that are computed), counts[12]: the two-lanes-per-node kernel for
 * the nodes inside a boundary face of a hexahedron mesh, counts[13..17]: the wide one-wavefront multifrontal kernel (interior nodes
 * of unstructured meshes: up to 16 fronts + 21 dense cells) by size class of its dense problem -- at most 96 x 40, 112 x 44,
 * 128 x 52, 144 x 60, 160 x 64 (rows x columns), counts[18]: its list of BOUNDARY nodes (computed only when flagged Neumann),
 * counts[19]: the multifrontal kernel whose dense problem lives in global-memory tiles (interior nodes beyond the wide kernel: up to 32 fronts +
 * 40 dense cells, 256 x 121), counts[20]: the wide kernel's SMALL class (interior nodes of 9 .. 14 cells that are not two-coloured: at most
 * 64 x 28), counts[21]: its class (7, 12) -- at most 112 x 48, between 112 x 44 and 128 x 52.  (Diagnostics and tests: the reference has one code path, gls.pyx:138-197,
 * for every node.) */
int nin_gls_plan(const nin_grid *g, int64_t counts[22]);

/* Measurement: the FP64 flops one GLS launch performs, kernel by kernel of the launch plan (numbered as in
 * nin_gls_plan): alg[k] = ALGORITHMIC flops of the formulation kernel k runs on its nodes (fronts + dense rest for the
 * multifrontal kernels, from each node's own descriptor; one dense Householder QR with the last-row identity for the small-node /
 * one-wavefront block / global-scratch kernels), ref[k] = the reference's dense dgels on the same nodes (gls.pyx:420-474),
 * computed[k] = the nodes that are computed at all (Dirichlet boundary nodes and nodes outside the parity set get the zero row).
 * Needs nin_fields_set (the Neumann flags decide which boundary nodes are computed). */
int nin_gls_plan_flops(nin_grid *g, double alg[22], double ref[22], int64_t computed[22]);

/* ---- multi-GPU: the all-gather of the path as direct peer-to-peer writes ------------------------------------------
 * Replaces nothing in the reference (it is single-process); it is the exchange step north_star adds -- "a single allgatherv to
 * reassemble the COO triplets" -- without a collective library: rank r writes its block straight into slot r of every peer's
 * gathered buffer, one device-to-device copy per peer, each on its own stream (an MI355X has one xGMI link to each of its seven
 * peers: seven copies in flight on seven links; a ring all-gather uses one link at a time).  The library does no rendezvous: the
 * caller exchanges the 64-byte handles by its own means (MPI, torch.distributed, a file) and puts a barrier of its own between
 * "every rank's pushes are complete" and "read the gathered buffer".  INTEGRATION.md shows the call sequence.
 *
 *   nin_exchange_create     a gathered buffer of world slots of slot_bytes (rounded up to 256) on `device`
 *   nin_exchange_handle     this rank's 64-byte IPC handle (hipIpcMemHandle_t) -> handle64
 *   nin_exchange_connect    all_handles: world x 64 bytes, rank-major; opens the peers' buffers (once)
 *   nin_exchange_push       bytes from dev_src (device memory of this rank) -> offset `offset` of slot `rank` in EVERY rank's buffer,
 *                           this rank's own included; asynchronous, ordered behind everything enqueued on `stream` so far
 *   nin_exchange_wait_sent  host_wait = 0: `stream` waits for this rank's pushes; 1: the host does
 *   nin_exchange_buffer     this rank's gathered buffer (device pointer): slot r at r * nin_exchange_slot_bytes() */
#define NIN_EXCHANGE_HANDLE_BYTES 64
typedef struct nin_exchange nin_exchange;
int nin_exchange_create(int device, int rank, int world, size_t slot_bytes, nin_exchange **out);
void nin_exchange_destroy(nin_exchange *x);
int nin_exchange_handle(nin_exchange *x, void *handle64);
int nin_exchange_connect(nin_exchange *x, const void *all_handles);
int nin_exchange_push(nin_exchange *x, const void *dev_src, size_t bytes, size_t offset, void *stream);
int nin_exchange_wait_sent(nin_exchange *x, void *stream, int host_wait);
void *nin_exchange_buffer(nin_exchange *x);
size_t nin_exchange_slot_bytes(const nin_exchange *x);

#ifdef __cplusplus
}
#endif
#endif /* NINPOL_AMD_H */
