// grid_host.cpp -- connectivity + geometry of the mesh, built once on the host (north_star: "the
// Grid connectivity is built once as CSR on host and pushed to HBM").
//
// Same results as the reference's Grid.build() + calculate_centroids() + calculate_normal_faces()
// (ninpol/_interpolator/grid.pyx:142-231, 669-809) on conforming meshes -- bit for bit, integers and
// geometry -- but not the same algorithm: the reference's serial sweeps (esup fill :254-263, face
// numbering :315-334, fsup/esuf transposes :347-416) are replaced by data-parallel formulations whose
// output order is fixed by construction:
//   esup   rows are ascending element ids           -> atomic scatter, then sort each (short) row
//   esuel  neighbour through face j of element e   -> independent search per (e, j), no cross writes
//   faces  id = rank of (creator element, local face) among "owned" pairs, creator = lower element id
//          (what the first-sight sweep of grid.pyx:315-334 produces) -> flag + exclusive scan
//   fsup   rows are ascending face ids              -> gathered per point from its elements' faces
//   esuf   rows are [creator, neighbour]            -> read off the owner table
// This translation unit is compiled with -ffp-contract=off: the reference is built for baseline
// x86-64 (no FMA) and its float32 normals (grid.pyx:732-767) must be reproduced exactly.
#include "grid_host.hpp"
#include "host_threads.hpp"

#include <omp.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <numeric>
#include <unordered_map>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace nin {

namespace {

// Sort by the library's own team: chunks sorted in parallel, then merged pairwise.  (libstdc++'s parallel-mode sort was
// used here in round 2; its multiway_mergesort.h lets every thread of the team store the team size into one shared
// variable without synchronisation -- a data race by the letter, reported by ThreadSanitizer -- and sizes its team from the
// process-wide OpenMP default.)
template <class T>
void team_sort(std::vector<T> &v) {
    const int nt = host_team();
    const size_t n = v.size();
    if (nt <= 1 || n < ((size_t)1 << 16)) { std::sort(v.begin(), v.end()); return; }
    const int chunks = nt;
    std::vector<size_t> cut((size_t)chunks + 1);
    for (int c = 0; c <= chunks; ++c) cut[(size_t)c] = n * (size_t)c / (size_t)chunks;
#pragma omp parallel for schedule(static, 1) num_threads(nt)
    for (int c = 0; c < chunks; ++c) std::sort(v.begin() + (ptrdiff_t)cut[c], v.begin() + (ptrdiff_t)cut[c + 1]);
    for (int w = 1; w < chunks; w *= 2) {
#pragma omp parallel for schedule(static, 1) num_threads(nt)
        for (int c = 0; c < chunks - w; c += 2 * w)
            std::inplace_merge(v.begin() + (ptrdiff_t)cut[c], v.begin() + (ptrdiff_t)cut[c + w],
                               v.begin() + (ptrdiff_t)cut[std::min(c + 2 * w, chunks)]);
    }
}

inline bool elem_has_point(const int32_t *el, int n, int32_t p) {
    for (int i = 0; i < n; ++i)
        if (el[i] == p) return true;
    return false;
}

}  // namespace

int HostGrid::build(const int64_t *connectivity, const int64_t *element_types, const double *xyz, int coords_dim) {
    const int64_t E = n_elems, P = n_points;
    const ScopedTeam team(num_threads);   // this build's OpenMP team (0: the CPUs this process may actually use); the process-wide default is not touched
    const bool timing = getenv("NIN_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[nin_grid] %-10s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };

    inpoel.resize((size_t)E * kMaxPointsPerElement);
    etype.resize((size_t)E);
    bool bad = false;
#pragma omp parallel for schedule(dynamic, 16384) reduction(|| : bad) num_threads(nin::host_team())
    for (int64_t e = 0; e < E; ++e) {
        int64_t t = element_types[e];
        if (t < 0 || t >= kNumElementTypes) { bad = true; t = 0; }
        etype[e] = (int8_t)t;
        for (int j = 0; j < kMaxPointsPerElement; ++j) {
            int64_t p = connectivity[e * kMaxPointsPerElement + j];
            if (j < npoel[t] && (p < 0 || p >= P)) bad = true;
            inpoel[e * kMaxPointsPerElement + j] = (int32_t)p;
        }
    }
    if (bad) return -1;

    coords.assign((size_t)P * 3, 0.0);
#pragma omp parallel for schedule(dynamic, 16384) num_threads(nin::host_team())
    for (int64_t p = 0; p < P; ++p)
        for (int k = 0; k < coords_dim && k < 3; ++k) coords[p * 3 + k] = xyz[p * coords_dim + k];

    lap("ingest");
    // ---- esup (grid.pyx:233-267) ------------------------------------------------------------
    esup_ptr.assign((size_t)P + 1, 0);
    {
        std::vector<std::atomic<int32_t>> cnt((size_t)P);
#pragma omp parallel for schedule(dynamic, 16384) num_threads(nin::host_team())
        for (int64_t p = 0; p < P; ++p) cnt[p].store(0, std::memory_order_relaxed);
#pragma omp parallel for schedule(dynamic, 16384) num_threads(nin::host_team())
        for (int64_t e = 0; e < E; ++e) {
            int n = npoel[etype[e]];
            for (int j = 0; j < n; ++j) cnt[inpoel[e * 8 + j]].fetch_add(1, std::memory_order_relaxed);
        }
        int64_t mx = 0, run = 0;
        for (int64_t p = 0; p < P; ++p) {
            int32_t c = cnt[p].load(std::memory_order_relaxed);
            mx = std::max<int64_t>(mx, c);
            esup_ptr[p] = run;
            run += c;
        }
        esup_ptr[P] = run;
        mx_elems_per_point = mx;
        if (run >= INT32_MAX) return -5;
        esup.resize((size_t)run);
#pragma omp parallel for schedule(dynamic, 16384) num_threads(nin::host_team())
        for (int64_t p = 0; p < P; ++p) cnt[p].store(0, std::memory_order_relaxed);
#pragma omp parallel for schedule(dynamic, 16384) num_threads(nin::host_team())
        for (int64_t e = 0; e < E; ++e) {
            int n = npoel[etype[e]];
            for (int j = 0; j < n; ++j) {
                int32_t p = inpoel[e * 8 + j];
                int32_t at = cnt[p].fetch_add(1, std::memory_order_relaxed);
                esup[esup_ptr[p] + at] = (int32_t)e;
            }
        }
#pragma omp parallel for schedule(dynamic, 16384) num_threads(nin::host_team())
        for (int64_t p = 0; p < P; ++p) std::sort(esup.begin() + esup_ptr[p], esup.begin() + esup_ptr[p + 1]);
    }

    lap("esup");
    // ---- esuel (grid.pyx:449-525) -----------------------------------------------------------
    esuel.assign((size_t)E * kMaxFacesPerElement, -1);
#pragma omp parallel for schedule(dynamic, 16384) num_threads(nin::host_team())
    for (int64_t ie = 0; ie < E; ++ie) {
        const int it = etype[ie];
        const int32_t *el = &inpoel[ie * 8];
        for (int j = 0; j < nfael[it]; ++j) {
            const int nf = lnofa[it][j];
            int32_t fp[kMaxPointsPerFace];
            for (int k = 0; k < nf; ++k) fp[k] = el[lpofa[it][j][k]];
            // the face point with the fewest surrounding elements (first minimum, grid.pyx:479-488)
            int32_t point = fp[0];
            int64_t nmin = esup_ptr[point + 1] - esup_ptr[point];
            for (int k = 0; k < nf; ++k) {
                int64_t n = esup_ptr[fp[k] + 1] - esup_ptr[fp[k]];
                if (n < nmin) { point = fp[k]; nmin = n; }
            }
            int32_t found = -1;
            for (int64_t q = esup_ptr[point]; q < esup_ptr[point + 1] && found < 0; ++q) {
                const int32_t je = esup[q];
                if (je == ie) continue;
                const int jt = etype[je];
                const int32_t *jl = &inpoel[(int64_t)je * 8];
                // cheap reject: a neighbour across this face holds every point of the face
                bool all = true;
                for (int k = 0; k < nf && all; ++k) all = elem_has_point(jl, npoel[jt], fp[k]);
                if (!all) continue;
                for (int l = 0; l < nfael[jt]; ++l) {  // the reference's own face test (:503-512)
                    int is_equal = 0;
                    for (int m = 0; m < lnofa[jt][l]; ++m) {
                        const int32_t jp = jl[lpofa[jt][l][m]];
                        for (int o = 0; o < nf; ++o)
                            if (jp == fp[o]) { ++is_equal; break; }
                    }
                    if (is_equal == nf) { found = je; break; }
                }
            }
            esuel[ie * 6 + j] = found;
        }
    }

    lap("esuel");
    // ---- infael / inpofa: global face numbering (grid.pyx:304-345) --------------------------
    infael.assign((size_t)E * kMaxFacesPerElement, -1);
    std::vector<int64_t> own_start((size_t)E + 1, 0);
#pragma omp parallel for schedule(dynamic, 16384) num_threads(nin::host_team())
    for (int64_t e = 0; e < E; ++e) {
        int c = 0;
        for (int j = 0; j < nfael[etype[e]]; ++j) {
            int32_t k = esuel[e * 6 + j];
            c += (k == -1 || k > e);
        }
        own_start[e + 1] = c;
    }
    for (int64_t e = 0; e < E; ++e) own_start[e + 1] += own_start[e];
    n_faces = own_start[E];
    if (n_faces * 4 >= INT32_MAX) return -5;
    std::vector<int32_t> face_owner((size_t)n_faces);   // creating element
    std::vector<int8_t> face_owner_lf((size_t)n_faces);  // its local face
#pragma omp parallel for schedule(dynamic, 16384) num_threads(nin::host_team())
    for (int64_t e = 0; e < E; ++e) {
        int64_t f = own_start[e];
        for (int j = 0; j < nfael[etype[e]]; ++j) {
            int32_t k = esuel[e * 6 + j];
            if (k == -1 || k > e) {
                infael[e * 6 + j] = (int32_t)f;
                face_owner[f] = (int32_t)e;
                face_owner_lf[f] = (int8_t)j;
                ++f;
            }
        }
    }
#pragma omp parallel for schedule(dynamic, 16384) num_threads(nin::host_team())
    for (int64_t e = 0; e < E; ++e) {
        for (int j = 0; j < nfael[etype[e]]; ++j) {
            int32_t k = esuel[e * 6 + j];
            if (k != -1 && k < e) {  // created by the lower element: mirror its id (first l with esuel[k,l]==e)
                for (int l = 0; l < nfael[etype[k]]; ++l)
                    if (esuel[(int64_t)k * 6 + l] == e) { infael[e * 6 + j] = infael[(int64_t)k * 6 + l]; break; }
            }
        }
    }
    const int64_t F = n_faces;
    inpofa.assign((size_t)F * kMaxPointsPerFace, -1);
#pragma omp parallel for schedule(dynamic, 16384) num_threads(nin::host_team())
    for (int64_t f = 0; f < F; ++f) {
        const int64_t e = face_owner[f];
        const int t = etype[e], j = face_owner_lf[f];
        for (int k = 0; k < lnofa[t][j]; ++k) inpofa[f * 4 + k] = inpoel[e * 8 + lpofa[t][j][k]];
    }

    lap("infael");
    // ---- fsup (grid.pyx:347-379): per point, the ascending unique faces of its elements ------
    fsup_ptr.assign((size_t)P + 1, 0);
    auto gather_faces = [&](int64_t p, int32_t *buf) -> int {
        int n = 0;
        for (int64_t q = esup_ptr[p]; q < esup_ptr[p + 1]; ++q) {
            const int64_t e = esup[q];
            const int t = etype[e];
            for (int j = 0; j < nfael[t]; ++j) {
                bool has = false;
                for (int k = 0; k < lnofa[t][j]; ++k) has |= (inpoel[e * 8 + lpofa[t][j][k]] == p);
                if (has) buf[n++] = infael[e * 6 + j];
            }
        }
        std::sort(buf, buf + n);
        return (int)(std::unique(buf, buf + n) - buf);
    };
    const size_t fbuf = (size_t)mx_elems_per_point * kMaxFacesPerElement + 1;
    int64_t mxf = 0;
#pragma omp parallel num_threads(nin::host_team()) reduction(max : mxf)
    {
        std::vector<int32_t> buf(fbuf);
#pragma omp for schedule(dynamic, 16384)
        for (int64_t p = 0; p < P; ++p) {
            int n = gather_faces(p, buf.data());
            fsup_ptr[p + 1] = n;
            mxf = std::max<int64_t>(mxf, n);
        }
    }
    mx_faces_per_point = mxf;
    for (int64_t p = 0; p < P; ++p) fsup_ptr[p + 1] += fsup_ptr[p];
    if (fsup_ptr[P] >= INT32_MAX) return -5;
    fsup.resize((size_t)fsup_ptr[P]);
#pragma omp parallel num_threads(nin::host_team())
    {
        std::vector<int32_t> buf(fbuf);
#pragma omp for schedule(dynamic, 16384)
        for (int64_t p = 0; p < P; ++p) {
            int n = gather_faces(p, buf.data());
            std::copy(buf.begin(), buf.begin() + n, fsup.begin() + fsup_ptr[p]);
        }
    }

    lap("fsup");
    // ---- esuf, boundary flags (grid.pyx:381-444) --------------------------------------------
    esuf_ptr.assign((size_t)F + 1, 0);
    boundary_faces.assign((size_t)F, 0);
    boundary_points.assign((size_t)P, 0);
    int64_t mxe = 0;
#pragma omp parallel for schedule(dynamic, 16384) reduction(max : mxe) num_threads(nin::host_team())
    for (int64_t f = 0; f < F; ++f) {
        const int32_t nb = esuel[(int64_t)face_owner[f] * 6 + face_owner_lf[f]];
        esuf_ptr[f + 1] = nb == -1 ? 1 : 2;
        mxe = std::max<int64_t>(mxe, esuf_ptr[f + 1]);
    }
    mx_elems_per_face = mxe;
    for (int64_t f = 0; f < F; ++f) esuf_ptr[f + 1] += esuf_ptr[f];
    esuf.resize((size_t)esuf_ptr[F]);
#pragma omp parallel for schedule(dynamic, 16384) num_threads(nin::host_team())
    for (int64_t f = 0; f < F; ++f) {
        const int32_t nb = esuel[(int64_t)face_owner[f] * 6 + face_owner_lf[f]];
        esuf[esuf_ptr[f]] = face_owner[f];
        if (nb != -1) esuf[esuf_ptr[f] + 1] = nb;
        else boundary_faces[f] = 1;
    }
    for (int64_t f = 0; f < F; ++f)
        if (boundary_faces[f])
            for (int k = 0; k < kMaxPointsPerFace && inpofa[f * 4 + k] != -1; ++k) boundary_points[inpofa[f * 4 + k]] = 1;

    lap("esuf");
    // ---- geometry (grid.pyx:669-809) -----------------------------------------------------------
    centroids.assign((size_t)E * 3, 0.0);
    const int d = (int)dim;
#pragma omp parallel for schedule(dynamic, 16384) num_threads(nin::host_team())
    for (int64_t e = 0; e < E; ++e) {
        const int n = npoel[etype[e]];
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < d; ++k) centroids[e * 3 + k] += coords[(int64_t)inpoel[e * 8 + j] * 3 + k] / (double)n;
    }
    faces_centers.assign((size_t)F * 3, 0.0);
    normal_faces.assign((size_t)F * 3, 0.0f);
    faces_areas.assign((size_t)F, 0.0);
    const double *X = coords.data();
#pragma omp parallel for schedule(dynamic, 16384) num_threads(nin::host_team())
    for (int64_t f = 0; f < F; ++f) {
        int npofa = 0;
        for (int j = 0; j < kMaxPointsPerFace && inpofa[f * 4 + j] != -1; ++j) {
            ++npofa;
            for (int k = 0; k < d; ++k) faces_centers[f * 3 + k] += X[(int64_t)inpofa[f * 4 + j] * 3 + k];
        }
        for (int k = 0; k < d; ++k) faces_centers[f * 3 + k] /= (double)npofa;
        const int64_t p1 = inpofa[f * 4 + 0], p2 = inpofa[f * 4 + 1];
        if (d == 3) {
            // float locals exactly as grid.pyx:732-736; the module is C++, so sqrt(float) is sqrtf
            const int64_t p3 = inpofa[f * 4 + 2];
            float v1x = (float)(X[p1 * 3 + 0] - X[p2 * 3 + 0]), v1y = (float)(X[p1 * 3 + 1] - X[p2 * 3 + 1]),
                  v1z = (float)(X[p1 * 3 + 2] - X[p2 * 3 + 2]);
            float v2x = (float)(X[p3 * 3 + 0] - X[p2 * 3 + 0]), v2y = (float)(X[p3 * 3 + 1] - X[p2 * 3 + 1]),
                  v2z = (float)(X[p3 * 3 + 2] - X[p2 * 3 + 2]);
            float nx = v1y * v2z - v1z * v2y, ny = v1z * v2x - v1x * v2z, nz = v1x * v2y - v1y * v2x;
            float norm = fabsf(sqrtf(nx * nx + ny * ny + nz * nz));
            normal_faces[f * 3 + 0] = nx / norm;
            normal_faces[f * 3 + 1] = ny / norm;
            normal_faces[f * 3 + 2] = nz / norm;
            if (inpofa[f * 4 + 3] == -1) {
                faces_areas[f] = (double)norm / 2.0;
            } else {
                const int64_t p4 = inpofa[f * 4 + 3];
                v1x = (float)(X[p1 * 3 + 0] - X[p4 * 3 + 0]); v1y = (float)(X[p1 * 3 + 1] - X[p4 * 3 + 1]);
                v1z = (float)(X[p1 * 3 + 2] - X[p4 * 3 + 2]);
                v2x = (float)(X[p3 * 3 + 0] - X[p4 * 3 + 0]); v2y = (float)(X[p3 * 3 + 1] - X[p4 * 3 + 1]);
                v2z = (float)(X[p3 * 3 + 2] - X[p4 * 3 + 2]);
                nx = v1y * v2z - v1z * v2y; ny = v1z * v2x - v1x * v2z; nz = v1x * v2y - v1y * v2x;
                faces_areas[f] = (double)(norm + sqrtf(nx * nx + ny * ny + nz * nz)) / 2.0;
            }
        } else {
            float v1x = (float)(X[p1 * 3 + 0] - X[p2 * 3 + 0]), v1y = (float)(X[p1 * 3 + 1] - X[p2 * 3 + 1]);
            float nx = -v1y, ny = v1x;
            float norm = fabsf(sqrtf(nx * nx + ny * ny));
            normal_faces[f * 3 + 0] = nx / norm;
            normal_faces[f * 3 + 1] = ny / norm;
            normal_faces[f * 3 + 2] = 0.0f;
            faces_areas[f] = (double)norm;
        }
    }
    lap("geometry");
    nnz_esup = (int64_t)esup.size();
    nnz_fsup = (int64_t)fsup.size();
    if (build_edges) build_inedel();
    return 0;
}

void HostGrid::widen(const std::vector<int32_t> &src, std::vector<int64_t> &dst) {
    const int64_t n = (int64_t)src.size();
    dst.resize(src.size());
#pragma omp parallel for schedule(static) num_threads(nin::host_team())
    for (int64_t i = 0; i < n; ++i) dst[i] = src[i];
}

void HostGrid::esuf_from_pairs(const std::vector<int32_t> &pairs) {
    const int64_t F = (int64_t)pairs.size() / 2;
    esuf_ptr.resize((size_t)F + 1);
    esuf_ptr[0] = 0;
    for (int64_t f = 0; f < F; ++f) esuf_ptr[f + 1] = esuf_ptr[f] + (pairs[2 * f + 1] == -1 ? 1 : 2);
    esuf.resize((size_t)esuf_ptr[F]);
#pragma omp parallel for schedule(static) num_threads(nin::host_team())
    for (int64_t f = 0; f < F; ++f) {
        esuf[esuf_ptr[f]] = pairs[2 * f];
        if (pairs[2 * f + 1] != -1) esuf[esuf_ptr[f] + 1] = pairs[2 * f + 1];
    }
}

// grid.pyx:269-302: unique neighbour points in order of first encounter (per point independent).
void HostGrid::build_psup() {
    if (psup_built) return;
    ensure(A_INPOEL | A_ETYPE | A_ESUP_PTR | A_ESUP);
    const int64_t P = n_points;
    psup_ptr.assign((size_t)P + 1, 0);
    auto gather = [&](int64_t p, std::vector<int32_t> &buf) {
        buf.clear();
        for (int64_t q = esup_ptr[p]; q < esup_ptr[p + 1]; ++q) {
            const int64_t e = esup[q];
            for (int k = 0; k < npoel[etype[e]]; ++k) {
                const int32_t r = inpoel[e * 8 + k];
                if (r != p && std::find(buf.begin(), buf.end(), r) == buf.end()) buf.push_back(r);
            }
        }
    };
    int64_t mx = 0;
#pragma omp parallel num_threads(nin::host_team()) reduction(max : mx)
    {
        std::vector<int32_t> buf;
#pragma omp for schedule(dynamic, 16384)
        for (int64_t p = 0; p < P; ++p) {
            gather(p, buf);
            psup_ptr[p + 1] = (int64_t)buf.size();
            mx = std::max<int64_t>(mx, (int64_t)buf.size());
        }
    }
    mx_points_per_point = mx;
    for (int64_t p = 0; p < P; ++p) psup_ptr[p + 1] += psup_ptr[p];
    psup.resize((size_t)psup_ptr[P]);
#pragma omp parallel num_threads(nin::host_team())
    {
        std::vector<int32_t> buf;
#pragma omp for schedule(dynamic, 16384)
        for (int64_t p = 0; p < P; ++p) {
            gather(p, buf);
            std::copy(buf.begin(), buf.end(), psup.begin() + psup_ptr[p]);
        }
    }
    psup_built = true;
}

// grid.pyx:29-43 myhash + :527-580 build_inedel.  The reference keys its map on the HASH of the sorted
// edge, truncated to C int (`unordered_map[int, int]`, :539) -- two edges that collide share one edge id.
// That is the contract, so the same hash and the same truncation are used here.
void HostGrid::build_inedel() {
    if (edges_built) return;
    ensure(A_INPOEL | A_ETYPE);
    const int64_t E = n_elems;
    inedel.assign((size_t)E * kMaxEdgesPerElement, -1);
    inpoed.clear();
    auto myhash = [](const int64_t *vec, int len) -> size_t {
        size_t seed = (size_t)len;
        for (int i = 0; i < len; ++i) {
            int x = (int)vec[i];
            x = (int)((unsigned)((x >> 16) ^ x) * 0x45d9f3bu);
            x = (int)((unsigned)((x >> 16) ^ x) * 0x45d9f3bu);
            x = (x >> 16) ^ x;
            seed ^= (size_t)((unsigned)x + 0x9e3779b9u) + (seed << 6) + (seed >> 2);
        }
        return seed;
    };
    // The reference walks (element, local edge) in order and gives a new id to every key it has not seen
    // (grid.pyx:527-580).  Same ids without the serial map: position of every pair in that walk, sort (key, position),
    // the head of each key group is its first sight, and the id is the rank of that first sight among all of them.
    std::vector<int64_t> start((size_t)E + 1, 0);
    for (int64_t i = 0; i < E; ++i) start[i + 1] = start[i] + nedel[etype[i]];
    const int64_t N = start[E];
    if (N >= (int64_t)1 << 32) {   // positions no longer fit the packed sort key: the reference's own serial walk
        std::unordered_map<int, int> dict;
        for (int64_t i = 0; i < E; ++i) {
            const int t = etype[i];
            for (int j = 0; j < nedel[t]; ++j) {
                int64_t ed[2] = {inpoel[i * 8 + lpoed[t][j][0]], inpoel[i * 8 + lpoed[t][j][1]]};
                int64_t sd[2] = {ed[0], ed[1]};
                if (ed[0] > ed[1]) std::swap(sd[0], sd[1]);
                const int key = (int)myhash(sd, 2);
                auto it = dict.find(key);
                int idx;
                if (it == dict.end()) {
                    idx = (int)dict.size();
                    dict.emplace(key, idx);
                    inpoed.push_back((int32_t)ed[0]);
                    inpoed.push_back((int32_t)ed[1]);
                } else {
                    idx = it->second;
                }
                inedel[i * kMaxEdgesPerElement + j] = idx;
            }
        }
        n_edges = (int64_t)dict.size();
        edges_built = true;
        return;
    }
    std::vector<uint64_t> kv((size_t)N);   // (key as unsigned 32 bits) << 32 | position in the walk
#pragma omp parallel for schedule(dynamic, 16384) num_threads(nin::host_team())
    for (int64_t i = 0; i < E; ++i) {
        const int t = etype[i];
        for (int j = 0; j < nedel[t]; ++j) {
            int64_t sd[2] = {inpoel[i * 8 + lpoed[t][j][0]], inpoel[i * 8 + lpoed[t][j][1]]};
            if (sd[0] > sd[1]) std::swap(sd[0], sd[1]);
            const uint32_t key = (uint32_t)(int)myhash(sd, 2);
            kv[(size_t)(start[i] + j)] = ((uint64_t)key << 32) | (uint64_t)(start[i] + j);
        }
    }
    team_sort(kv);
    std::vector<uint64_t> firsts;          // position of the first sight of every key, then sorted ascending
    firsts.reserve((size_t)N / 3 + 16);
    for (int64_t q = 0; q < N; ++q)
        if (q == 0 || (kv[q] >> 32) != (kv[q - 1] >> 32)) firsts.push_back(kv[q] & 0xffffffffu);
    std::vector<uint64_t> by_pos(firsts);
    team_sort(by_pos);
    n_edges = (int64_t)by_pos.size();
    // element / local edge of a walk position (binary search over the per-element starts)
    auto locate = [&](int64_t pos, int64_t *e, int *j) {
        const int64_t i = (std::upper_bound(start.begin(), start.end(), pos) - start.begin()) - 1;
        *e = i;
        *j = (int)(pos - start[i]);
    };
    inpoed.resize((size_t)n_edges * 2);
#pragma omp parallel for schedule(static) num_threads(nin::host_team())
    for (int64_t id = 0; id < n_edges; ++id) {
        int64_t e; int j;
        locate((int64_t)by_pos[id], &e, &j);
        const int t = etype[e];
        inpoed[2 * id] = inpoel[e * 8 + lpoed[t][j][0]];       // the endpoints as first met, unsorted (grid.pyx:566-570)
        inpoed[2 * id + 1] = inpoel[e * 8 + lpoed[t][j][1]];
    }
#pragma omp parallel for schedule(dynamic, 65536) num_threads(nin::host_team())
    for (int64_t q = 0; q < N; ++q) {
        int64_t h0 = q;                                          // head of this key's group: walk back (groups are short)
        while (h0 > 0 && (kv[h0 - 1] >> 32) == (kv[q] >> 32)) --h0;
        const uint64_t first = kv[h0] & 0xffffffffu;
        const int64_t id = std::lower_bound(by_pos.begin(), by_pos.end(), first) - by_pos.begin();
        int64_t e; int j;
        locate((int64_t)(kv[q] & 0xffffffffu), &e, &j);
        inedel[e * kMaxEdgesPerElement + j] = (int32_t)id;
    }
    edges_built = true;
}

}  // namespace nin
