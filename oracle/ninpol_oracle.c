/*
 * ninpol_oracle.c -- CPU restatement of the reference hot path (daviyan5/ninpol v1.0.2).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  Nothing under ninpol_amd/ links, imports or calls it.
 *
 * Parity is PINNED: every function here is validated in tests/test_oracle.py against the
 * reference's own compiled Grid / IDW / LS / GLS (oracle/_ref, built by oracle/build_ref.py from the
 * sources under /root/reference) on generated meshes, and against the golden fixtures in
 * tests/golden/ that were produced by that same reference build (tests/golden/make_golden.py).
 *
 * Each function cites the reference file:line it restates.  Plain C99, serial where the reference
 * is serial; the three method loops carry an OpenMP pragma like the reference's prange.
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/build_oracle.py); contraction
 * is off because the reference is built for baseline x86-64 (no FMA) and the float32 normals of
 * grid.pyx:721-809 must come out bit-identical.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef long long i64;

#define MAX_POINTS_PER_ELEMENT 8 /* ninpol_defines.pxd:2 */
#define MAX_FACES_PER_ELEMENT 6  /* ninpol_defines.pxd:3 */
#define MAX_POINTS_PER_FACE 4    /* ninpol_defines.pxd:4 */
#define NUM_ELEMENT_TYPES 8      /* ninpol_defines.pxd:5 */

typedef struct {
    i64 dim, n_elems, n_points, n_faces;
    i64 MX_ELEMENTS_PER_POINT, MX_POINTS_PER_POINT, MX_ELEMENTS_PER_FACE, MX_FACES_PER_POINT;
    /* topology tables (ctor args, grid.pyx:47-53) */
    i64 npoel[NUM_ELEMENT_TYPES], nfael[NUM_ELEMENT_TYPES];
    i64 lnofa[NUM_ELEMENT_TYPES][MAX_FACES_PER_ELEMENT];
    i64 lpofa[NUM_ELEMENT_TYPES][MAX_FACES_PER_ELEMENT][MAX_POINTS_PER_FACE];
    i64 *inpoel;        /* [E][8]  */
    i64 *element_types; /* [E]     */
    i64 *esup_ptr, *esup, *psup_ptr, *psup, *fsup_ptr, *fsup, *esuf_ptr, *esuf;
    i64 n_psup;
    i64 *esuel, *infael; /* [E][6] */
    i64 *inpofa;         /* [F][4] */
    i64 *boundary_faces, *boundary_points;
    double *point_coords, *centroids, *faces_centers, *normal_faces, *faces_areas;
} oracle_grid;

#define INPOEL(g, e, j) ((g)->inpoel[(e) * MAX_POINTS_PER_ELEMENT + (j)])
#define ESUEL(g, e, j) ((g)->esuel[(e) * MAX_FACES_PER_ELEMENT + (j)])
#define INFAEL(g, e, j) ((g)->infael[(e) * MAX_FACES_PER_ELEMENT + (j)])
#define INPOFA(g, f, j) ((g)->inpofa[(f) * MAX_POINTS_PER_FACE + (j)])

static i64 *alloc_i64(i64 n, i64 fill) {
    i64 *p = (i64 *)malloc(sizeof(i64) * (size_t)(n > 0 ? n : 1));
    for (i64 i = 0; i < n; ++i) p[i] = fill;
    return p;
}
static double *alloc_f64(i64 n) { return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }

/* grid.pyx:233-267 build_esup: count, prefix sum, fill in element order, shift pointers back */
static void build_esup(oracle_grid *g) {
    g->esup_ptr = alloc_i64(g->n_points + 1, 0);
    for (i64 i = 0; i < g->n_elems; ++i) {
        i64 t = g->element_types[i];
        for (i64 j = 0; j < g->npoel[t]; ++j) {
            i64 p = INPOEL(g, i, j);
            g->esup_ptr[p + 1] += 1;
            if (g->esup_ptr[p + 1] > g->MX_ELEMENTS_PER_POINT) g->MX_ELEMENTS_PER_POINT = g->esup_ptr[p + 1];
        }
    }
    for (i64 i = 0; i < g->n_points; ++i) g->esup_ptr[i + 1] += g->esup_ptr[i];
    g->esup = alloc_i64(g->esup_ptr[g->n_points], 0);
    for (i64 i = 0; i < g->n_elems; ++i) {
        i64 t = g->element_types[i];
        for (i64 j = 0; j < g->npoel[t]; ++j) {
            i64 p = INPOEL(g, i, j);
            g->esup[g->esup_ptr[p]] = i;
            g->esup_ptr[p] += 1;
        }
    }
    for (i64 i = g->n_points; i > 0; --i) g->esup_ptr[i] = g->esup_ptr[i - 1];
    g->esup_ptr[0] = 0;
}

/* grid.pyx:269-302 build_psup: unique neighbours in order of first encounter (marker array) */
static void build_psup(oracle_grid *g) {
    i64 *mark = alloc_i64(g->n_points, -1);
    g->psup_ptr = alloc_i64(g->n_points + 1, 0);
    g->psup = alloc_i64(g->esup_ptr[g->n_points] * (MAX_POINTS_PER_ELEMENT - 1), 0);
    i64 stor = 0;
    for (i64 i = 0; i < g->n_points; ++i) {
        for (i64 j = g->esup_ptr[i]; j < g->esup_ptr[i + 1]; ++j) {
            i64 e = g->esup[j], t = g->element_types[e];
            for (i64 k = 0; k < g->npoel[t]; ++k) {
                i64 q = INPOEL(g, e, k);
                if (q != i && mark[q] != i) {
                    g->psup[stor++] = q;
                    mark[q] = i;
                }
            }
        }
        g->psup_ptr[i + 1] = stor;
        i64 d = g->psup_ptr[i + 1] - g->psup_ptr[i];
        if (d > g->MX_POINTS_PER_POINT) g->MX_POINTS_PER_POINT = d;
    }
    g->n_psup = stor;
    free(mark);
}

/* grid.pyx:449-525 build_esuel (serial here; the reference's prange writes the same values on a
 * conforming mesh): for each element face pick the face point with fewest surrounding elements and
 * scan that point's elements for a face holding every point of this face. */
static void build_esuel(oracle_grid *g) {
    g->esuel = alloc_i64(g->n_elems * MAX_FACES_PER_ELEMENT, -1);
    for (i64 ie = 0; ie < g->n_elems; ++ie) {
        i64 it = g->element_types[ie];
        for (i64 j = 0; j < g->nfael[it]; ++j) {
            if (ESUEL(g, ie, j) != -1) continue;
            i64 point = INPOEL(g, ie, g->lpofa[it][j][0]);
            i64 nmin = g->esup_ptr[point + 1] - g->esup_ptr[point];
            for (i64 k = 0; k < g->lnofa[it][j]; ++k) {
                i64 kp = INPOEL(g, ie, g->lpofa[it][j][k]);
                i64 n = g->esup_ptr[kp + 1] - g->esup_ptr[kp];
                if (n < nmin) { point = kp; nmin = n; }
            }
            int found = 0;
            for (i64 k = g->esup_ptr[point]; k < g->esup_ptr[point + 1]; ++k) {
                i64 je = g->esup[k], jt = g->element_types[je];
                if (je != ie) {
                    for (i64 l = 0; l < g->nfael[jt]; ++l) {
                        i64 is_equal = 0;
                        for (i64 m = 0; m < g->lnofa[jt][l]; ++m) {
                            i64 jp = INPOEL(g, je, g->lpofa[jt][l][m]);
                            for (i64 o = 0; o < g->lnofa[it][j]; ++o) {
                                if (jp == INPOEL(g, ie, g->lpofa[it][j][o])) { is_equal += 1; break; }
                            }
                        }
                        if (is_equal == g->lnofa[it][j]) {
                            ESUEL(g, ie, j) = je;
                            ESUEL(g, je, l) = ie;
                            found = 1;
                        }
                        if (found) break;
                    }
                }
                if (found) break;
            }
        }
    }
}

/* grid.pyx:304-345 build_infael: serial sweep assigning global face ids on first sight */
static void build_infael(oracle_grid *g) {
    i64 ub = g->n_elems * MAX_FACES_PER_ELEMENT;
    i64 *f2e = alloc_i64(ub * 2, -1);
    i64 face_index = 0;
    g->infael = alloc_i64(g->n_elems * MAX_FACES_PER_ELEMENT, -1);
    for (i64 i = 0; i < g->n_elems; ++i) {
        i64 it = g->element_types[i];
        for (i64 j = 0; j < g->nfael[it]; ++j) {
            if (INFAEL(g, i, j) != -1) continue;
            INFAEL(g, i, j) = face_index++;
            f2e[INFAEL(g, i, j) * 2 + 0] = i;
            f2e[INFAEL(g, i, j) * 2 + 1] = j;
            i64 k = ESUEL(g, i, j);
            if (k == -1) continue;
            i64 kt = g->element_types[k];
            for (i64 l = 0; l < g->nfael[kt]; ++l) {
                if (ESUEL(g, k, l) == i) { INFAEL(g, k, l) = INFAEL(g, i, j); break; }
            }
        }
    }
    g->n_faces = face_index;
    g->inpofa = alloc_i64(g->n_faces * MAX_POINTS_PER_FACE, -1);
    for (i64 f = 0; f < g->n_faces; ++f) {
        i64 i = f2e[f * 2], j = f2e[f * 2 + 1], it = g->element_types[i];
        for (i64 k = 0; k < g->lnofa[it][j]; ++k) INPOFA(g, f, k) = INPOEL(g, i, g->lpofa[it][j][k]);
    }
    free(f2e);
}

/* grid.pyx:347-379 build_fsup: CSR transpose of inpofa */
static void build_fsup(oracle_grid *g) {
    g->fsup_ptr = alloc_i64(g->n_points + 1, 0);
    for (i64 i = 0; i < g->n_faces; ++i)
        for (i64 j = 0; j < MAX_POINTS_PER_FACE; ++j) {
            if (INPOFA(g, i, j) == -1) break;
            i64 p = INPOFA(g, i, j);
            g->fsup_ptr[p + 1] += 1;
            if (g->fsup_ptr[p + 1] > g->MX_FACES_PER_POINT) g->MX_FACES_PER_POINT = g->fsup_ptr[p + 1];
        }
    for (i64 i = 0; i < g->n_points; ++i) g->fsup_ptr[i + 1] += g->fsup_ptr[i];
    g->fsup = alloc_i64(g->fsup_ptr[g->n_points], 0);
    for (i64 i = 0; i < g->n_faces; ++i)
        for (i64 j = 0; j < MAX_POINTS_PER_FACE; ++j) {
            if (INPOFA(g, i, j) == -1) break;
            i64 p = INPOFA(g, i, j);
            g->fsup[g->fsup_ptr[p]] = i;
            g->fsup_ptr[p] += 1;
        }
    for (i64 i = g->n_points; i > 0; --i) g->fsup_ptr[i] = g->fsup_ptr[i - 1];
    g->fsup_ptr[0] = 0;
}

/* grid.pyx:381-444 build_esuf: CSR elements-per-face, inpofa re-derived from the first element,
 * boundary_faces = (count == 1), boundary_points = points of boundary faces */
static void build_esuf(oracle_grid *g) {
    g->esuf_ptr = alloc_i64(g->n_faces + 1, 0);
    for (i64 i = 0; i < g->n_elems; ++i) {
        i64 t = g->element_types[i];
        for (i64 j = 0; j < g->nfael[t]; ++j) {
            i64 f = INFAEL(g, i, j);
            g->esuf_ptr[f + 1] += 1;
            if (g->esuf_ptr[f + 1] > g->MX_ELEMENTS_PER_FACE) g->MX_ELEMENTS_PER_FACE = g->esuf_ptr[f + 1];
        }
    }
    for (i64 i = 0; i < g->n_faces; ++i) g->esuf_ptr[i + 1] += g->esuf_ptr[i];
    g->esuf = alloc_i64(g->esuf_ptr[g->n_faces], 0);
    for (i64 i = 0; i < g->n_elems; ++i) {
        i64 t = g->element_types[i];
        for (i64 j = 0; j < g->nfael[t]; ++j) {
            i64 f = INFAEL(g, i, j);
            g->esuf[g->esuf_ptr[f]] = i;
            g->esuf_ptr[f] += 1;
        }
    }
    for (i64 i = g->n_faces; i > 0; --i) g->esuf_ptr[i] = g->esuf_ptr[i - 1];
    g->esuf_ptr[0] = 0;
    for (i64 f = 0; f < g->n_faces; ++f) {
        i64 e = g->esuf[g->esuf_ptr[f]];
        if (e != -1) {
            i64 t = g->element_types[e], j;
            for (j = 0; j < g->nfael[t]; ++j)
                if (INFAEL(g, e, j) == f) break;
            for (i64 k = 0; k < g->lnofa[t][j]; ++k) INPOFA(g, f, k) = INPOEL(g, e, g->lpofa[t][j][k]);
        }
    }
    g->boundary_faces = alloc_i64(g->n_faces, 0);
    g->boundary_points = alloc_i64(g->n_points, 0);
    for (i64 i = 0; i < g->n_faces; ++i) {
        if (g->esuf_ptr[i + 1] - g->esuf_ptr[i] == 1) {
            g->boundary_faces[i] = 1;
            for (i64 j = 0; j < MAX_POINTS_PER_FACE; ++j) {
                i64 p = INPOFA(g, i, j);
                if (p == -1) break;
                g->boundary_points[p] = 1;
            }
        }
    }
}

/* grid.pyx:669-719 calculate_centroids: centroid = sum_j x_j / npoel (divide, then add, vertex
 * order); face centre = (sum_j x_j) / npofa.  Only the first `dim` coordinates are touched. */
static void calculate_centroids(oracle_grid *g) {
    g->centroids = alloc_f64(g->n_elems * 3);
    for (i64 i = 0; i < g->n_elems; ++i) {
        i64 t = g->element_types[i], n = g->npoel[t];
        for (i64 j = 0; j < n; ++j)
            for (i64 k = 0; k < g->dim; ++k)
                g->centroids[i * 3 + k] += g->point_coords[INPOEL(g, i, j) * 3 + k] / (double)n;
    }
    g->faces_centers = alloc_f64(g->n_faces * 3);
    for (i64 i = 0; i < g->n_faces; ++i) {
        i64 npofa = 0;
        for (i64 j = 0; j < MAX_POINTS_PER_FACE; ++j) {
            if (INPOFA(g, i, j) == -1) break;
            npofa += 1;
            for (i64 k = 0; k < g->dim; ++k)
                g->faces_centers[i * 3 + k] += g->point_coords[INPOFA(g, i, j) * 3 + k];
        }
        for (i64 k = 0; k < g->dim; ++k) g->faces_centers[i * 3 + k] /= (double)npofa;
    }
}

/* grid.pyx:721-809 calculate_normal_faces.  v1*, v2*, normal*, norm are C `float` in the
 * reference (:732-736): the coordinate differences are taken in double and truncated on
 * assignment, the cross product, the sum of squares, the square root (C++ std::sqrt(float), the
 * module is built as C++) and the division are all float operations. */
static void calculate_normal_faces(oracle_grid *g) {
    g->normal_faces = alloc_f64(g->n_faces * 3);
    g->faces_areas = alloc_f64(g->n_faces);
    const double *X = g->point_coords;
    if (g->dim == 3) {
        for (i64 f = 0; f < g->n_faces; ++f) {
            int npofa = INPOFA(g, f, 3) == -1 ? 3 : 4;
            i64 p1 = INPOFA(g, f, 0), p2 = INPOFA(g, f, 1), p3 = INPOFA(g, f, 2);
            float v1x = (float)(X[p1 * 3 + 0] - X[p2 * 3 + 0]);
            float v1y = (float)(X[p1 * 3 + 1] - X[p2 * 3 + 1]);
            float v1z = (float)(X[p1 * 3 + 2] - X[p2 * 3 + 2]);
            float v2x = (float)(X[p3 * 3 + 0] - X[p2 * 3 + 0]);
            float v2y = (float)(X[p3 * 3 + 1] - X[p2 * 3 + 1]);
            float v2z = (float)(X[p3 * 3 + 2] - X[p2 * 3 + 2]);
            float nx = v1y * v2z - v1z * v2y;
            float ny = v1z * v2x - v1x * v2z;
            float nz = v1x * v2y - v1y * v2x;
            float norm = sqrtf(nx * nx + ny * ny + nz * nz); /* C++ float overload, see below */
            norm = fabsf(norm);
            g->normal_faces[f * 3 + 0] = (double)(nx / norm);
            g->normal_faces[f * 3 + 1] = (double)(ny / norm);
            g->normal_faces[f * 3 + 2] = (double)(nz / norm);
            if (npofa == 3) {
                g->faces_areas[f] = (double)norm / 2.0;
            } else {
                i64 p4 = INPOFA(g, f, 3);
                v1x = (float)(X[p1 * 3 + 0] - X[p4 * 3 + 0]);
                v1y = (float)(X[p1 * 3 + 1] - X[p4 * 3 + 1]);
                v1z = (float)(X[p1 * 3 + 2] - X[p4 * 3 + 2]);
                v2x = (float)(X[p3 * 3 + 0] - X[p4 * 3 + 0]);
                v2y = (float)(X[p3 * 3 + 1] - X[p4 * 3 + 1]);
                v2z = (float)(X[p3 * 3 + 2] - X[p4 * 3 + 2]);
                nx = v1y * v2z - v1z * v2y;
                ny = v1z * v2x - v1x * v2z;
                nz = v1x * v2y - v1y * v2x;
                /* grid.pyx is compiled as C++ (setup.py:34): sqrt(float) is the float overload, so the
                 * sum norm + sqrt(..) is a float sum; only the division by 2.0 is double */
                g->faces_areas[f] = (double)(norm + sqrtf(nx * nx + ny * ny + nz * nz)) / 2.0;
            }
        }
    } else {
        for (i64 f = 0; f < g->n_faces; ++f) {
            i64 p1 = INPOFA(g, f, 0), p2 = INPOFA(g, f, 1);
            float v1x = (float)(X[p1 * 3 + 0] - X[p2 * 3 + 0]);
            float v1y = (float)(X[p1 * 3 + 1] - X[p2 * 3 + 1]);
            float nx = -v1y, ny = v1x;
            float norm = sqrtf(nx * nx + ny * ny);
            norm = fabsf(norm);
            g->normal_faces[f * 3 + 0] = (double)(nx / norm);
            g->normal_faces[f * 3 + 1] = (double)(ny / norm);
            g->normal_faces[f * 3 + 2] = 0.0;
            g->faces_areas[f] = (double)norm;
        }
    }
}

/* Grid(*args); build(); load_point_coords(); calculate_centroids(); calculate_normal_faces()
 * -- interpolator.pyx:194,204-207 / grid.pyx:142-231.  coords is (P,3). */
oracle_grid *oracle_grid_create(i64 dim, i64 n_elems, i64 n_points, const i64 *npoel, const i64 *nfael,
                                const i64 *lnofa, const i64 *lpofa, const i64 *connectivity,
                                const i64 *element_types, const double *coords) {
    oracle_grid *g = (oracle_grid *)calloc(1, sizeof(oracle_grid));
    g->dim = dim; g->n_elems = n_elems; g->n_points = n_points;
    memcpy(g->npoel, npoel, sizeof(g->npoel));
    memcpy(g->nfael, nfael, sizeof(g->nfael));
    memcpy(g->lnofa, lnofa, sizeof(g->lnofa));
    memcpy(g->lpofa, lpofa, sizeof(g->lpofa));
    g->inpoel = alloc_i64(n_elems * MAX_POINTS_PER_ELEMENT, 0);
    memcpy(g->inpoel, connectivity, sizeof(i64) * (size_t)(n_elems * MAX_POINTS_PER_ELEMENT));
    g->element_types = alloc_i64(n_elems, 0);
    memcpy(g->element_types, element_types, sizeof(i64) * (size_t)n_elems);
    g->point_coords = alloc_f64(n_points * 3);
    memcpy(g->point_coords, coords, sizeof(double) * (size_t)(n_points * 3));
    build_esup(g);
    build_psup(g);
    build_esuel(g);
    build_infael(g);
    build_fsup(g);
    build_esuf(g);
    calculate_centroids(g);
    calculate_normal_faces(g);
    return g;
}

void oracle_grid_destroy(oracle_grid *g) {
    if (!g) return;
    free(g->inpoel); free(g->element_types); free(g->esup_ptr); free(g->esup); free(g->psup_ptr);
    free(g->psup); free(g->fsup_ptr); free(g->fsup); free(g->esuf_ptr); free(g->esuf); free(g->esuel);
    free(g->infael); free(g->inpofa); free(g->boundary_faces); free(g->boundary_points);
    free(g->point_coords); free(g->centroids); free(g->faces_centers); free(g->normal_faces);
    free(g->faces_areas); free(g);
}

/* name -> (pointer, element count); is_float tells the element type (i64 / double) */
int oracle_grid_get(const oracle_grid *g, const char *name, const void **ptr, i64 *count, int *is_float) {
    *is_float = 0;
#define RET(n, p, c) if (!strcmp(name, n)) { *ptr = (p); *count = (c); return 0; }
#define RETF(n, p, c) if (!strcmp(name, n)) { *ptr = (p); *count = (c); *is_float = 1; return 0; }
    RET("esup_ptr", g->esup_ptr, g->n_points + 1) RET("esup", g->esup, g->esup_ptr[g->n_points])
    RET("psup_ptr", g->psup_ptr, g->n_points + 1) RET("psup", g->psup, g->n_psup)
    RET("fsup_ptr", g->fsup_ptr, g->n_points + 1) RET("fsup", g->fsup, g->fsup_ptr[g->n_points])
    RET("esuf_ptr", g->esuf_ptr, g->n_faces + 1) RET("esuf", g->esuf, g->esuf_ptr[g->n_faces])
    RET("esuel", g->esuel, g->n_elems * MAX_FACES_PER_ELEMENT)
    RET("infael", g->infael, g->n_elems * MAX_FACES_PER_ELEMENT)
    RET("inpofa", g->inpofa, g->n_faces * MAX_POINTS_PER_FACE)
    RET("inpoel", g->inpoel, g->n_elems * MAX_POINTS_PER_ELEMENT)
    RET("boundary_faces", g->boundary_faces, g->n_faces) RET("boundary_points", g->boundary_points, g->n_points)
    RETF("point_coords", g->point_coords, g->n_points * 3) RETF("centroids", g->centroids, g->n_elems * 3)
    RETF("faces_centers", g->faces_centers, g->n_faces * 3) RETF("normal_faces", g->normal_faces, g->n_faces * 3)
    RETF("faces_areas", g->faces_areas, g->n_faces)
#undef RET
#undef RETF
    return -1;
}

i64 oracle_grid_scalar(const oracle_grid *g, const char *name) {
#define S(n, v) if (!strcmp(name, n)) return (v);
    S("dim", g->dim) S("n_elems", g->n_elems) S("n_points", g->n_points) S("n_faces", g->n_faces)
    S("MX_ELEMENTS_PER_POINT", g->MX_ELEMENTS_PER_POINT) S("MX_POINTS_PER_POINT", g->MX_POINTS_PER_POINT)
    S("MX_ELEMENTS_PER_FACE", g->MX_ELEMENTS_PER_FACE) S("MX_FACES_PER_POINT", g->MX_FACES_PER_POINT)
#undef S
    return -1;
}

/* ------------------------------------------------------------------------------------------------
 * Methods.  Shared convention (interpolator.pyx:631-665): weights is the dense pre-zeroed
 * [n_points][MX_ELEMENTS_PER_POINT] table indexed by POINT id (idw.pyx:72-84, ls.pyx:98-135,
 * gls.pyx:467-472 all write weights[point, ...]), neumann_ws is [n_points].  neumann_point is the
 * points_data flag row cast to integer (idw.pyx:28).
 * ---------------------------------------------------------------------------------------------- */

/* idw.pyx:35-84 inverse_distance */
void oracle_idw(const oracle_grid *g, const i64 *targets, i64 n_target, const i64 *neumann_point,
                double *weights, int num_threads) {
    const i64 W = g->MX_ELEMENTS_PER_POINT;
    const int dim = (int)g->dim;
    /* idw.pyx:53: `float machine_epsilon = 10 ** int(np.log10(np.finfo(np.float64).eps))` = (float)1e-15 */
    const float machine_epsilon = (float)1e-15;
    (void)num_threads;
#pragma omp parallel for schedule(static) num_threads(num_threads)
    for (i64 d = 0; d < n_target; ++d) {
        i64 point = targets[d];
        int zero_found = 0;
        double total = 0.0;
        i64 n_source = 0;
        if (g->boundary_points[point] && !neumann_point[point]) continue;
        double *w = weights + point * W;
        const double *xt = g->point_coords + point * 3; /* target_coordinates[dest_idx] (idw.pyx:24) */
        for (i64 jj = g->esup_ptr[point], j = 0; jj < g->esup_ptr[point + 1]; ++jj, ++j) {
            i64 s = g->esup[jj];
            double dist = 0.0;
            for (int k = 0; k < dim; ++k) {
                double df = xt[k] - g->centroids[s * 3 + k];
                dist = dist + df * df;
            }
            if (dist <= (double)machine_epsilon) {
                zero_found = 1;
                for (i64 k = 0; k < n_source; ++k) w[k] = 0.0;
                w[j] = 1.0;
                break;
            }
            dist = sqrt(dist);
            w[j] += 1 / dist;
            total += 1 / dist;
            n_source += 1;
        }
        if (!zero_found)
            for (i64 k = 0; k < n_source; ++k) w[k] /= total;
    }
}

/* ls.pyx:33-135 LS */
void oracle_ls(const oracle_grid *g, const i64 *targets, i64 n_target, const i64 *neumann_point,
               double *weights, int num_threads) {
    const i64 W = g->MX_ELEMENTS_PER_POINT;
    (void)num_threads;
#pragma omp parallel for schedule(static) num_threads(num_threads)
    for (i64 idx = 0; idx < n_target; ++idx) {
        i64 point = targets[idx];
        if (g->boundary_points[point] && !neumann_point[point]) continue;
        double Ix = 0, Iy = 0, Iz = 0, Ixx = 0, Ixy = 0, Ixz = 0, Iyy = 0, Iyz = 0, Izz = 0;
        i64 b = g->esup_ptr[point], e = g->esup_ptr[point + 1];
        i64 n_vols = e - b;
        const double *xp = g->point_coords + point * 3;
        double *w = weights + point * W;
        for (i64 q = b; q < e; ++q) {
            i64 vol = g->esup[q];
            double vx = g->centroids[vol * 3 + 0] - xp[0];
            double vy = g->centroids[vol * 3 + 1] - xp[1];
            double vz = g->centroids[vol * 3 + 2] - xp[2];
            Ix = Ix + vx; Iy = Iy + vy; Iz = Iz + vz;
            Ixx = Ixx + vx * vx; Ixy = Ixy + vx * vy; Ixz = Ixz + vx * vz;
            Iyy = Iyy + vy * vy; Iyz = Iyz + vy * vz; Izz = Izz + vz * vz;
        }
        if (Iz == 0.0 && Izz == 0.0 && Ixz == 0.0 && Iyz == 0.0) Izz = 1.0;
        double D = (Ixx * (Iyy * Izz - Iyz * Iyz) + Ixy * (Iyz * Ixz - Ixy * Izz) + Ixz * (Ixy * Iyz - Iyy * Ixz));
        if (D == 0.0) {
            double total = 0.0;
            for (i64 q = b, i = 0; q < e; ++q, ++i) {
                i64 vol = g->esup[q];
                double vx = g->centroids[vol * 3 + 0] - xp[0];
                double vy = g->centroids[vol * 3 + 1] - xp[1];
                double vz = g->centroids[vol * 3 + 2] - xp[2];
                w[i] = 1.0 / sqrt(vx * vx + vy * vy + vz * vz);
                total = total + 1.0 / sqrt(vx * vx + vy * vy + vz * vz);
            }
            for (i64 i = 0; i < n_vols; ++i) w[i] = w[i] / total;
            continue;
        }
        if (Iz == 0.0 && Izz == 0.0 && Ixz == 0.0 && Iyz == 0.0) Izz = -1.0;
        double lx = (Ix * (Iyz * Iyz - Iyy * Izz) + Iy * (Ixy * Izz - Iyz * Ixz) + Iz * (Iyy * Ixz - Ixy * Iyz)) / D;
        double ly = (Ix * (Ixy * Izz - Iyz * Ixz) + Iy * (Ixz * Ixz - Ixx * Izz) + Iz * (Ixx * Iyz - Ixy * Ixz)) / D;
        double lz = (Ix * (Iyy * Ixz - Ixy * Iyz) + Iy * (Ixx * Iyz - Ixy * Ixz) + Iz * (Ixy * Ixy - Ixx * Iyy)) / D;
        double denom = (double)n_vols + lx * Ix + ly * Iy + lz * Iz;
        for (i64 q = b, i = 0; q < e; ++q, ++i) {
            i64 vol = g->esup[q];
            double vx = g->centroids[vol * 3 + 0] - xp[0];
            double vy = g->centroids[vol * 3 + 1] - xp[1];
            double vz = g->centroids[vol * 3 + 2] - xp[2];
            w[i] = (1. + lx * vx + ly * vy + lz * vz);
            w[i] /= denom;
        }
    }
}

/* ---- dgels('N') for m >= n, restated: Householder QR (LAPACK dgeqr2 / dlarfg / dlarf), B := Q^T B
 * (dorm2r), then the triangular solve of dtrtrs, which returns early when a diagonal entry of R is
 * exactly zero (dgels then leaves Q^T B in B and reports info > 0; gls.pyx:457 ignores info).
 * SciPy's LAPACK (third party, not under /root/reference; here OpenBLAS 0.3.28) is the reference's
 * actual solver; this is the same algorithm, agreeing to rounding (checked <= 1e-12 in tests).
 * A is column-major m x n with leading dimension lda, B is m x nrhs with ldb. */
static double dnrm2_scaled(i64 n, const double *x) {
    double scale = 0.0, ssq = 1.0;
    for (i64 i = 0; i < n; ++i) {
        if (x[i] != 0.0) {
            double a = fabs(x[i]);
            if (scale < a) { ssq = 1.0 + ssq * (scale / a) * (scale / a); scale = a; }
            else ssq += (a / scale) * (a / scale);
        }
    }
    return scale * sqrt(ssq);
}

static int oracle_dgels(i64 m, i64 n, i64 nrhs, double *A, i64 lda, double *B, i64 ldb, double *tau) {
    for (i64 k = 0; k < n; ++k) {
        /* dlarfg on A[k:m, k] */
        double alpha = A[k + k * lda];
        double xnorm = (m - k - 1 > 0) ? dnrm2_scaled(m - k - 1, &A[k + 1 + k * lda]) : 0.0;
        double t = 0.0;
        if (xnorm != 0.0) {
            double beta = -copysign(hypot(alpha, xnorm), alpha);
            t = (beta - alpha) / beta;
            double sc = 1.0 / (alpha - beta);
            for (i64 i = k + 1; i < m; ++i) A[i + k * lda] *= sc;
            alpha = beta;
        }
        tau[k] = t;
        A[k + k * lda] = alpha;
        if (t != 0.0) {
            /* apply H = I - t v v^T (v[0] = 1) to the remaining columns of A and to B */
            for (i64 j = k + 1; j < n; ++j) {
                double s = A[k + j * lda];
                for (i64 i = k + 1; i < m; ++i) s += A[i + k * lda] * A[i + j * lda];
                s *= t;
                A[k + j * lda] -= s;
                for (i64 i = k + 1; i < m; ++i) A[i + j * lda] -= s * A[i + k * lda];
            }
            for (i64 j = 0; j < nrhs; ++j) {
                double s = B[k + j * ldb];
                for (i64 i = k + 1; i < m; ++i) s += A[i + k * lda] * B[i + j * ldb];
                s *= t;
                B[k + j * ldb] -= s;
                for (i64 i = k + 1; i < m; ++i) B[i + j * ldb] -= s * A[i + k * lda];
            }
        }
    }
    for (i64 k = 0; k < n; ++k)
        if (A[k + k * lda] == 0.0) return (int)(k + 1);
    for (i64 j = 0; j < nrhs; ++j) {
        for (i64 k = n - 1; k >= 0; --k) {
            double x = B[k + j * ldb] / A[k + k * lda];
            B[k + j * ldb] = x;
            for (i64 i = 0; i < k; ++i) B[i + j * ldb] -= x * A[i + k * lda];
        }
    }
    return 0;
}

static void cross3(const double *a, const double *b, double *c) { /* gls.pyx:365-369 */
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

/* dgemv("T", 3, 3, 1, K, 3, N, 1, 0, out, 1) on the row-major 3x3 K == K . N  (gls.pyx:320-321) */
static void k_dot_n(const double *K, const double *N, double *out) {
    for (int r = 0; r < 3; ++r) out[r] = K[r * 3 + 0] * N[0] + K[r * 3 + 1] * N[1] + K[r * 3 + 2] * N[2];
}

/* gls.pyx:75-219 GLS + :234-249 build_ks_sv_arrays + :252-356 build_ls_matrices + :374-416
 * set_neumann_rows + :420-474 solve_ls.  permeability is [E][3][3] row-major, diff_mag [E]. */
void oracle_gls(const oracle_grid *g, const i64 *targets, i64 n_target, const double *permeability,
                const double *diff_mag, const i64 *neumann_point, const double *neumann_val,
                double *weights, double *neumann_ws, int num_threads) {
    const i64 W = g->MX_ELEMENTS_PER_POINT;
    const i64 NE = g->MX_ELEMENTS_PER_POINT, NF = g->MX_FACES_PER_POINT;
    const i64 M_MAX = NE + 3 * NF + NF, N_MAX = 3 * NE + 1, R_MAX = NE + 1;
    (void)num_threads;
#pragma omp parallel num_threads(num_threads)
    {
        double *A = (double *)malloc(sizeof(double) * (size_t)(M_MAX * N_MAX));
        double *B = (double *)malloc(sizeof(double) * (size_t)(M_MAX * R_MAX));
        double *tau = (double *)malloc(sizeof(double) * (size_t)N_MAX);
        i64 *Sv = (i64 *)malloc(sizeof(i64) * (size_t)NF), *Svb = (i64 *)malloc(sizeof(i64) * (size_t)NF);
#pragma omp for schedule(static)
        for (i64 it = 0; it < n_target; ++it) {
            i64 point = targets[it];
            if (g->boundary_points[point] && !neumann_point[point]) continue;
            const i64 *KSetv = g->esup + g->esup_ptr[point];
            i64 n_elem = g->esup_ptr[point + 1] - g->esup_ptr[point];
            i64 n_face = g->fsup_ptr[point + 1] - g->fsup_ptr[point];
            i64 n_bface = 0;
            for (i64 q = g->fsup_ptr[point], j = 0; q < g->fsup_ptr[point + 1]; ++q, ++j) {
                i64 f = g->fsup[q];
                Sv[j] = f;
                if (g->boundary_faces[f] == 1) Svb[n_bface++] = f;
            }
            i64 m = n_elem + 3 * n_face + n_bface, n = 3 * n_elem + 1;
            i64 is_neu = neumann_point[point];
            i64 nrhs = n_elem + is_neu;
            i64 lda = m > 1 ? m : 1, ldb = lda;
            /* Mi / Ni are assembled directly in the column-major A / B of solve_ls (gls.pyx:446-452) */
            memset(A, 0, sizeof(double) * (size_t)(lda * n));
            memset(B, 0, sizeof(double) * (size_t)(ldb * nrhs));
#define MI(r, c) A[(r) + (c) * lda]
#define NI(r, c) B[(r) + (c) * ldb]
            const double *xv = g->point_coords + point * 3;
            if (!(n_bface >= n_face)) { /* gls.pyx:266-267 early return leaves Mi = 0 */
                for (i64 i = 0; i < n_elem; ++i) {
                    const double *xK = g->centroids + KSetv[i] * 3;
                    MI(i, 3 * i + 0) = xK[0] - xv[0];
                    MI(i, 3 * i + 1) = xK[1] - xv[1];
                    MI(i, 3 * i + 2) = xK[2] - xv[2];
                    MI(i, 3 * n_elem) = 1.0;
                    NI(i, i) = 1.0;
                }
                i64 row = n_elem;
                for (i64 i = 0; i < n_face; ++i) {
                    i64 f = Sv[i];
                    i64 n_esuf = g->esuf_ptr[f + 1] - g->esuf_ptr[f];
                    if (n_esuf < 2) continue;
                    const double *xS = g->faces_centers + f * 3, *N = g->normal_faces + f * 3;
                    double eta = 0.0;
                    i64 Ks[2];
                    for (i64 k = 0; k < n_esuf; ++k) {
                        Ks[k] = g->esuf[g->esuf_ptr[f] + k];
                        if (diff_mag[Ks[k]] > eta) eta = diff_mag[Ks[k]]; /* max(eta, diff_mag) :304 */
                    }
                    double T1[3] = {xv[0] - xS[0], xv[1] - xS[1], xv[2] - xS[2]}, T2[3], tT2[3], nL1[3], nL2[3];
                    cross3(N, T1, T2);
                    double tau2 = pow(sqrt(T2[0] * T2[0] + T2[1] * T2[1] + T2[2] * T2[2]), -eta);
                    tT2[0] = tau2 * T2[0]; tT2[1] = tau2 * T2[1]; tT2[2] = tau2 * T2[2];
                    k_dot_n(permeability + Ks[0] * 9, N, nL1);
                    k_dot_n(permeability + Ks[1] * 9, N, nL2);
                    i64 I1 = -1, I2 = -1; /* KSetv_map lookups (gls.pyx:325-333) */
                    for (i64 k = 0; k < n_elem; ++k) {
                        if (KSetv[k] == Ks[0]) I1 = k;
                        if (KSetv[k] == Ks[1]) I2 = k;
                    }
                    for (int c = 0; c < 3; ++c) {
                        MI(row + 0, 3 * I1 + c) = nL1[c] * -1; MI(row + 0, 3 * I2 + c) = nL2[c] * 1;
                        MI(row + 1, 3 * I1 + c) = T1[c] * -1;  MI(row + 1, 3 * I2 + c) = T1[c] * 1;
                        MI(row + 2, 3 * I1 + c) = tT2[c] * -1; MI(row + 2, 3 * I2 + c) = tT2[c] * 1;
                    }
                    row += 3;
                }
            }
            if (is_neu) { /* set_neumann_rows gls.pyx:374-416 */
                i64 start = n_elem + 3 * n_face;
                for (i64 i = 0; i < n_bface; ++i) {
                    i64 f = Svb[i];
                    i64 K0 = g->esuf[g->esuf_ptr[f]];
                    double nL[3];
                    k_dot_n(permeability + K0 * 9, g->normal_faces + f * 3, nL);
                    i64 total = 0;
                    for (int q = 0; q < MAX_POINTS_PER_FACE; ++q) {
                        i64 bp = INPOFA(g, f, q);
                        if (bp == -1) break;
                        total += 1;
                        NI(start + i, n_elem) += neumann_val[bp];
                    }
                    NI(start + i, n_elem) /= (double)total;
                    i64 Ik = -1;
                    for (i64 k = 0; k < n_elem; ++k) if (KSetv[k] == K0) Ik = k;
                    MI(start + i, 3 * Ik + 0) = -nL[0];
                    MI(start + i, 3 * Ik + 1) = -nL[1];
                    MI(start + i, 3 * Ik + 2) = -nL[2];
                }
            }
#undef MI
#undef NI
            oracle_dgels(m, n, nrhs, A, lda, B, ldb, tau);
            i64 w_total = nrhs - is_neu;
            double *w = weights + point * W;
            for (i64 i = 0; i < w_total; ++i) w[i] = B[(n - 1) + i * ldb];
            if (is_neu) neumann_ws[point] = B[(n - 1) + (w_total - 1) * ldb];
        }
        free(A); free(B); free(tau); free(Sv); free(Svb);
    }
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
