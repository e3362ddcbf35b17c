// mfx_desc.hpp -- launch-plan descriptor of a node for kernels_gls_mfx.hip, the WIDE one-wavefront multifrontal GLS kernel
// (internal, device code): interior nodes of unstructured meshes -- a Delaunay tetrahedrisation has 14 .. 40 cells around a
// node, no two-colouring, and a third to a half of its nodes are beyond mfw_desc.hpp's 12 + 15 cells.
//
// Same decomposition as mfw_desc.hpp's general kind: F <= 16 FRONTS (cells with exactly 3 faces at the node that share no
// face with each other: a minimum-degree greedy independent set, the exact maximum on almost every Delaunay node), D <= 21 DENSE cells, faces between two dense cells are FREE faces (<= 16).
// Dense problem: (7 F + D + 3 free + boundary) x (3 D + 1) <= 160 x 64.
// Boundary nodes too (round 4): a cell with a boundary face at the node has fewer than 3 internal faces there and is a dense cell; each
// boundary face adds ONE row -- the Neumann row -(K N) on its cell's columns (gls.pyx:394-416) -- and the node is computed only when the
// variable flags it Neumann (gls.pyx:165-166: the kernel gives a Dirichlet boundary node its zero row at launch time).
//
// 56 words per node:
//   word 0            F | D << 8 | free faces << 16 | boundary faces << 24
//   word 1 + f        front f: position in the esup row (6 bits) | fsup positions of its faces 0, 1, 2 (6 bits each, << 6, 12, 18) |
//                     bit 24 + i: the front is face i's FIRST cell (side a: row = [-B_a | +B_b], gls.pyx:340-356)
//   word 17 + f       front f: dense slots of the cells across its faces 0, 1, 2 (5 bits each)
//   word 33 .. 38     esup position of dense slot d, one byte each (d = 0 .. 20, slots in esup order)
//   word 39 + q       free face q: fsup position (6 bits) | dense slot of its first cell << 6 | of its second cell << 11;
//                     then the boundary faces: fsup position | dense slot of its cell << 6 | bit 31  (free + boundary faces <= 16)
// Fronts and dense cells are numbered in esup order, free and boundary faces in fsup order.
#pragma once
#include <cstdint>

#include "device_grid.hpp"

namespace nin {

constexpr int kMfxMaxFronts = 16, kMfxMaxDense = 21, kMfxMaxFree = 16, kMfxDescWords = 56;
constexpr int kMfxMaxRows = 160, kMfxMaxCells = kMfxMaxFronts + kMfxMaxDense, kMfxMaxFaces = 63;
constexpr int kMfxW0 = 1, kMfxW1 = 17, kMfxSlotTable = 33, kMfxFree0 = 39;
// Size classes of the dense problem (kernels_gls_mfx.hip: one instantiation and one list of the launch plan each): rows <= 16 TQ,
// pivot columns nc = 3 D < 4 TCB for (TQ, TCB) = (6, 10), (7, 11), (8, 13), (9, 15), (10, 16)
constexpr int kMfxClasses = 5;
// ... and a SMALL class, (4, 7): rows <= 64, nc < 28 -- interior nodes of 10 .. 14 cells that are not two-coloured (an unstructured prism mesh:
// a node of odd valence 5 or 7 has 10 / 14 wedges in a ring no two-colouring closes: 4 + 6 or 6 + 8 cells, 43 x 19 or 59 x 25).  Its own
// list of the launch plan (node_class 240), three wavefronts per SIMD.  mfx_descriptor returns kMfxSmallCode for it.
constexpr int kMfxSmallCode = 7;
// ... and one BETWEEN (7, 11) and (8, 13): (7, 12) -- rows <= 112, nc < 48.  45 % of the nodes of a body-centred Delaunay mesh that miss (7, 11)
// fit it and would sweep 104 tiles for 84 otherwise (node_class 239, plan entry mfx_7x12; mfx_descriptor returns kMfxMidCode).
constexpr int kMfxMidCode = 8;
NIN_HD inline bool mfx_fits_mid(int F, int D, int nfree, int nbnd) { return nbnd == 0 && 7 * F + D + 3 * nfree <= 112 && 3 * D < 48; }
NIN_HD inline bool mfx_fits_small(int F, int D, int nfree, int nbnd) { return nbnd == 0 && 7 * F + D + 3 * nfree <= 64 && 3 * D < 28; }
NIN_HD inline int mfx_size_class(int F, int D, int nfree, int nbnd) {
    const int rows = 7 * F + D + 3 * nfree + nbnd, nc = 3 * D;
    if (rows <= 96 && nc < 40) return 0;
    if (rows <= 112 && nc < 44) return 1;
    if (rows <= 128 && nc < 52) return 2;
    if (rows <= 144 && nc < 60) return 3;
    if (rows <= 160 && nc < 64) return 4;
    return -1;
}

#ifdef __HIPCC__
struct MfxGraph {
    int ne, nf, nbnd;
    uint64_t adj[kMfxMaxCells];
    uint8_t deg[kMfxMaxCells], fa[kMfxMaxFaces], fbb[kMfxMaxFaces];   // fa / fbb: esup positions of a face's first / second cell (0xFF: boundary face)
};

// false: two faces between the same pair of cells, or too many cells / faces
__device__ inline bool mfx_graph(const GridView &g, int32_t p, MfxGraph &G) {
    const int32_t eb = g.esup_ptr[p], fb = g.fsup_ptr[p];
    G.ne = g.esup_ptr[p + 1] - eb;
    G.nf = g.fsup_ptr[p + 1] - fb;
    if (g.dim != 3 || G.ne < 2 || G.ne > kMfxMaxCells || G.nf > kMfxMaxFaces || G.nf < 1) return false;
    for (int i = 0; i < G.ne; ++i) { G.adj[i] = 0ull; G.deg[i] = 0; }
    G.nbnd = 0;
    for (int fi = 0; fi < G.nf; ++fi) {
        const int64_t f = g.fsup[fb + fi];
        const int32_t a = g.face_cells[2 * f], b = g.face_cells[2 * f + 1];
        int ia = -1, ib = -1;
        for (int i = 0; i < G.ne; ++i) {
            const int32_t c = g.esup[eb + i];
            ia = c == a ? i : ia;
            ib = c == b ? i : ib;
        }
        if (b < 0) {                                       // a boundary face: no coupling, one Neumann row on its cell
            if (ia < 0) return false;
            G.fa[fi] = (uint8_t)ia;
            G.fbb[fi] = 0xFF;
            ++G.nbnd;
            continue;
        }
        if (ia < 0 || ib < 0 || ia == ib) return false;
        if ((G.adj[ia] >> ib) & 1ull) return false;
        G.adj[ia] |= 1ull << ib;
        G.adj[ib] |= 1ull << ia;
        ++G.deg[ia];
        ++G.deg[ib];
        G.fa[fi] = (uint8_t)ia;
        G.fbb[fi] = (uint8_t)ib;
    }
    return true;
}

// 0: not for this kernel; 1 + size class (or kMfxSmallCode): the words are filled
__device__ inline int mfx_descriptor(const GridView &g, int32_t p, uint32_t w[kMfxDescWords]) {
    MfxGraph G;
    if (!mfx_graph(g, p, G)) return 0;
    const int ne = G.ne, nbnd = G.nbnd, nf = G.nf - nbnd;    // nf: internal faces
    if (nf < 1 || 3 * nf + nbnd < 2 * ne) return 0;        // no internal face / fewer rows than unknowns next to the node value: the zero row
    // a front: a cell with exactly 3 faces at the node, all internal (a cell owning a Neumann row has a non-zero outside its front)
    uint64_t elig = 0ull, has_bnd = 0ull;
    for (int fi = 0; fi < G.nf; ++fi)
        if (G.fbb[fi] == 0xFF) has_bnd |= 1ull << G.fa[fi];
    for (int i = 0; i < ne; ++i)
        if (G.deg[i] == 3 && !((has_bnd >> i) & 1ull)) elig |= 1ull << i;
    // The fronts: an independent set of 3-face cells, as large as a cheap search finds -- every front takes three rows and three
    // columns out of the dense problem.  Minimum-residual-degree greedy (always the eligible cell with the fewest eligible
    // neighbours left; ties to the first in cyclic order from `start`), from every fourth starting cell: on Delaunay nodes its
    // best set is the exact maximum but for 0.02 cells a node (tools/front_sets.py: branch and bound), where the first-fit greedy
    // of mfw_desc.hpp is 0.18 below it.
    uint64_t best = 0ull;
    for (int start = 0; start < ne; start += 4) {
        uint64_t chosen = 0ull, avail = elig;
        int n = 0;
        while (avail && n < kMfxMaxFronts) {
            int pick = -1, pd = 99;
            for (int k = 0; k < ne; ++k) {
                const int c = start + k < ne ? start + k : start + k - ne;
                if (!((avail >> c) & 1ull)) continue;
                const int d = __popcll(G.adj[c] & avail);
                if (d < pd) { pd = d; pick = c; }
            }
            chosen |= 1ull << pick;
            avail &= ~(G.adj[pick] | (1ull << pick));
            ++n;
        }
        if (n > __popcll(best)) best = chosen;
    }
    const int F = __popcll(best), D = ne - F, nfree = nf - 3 * F;
    if (F < 1 || D < 1 || D > kMfxMaxDense || nfree < 0 || nfree + nbnd > kMfxMaxFree) return 0;
    if (7 * F + D + 3 * nfree + nbnd > kMfxMaxRows) return 0;
    int rank[kMfxMaxCells];                                // front number or dense slot of a cell
    {
        int f = 0, d = 0;
        for (int i = 0; i < ne; ++i) rank[i] = ((best >> i) & 1ull) ? f++ : d++;
    }
    for (int k = 0; k < kMfxDescWords; ++k) w[k] = 0u;
    w[0] = (uint32_t)F | ((uint32_t)D << 8) | ((uint32_t)nfree << 16) | ((uint32_t)nbnd << 24);
    for (int i = 0; i < ne; ++i) {
        if ((best >> i) & 1ull) w[kMfxW0 + rank[i]] |= (uint32_t)i;
        else w[kMfxSlotTable + (rank[i] >> 2)] |= (uint32_t)i << (8 * (rank[i] & 3));
    }
    int nface[kMfxMaxFronts];
    for (int f = 0; f < kMfxMaxFronts; ++f) nface[f] = 0;
    int q = 0, qb = nfree;
    for (int fi = 0; fi < G.nf; ++fi) {
        const int ia = G.fa[fi], ib = G.fbb[fi];
        if (ib == 0xFF) {                                  // boundary face: behind the free faces
            w[kMfxFree0 + qb++] = (uint32_t)fi | ((uint32_t)rank[ia] << 6) | 0x80000000u;
            continue;
        }
        const bool a_front = ((best >> ia) & 1ull) != 0, b_front = ((best >> ib) & 1ull) != 0;
        if (!a_front && !b_front) {
            w[kMfxFree0 + q++] = (uint32_t)fi | ((uint32_t)rank[ia] << 6) | ((uint32_t)rank[ib] << 11);
            continue;
        }
        const int fc = a_front ? ia : ib, oc = a_front ? ib : ia;
        const int f = rank[fc], k = nface[f]++;
        w[kMfxW0 + f] |= ((uint32_t)fi << (6 + 6 * k)) | ((a_front ? 1u : 0u) << (24 + k));
        w[kMfxW1 + f] |= (uint32_t)rank[oc] << (5 * k);
    }
    if (mfx_fits_small(F, D, nfree, nbnd)) return kMfxSmallCode;
    const int cls = mfx_size_class(F, D, nfree, nbnd);
    if (cls == 2 && mfx_fits_mid(F, D, nfree, nbnd)) return kMfxMidCode;
    return 1 + cls;
}
#endif

}  // namespace nin
