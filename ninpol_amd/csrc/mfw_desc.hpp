// mfw_desc.hpp -- launch-plan descriptor of a node for kernels_gls_mfw.hip, the one-wavefront multifrontal GLS kernel
// (internal, device code).
//
// Cells around the node = vertices, its internal faces = edges (a face row of the GLS system couples exactly its two
// cells, gls.pyx:340-356).  The kernel takes interior nodes (no boundary face at the node) whose cells split into
//     F <= 12 FRONTS  -- cells with exactly 3 faces at the node (true of every tetrahedron, hexahedron and wedge corner)
//                        that share no face with each other -- and
//     D DENSE cells   -- all the others (any number of faces).
// Two kinds:
//   two-coloured (return 1)  the graph is bipartite and one colour class qualifies as the fronts: every face joins a front
//                            to a dense cell (nf = 3 F), D <= 12.  Interior nodes of Kuhn-type tetrahedron meshes (the
//                            truncated octahedron, 12 + 12), of wedge meshes (hexagonal prism, 6 + 6), cube nodes (4 + 4).
//   general (return 2)       anything else that fits: the fronts are a maximal independent set of 3-face cells (greedy from
//                            every starting cell, the largest kept), faces between two dense cells are FREE faces (<= 14;
//                            their three rows go straight into the dense problem), D <= 15.  The hex | pyramid | tet
//                            interface and pyramid-apex nodes of mixed meshes.
// Anything else stays with kernels_gls_block.hip.
//
// 40 words per node:
//   word f      (f < 12)   front f:  bits 0-4 its position in the node's esup row, bits 5-20 face 0,
//                          bits 21-25 the esup position of DENSE cell f (the word doubles as that cell's record)
//   word 12 + f (f < 12)   front f:  bits 0-15 face 1, bits 16-31 face 2
//   word 24                F | D << 8 | free faces << 16
//   word 25                esup positions of the dense cells 12, 13, 14 (5 bits each)
//   word 26 + q (q < 14)   free face q: 6 bits position in the fsup row | dense slot of its first cell (side a) << 6 |
//                          dense slot of its second cell << 10 | their esup positions << 14, << 19
//   a front's face (16 bits): 6 bits position in the fsup row | 4 bits dense slot of the cell on the other side << 6 |
//                     1 bit "the front is the face's first cell (side a: row = [-B_a | +B_b])" << 10 |
//                     5 bits esup position of that other cell << 11
// Fronts and dense cells are numbered in esup order; in the two-coloured kind the front class is the colour of the
// row's first cell if that class qualifies, the other one otherwise.
#pragma once
#include <cstdint>

#include "device_grid.hpp"

namespace nin {

constexpr int kMfwMaxFronts = 12, kMfwMaxDense = 12, kMfwDescWords = 40;
constexpr int kMfwSmallFronts = 6, kMfwSmallDense = 6;   // the kernel's second instantiation (wedge nodes: 6 + 6, cube nodes: 4 + 4)
constexpr int kMfwWideDense = 15, kMfwMaxFree = 14;      // the third one: the general kind
constexpr int kMfwMaxRows = 128;                         // dense rows a wavefront holds (two per lane)
constexpr int kMfwMaxCells = kMfwMaxFronts + kMfwWideDense, kMfwMaxFaces = 63;

#ifdef __HIPCC__
struct MfwGraph {
    int ne, nf;
    uint32_t adj[kMfwMaxCells];
    uint8_t deg[kMfwMaxCells], fa[kMfwMaxFaces], fbb[kMfwMaxFaces];   // fa / fbb: esup positions of a face's first / second cell
};

// false: a boundary face at the node, two faces between the same pair of cells, or too many cells / faces
__device__ inline bool mfw_graph(const GridView &g, int32_t p, MfwGraph &G) {
    const int32_t eb = g.esup_ptr[p], fb = g.fsup_ptr[p];
    G.ne = g.esup_ptr[p + 1] - eb;
    G.nf = g.fsup_ptr[p + 1] - fb;
    if (g.dim != 3 || G.ne < 2 || G.ne > kMfwMaxCells || G.nf > kMfwMaxFaces || G.nf < 1) return false;
    int32_t cells[kMfwMaxCells];
    for (int i = 0; i < G.ne; ++i) { cells[i] = g.esup[eb + i]; G.adj[i] = 0u; G.deg[i] = 0; }
    for (int fi = 0; fi < G.nf; ++fi) {
        const int64_t f = g.fsup[fb + fi];
        const int32_t a = g.face_cells[2 * f], b = g.face_cells[2 * f + 1];
        if (b < 0) return false;
        int ia = -1, ib = -1;
        for (int i = 0; i < G.ne; ++i) {
            ia = cells[i] == a ? i : ia;
            ib = cells[i] == b ? i : ib;
        }
        if (ia < 0 || ib < 0 || ia == ib) return false;
        if ((G.adj[ia] >> ib) & 1u) return false;
        G.adj[ia] |= 1u << ib;
        G.adj[ib] |= 1u << ia;
        ++G.deg[ia];
        ++G.deg[ib];
        G.fa[fi] = (uint8_t)ia;
        G.fbb[fi] = (uint8_t)ib;
    }
    return true;
}

// the words for a given set of fronts (every front has 3 faces and no front neighbour)
__device__ inline void mfw_pack(const MfwGraph &G, uint32_t fronts, uint32_t w[kMfwDescWords]) {
    int rank[kMfwMaxCells];                                // front number or dense slot of a cell
    int F = 0, D = 0;
    for (int i = 0; i < G.ne; ++i) rank[i] = ((fronts >> i) & 1u) ? F++ : D++;
    for (int k = 0; k < kMfwDescWords; ++k) w[k] = 0u;
    for (int i = 0; i < G.ne; ++i) {
        if ((fronts >> i) & 1u) w[rank[i]] |= (uint32_t)i;
        else if (rank[i] < 12) w[rank[i]] |= (uint32_t)i << 21;
        else w[25] |= (uint32_t)i << (5 * (rank[i] - 12));
    }
    int nface[kMfwMaxFronts];
    for (int f = 0; f < kMfwMaxFronts; ++f) nface[f] = 0;
    int nfree = 0;
    for (int fi = 0; fi < G.nf; ++fi) {
        const int ia = G.fa[fi], ib = G.fbb[fi];
        const bool a_front = ((fronts >> ia) & 1u) != 0, b_front = ((fronts >> ib) & 1u) != 0;
        if (!a_front && !b_front) {
            w[26 + nfree++] = (uint32_t)fi | ((uint32_t)rank[ia] << 6) | ((uint32_t)rank[ib] << 10) | ((uint32_t)ia << 14) |
                              ((uint32_t)ib << 19);
            continue;
        }
        const int fc = a_front ? ia : ib, oc = a_front ? ib : ia;
        const int f = rank[fc], k = nface[f]++;
        const uint32_t rec = (uint32_t)fi | ((uint32_t)rank[oc] << 6) | ((a_front ? 1u : 0u) << 10) | ((uint32_t)oc << 11);
        if (k == 0) w[f] |= rec << 5;
        else if (k == 1) w[12 + f] |= rec;
        else w[12 + f] |= rec << 16;
    }
    w[24] = (uint32_t)F | ((uint32_t)D << 8) | ((uint32_t)nfree << 16);
}

// 0: not for this kernel; 1: two-coloured; 2: general
__device__ inline int mfw_descriptor(const GridView &g, int32_t p, uint32_t w[kMfwDescWords]) {
    MfwGraph G;
    if (!mfw_graph(g, p, G)) return 0;
    const int ne = G.ne, nf = G.nf;
    if (3 * nf < 2 * ne) return 0;                         // fewer rows than unknowns next to the node value: the zero row
    const uint32_t all = (1u << ne) - 1u;
    // ---- two-colouring from cell 0
    uint32_t col0 = 1u, col1 = 0u, frontier = 1u;
    for (int sweep = 0; sweep < ne && frontier; ++sweep) {
        uint32_t next = 0u;
        for (int i = 0; i < ne; ++i)
            if ((frontier >> i) & 1u) next |= G.adj[i];
        next &= ~(col0 | col1);
        if (sweep & 1) col0 |= next; else col1 |= next;
        frontier = next;
    }
    bool bipartite = (col0 | col1) == all && !(col0 & col1);
    for (int i = 0; i < ne && bipartite; ++i) {
        const uint32_t mine = ((col0 >> i) & 1u) ? col0 : col1;
        if (G.adj[i] & mine) bipartite = false;            // an odd cycle
    }
    if (bipartite) {
        auto qualifies = [&](uint32_t cls) {
            const int F = __popc(cls), D = ne - F;
            if (F < 1 || F > kMfwMaxFronts || D < 1 || D > kMfwMaxDense || nf != 3 * F) return false;
            for (int i = 0; i < ne; ++i)
                if (((cls >> i) & 1u) && G.deg[i] != 3) return false;
            return true;
        };
        const uint32_t fronts = qualifies(col0) ? col0 : qualifies(col1) ? col1 : 0u;
        if (fronts) {
            mfw_pack(G, fronts, w);
            return 1;
        }
    }
    // ---- general: the largest of the greedy independent sets of 3-face cells, one per starting cell
    uint32_t elig = 0u;
    for (int i = 0; i < ne; ++i)
        if (G.deg[i] == 3) elig |= 1u << i;
    uint32_t best = 0u;
    for (int start = 0; start < ne; ++start) {
        uint32_t chosen = 0u;
        for (int k = 0; k < ne; ++k) {
            const int c = start + k < ne ? start + k : start + k - ne;
            if (((elig >> c) & 1u) && !(G.adj[c] & chosen)) chosen |= 1u << c;
        }
        if (__popc(chosen) > __popc(best)) best = chosen;
    }
    const int F = __popc(best), D = ne - F, nfree = nf - 3 * F;
    if (F < 1 || F > kMfwMaxFronts || D < 1 || D > kMfwWideDense || nfree < 0 || nfree > kMfwMaxFree) return 0;
    if (7 * kMfwMaxFronts + D + 3 * nfree > kMfwMaxRows) return 0;   // (the fronts' 7 fill rows have fixed places)
    mfw_pack(G, best, w);
    return 2;
}
#endif

}  // namespace nin
