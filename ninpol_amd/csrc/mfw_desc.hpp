// mfw_desc.hpp -- launch-plan descriptor of a "two-coloured" node for kernels_gls_mfw.hip (internal, device code).
//
// Cells around the node = vertices, its internal faces = edges (a face row of the GLS system couples exactly its two
// cells, gls.pyx:340-356).  The one-wavefront multifrontal kernel takes nodes whose graph is BIPARTITE with one colour
// class -- the "fronts" -- made of cells that touch exactly 3 faces at the node (true of every tetrahedron, hexahedron
// and wedge corner), and with no boundary face at the node:
//     interior nodes of Kuhn-type tetrahedron meshes (24 cells: the truncated octahedron, 12 + 12),
//     interior nodes of wedge meshes (12 cells: the hexagonal prism, 6 + 6), cube nodes (4 + 4), ...
// F fronts (<= 12) and D dense cells (<= 12); every face then joins one front to one dense cell, so nf = 3 F.
// Anything else stays with kernels_gls_block.hip.
//
// 32 words per node:
//   word f      (f < 12)   front f:  bits 0-4 its position in the node's esup row, bits 5-20 face 0,
//                          bits 21-25 the esup position of DENSE cell f (the word doubles as that cell's record)
//   word 12 + f (f < 12)   front f:  bits 0-15 face 1, bits 16-31 face 2
//   word 24                F | D << 8
//   a face (16 bits): 6 bits position in the fsup row | 4 bits dense slot of the cell on the other side << 6 |
//                     1 bit "the front is the face's first cell (side a: row = [-B_a | +B_b])" << 10 |
//                     5 bits esup position of that other cell << 11
// Fronts and dense cells are numbered in esup order; the front class is the colour of the row's first cell if that
// class qualifies, the other one otherwise.
#pragma once
#include <cstdint>

#include "device_grid.hpp"

namespace nin {

constexpr int kMfwMaxFronts = 12, kMfwMaxDense = 12, kMfwDescWords = 32;
constexpr int kMfwSmallFronts = 6, kMfwSmallDense = 6;   // the kernel's second instantiation (wedge nodes: 6 + 6, cube nodes: 4 + 4)

#ifdef __HIPCC__
__device__ inline bool mfw_descriptor(const GridView &g, int32_t p, uint32_t w[kMfwDescWords]) {
    const int32_t eb = g.esup_ptr[p], fb = g.fsup_ptr[p];
    const int ne = g.esup_ptr[p + 1] - eb, nf = g.fsup_ptr[p + 1] - fb;
    if (g.dim != 3 || ne < 2 || ne > kMfwMaxFronts + kMfwMaxDense || nf > 3 * kMfwMaxFronts || nf < 1) return false;
    int32_t cells[24];
    for (int i = 0; i < ne; ++i) cells[i] = g.esup[eb + i];
    uint32_t adj[24];
    uint8_t deg[24], fa[36], fbb[36];
    for (int i = 0; i < ne; ++i) { adj[i] = 0u; deg[i] = 0; }
    for (int fi = 0; fi < nf; ++fi) {
        const int64_t f = g.fsup[fb + fi];
        const int32_t a = g.face_cells[2 * f], b = g.face_cells[2 * f + 1];
        if (b < 0) return false;                         // a boundary face at the node
        int ia = -1, ib = -1;
        for (int i = 0; i < ne; ++i) {
            ia = cells[i] == a ? i : ia;
            ib = cells[i] == b ? i : ib;
        }
        if (ia < 0 || ib < 0 || ia == ib) return false;
        if ((adj[ia] >> ib) & 1u) return false;          // two faces between the same pair of cells
        adj[ia] |= 1u << ib;
        adj[ib] |= 1u << ia;
        ++deg[ia];
        ++deg[ib];
        fa[fi] = (uint8_t)ia;
        fbb[fi] = (uint8_t)ib;
    }
    // two-colouring from cell 0 (the graph must be connected and bipartite)
    uint32_t col0 = 1u, col1 = 0u, frontier = 1u;
    for (int sweep = 0; sweep < ne && frontier; ++sweep) {
        uint32_t next = 0u;
        for (int i = 0; i < ne; ++i)
            if ((frontier >> i) & 1u) next |= adj[i];
        next &= ~(col0 | col1);
        if (sweep & 1) col0 |= next; else col1 |= next;
        frontier = next;
    }
    const uint32_t all = ne >= 32 ? ~0u : ((1u << ne) - 1u);
    if ((col0 | col1) != all || (col0 & col1)) return false;
    for (int i = 0; i < ne; ++i) {
        const uint32_t mine = ((col0 >> i) & 1u) ? col0 : col1;
        if (adj[i] & mine) return false;                 // an odd cycle
    }
    auto qualifies = [&](uint32_t cls) {
        const int F = __popc(cls), D = ne - F;
        if (F < 1 || F > kMfwMaxFronts || D < 1 || D > kMfwMaxDense || 7 * F < 2 * D) return false;   // (rows >= unknowns)
        for (int i = 0; i < ne; ++i)
            if (((cls >> i) & 1u) && deg[i] != 3) return false;
        return true;
    };
    uint32_t fronts;
    if (qualifies(col0)) fronts = col0;
    else if (qualifies(col1)) fronts = col1;
    else return false;
    const int F = __popc(fronts), D = ne - F;
    if (nf != 3 * F) return false;
    int rank[24];                                          // front number or dense slot of a cell
    for (int i = 0, nfr = 0, nd = 0; i < ne; ++i) rank[i] = ((fronts >> i) & 1u) ? nfr++ : nd++;
    for (int k = 0; k < kMfwDescWords; ++k) w[k] = 0u;
    int nface[12];
    for (int f = 0; f < 12; ++f) nface[f] = 0;
    for (int i = 0; i < ne; ++i) {
        if ((fronts >> i) & 1u) w[rank[i]] |= (uint32_t)i;
        else w[rank[i]] |= (uint32_t)i << 21;
    }
    for (int fi = 0; fi < nf; ++fi) {
        const int ia = fa[fi], ib = fbb[fi];
        const bool a_front = ((fronts >> ia) & 1u) != 0;
        const int fc = a_front ? ia : ib, oc = a_front ? ib : ia;
        const int f = rank[fc], k = nface[f]++;
        const uint32_t rec = (uint32_t)fi | ((uint32_t)rank[oc] << 6) | ((a_front ? 1u : 0u) << 10) | ((uint32_t)oc << 11);
        if (k == 0) w[f] |= rec << 5;
        else if (k == 1) w[12 + f] |= rec;
        else w[12 + f] |= rec << 16;
    }
    w[24] = (uint32_t)F | ((uint32_t)D << 8);
    return true;
}
#endif

}  // namespace nin
