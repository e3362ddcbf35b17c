// hex8_desc.hpp -- launch-plan descriptor of a "cube" node for kernels_gls_hex8mf.hip (internal, device code).
//
// A node with 8 cells and 12 faces, all internal, each cell touching exactly 3 of those faces, has a 3-regular
// cell-adjacency graph on 8 vertices (cells = vertices, faces = edges).  The multifrontal kernel needs that graph
// to be the CUBE graph: bipartite, 4 "even" cells E0..E3 sharing no face, 4 "odd" cells O0..O3, every even cell
// adjacent to all odd cells but one.  Every interior node of a hexahedron mesh is one (so is a node surrounded
// by 8 tetrahedra in the octahedral arrangement); anything else falls back to kernels_gls_block.hip.
//
// The labelling is canonical so that the kernel's register layout is fixed at compile time:
//   E_l  = the l-th even cell in esup order (even = the colour class of the row's first cell),
//   O_(3-l) = the odd cell NOT adjacent to E_l,
//   lane l of the node's quad works on E_l, owns the cell row of O_l, and holds the 3 faces of E_l ordered by the
//   odd slot of the cell on their other side (ascending: the slots {0,1,2,3} \ {3 - l}).
// One 32-bit word per lane:
//   bits 0-2   position of E_l in the node's esup row        bits 3-5   position of O_l
//   bits 6+8i .. 13+8i (i = 0,1,2), face i of E_l:  4 bits position in the fsup row, 3 bits position of the
//   neighbour cell in the esup row, 1 bit "E_l is the first cell (side a) of the face" (row = [-B_a | +B_b]).
#pragma once
#include <cstdint>

#include "device_grid.hpp"

namespace nin {

#ifdef __HIPCC__
__device__ inline bool hex8_descriptor(const GridView &g, int32_t p, int32_t d[4]) {
    const int32_t eb = g.esup_ptr[p], fb = g.fsup_ptr[p];
    if (g.esup_ptr[p + 1] - eb != 8 || g.fsup_ptr[p + 1] - fb != 12 || g.dim != 3) return false;
    int32_t cells[8];
    for (int i = 0; i < 8; ++i) cells[i] = g.esup[eb + i];
    // per cell: up to 3 (neighbour, face, side) entries packed 8 bits each, and the degree
    uint32_t adj[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int deg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int fi = 0; fi < 12; ++fi) {
        const int64_t f = g.fsup[fb + fi];
        const int32_t a = g.face_cells[2 * f], b = g.face_cells[2 * f + 1];
        if (b < 0) return false;
        int ia = -1, ib = -1;
        for (int i = 0; i < 8; ++i) {
            ia = cells[i] == a ? i : ia;
            ib = cells[i] == b ? i : ib;
        }
        if (ia < 0 || ib < 0 || ia == ib || deg[ia] >= 3 || deg[ib] >= 3) return false;
        adj[ia] |= (uint32_t)(fi | (ib << 4) | (1 << 7)) << (8 * deg[ia]);
        adj[ib] |= (uint32_t)(fi | (ia << 4)) << (8 * deg[ib]);
        ++deg[ia];
        ++deg[ib];
    }
    for (int i = 0; i < 8; ++i)
        if (deg[i] != 3) return false;
    // two-colouring from cell 0
    int color[8] = {0, -1, -1, -1, -1, -1, -1, -1};
    for (int sweep = 0; sweep < 4; ++sweep)
        for (int i = 0; i < 8; ++i)
            if (color[i] >= 0)
                for (int k = 0; k < 3; ++k) {
                    const int j = (adj[i] >> (8 * k + 4)) & 7;
                    if (color[j] < 0) color[j] = 1 - color[i];
                }
    int n_even = 0;
    for (int i = 0; i < 8; ++i) {
        if (color[i] < 0) return false;
        n_even += color[i] == 0;
        for (int k = 0; k < 3; ++k)
            if (color[(adj[i] >> (8 * k + 4)) & 7] == color[i]) return false;
    }
    if (n_even != 4) return false;
    int even[4], oslot[8] = {-1, -1, -1, -1, -1, -1, -1, -1}, odd[4] = {-1, -1, -1, -1};
    for (int i = 0, l = 0; i < 8; ++i)
        if (color[i] == 0) even[l++] = i;
    for (int l = 0; l < 4; ++l) {
        const int e = even[l];
        unsigned seen = 0;
        for (int k = 0; k < 3; ++k) seen |= 1u << ((adj[e] >> (8 * k + 4)) & 7);
        int missing = -1, n_missing = 0;
        for (int i = 0; i < 8; ++i)
            if (color[i] == 1 && !((seen >> i) & 1)) { missing = i; ++n_missing; }
        if (n_missing != 1 || oslot[missing] >= 0) return false;   // (a doubled face would leave two cells out)
        oslot[missing] = 3 - l;
        odd[3 - l] = missing;
    }
    for (int l = 0; l < 4; ++l) {
        const int e = even[l];
        uint32_t w = (uint32_t)e | ((uint32_t)odd[l] << 3);
        // the three faces ordered by the odd slot of their other cell
        int ord[3] = {0, 1, 2};
        for (int x = 0; x < 2; ++x)
            for (int y = 0; y < 2 - x; ++y) {
                const int sa = oslot[(adj[e] >> (8 * ord[y] + 4)) & 7], sb = oslot[(adj[e] >> (8 * ord[y + 1] + 4)) & 7];
                if (sa > sb) { const int t = ord[y]; ord[y] = ord[y + 1]; ord[y + 1] = t; }
            }
        for (int i = 0; i < 3; ++i) w |= ((adj[e] >> (8 * ord[i])) & 0xFFu) << (6 + 8 * i);
        d[l] = (int32_t)w;
    }
    return true;
}
#endif

}  // namespace nin
