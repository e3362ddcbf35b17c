// quad4_desc.hpp -- launch-plan descriptor of a "quad" node for kernels_gls_quad4.hip (internal, device code).
//
// A node with 4 cells and 8 faces, 4 of them internal and 4 on the boundary, every cell touching exactly two of the internal
// faces and one of the boundary faces, has a cell-adjacency graph that is a 4-CYCLE: bipartite, two "even" cells E0, E1 that
// share no face, two "odd" cells O0, O1, every even cell adjacent to BOTH odd cells.  Every node inside a boundary face of a
// hexahedron mesh is one; such nodes are computed only when the variable flags them Neumann (gls.pyx:165-166), and then a whole
// boundary plane of them is.  Anything else stays with the small-node kernel (kernels_gls_mfw.hip).
//
// The labelling is canonical: E0 = the row's first cell, O0 / O1 = its two neighbours in esup order, E1 = the fourth cell.
// Lane e of the node's pair works on E_e and owns the rows of O_e.  One 32-bit word per lane:
//   bits 0-1  position of E_e in the node's esup row     bits 2-3  position of O_e
//   bits 4-6  face A (between E_e and O0): position in the fsup row, bit 7: E_e is its first cell (row = [-B_a | +B_b])
//   bits 8-10 face B (between E_e and O1), bit 11 likewise
//   bits 12-14 the boundary face of E_e, bits 15-17 the boundary face of O_e (positions in the fsup row)
//   bits 18-19 position of O0, bits 20-21 position of O1 in the esup row
#pragma once
#include <cstdint>

#include "device_grid.hpp"

namespace nin {

#ifdef __HIPCC__
__device__ inline bool quad4_descriptor(const GridView &g, int32_t p, int32_t d[2]) {
    const int32_t eb = g.esup_ptr[p], fb = g.fsup_ptr[p];
    if (g.esup_ptr[p + 1] - eb != 4 || g.fsup_ptr[p + 1] - fb != 8 || g.dim != 3) return false;
    int32_t cells[4];
    for (int i = 0; i < 4; ++i) cells[i] = g.esup[eb + i];
    // per cell: its two internal faces as (face position | other cell << 3 | side a << 5), 6 bits each; its boundary face
    uint32_t adj[4] = {0, 0, 0, 0};
    int deg[4] = {0, 0, 0, 0}, bnd[4] = {-1, -1, -1, -1};
    for (int fi = 0; fi < 8; ++fi) {
        const int64_t f = g.fsup[fb + fi];
        const int32_t a = g.face_cells[2 * f], b = g.face_cells[2 * f + 1];
        int ia = -1, ib = -1;
        for (int i = 0; i < 4; ++i) {
            ia = cells[i] == a ? i : ia;
            ib = cells[i] == b ? i : ib;
        }
        if (ia < 0) return false;
        if (b < 0) {
            if (bnd[ia] >= 0) return false;
            bnd[ia] = fi;
            continue;
        }
        if (ib < 0 || ia == ib || deg[ia] >= 2 || deg[ib] >= 2) return false;
        adj[ia] |= (uint32_t)(fi | (ib << 3) | (1 << 5)) << (6 * deg[ia]);
        adj[ib] |= (uint32_t)(fi | (ia << 3)) << (6 * deg[ib]);
        ++deg[ia];
        ++deg[ib];
    }
    for (int i = 0; i < 4; ++i)
        if (deg[i] != 2 || bnd[i] < 0) return false;
    const int n0 = (adj[0] >> 3) & 3, n1 = (adj[0] >> 9) & 3;
    if (n0 == n1 || n0 == 0 || n1 == 0) return false;
    const int o0 = n0 < n1 ? n0 : n1, o1 = n0 < n1 ? n1 : n0;
    const int e1 = 6 - o0 - o1;   // 0 + 1 + 2 + 3 minus the three others
    if (e1 < 1 || e1 > 3 || e1 == o0 || e1 == o1) return false;
    const int even[2] = {0, e1}, odd[2] = {o0, o1};
    for (int e = 0; e < 2; ++e) {
        const int c = even[e];
        uint32_t fa = 0xFFu, fbw = 0xFFu;
        for (int k = 0; k < 2; ++k) {
            const uint32_t rec = (adj[c] >> (6 * k)) & 63u;
            const int other = (rec >> 3) & 3;
            const uint32_t packed = (rec & 7u) | (((rec >> 5) & 1u) << 3);   // position | side a << 3
            if (other == o0) fa = packed;
            else if (other == o1) fbw = packed;
            else return false;
        }
        if (fa == 0xFFu || fbw == 0xFFu) return false;
        d[e] = (int32_t)((uint32_t)c | ((uint32_t)odd[e] << 2) | (fa << 4) | (fbw << 8) | ((uint32_t)bnd[c] << 12) |
                         ((uint32_t)bnd[odd[e]] << 15) | ((uint32_t)o0 << 18) | ((uint32_t)o1 << 20));
    }
    return true;
}
#endif

}  // namespace nin
