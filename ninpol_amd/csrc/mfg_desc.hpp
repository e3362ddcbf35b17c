// mfg_desc.hpp -- launch-plan descriptor of a node for kernels_gls_mfg.hip, the multifrontal GLS kernel whose dense problem lives in
// GLOBAL-memory tiles (internal, device code): interior nodes with more cells than kernels_gls_mfx.hip's registers hold -- up to 64
// cells (a Poisson-Delaunay cloud has 6 % of its nodes beyond 37 cells, 0.02 % beyond 50).
//
// The decomposition is mfx_desc.hpp's -- fronts = a minimum-degree greedy independent set of the cells with exactly 3 faces at the
// node, the other cells dense, faces between two dense cells free -- with wider limits: F <= 32 fronts, D <= 40 dense cells, 48 free
// faces, dense problem (7 F + D + 3 free) x (3 D + 1) <= 256 x 121.
//
// 124 words per node:
//   word 0            F | D << 8 | free faces << 16
//   word 1 + f        front f: position in the esup row (6 bits) | fsup positions of its faces 0, 1, 2 (7 bits each, << 6, 13, 20) |
//                     bit 27 + i: the front is face i's FIRST cell (side a: row = [-B_a | +B_b], gls.pyx:340-356)
//   word 33 + f       front f: dense slots of the cells across its faces 0, 1, 2 (6 bits each)
//   word 65 .. 74     esup position of dense slot d, one byte each (slots in esup order)
//   word 75 + q       free face q: fsup position (7 bits) | dense slot of its first cell << 7 | of its second cell << 13
#pragma once
#include <cstdint>

#include "device_grid.hpp"

namespace nin {

constexpr int kMfgMaxFronts = 32, kMfgMaxDense = 40, kMfgMaxFree = 48, kMfgDescWords = 124;
constexpr int kMfgMaxRows = 256, kMfgMaxCells = 64, kMfgMaxFaces = 127;
constexpr int kMfgW0 = 1, kMfgW1 = 33, kMfgSlotTable = 65, kMfgFree0 = 75;
constexpr int kMfgRowTiles = 16, kMfgColBlocks = 32;                   // tiles of 16 rows x 4 columns: 256 x 128
constexpr int kMfgSlotDoubles = (kMfgRowTiles * kMfgColBlocks + 2 * kMfgColBlocks) * 64;   // one node's tiles + two auxiliary tiles per panel: 288 KB of global scratch
constexpr int kMfgResidentWaves = 4 * 256;                             // slots at most: one wavefront per SIMD (288 MB)

#ifdef __HIPCC__
// 0: not for this kernel; 1: the words are filled.  Interior nodes only (a boundary face at the node: 0).
__device__ inline int mfg_descriptor(const GridView &g, int32_t p, uint32_t w[kMfgDescWords]) {
    const int32_t eb = g.esup_ptr[p], fb = g.fsup_ptr[p];
    const int ne = g.esup_ptr[p + 1] - eb, nf = g.fsup_ptr[p + 1] - fb;
    if (g.dim != 3 || ne < 2 || ne > kMfgMaxCells || nf > kMfgMaxFaces || nf < 1) return 0;
    if (3 * nf < 2 * ne) return 0;                         // fewer rows than unknowns next to the node value: the zero row
    uint64_t adj[kMfgMaxCells];
    uint8_t deg[kMfgMaxCells], fa[kMfgMaxFaces], fbb[kMfgMaxFaces];
    for (int i = 0; i < ne; ++i) { adj[i] = 0ull; deg[i] = 0; }
    for (int fi = 0; fi < nf; ++fi) {
        const int64_t f = g.fsup[fb + fi];
        const int32_t a = g.face_cells[2 * f], b = g.face_cells[2 * f + 1];
        if (b < 0) return 0;
        int ia = -1, ib = -1;
        for (int i = 0; i < ne; ++i) {
            const int32_t c = g.esup[eb + i];
            ia = c == a ? i : ia;
            ib = c == b ? i : ib;
        }
        if (ia < 0 || ib < 0 || ia == ib) return 0;
        if ((adj[ia] >> ib) & 1ull) return 0;               // two faces between the same pair of cells
        adj[ia] |= 1ull << ib;
        adj[ib] |= 1ull << ia;
        ++deg[ia];
        ++deg[ib];
        fa[fi] = (uint8_t)ia;
        fbb[fi] = (uint8_t)ib;
    }
    uint64_t elig = 0ull;
    for (int i = 0; i < ne; ++i)
        if (deg[i] == 3) elig |= 1ull << i;
    uint64_t best = 0ull;                                  // minimum-residual-degree greedy, as mfx_desc.hpp
    for (int start = 0; start < ne; start += 4) {
        uint64_t chosen = 0ull, avail = elig;
        int n = 0;
        while (avail && n < kMfgMaxFronts) {
            int pick = -1, pd = 99;
            for (int k = 0; k < ne; ++k) {
                const int c = start + k < ne ? start + k : start + k - ne;
                if (!((avail >> c) & 1ull)) continue;
                const int d = __popcll(adj[c] & avail);
                if (d < pd) { pd = d; pick = c; }
            }
            chosen |= 1ull << pick;
            avail &= ~(adj[pick] | (1ull << pick));
            ++n;
        }
        if (n > __popcll(best)) best = chosen;
    }
    const int F = __popcll(best), D = ne - F, nfree = nf - 3 * F;
    if (F < 1 || D < 1 || D > kMfgMaxDense || nfree < 0 || nfree > kMfgMaxFree) return 0;
    if (7 * F + D + 3 * nfree > kMfgMaxRows) return 0;
    uint8_t rank[kMfgMaxCells];                            // front number or dense slot of a cell
    {
        int f = 0, d = 0;
        for (int i = 0; i < ne; ++i) rank[i] = (uint8_t)(((best >> i) & 1ull) ? f++ : d++);
    }
    for (int k = 0; k < kMfgDescWords; ++k) w[k] = 0u;
    w[0] = (uint32_t)F | ((uint32_t)D << 8) | ((uint32_t)nfree << 16);
    for (int i = 0; i < ne; ++i) {
        if ((best >> i) & 1ull) w[kMfgW0 + rank[i]] |= (uint32_t)i;
        else w[kMfgSlotTable + (rank[i] >> 2)] |= (uint32_t)i << (8 * (rank[i] & 3));
    }
    uint8_t nface[kMfgMaxFronts];
    for (int f = 0; f < kMfgMaxFronts; ++f) nface[f] = 0;
    int q = 0;
    for (int fi = 0; fi < nf; ++fi) {
        const int ia = fa[fi], ib = fbb[fi];
        const bool a_front = ((best >> ia) & 1ull) != 0, b_front = ((best >> ib) & 1ull) != 0;
        if (!a_front && !b_front) {
            w[kMfgFree0 + q++] = (uint32_t)fi | ((uint32_t)rank[ia] << 7) | ((uint32_t)rank[ib] << 13);
            continue;
        }
        const int fc = a_front ? ia : ib, oc = a_front ? ib : ia;
        const int f = rank[fc], k = nface[f]++;
        w[kMfgW0 + f] |= ((uint32_t)fi << (6 + 7 * k)) | ((a_front ? 1u : 0u) << (27 + k));
        w[kMfgW1 + f] |= (uint32_t)rank[oc] << (6 * k);
    }
    return 1;
}
#endif

}  // namespace nin
