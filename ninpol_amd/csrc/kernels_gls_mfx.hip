// kernels_gls_mfx.hip -- GLS weights of interior nodes of UNSTRUCTURED meshes (mfx_desc.hpp: up to 16 fronts + 21 dense cells,
// 16 free faces; a Delaunay tetrahedrisation has 14 .. 40 cells around a node and no two-colouring), gfx950: one wavefront per
// node, the dense problem -- up to 160 x 64 -- in 16-row x 4-column tiles of the unified register file: at ONE wavefront per SIMD
// the 512 registers of a lane hold up to 160 tiles (the matrix unit reads and writes the accumulation registers directly).
//
// The system and phase 1 are kernels_gls_mfw.hip's (gls.pyx:252-356; fronts = cells with exactly 3 faces at the node that share
// no face, four lanes per front).  The dense phase is mfw_strips.hpp's strip form (blocked Householder QR, compact WY panels of
// four, the FP64 matrix unit as the cross-lane adder), instantiated per SIZE CLASS -- (6, 10), (7, 11), (8, 13), (9, 15), (10, 16)
// row tiles x column blocks -- each class its own kernel and its own list of the launch plan: straight-line code of the class's
// size.  (Round 4 first built ONE body for every panel of every node, driven by wave-uniform branches so that the work followed
// the node's own size -- mfx_strips.hpp, kept for tools/test_xstrip.hip: a lone wavefront pays ~50 cycles for every taken branch,
// and that form is 1.9 x slower on the same 118 x 48 problem than the unrolled class it falls into.  And a lone wavefront cannot
// overlap anything: tools/micro_overlap.hip -- an FP64 MFMA issued into the shadow of a dependent FP64 chain costs the SUM of the
// two, so look-ahead between the panel factorisation and the trailing update buys nothing here.)  The smallest class, 60 tiles =
// the Kuhn problem's size, runs at two wavefronts per SIMD.  Same mathematics as dgels on the reference's matrix: a Householder QR
// under a column / row order that exposes the zeros.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device_grid.hpp"
#include "gls_device_math.hpp"
#include "launch.hpp"
#include "mfw_strips.hpp"
#include "mfx_desc.hpp"

namespace nin {

namespace {

using namespace glsmath;
using namespace mfwstrips;

// Staging area (phase 1 -> tiles; it lies under R): row r of the dense problem as 13 doubles at 13 r,
//     [0 0 0 | cell 1 (3) | cell 2 (3) | cell 3 (3) | c]
// so that the entry of a column whose cell has code k in the row (0: not one of the row's cells) is at 3 k + component -- code 0
// reads the zeros in front, no select.  A front's 7 fill rows have 3 cells, a free face's 3 rows 2, a dense cell's row 1 (c = 1).
// Rows: 7 f + i (fill row i of front f), 7 F + d (dense cell d), 7 F + D + 3 q + k (row k of free face q), then one per boundary face.
constexpr int XROW = 13;
#ifndef NIN_MFX_TWO_WAVE_TILES
#define NIN_MFX_TWO_WAVE_TILES 104     // classes of up to this many tiles run at two wavefronts per SIMD (256 registers: the 7 x 11 class spills 256 B,
                                       // the 8 x 13 class 528 B a lane -- and still gain 22 % / 12 % from the second wave: A/B in one session)
#endif
#ifndef NIN_MFX_SMALL_WAVES
#define NIN_MFX_SMALL_WAVES 3          // the (4, 7) class: 28 tiles
#endif
template <int TQ, int TCB>
struct XDims {
    static constexpr int RP = 4 * TCB + 1;               // pitch of R in LDS (odd: lane = row reads are conflict-free)
    static constexpr int NP = 4 * TCB - 1;               // pivot rows at most (nc <= 4 TCB - 1: c has a column)
    static constexpr int STAGE = 16 * TQ * XROW, RSZ = NP * RP;
    static constexpr int MAIN = ((STAGE > RSZ ? STAGE : RSZ) + 1) & ~1;
    static constexpr int Y = MAIN, W = Y + 64, Z = W + 40, DESC = Z + 2, PER_WAVE = DESC + kMfxDescWords / 2;
    static constexpr int WAVES = TQ * TCB <= 32 ? NIN_MFX_SMALL_WAVES : TQ * TCB <= NIN_MFX_TWO_WAVE_TILES ? 2 : 1;   // wavefronts per SIMD (60 tiles: the registers of half a SIMD lane hold them)
};

// The dense phase of one node: gather the rows from the staging area into TQ x TCB tiles (16 rows x 4 columns each) and factor
// (mfw_strips.hpp's unrolled strip_factor<TQ, TCB>: straight-line, sweeps the whole class size).
template <int TQ, int TCB, bool BND>
__device__ __forceinline__ double dense_phase(double *Rm, const uint32_t *dl, int nc, int nrows, int F, int D, int nfree, int nbnd, int lane_) {
    using Dm = XDims<TQ, TCB>;
    // (the lane number behind an opaque move, fresh per node: everything the dense phase derives from it -- row / quad / column indices, the
    //  identity strip, the masks of the panel steps -- is otherwise hoisted out of the node loop, lives across phase 1 and ends up in scratch,
    //  to come back word by word inside the panels, each behind a full wait)
    int lane;
    asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(lane_));
    const int si = lane >> 4, sb = (lane >> 2) & 3, sj = lane & 3;
    double C[TQ][TCB];
    // code(sd) = 1 + the index of dense slot sd among the row's cells (0: not one of them), two bits per slot; a front's
    // table is made once (lane f) and shuffled
    uint32_t tlo = 0u, thi = 0u;
    {
        const uint32_t wbl = dl[kMfxW1 + (lane < kMfxMaxFronts ? lane : 0)];
        const uint64_t t = (1ull << (2 * (wbl & 31u))) | (2ull << (2 * ((wbl >> 5) & 31u))) | (3ull << (2 * ((wbl >> 10) & 31u)));
        tlo = lane < F ? (uint32_t)t : 0u;
        thi = lane < F ? (uint32_t)(t >> 32) : 0u;
    }
    // per tile row q, this lane's row: rtl / rth = the 2-bit codes of its cells' slots (slots 0 .. 15 / 16 .. 20), rbz = the byte
    // address of the row in the staging area (a row beyond the node's: row 0, whose codes are all zero then), rbc = of its c
    uint32_t rtl[TQ], rth[TQ], rbz[TQ], rbc[TQ];
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
        const int row = 16 * q + 4 * sb + si;
        const bool fill = row < 7 * F;
        const int f = fill ? (row * 9363) >> 16 : 0, d = row - 7 * F, x = d - D;
        const bool cell = d >= 0 && d < D, fre = x >= 0 && x < 3 * nfree, bnd = BND && x >= 3 * nfree && x < 3 * nfree + nbnd;
        const int qf = fre ? (x * 43) >> 7 : bnd ? x - 2 * nfree : 0;     // (a boundary face's entry lies behind the free faces': nfree + (x - 3 nfree))
        const uint32_t flo = (uint32_t)__shfl((int)tlo, f), fhi = (uint32_t)__shfl((int)thi, f);
        const uint32_t fw = dl[kMfxFree0 + qf];
        const uint64_t qt = (1ull << (2 * ((fw >> 6) & 31u))) | (bnd ? 0ull : 2ull << (2 * ((fw >> 11) & 31u)));
        rtl[q] = fill ? flo : cell ? (d < 16 ? 1u << (2 * d) : 0u) : (fre || bnd) ? (uint32_t)qt : 0u;
        rth[q] = fill ? fhi : cell ? (d >= 16 ? 1u << (2 * (d - 16)) : 0u) : (fre || bnd) ? (uint32_t)(qt >> 32) : 0u;
        const bool have = row < nrows;
        rbz[q] = have ? 8u * XROW * (uint32_t)row : 0u;
        rbc[q] = have ? 8u * (XROW * (uint32_t)row + 12u) : 8u * (uint32_t)Dm::Z;
    }
    const char *const Rb = reinterpret_cast<const char *>(Rm);
    const int cbc = nc >> 2;                             // c: column nc & 3 of block cbc
    const bool is_c_lane = sj == (nc & 3);
#pragma unroll
    for (int cb = 0; cb < TCB; ++cb) {
        // (sjq: sj behind an opaque move, fresh per column block -- otherwise the column arithmetic of all blocks is
        //  hoisted: lane constants that end up in scratch and come back one by one, each behind a full wait)
        int sjq;
        asm volatile("v_mov_b32 %0, %1" : "=v"(sjq) : "v"(sj));
        const int col = 4 * cb + sjq;
        const int sd = (col * 43) >> 7;                  // column = component tt of dense slot sd
        const uint32_t tt8 = 8u * (uint32_t)(col - 3 * sd);
        const int sh = cb < 12 ? 2 * sd : 2 * sd - 32;   // (blocks 0 .. 11: slots 0 .. 15, the low word)
        const uint32_t cmask = (cb == cbc && is_c_lane) ? 0xFFFFFFFFu : 0u;   // this lane's column of this block is c
#pragma unroll
        for (int q = 0; q < TQ; ++q) {
            const uint32_t code = ((cb < 12 ? rtl[q] : rth[q]) >> sh) & 3u;
            const uint32_t a = rbz[q] + 24u * code + tt8;
            C[q][cb] = *reinterpret_cast<const double *>(Rb + ((rbc[q] & cmask) | (a & ~cmask)));
        }
    }
    wave_lds_sync();          // the staging area is R's from here on
    SubStamps ST;
    return strip_factor<TQ, TCB>(C, nc, lane, Rm, Dm::RP, ST);
}

// BND: the list holds BOUNDARY nodes (one instantiation, 7 x 11 tiles: a boundary node has half a node's cells): a Dirichlet one gets its
// zero row at once (gls.pyx:165-166), a Neumann one one more row per boundary face.  The interior instantiations carry none of that code.
template <int TQ, int TCB, bool BND>
__global__ __launch_bounds__(64, (XDims<TQ, TCB>::WAVES)) void nin_gls_mfx_kernel(GridView g, const int32_t *__restrict__ nodes,
                                                                                 const uint32_t *__restrict__ desc, int32_t count,
                                                                                 int add_neumann, double *__restrict__ out,
                                                                                 double *__restrict__ nws, int32_t *__restrict__ queue) {
    using Dm = XDims<TQ, TCB>;
    constexpr int RP = Dm::RP;
    __shared__ double Rm[Dm::PER_WAVE];
    const int lane = threadIdx.x;
    double *const yb = Rm + Dm::Y, *const wbuf = Rm + Dm::W;
    uint32_t *const dl = reinterpret_cast<uint32_t *>(Rm + Dm::DESC);
    const uint8_t *const slotpos = reinterpret_cast<const uint8_t *>(dl + kMfxSlotTable);
    if (lane == 0) { Rm[Dm::Z] = 0.0; Rm[Dm::Z + 1] = 1.0; }

    auto ticket = [&]() -> int32_t {
        int32_t v = 0;
        if (lane == 0) v = atomicAdd(queue, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    for (int32_t idx = ticket(); idx < count; idx = ticket()) {
        const int32_t p = __builtin_amdgcn_readfirstlane(nodes ? nodes[idx] : idx);
        const uint32_t *dw = desc + (size_t)kMfxDescWords * idx;
        if (lane < kMfxDescWords) dl[lane] = dw[lane];
        const uint32_t fd = (uint32_t)__builtin_amdgcn_readfirstlane((int)dw[0]);
        const int F = fd & 255, D = (fd >> 8) & 255, nfree = (fd >> 16) & 255, nbnd = BND ? (int)(fd >> 24) : 0, ne = F + D;
        const uint32_t eb = (uint32_t)__builtin_amdgcn_readfirstlane(g.esup_ptr[p]);
        const uint32_t fb = (uint32_t)__builtin_amdgcn_readfirstlane(g.fsup_ptr[p]);
        const int flg = __builtin_amdgcn_readfirstlane((int)g.flags[p]);
        const bool is_neu = (flg & 2) != 0;
        if (BND && (flg & 1) && !is_neu) {                       // a Dirichlet boundary node (gls.pyx:165-166): the zero row, nothing computed
            if (lane < ne) out[eb + lane] = 0.0;
            if (lane == 0) nws[p] = 0.0;
            wave_lds_sync();                                     // (the descriptor words of the next node go where this one's are being written)
            continue;
        }
        const double xv0 = g.coords[3 * (size_t)p + 0], xv1 = g.coords[3 * (size_t)p + 1], xv2 = g.coords[3 * (size_t)p + 2];
        wave_lds_sync();
        // phase 1 works with FOUR lanes per front: lane 4 f + j applies the front's reflectors to c (j = 0) or to the columns of
        // the front's neighbour j - 1; dense cell d's centroid is fetched by lane d
        const int fq = lane >> 2, jq = lane & 3;
        const uint32_t wa = dl[kMfxW0 + fq], wb = dl[kMfxW1 + fq];
        const uint32_t pe = wa & 63u;
        const uint32_t po = slotpos[lane < kMfxMaxDense ? lane : 0];
        const int my = jq > 0 ? jq - 1 : 0;                      // this lane's face (the c lane computes face 0's neighbour side in vain)
        const uint32_t myslot = (wb >> (5 * my)) & 31u;
        // ---- phase 1: the front of cell E_f (rows 0 = cell row, 1 + 3 i + r = row r of face i) ------------------------------
        // u: this lane's block of u = z^T R_ed (3 columns), or s = z . b_e in u[0] of the c lane
        double u[3], de[3], dod[3];
        {
            const uint32_t ce = (uint32_t)g.esup[eb + pe], co = (uint32_t)g.esup[eb + po];
            const uint32_t cm = (uint32_t)g.esup[eb + slotpos[myslot]];
            double P[10][3], B[10][3];
            double Ke[9], Km[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) { Ke[k] = g.perm[9 * (size_t)ce + k]; Km[k] = g.perm[9 * (size_t)cm + k]; }
            const double dme = g.diff_mag[ce];
            de[0] = g.centroids[3 * (size_t)ce + 0] - xv0;      // gls.pyx:269-277
            de[1] = g.centroids[3 * (size_t)ce + 1] - xv1;
            de[2] = g.centroids[3 * (size_t)ce + 2] - xv2;
            dod[0] = g.centroids[3 * (size_t)co + 0] - xv0;
            dod[1] = g.centroids[3 * (size_t)co + 1] - xv1;
            dod[2] = g.centroids[3 * (size_t)co + 2] - xv2;
#pragma unroll
            for (int t = 0; t < 3; ++t) { P[0][t] = de[t]; B[0][t] = (jq == 0 && t == 0) ? 1.0 : 0.0; }   // c = e_0 on entry
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                // B = [K N; T1; tau T2] (gls.pyx:293-321), row = [-B_a | +B_b] (gls.pyx:340-356)
                const uint32_t f = (uint32_t)g.fsup[fb + ((wa >> (6 + 6 * i)) & 63u)];
                const uint32_t cn = (uint32_t)g.esup[eb + slotpos[(wb >> (5 * i)) & 31u]];
                const double N0 = (double)g.face_normal[3 * (size_t)f + 0], N1 = (double)g.face_normal[3 * (size_t)f + 1],
                             N2 = (double)g.face_normal[3 * (size_t)f + 2];
                const double T0 = xv0 - g.face_center[3 * (size_t)f + 0], T1 = xv1 - g.face_center[3 * (size_t)f + 1],
                             T2 = xv2 - g.face_center[3 * (size_t)f + 2];
                const double U0 = N1 * T2 - N2 * T1, U1 = N2 * T0 - N0 * T2, U2 = N0 * T1 - N1 * T0;
                const double dmn = g.diff_mag[cn];
                double eta = 0.0;
                eta = dme > eta ? dme : eta;
                eta = dmn > eta ? dmn : eta;
                const double tj = face_tau(sqrt(U0 * U0 + U1 * U1 + U2 * U2), eta);
                const double sg = ((wa >> (24 + i)) & 1u) ? -1.0 : 1.0;
                const bool mine = jq > 0 && my == i;
                const double s0[3] = {sg * T0, sg * T1, sg * T2}, s1[3] = {sg * (tj * U0), sg * (tj * U1), sg * (tj * U2)};
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    P[1 + 3 * i][t] = sg * (Ke[t * 3 + 0] * N0 + Ke[t * 3 + 1] * N1 + Ke[t * 3 + 2] * N2);
                    P[2 + 3 * i][t] = s0[t];
                    P[3 + 3 * i][t] = s1[t];
                    // the neighbour's side of the face rows, in the lane that owns this face only (zero elsewhere)
                    const double nb = -sg * (Km[t * 3 + 0] * N0 + Km[t * 3 + 1] * N1 + Km[t * 3 + 2] * N2);
                    B[1 + 3 * i][t] = mine ? nb : 0.0;
                    B[2 + 3 * i][t] = mine ? -s0[t] : 0.0;
                    B[3 + 3 * i][t] = mine ? -s1[t] : 0.0;
                }
            }
            // panel: three Householder steps on the own columns; v_k stays in P[k..9][k]; z = R_ee^-T d_e
            double g3[3], z[3];
            front_panel(P, de, g3, z);
            // the reflectors on this lane's block; rows 0..2 -> u (or s); rows 3..9 -> the staging area
            __builtin_amdgcn_sched_barrier(0);
            apply_panel<3, true, true, true, true>(P, g3, B);
#pragma unroll
            for (int t = 0; t < 3; ++t) u[t] = fma(z[2], B[2][t], fma(z[1], B[1][t], z[0] * B[0][t]));
            if (fq < F) {
                // the c lane writes c and the row's three zeros, the others their cell's three entries
                double *stg = Rm + 7 * XROW * fq + (jq == 0 ? 0 : 3 + 3 * my);   // rows 7 f .. 7 f + 6 of the dense problem
#pragma unroll
                for (int r = 0; r < 7; ++r) {
                    stg[r * XROW + 0] = jq == 0 ? 0.0 : B[3 + r][0];
                    stg[r * XROW + 1] = jq == 0 ? 0.0 : B[3 + r][1];
                    stg[r * XROW + 2] = jq == 0 ? 0.0 : B[3 + r][2];
                }
                if (jq == 0) {
#pragma unroll
                    for (int r = 0; r < 7; ++r) stg[r * XROW + 12] = B[3 + r][0];
                }
            }
        }
        if (lane < nfree) {
            // a free face (both its cells dense): its three rows [-B_a | +B_b] (gls.pyx:293-356) go straight into the dense problem
            const uint32_t fw = dl[kMfxFree0 + lane];
            const uint32_t f = (uint32_t)g.fsup[fb + (fw & 63u)];
            const uint32_t ca_ = (uint32_t)g.esup[eb + slotpos[(fw >> 6) & 31u]], cb_ = (uint32_t)g.esup[eb + slotpos[(fw >> 11) & 31u]];
            const double N0 = (double)g.face_normal[3 * (size_t)f + 0], N1 = (double)g.face_normal[3 * (size_t)f + 1],
                         N2 = (double)g.face_normal[3 * (size_t)f + 2];
            const double T0 = xv0 - g.face_center[3 * (size_t)f + 0], T1 = xv1 - g.face_center[3 * (size_t)f + 1],
                         T2 = xv2 - g.face_center[3 * (size_t)f + 2];
            const double U0 = N1 * T2 - N2 * T1, U1 = N2 * T0 - N0 * T2, U2 = N0 * T1 - N1 * T0;
            const double da = g.diff_mag[ca_], db = g.diff_mag[cb_];
            double eta = 0.0;
            eta = da > eta ? da : eta;
            eta = db > eta ? db : eta;
            const double tj = face_tau(sqrt(U0 * U0 + U1 * U1 + U2 * U2), eta);
            const double *Ka = g.perm + 9 * (size_t)ca_, *Kb = g.perm + 9 * (size_t)cb_;
            double *fs = Rm + XROW * (7 * F + D + 3 * lane);    // rows 7 F + D + 3 q .. + 2
            const double Tv[3] = {T0, T1, T2}, Uv[3] = {tj * U0, tj * U1, tj * U2};
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                fs[0 * XROW + t] = 0.0; fs[1 * XROW + t] = 0.0; fs[2 * XROW + t] = 0.0;
                fs[0 * XROW + 3 + t] = -(Ka[t * 3 + 0] * N0 + Ka[t * 3 + 1] * N1 + Ka[t * 3 + 2] * N2);
                fs[0 * XROW + 6 + t] = Kb[t * 3 + 0] * N0 + Kb[t * 3 + 1] * N1 + Kb[t * 3 + 2] * N2;
                fs[1 * XROW + 3 + t] = -Tv[t];
                fs[1 * XROW + 6 + t] = Tv[t];
                fs[2 * XROW + 3 + t] = -Uv[t];
                fs[2 * XROW + 6 + t] = Uv[t];
            }
            fs[0 * XROW + 12] = 0.0; fs[1 * XROW + 12] = 0.0; fs[2 * XROW + 12] = 0.0;
        } else if (BND && lane < nfree + nbnd) {
            // a boundary face of a Neumann node: ONE row, -(K N) on its cell's columns (gls.pyx:394-416; the right-hand side it carries
            // in the reference sits in a column the last-row identity never reads)
            const uint32_t fw = dl[kMfxFree0 + lane];
            const uint32_t f = (uint32_t)g.fsup[fb + (fw & 63u)];
            const uint32_t ca_ = (uint32_t)g.esup[eb + slotpos[(fw >> 6) & 31u]];
            const double N0 = (double)g.face_normal[3 * (size_t)f + 0], N1 = (double)g.face_normal[3 * (size_t)f + 1],
                         N2 = (double)g.face_normal[3 * (size_t)f + 2];
            const double *Ka = g.perm + 9 * (size_t)ca_;
            double *fs = Rm + XROW * (7 * F + D + 3 * nfree + (lane - nfree));
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                fs[t] = 0.0;
                fs[3 + t] = -(Ka[t * 3 + 0] * N0 + Ka[t * 3 + 1] * N1 + Ka[t * 3 + 2] * N2);
            }
            fs[12] = 0.0;
        }
        // the dense cells' rows: (x_K - x_v) on the cell's own columns, c = 1
        if (lane < D) {
            double *cd = Rm + XROW * (7 * F + lane);
            cd[0] = 0.0; cd[1] = 0.0; cd[2] = 0.0;
            cd[3] = dod[0]; cd[4] = dod[1]; cd[5] = dod[2];
            cd[12] = 1.0;
        }
        wave_lds_sync();

        const int nc = 3 * D;                                  // the columns 0 .. nc - 1 are pivoted, column nc is c
        const int nrows = 7 * F + D + 3 * nfree + nbnd;
        const double rr = dense_phase<TQ, TCB, BND>(Rm, dl, nc, nrows, F, D, nfree, nbnd, lane);
        wave_lds_sync();
        // ---- R y = (Q^T c)(0:nc) by columns: lane = row ------------------------------------------------------------------
        {
            const int li = lane < nc ? lane : 0;
            double ct = lane < nc ? Rm[li * RP + nc] : 0.0;
            const double ri = fast_rcp(Rm[li * RP + li]);
            const double *Rl = Rm + li * RP;
            int kb = (nc - 1) & ~3;
            double c4[4], n4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) c4[j] = Rl[kb + j];   // (unguarded: a column right of nc is a word of the row nobody uses, a lane beyond nc reads row 0 in vain)
            for (; kb >= 0; kb -= 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) n4[j] = Rl[(kb >= 4 ? kb - 4 : 0) + j];
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    const int k = kb + j;
                    if (k < nc) {                                             // (wave-uniform: only the first round can be short)
                        const double yk = rl64(ct * ri, k);
                        ct = lane < k ? fma(-yk, c4[j], ct) : ct;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) c4[j] = n4[j];
            }
            if (lane < nc) yb[lane] = ct * ri;
        }
        wave_lds_sync();
        // ---- residuals on the cell rows, weights ---------------------------------------------------------------------------
        {
            // r_e = 1 - d_e . y_e = 1 - z . b_e + u . y_dense: the c lane brings 1 - s, the three others their block of u . y
            const int sl = (int)myslot;
            double part = fma(u[2], yb[3 * sl + 2], fma(u[1], yb[3 * sl + 1], u[0] * yb[3 * sl]));
            part = jq == 0 ? 1.0 - u[0] : part;
            part += dpp_mov<0xB1>(part);   // quad_perm [1,0,3,2]
            part += dpp_mov<0x4E>(part);   // quad_perm [2,3,0,1]
            const double re = part;
            const int ld = lane < D ? lane : 0;
            const double ro = 1.0 - fma(dod[2], yb[3 * ld + 2], fma(dod[1], yb[3 * ld + 1], dod[0] * yb[3 * ld]));
            const double rri = fast_rcp(rr);
            double we = re * rri, wo = ro * rri;
            // rank-deficient system (or NaN from a zero column): undefined in the reference, the zero row here
            const bool ok = rr > 0.0;
            we = (ok && __builtin_isfinite(we)) ? we : 0.0;
            wo = (ok && __builtin_isfinite(wo)) ? wo : 0.0;
            if (fq < F && jq == 0) wbuf[pe] = we;
            if (lane < D) wbuf[po] = wo;
        }
        wave_lds_sync();
        {
            // gls.pyx:470-472 (only if an interior node carries the Neumann flag): neumann_ws = the last cell's weight
            const double nwv = is_neu ? wbuf[ne - 1] : 0.0;
            const double addv = add_neumann ? nwv : 0.0;
            if (lane < ne) out[eb + lane] = wbuf[lane] + addv;
            if (lane == 0) nws[p] = nwv;
        }
        wave_lds_sync();
    }
}

__global__ void k_mfx_desc(GridView g, const int32_t *__restrict__ nodes, int32_t count, uint32_t *__restrict__ desc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t w[kMfxDescWords];
    if (!mfx_descriptor(g, nodes ? nodes[i] : (int32_t)i, w)) {   // (the list holds classified nodes only)
#pragma unroll
        for (int k = 0; k < kMfxDescWords; ++k) w[k] = 0u;
    }
#pragma unroll
    for (int k = 0; k < kMfxDescWords; ++k) desc[kMfxDescWords * i + k] = w[k];
}

}  // namespace

int launch_mfx_desc(const GridView &g, const int32_t *nodes, int32_t count, uint32_t *desc, hipStream_t stream) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(k_mfx_desc, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, stream, g, nodes, count, desc);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// `cls`: the size class of every node of the list (mfx_desc.hpp: mfx_size_class)
int launch_gls_mfx(const GridView &g, const int32_t *nodes, const uint32_t *desc, int32_t count, int cls, int add_neumann, double *out,
                   double *nws, int32_t *queue, hipStream_t stream) {
    if (count <= 0) return 0;
    if (cls < 0 || cls > kMfxClasses + 2) return -1;     // (cls == kMfxClasses: the boundary nodes' list, + 1: the small interior class, + 2: (7, 12))
    int64_t blocks = count;
    // persistent: one wavefront per workgroup, one (class 0: two) per SIMD -- the register file of a SIMD lane belongs to one (two) node(s)
    constexpr int tiles[kMfxClasses + 3] = {6 * 10, 7 * 11, 8 * 13, 9 * 15, 10 * 16, 7 * 11, 4 * 7, 7 * 12};
    const int64_t cap = 4 * 256 * (tiles[cls] <= 32 ? NIN_MFX_SMALL_WAVES : tiles[cls] <= NIN_MFX_TWO_WAVE_TILES ? 2 : 1);
    if (blocks > cap) blocks = cap;
#define NIN_MFX_LAUNCH(TQ, TCB, BND)                                                                                                      \
    hipLaunchKernelGGL((nin_gls_mfx_kernel<TQ, TCB, BND>), dim3((unsigned)blocks), dim3(64), 0, stream, g, nodes, desc, count, add_neumann, out, \
                       nws, queue)
    if (cls == 0) NIN_MFX_LAUNCH(6, 10, false);
    else if (cls == 1) NIN_MFX_LAUNCH(7, 11, false);
    else if (cls == 2) NIN_MFX_LAUNCH(8, 13, false);
    else if (cls == 3) NIN_MFX_LAUNCH(9, 15, false);
    else if (cls == 4) NIN_MFX_LAUNCH(10, 16, false);
    else if (cls == 5) NIN_MFX_LAUNCH(7, 11, true);
    else if (cls == 6) NIN_MFX_LAUNCH(4, 7, false);
    else NIN_MFX_LAUNCH(7, 12, false);
#undef NIN_MFX_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

const char *kernel_name_gls_mfx() { return "nin_gls_mfx_kernel"; }

}  // namespace nin
