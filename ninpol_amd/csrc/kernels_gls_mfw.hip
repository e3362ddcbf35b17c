// kernels_gls_mfw.hip -- GLS weights of interior nodes whose cells split into fronts and dense cells (mfw_desc.hpp:
// Kuhn-type tetrahedron meshes, wedge meshes, the interfaces and pyramid apexes of mixed meshes, ...), gfx950: ONE
// wavefront per node, the whole factorisation in registers.
//
// The system (gls.pyx:252-356) as in kernels_gls_hex8mf.hip: unknowns = a gradient per cell (3 columns) + the node
// value (the column c that the last-row identity turns into a right-hand side); rows = one per cell and three per
// internal face, a face row coupling exactly its two cells.  F "front" cells share no face (each has exactly 3 faces
// at the node), the D others are "dense" cells; a face joins a front to a dense cell (all of them if the cell graph is
// bipartite: the two-coloured kind) or two dense cells (a free face, general kind: its rows join the dense problem).
//
//   phase 1  the 3 Householder steps on a front cell's own columns touch only its 10 rows (cell row + 3 x 3 face rows):
//            all F fronts at once, in-lane as the quad lanes of the hex8 kernel do it, FOUR lanes per front -- each
//            factors the 10 x 3 panel (redundantly) and applies it to one block of the front's other columns: c, or
//            the three columns of one dense neighbour.  Each front leaves 3 rows of R, folded at once into what the
//            weights need of them (z = R_ee^-T d_e, u = z^T R_ed, s = z . b_e), and 7 fill rows over the 9 columns
//            of its three dense neighbours + c, which go through LDS into
//   phase 2  the dense (7 F + D) x (3 D + 1) problem -- Kuhn tetrahedra: 96 x 37 -- in one wavefront's registers, no
//            barrier, no partial sums (the block kernel spends 4 wavefronts, an LDS round trip per row and two workgroup
//            barriers per step on the same sweep, and ~5 x the instructions).  Two forms:
//            lane = ROW: lane r holds a non-pivot row in a[36] and, r < 36, pivot row r in b[36] (large instantiation,
//              up to 12 + 12 cells; the small one -- wedge 6 + 6, cube 4 + 4 nodes -- has all its 48 rows in ONE array,
//              pivot rows first); the columns are static register indices.  With v = the pivot column's entries and
//              alpha - beta in the pivot lane, ONE reduction per column gives w_j = g (v . C_j), w_j goes to a scalar
//              register pair and every row's update is one FMA per array.  Four columns are reduced together
//              (v_permlane32_swap / v_permlane16_swap bring them into the four 16-lane rows of one register, a DPP row
//              reduction finishes them); the arrays move down by three registers after every cell so that one step body
//              serves all cells, column groups that hold only zeros being skipped; a retired pivot row keeps row t of
//              R in its lane's registers, three columns of it are saved to LDS per cell.
//            lane = COLUMN (the first form, kept behind NIN_MFW_LANE_COLUMNS as the A/B baseline): a step broadcasts
//              the pivot column's entry of a row with v_readlane, updates the row and accumulates the next pivot
//              column's dots in the same pass; the non-pivot rows are register pairs, the 3 D pivot rows live in LDS
//              where row t of R replaces pivot row t in place.  v_readlane costs ~2 FP64 FMAs of issue time
//              (tools/micro_readlane.hip), 4 of them per row and step: that is why the rows went into the lanes
//              (tet40: 3.8 -> 1.8 ms, wedge60: 4.1 -> 2.95 ms);
//   then     R y = Q^T c by columns (lane = row, R through LDS), r_i = 1 - d_i . y_i per cell, weights r_i / (r . r)
//            (the identity X[n-1, i] = r_i / (r.r)).
// Same mathematics as dgels on the reference's matrix -- a Householder QR in a column order that exposes the zeros.
// Two wavefronts per SIMD (<= 256 registers; three for the small instantiation), 8 nodes in flight per CU against 2 for
// the block kernel.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device_grid.hpp"
#include "gls_device_math.hpp"
#include "launch.hpp"
#include "mfw_desc.hpp"
#include "mfw_strips.hpp"

namespace nin {

namespace {

using namespace glsmath;
using namespace mfwstrips;

// Diagnostic build (-DNIN_MFW_STAMPS, tools/stamps_mfw.py): wavefront 0 of workgroup 0 records s_memtime at the phase
// boundaries of its 5th node into the neumann_ws entries of the first listed nodes instead of results.
#ifdef NIN_MFW_STAMPS
#define NIN_MFW_STAMP(J) do { if (stamping) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); stamps[J] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define NIN_MFW_STAMP(J) do { } while (0)
#endif

constexpr int STAGE_F = 70;   // per front: 7 fill rows x (9 neighbour columns + c)

// Sizes for nodes with at most FM fronts, DM dense cells and (GENERAL) kMfwMaxFree free faces.  Dense rows: 7 per front,
// then the D dense cells' rows, then 3 per free face (<= kMfwMaxRows in all).  ROWS_IN_LANES: rows 0 .. NP-1 (NP = 3 DM,
// the ones that get pivoted) in b[], the next 64 in a[], the rest in b[]'s lanes from NP on.  Otherwise (the first form)
// the pivot rows live in LDS -- row t of R replaces pivot row t in place -- and the other NREG in registers.
template <int FM, int DM, bool ROWS_IN_LANES, bool GENERAL>
struct MfwDims {
    static constexpr int NP = 3 * DM, NREG = 7 * FM - 2 * DM, DROW0 = 7 * FM;
    static constexpr int RP = 4 * ((NP + 1 + 3) / 4) + 1;   // pitch of R in LDS: whole column blocks of four (the strip form stores a pivot row's blocks unguarded) + 1 (odd: no bank conflicts)
    static constexpr int TOTAL = GENERAL ? kMfwMaxRows : 7 * FM + DM;
    static constexpr int PAD = ROWS_IN_LANES ? 0 : 8;   // zero rows behind the pivot rows: the LDS sweep runs in whole groups of 4 and reads one group ahead
    // the strip form stages whole rows of the dense problem, 13 doubles each: [0 0 0 | cell 1 | cell 2 | cell 3 | c] at 13 row (kernels_gls_mfx.hip's
    // layout: the entry of a column whose cell has code k in the row is at 3 k + component, code 0 = the zeros in front -- no select, no branch)
    static constexpr int XROW = 13, XSTAGE = (ROWS_IN_LANES && !GENERAL) ? (7 * FM + DM) * XROW : 0;
    static constexpr int STAGE0 = FM * STAGE_F + (GENERAL ? 18 * kMfwMaxFree : 0);   // phase 1's staging area (fronts' fill rows, free faces' rows)
    static constexpr int STAGE = STAGE0 > XSTAGE ? STAGE0 : XSTAGE;
    static constexpr int LDS_FS = FM * STAGE_F;
    static constexpr int LDS_R = ((NP + PAD) * RP > STAGE ? (NP + PAD) * RP : STAGE);   // R rows; the staging area lies under them
    static constexpr int LDS_Y = LDS_R, LDS_D = LDS_Y + 3 * DM + 4, LDS_W = LDS_D + 3 * DM + 4, LDS_Z = LDS_W + FM + DM,
                         LDS_PER_WAVE = LDS_Z + 66;   // (LDS_Z: 64 zeros, what a lane without an entry in a fill row reads; then a one)
    static_assert(NREG > 0 && NP % 2 == 0 || ROWS_IN_LANES, "row split");
    static_assert(!GENERAL || (ROWS_IN_LANES && TOTAL <= 128 && NP <= 64 && FM * 4 + kMfwMaxFree <= 64 && 26 + kMfwMaxFree <= kMfwDescWords), "lanes");
};

// One Householder step with the rows in the lanes: lane r holds row 36 + r of the dense problem in a[] (r < 60) and,
// r < 36, pivot row r in b[]; registers S .. live - 1 of a[] / b[] are the live columns (S the pivot column; the
// registers from `live` on are zero), ca / cb the right-hand side c.  The reflector is v = (the pivot column's entries;
// alpha - beta in the pivot row's lane), so one reduction per column gives w_j = g (v . C_j) directly, w_j goes to a
// scalar register pair and every row's update is one FMA per array: two v_readlane per COLUMN and step instead of four
// per ROW and step.  Columns go four at a time (c rides with the first three); groups that lie in the zero registers
// altogether are skipped (wave-uniform), so ONE body serves every cell.
// the register behind slot q of column group m of step S: -1 = the right-hand side c, -2 = the squared norm of the NEXT
// pivot column (it rides along, so only a node's first step needs a reduction of its own), >= 36 = empty
__device__ constexpr int slot_reg(int S, int m, int q) {
    return m == 0 ? (q == 0 ? -1 : S + q) : m == 1 ? (q == 0 ? -2 : S + 3 + q) : S + 4 * m - 1 + q;
}
// NC = 3 DM columns.  TWO: the rows fill two arrays (large instantiation: 60 non-pivot rows in a[], 36 pivot rows in
// b[]); otherwise all 7 FM + DM <= 64 rows sit in b[], pivot rows first (small instantiation: 48 rows), and a[] is unused.
// a reflector's scalars from its pivot entry alpha and dk = |(alpha, x)|^2: beta = R(t, t), vk = alpha - beta = v's pivot
// entry, inv = g = 1 / (beta (beta - alpha))
struct Reflector {
    double beta, vk, inv;
};
__device__ __forceinline__ Reflector reflector(double alpha, double dk) {
    const double sq = dk * fast_rsqrt(dk);
    Reflector h;
    h.beta = -copysign(sq, alpha);
    h.inv = fast_rcp(fma(fabs(alpha), sq, dk));
    h.vk = alpha - h.beta;
    return h;
}
// `h`: this step's reflector on entry, the NEXT step's on exit -- its inputs (the next pivot column's norm, which rides in
// the second column group, and the next pivot entry, updated in the first) are there long before this step ends, and
// the ~25 dependent operations of its scalar chain overlap the remaining column groups instead of opening the next step.
template <int NC, bool TWO, int S>
__device__ __forceinline__ void rows_step(double (&a)[NC], double (&b)[NC], double &ca, double &cb, Reflector &h, int t, int live,
                                          int lane) {
    const bool live_b = lane > t, piv = lane == t;
    const double xa = TWO ? a[S] : 0.0;
    const double inv = h.inv;
    const double xb = piv ? h.vk : (live_b ? b[S] : 0.0);
    b[S] = piv ? h.beta : b[S];                                      // R(t, t); the rest of row t of R takes shape in the pivot lane's b[]
    constexpr int NG = (NC - S) / 4 + 1;
#pragma unroll
    for (int m = 0; m < NG; ++m) {
        if (m > 1 && slot_reg(S, m, 0) >= live) continue;            // (wave-uniform) nothing but zeros in this group
        double p[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = slot_reg(S, m, q);
            if (r == -1) p[q] = TWO ? fma(xb, cb, xa * ca) : xb * cb;
            else if (r == -2) {
                const double bn = live_b ? b[S + 1 < NC ? S + 1 : NC - 1] : 0.0;
                p[q] = TWO ? fma(a[S + 1 < NC ? S + 1 : NC - 1], a[S + 1 < NC ? S + 1 : NC - 1], bn * bn) : bn * bn;
            } else if (r < NC) p[q] = TWO ? fma(xb, b[r], xa * a[r]) : xb * b[r];
            else p[q] = 0.0;
        }
        const double tot = reduce4(p[0], p[1], p[2], p[3]), wu = tot * inv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = slot_reg(S, m, q);
            if (r >= NC) continue;
            if (r == -2) {
                h = reflector(rl64(b[S + 1 < NC ? S + 1 : NC - 1], t + 1), rl64(tot, kReduce4Lane[q]));
                continue;
            }
            const double wj = rl64(wu, kReduce4Lane[q]);
            if (r == -1) {
                if (TWO) ca = fma(-xa, wj, ca);
                cb = fma(-xb, wj, cb);
            } else {
                if (TWO) a[r] = fma(-xa, wj, a[r]);
                b[r] = fma(-xb, wj, b[r]);
            }
        }
    }
}
// three steps (one dense cell's columns), then the columns move down by three
template <int NC, bool TWO>
__device__ __forceinline__ void rows_block(double (&a)[NC], double (&b)[NC], double &ca, double &cb, Reflector &h, int k, int nc,
                                           int lane, double *Rm, int RP) {
    const int live = nc - 3 * k;
    rows_step<NC, TWO, 0>(a, b, ca, cb, h, 3 * k + 0, live, lane);
    rows_step<NC, TWO, 1>(a, b, ca, cb, h, 3 * k + 1, live, lane);
    rows_step<NC, TWO, 2>(a, b, ca, cb, h, 3 * k + 2, live, lane);
    // A retired pivot row is never touched again (its entry of every later reflector is zero), so row t of R simply
    // stays in lane t's b[] -- until the columns move down.  The three that are about to leave, R(0 .. 3k+2, 3k .. 3k+2):
    if (lane < 3 * k + 3) {
        double *Rt = Rm + lane * RP + 3 * k;
        Rt[0] = b[0]; Rt[1] = b[1]; Rt[2] = b[2];
    }
#pragma unroll
    for (int i0 = 0; i0 < NC; i0 += 3) {
        if (i0 >= live) continue;                                    // (zeros would move onto zeros)
#pragma unroll
        for (int i = i0; i < i0 + 3; ++i) {
            if (TWO) a[i] = i + 3 < NC ? a[i + 3] : 0.0;
            b[i] = i + 3 < NC ? b[i + 3] : 0.0;
        }
    }
}

template <int FM, int DM, bool ROWS_IN_LANES, bool GENERAL, bool STRIPS = false>
// (3 wavefronts per SIMD pay for the small instantiation -- wedge60 2.95 -> 2.72 ms, 160 B of spills; the large one spills
//  inside its steps at 168 registers: 1.80 -> 2.32 ms)
__global__ __launch_bounds__(256, (FM <= kMfwSmallFronts ? 3 : 2)) void nin_gls_mfw_kernel(GridView g, const int32_t *__restrict__ nodes,
                                                              const uint32_t *__restrict__ desc, int32_t count,
                                                              int add_neumann, double *__restrict__ out,
                                                              double *__restrict__ nws, int32_t *__restrict__ queue) {
    using Dm = MfwDims<FM, DM, ROWS_IN_LANES, GENERAL>;
    constexpr int NP = Dm::NP, NREG = Dm::NREG, RP = Dm::RP;
    __shared__ double lds_all[4][Dm::LDS_PER_WAVE];
    const int lane = threadIdx.x & 63;
    double *const Rm = lds_all[threadIdx.x >> 6];
    double *const yb = Rm + Dm::LDS_Y, *const dbuf = Rm + Dm::LDS_D, *const wbuf = Rm + Dm::LDS_W;
    Rm[Dm::LDS_Z + lane] = 0.0;
    if (lane == 0) Rm[Dm::LDS_Z + 64] = 1.0;
    const int sc = (lane * 43) >> 7, tc = lane - 3 * sc;          // this lane's dense column = component tc of dense slot sc

    auto ticket = [&]() -> int32_t {
        int32_t v = 0;
        if (lane == 0) v = atomicAdd(queue, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
#ifdef NIN_MFW_STAMPS
    unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int n_done = 0;
#endif
    for (int32_t idx = ticket(); idx < count; idx = ticket()) {
#ifdef NIN_MFW_STAMPS
        const bool stamping = blockIdx.x == 0 && threadIdx.x < 64 && n_done == 4;
#endif
        NIN_MFW_STAMP(0);
        const int32_t p = __builtin_amdgcn_readfirstlane(nodes ? nodes[idx] : idx);
        const uint32_t *dw = desc + (size_t)kMfwDescWords * idx;
        const uint32_t fd = (uint32_t)__builtin_amdgcn_readfirstlane((int)dw[24]);
        const int F = fd & 255, D = (fd >> 8) & 255, nfree = GENERAL ? (fd >> 16) & 255 : 0;
        // phase 1 works with FOUR lanes per front: lane 4 f + j applies the front's reflectors to c (j = 0) or to the
        // columns of the front's neighbour j - 1; dense cell d's centroid is fetched by lane d
        const int fq = (lane >> 2) < FM ? (lane >> 2) : 0, jq = lane & 3;
        const uint32_t w0 = dw[fq], w1 = dw[12 + fq];
        const int fl = lane < FM ? lane : 0;
        const uint32_t w0l = dw[fl], w1l = dw[12 + fl];   // word f in lane f: what the gathers shuffle
        const uint32_t eb = (uint32_t)__builtin_amdgcn_readfirstlane(g.esup_ptr[p]);
        const uint32_t fb = (uint32_t)__builtin_amdgcn_readfirstlane(g.fsup_ptr[p]);
        const bool is_neu = (__builtin_amdgcn_readfirstlane((int)g.flags[p]) & 2) != 0;
        const double xv0 = g.coords[3 * (size_t)p + 0], xv1 = g.coords[3 * (size_t)p + 1], xv2 = g.coords[3 * (size_t)p + 2];
        uint32_t po = (w0l >> 21) & 31;
        if (GENERAL && lane >= 12) po = (dw[25] >> (5 * ((lane < DM ? lane : 12) - 12))) & 31;   // dense cells 12 .. 14
        const uint32_t pe = w0 & 31;
        const uint32_t frec[3] = {(w0 >> 5) & 0xFFFFu, w1 & 0xFFFFu, w1 >> 16};

        NIN_MFW_STAMP(1);   // node, descriptor, CSR row starts read
        // ---- phase 1: the front of cell E_f in lane f (rows 0 = cell row, 1 + 3 i + r = row r of face i) -------------
        // u: this lane's block of u = z^T R_ed (3 columns), or s = z . b_e in u[0] of the c lane
        double u[3], de[3], dod[3];
        const int my = jq > 0 ? jq - 1 : 0;                      // this lane's face (the c lane computes face 0's neighbour side in vain)
        const uint32_t myrec = jq <= 1 ? frec[0] : jq == 2 ? frec[1] : frec[2];
        {
            const uint32_t ce = (uint32_t)g.esup[eb + pe], co = (uint32_t)g.esup[eb + po];
            const uint32_t cm = (uint32_t)g.esup[eb + ((myrec >> 11) & 31)];
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
                const uint32_t f = (uint32_t)g.fsup[fb + (frec[i] & 63)];
                const uint32_t cn = (uint32_t)g.esup[eb + ((frec[i] >> 11) & 31)];
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
                const double sg = ((frec[i] >> 10) & 1) ? -1.0 : 1.0;
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
            if constexpr (STRIPS) {
                if ((lane >> 2) < F) {
                    // rows 7 f .. 7 f + 6 of the dense problem: the c lane writes c and the row's three zeros, the others their cell's entries
                    double *stg = Rm + 7 * Dm::XROW * (lane >> 2) + (jq == 0 ? 0 : 3 + 3 * my);
#pragma unroll
                    for (int r = 0; r < 7; ++r) {
                        stg[r * Dm::XROW + 0] = jq == 0 ? 0.0 : B[3 + r][0];
                        stg[r * Dm::XROW + 1] = jq == 0 ? 0.0 : B[3 + r][1];
                        stg[r * Dm::XROW + 2] = jq == 0 ? 0.0 : B[3 + r][2];
                    }
                    if (jq == 0) {
#pragma unroll
                        for (int r = 0; r < 7; ++r) stg[r * Dm::XROW + 12] = B[3 + r][0];
                    }
                }
            } else if ((lane >> 2) < F) {
                double *stg = Rm + (lane >> 2) * STAGE_F + (jq == 0 ? 9 : 3 * my);
#pragma unroll
                for (int r = 0; r < 7; ++r) stg[r * 10] = B[3 + r][0];
                if (jq > 0) {
#pragma unroll
                    for (int r = 0; r < 7; ++r) { stg[r * 10 + 1] = B[3 + r][1]; stg[r * 10 + 2] = B[3 + r][2]; }
                }
            }
        }
        if (GENERAL && lane >= 4 * FM && lane - 4 * FM < nfree) {
            // a free face (both its cells dense): its three rows [-B_a | +B_b] (gls.pyx:293-356) go straight into the dense
            // problem; lane 4 FM + q stages the rows of free face q
            const int q = lane - 4 * FM;
            const uint32_t fw = dw[26 + q];
            const uint32_t f = (uint32_t)g.fsup[fb + (fw & 63)];
            const uint32_t ca_ = (uint32_t)g.esup[eb + ((fw >> 14) & 31)], cb_ = (uint32_t)g.esup[eb + ((fw >> 19) & 31)];
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
            double *fs = Rm + Dm::LDS_FS + 18 * q;
            const double Tv[3] = {T0, T1, T2}, Uv[3] = {tj * U0, tj * U1, tj * U2};
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                fs[0 + t] = -(Ka[t * 3 + 0] * N0 + Ka[t * 3 + 1] * N1 + Ka[t * 3 + 2] * N2);
                fs[3 + t] = Kb[t * 3 + 0] * N0 + Kb[t * 3 + 1] * N1 + Kb[t * 3 + 2] * N2;
                fs[6 + t] = -Tv[t];
                fs[9 + t] = Tv[t];
                fs[12 + t] = -Uv[t];
                fs[15 + t] = Uv[t];
            }
        }
        // the dense cells' rows, (x_K - x_v) on the cell's own columns: column 3 d + t <- lane d's component t
        if (lane < D) { dbuf[3 * lane + 0] = dod[0]; dbuf[3 * lane + 1] = dod[1]; dbuf[3 * lane + 2] = dod[2]; }
        if constexpr (STRIPS) {
            if (lane < D) {   // ... and, strip form, as row 7 FM + d of the staged problem (c = 1)
                double *cd = Rm + Dm::XROW * (Dm::DROW0 + lane);
                cd[0] = 0.0; cd[1] = 0.0; cd[2] = 0.0;
                cd[3] = dod[0]; cd[4] = dod[1]; cd[5] = dod[2];
                cd[12] = 1.0;
            }
        }
        NIN_MFW_STAMP(2);   // phase 1 done
        wave_lds_sync();

        const int nc = 3 * D;                                  // the columns 0 .. nc - 1 are pivoted, column nc is c
        double rr;
        if constexpr (STRIPS) {
            // ---- the dense problem in 16 x 4 tiles, panels of four reflectors through the matrix unit (strip_factor) ---------
            static_assert(ROWS_IN_LANES && !GENERAL, "the strip form serves the two-coloured kinds");
            constexpr int NQ = (Dm::TOTAL + 15) / 16, NCB = (NP + 1 + 3) / 4;
            // (the lane number behind an opaque move, fresh per node: what the dense phase derives from it -- row / quad / column indices, the
            //  identity strip, the panel steps' masks -- is otherwise hoisted out of the node loop, lives across phase 1 and comes back from
            //  scratch word by word inside the panels; kernels_gls_mfx.hip: -9 % on a Delaunay mesh)
            int ln;
            asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));
            const int si = ln >> 4, sb = (ln >> 2) & 3, sj = ln & 3;
            double C[NQ][NCB];
            // per tile row: where this lane's row lives in the staging area.  code(sd) = 1 + the index of dense slot sd among
            // the front's three neighbours (0: not a neighbour), two bits per slot, made once per front (lane f) and shuffled
            uint32_t tbl = 0;
            {
                const int s0 = (w0l >> 11) & 15, s1 = (w1l >> 6) & 15, s2 = (w1l >> 22) & 15;
                tbl = (1u << (2 * s0)) | (2u << (2 * s1)) | (3u << (2 * s2));
                tbl = lane < F ? tbl : 0u;
            }
            // per tile row q, this lane's row of the dense problem: rtbl = the 2-bit codes of its cells' column slots, rbz = the byte address
            // of the staged row (a row the node does not have: row 0, with no codes -- its three leading zeros), rbc = of its c entry
            uint32_t rtbl[NQ], rbz[NQ], rbc[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int row = 16 * q + 4 * sb + si;
                const bool fill = row < Dm::DROW0;
                const int f = fill ? (row * 37) >> 8 : 0, d = row - Dm::DROW0;
                const uint32_t ft = (uint32_t)__shfl((int)tbl, f);
                const bool ok = fill ? f < F : d < D;
                rtbl[q] = ok ? (fill ? ft : 1u << (2 * d)) : 0u;
                rbz[q] = ok ? 8u * Dm::XROW * (uint32_t)row : 0u;
                rbc[q] = ok ? 8u * (Dm::XROW * (uint32_t)row + 12u) : 8u * (uint32_t)Dm::LDS_Z;
            }
            const char *const Rb = reinterpret_cast<const char *>(Rm);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                // (sjq: sj behind an opaque move, fresh per column block -- otherwise the column arithmetic of all ten blocks is
                //  hoisted, ~60 lane constants that end up in scratch and come back one by one, each behind a full wait)
                int sjq;
                asm volatile("v_mov_b32 %0, %1" : "=v"(sjq) : "v"(sj));
                const int col = 4 * cb + sjq;
                const int sd = (col * 43) >> 7;                                                 // column = component tt of dense slot sd
                const uint32_t tt8 = 8u * (uint32_t)(col - 3 * sd);
                const int sh = col < nc ? 2 * sd : 30;                                          // (slot 15: nobody's neighbour)
                const uint32_t cmask = col == nc ? 0xFFFFFFFFu : 0u;                            // this lane's column of this block is c
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const uint32_t code = (rtbl[q] >> sh) & 3u;
                    const uint32_t a = rbz[q] + 24u * code + tt8;
                    C[q][cb] = *reinterpret_cast<const double *>(Rb + ((rbc[q] & cmask) | (a & ~cmask)));
                }
            }
            wave_lds_sync();          // the staging area is R's from here on
            NIN_MFW_STAMP(3);   // rows gathered
            SubStamps ST;
#ifdef NIN_MFW_STAMPS
            ST.on = stamping; ST.last = __builtin_amdgcn_s_memtime();
            for (int j = 0; j < 6; ++j) ST.acc[j] = 0;
#endif
            rr = strip_factor<NQ, NCB>(C, nc, ln, Rm, RP, ST);
#ifdef NIN_MFW_STAMPS
            if (stamping && lane == 0) for (int j = 0; j < 5; ++j) nws[nodes[8 + j]] = (double)ST.acc[j];
#endif
        } else if constexpr (ROWS_IN_LANES) {
            // ---- the dense problem, lane = ROW.  TWO arrays (large): lane r holds row NP + r (r < NREG) in a[] and pivot row
            //      r (r < NP) in b[]; one array (small): lane r holds row r in b[], the pivot rows first --------------------------
            constexpr int TOTAL = Dm::TOTAL;
            int ln;                                                  // (opaque, fresh per node: see the strip form above)
            asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));
            constexpr bool TWO = TOTAL > 64;
            static_assert(TOTAL <= 128 && NP <= 64, "one row per ln and array");
            double a[NP], b[NP], ca = 0.0, cb;
            {
                auto gather = [&](int row, bool have, double (&x)[NP], double &xc) {
                    const int free0 = Dm::DROW0 + D;                          // the free faces' rows follow the D dense cells' rows
                    const bool fill = row < Dm::DROW0, free_row = GENERAL && row >= free0;
                    const int f = fill ? (row * 37) >> 8 : 0, i = row - 7 * f, d = row - Dm::DROW0;
                    const uint32_t q0 = (uint32_t)__shfl((int)w0l, f), q1 = (uint32_t)__shfl((int)w1l, f);
                    int s0 = (q0 >> 11) & 15, s1 = (q1 >> 6) & 15, s2 = (q1 >> 22) & 15;   // dense slots of the front's 3 neighbours
                    bool ok = have && (fill ? f < F : d < D);
                    int fbase = f * STAGE_F + i * 10;
                    if (GENERAL) {
                        // a free face's row: its two cells' blocks (side a, side b) lie 3 apart in the staging area
                        const int x = free_row ? row - free0 : 0, q = (x * 43) >> 7, rr = x - 3 * q;
                        const uint32_t fw = dw[26 + (q < kMfwMaxFree ? q : 0)];
                        if (free_row) {
                            s0 = (fw >> 6) & 15; s1 = (fw >> 10) & 15; s2 = -1;
                            fbase = Dm::LDS_FS + 18 * q + 6 * rr;
                            ok = have && q < nfree;
                        }
                    }
                    const bool via_slots = fill || free_row;
#pragma unroll
                    for (int sj = 0; sj < DM; ++sj) {
                        int off = Dm::LDS_Z;                                  // zeros
                        if (via_slots) {
                            off = s0 == sj ? fbase : off;
                            off = s1 == sj ? fbase + 3 : off;
                            off = s2 == sj ? fbase + 6 : off;
                        } else {
                            off = d == sj ? Dm::LDS_D + 3 * sj : off;        // the cell row of dense cell d: (x_K - x_v) on its own columns
                        }
                        off = ok ? off : Dm::LDS_Z;
#pragma unroll
                        for (int tt = 0; tt < 3; ++tt) x[3 * sj + tt] = Rm[off + tt];
                    }
                    xc = ok ? (fill ? Rm[fbase + 9] : free_row ? 0.0 : 1.0) : 0.0;
                };
                if constexpr (TWO) {
                    gather(NP + ln, NP + ln < TOTAL, a, ca);
                    const int rb = ln < NP ? ln : ln + 64;         // b[]: the pivot rows, then the rows a[] has no lane for
                    gather(rb < TOTAL ? rb : 0, rb < TOTAL, b, cb);
                } else {
#pragma unroll
                    for (int i = 0; i < NP; ++i) a[i] = 0.0;
                    gather(ln < TOTAL ? ln : 0, ln < TOTAL, b, cb);
                }
            }
            wave_lds_sync();          // the staging area is R's from here on
            NIN_MFW_STAMP(3);   // rows gathered
            // the first reflector: |column 0|^2 by a wave reduction (the later ones come out of the steps)
            Reflector h = reflector(rl64(b[0], 0), wave_allsum(TWO ? fma(a[0], a[0], b[0] * b[0]) : b[0] * b[0]));
            for (int k = 0; k < D; ++k) rows_block<NP, TWO>(a, b, ca, cb, h, k, nc, ln, Rm, RP);
            if (ln < nc) Rm[ln * RP + nc] = cb;                  // (Q^T c)(0:nc), the last column of R
            const double cbl = ln >= nc ? cb : 0.0;                // rows that never were pivot rows (one array; or D < DM) count too
            rr = wave_allsum(TWO ? fma(ca, ca, cbl * cbl) : cbl * cbl);   // r . r = |(Q^T c)(nc:)|^2
        } else {
        // ---- the dense problem, lane = column: rows 0 .. NP-1 (fill rows of the first fronts) go to LDS, the others
        //      stay in the register pairs a[] ---------------------------------------------------------------------------------
        const bool is_rhs = lane == nc;
        double a[NREG];
        {
            double top[NP];
            const double dodv = dbuf[lane < 3 * DM ? lane : 0];
#pragma unroll
            for (int row = 0; row < NP + NREG; ++row) {
                double v = 0.0;
                if (row < 7 * FM) {
                    const int f = row / 7, r = row % 7;
                    // (the slot computation below is the same for the 7 rows of a front: the compiler keeps one copy)
                    int j = -1;
                    if (f < F) {
                        const uint32_t q0 = rl32(w0l, f), q1 = rl32(w1l, f);
                        const int s0 = (q0 >> 11) & 15, s1 = (q1 >> 6) & 15, s2 = (q1 >> 22) & 15;   // dense slots of the front's 3 neighbours
                        j = sc == s0 ? tc : j;
                        j = sc == s1 ? 3 + tc : j;
                        j = sc == s2 ? 6 + tc : j;
                        j = lane < nc ? j : -1;
                        j = is_rhs ? 9 : j;
                    }
                    const double *src = j >= 0 ? Rm + f * STAGE_F + j : Rm + Dm::LDS_Z;
                    v = src[r * 10];
                } else {
                    const int d = row - 7 * FM;                // the cell row of dense cell d: (x_K - x_v) on its own columns, c = 1
                    v = (sc == d && lane < nc) ? dodv : 0.0;
                    v = is_rhs ? 1.0 : v;
                    v = d < D ? v : 0.0;
                }
                if (row < NP) top[row] = v; else a[row - NP] = v;
            }
            wave_lds_sync();          // the staging area is R's from here on
            if (lane < RP) {
#pragma unroll
                for (int r = 0; r < NP; ++r) Rm[r * RP + lane] = top[r];
#pragma unroll
                for (int r = NP; r < NP + Dm::PAD; ++r) Rm[r * RP + lane] = 0.0;
            }
        }
        wave_lds_sync();

        double dd = 0.0;           // dots of the pivot column with every column (lane = column)
        if (lane < RP) {           // (the other lanes sit the factorisation out)
            double *const Rl = Rm + lane;
            {
                double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                for (int r = 0; r < NP; r += 2) {
                    const double v0 = Rl[r * RP], v1 = Rl[(r + 1) * RP];
                    acc0 = fma(rl64(v0, 0), v0, acc0);
                    acc1 = fma(rl64(v1, 0), v1, acc1);
                }
#pragma unroll
                for (int r = 0; r + 1 < NREG; r += 2) {
                    acc0 = fma(rl64(a[r], 0), a[r], acc0);
                    acc1 = fma(rl64(a[r + 1], 0), a[r + 1], acc1);
                }
                if (NREG & 1) acc0 = fma(rl64(a[NREG - 1], 0), a[NREG - 1], acc0);
                dd = acc0 + acc1;
            }
            for (int t = 0; t < nc; ++t) {
                const double rowk = Rl[t * RP];
                // the rows still waiting in LDS, t + 1 .. NP - 1, in groups of G (the zero rows behind them fill the last
                // group); the first group's loads overlap the scalar chain below
                constexpr int G = 4;
                double *lp = Rl + (t + 1) * RP;
                const int groups = (NP - 1 - t + G - 1) / G;
                double cur[G];
#pragma unroll
                for (int q = 0; q < G; ++q) cur[q] = lp[q * RP];
                const double dk = rl64(dd, t), alpha = rl64(rowk, t);
                const double sq = dk * fast_rsqrt(dk);                   // |(alpha, x)|
                const double beta = -copysign(sq, alpha);
                const double inv = fast_rcp(fma(fabs(alpha), sq, dk));   // 1 / (beta (beta - alpha))
                const double vk = alpha - beta;
                const double w = lane > t ? (dd - beta * rowk) * inv : 0.0;
                double rrow = fma(-vk, w, rowk);                          // row t of R, in place of the pivot row
                rrow = lane == t ? beta : rrow;
                rrow = lane < t ? 0.0 : rrow;
                Rl[t * RP] = rrow;
                double acc0 = 0.0, acc1 = 0.0;
                for (int gi = 0; gi < groups; ++gi) {
                    double nx[G], x[G], xn[G];
#pragma unroll
                    for (int q = 0; q < G; ++q) nx[q] = lp[(G + q) * RP];
#pragma unroll
                    for (int q = 0; q < G; ++q) x[q] = rl64(cur[q], t);
#pragma unroll
                    for (int q = 0; q < G; ++q) cur[q] = fma(-x[q], w, cur[q]);
#pragma unroll
                    for (int q = 0; q < G; ++q) lp[q * RP] = cur[q];
#pragma unroll
                    for (int q = 0; q < G; ++q) xn[q] = rl64(cur[q], t + 1);
#pragma unroll
                    for (int q = 0; q < G; q += 2) {
                        acc0 = fma(xn[q], cur[q], acc0);
                        acc1 = fma(xn[q + 1], cur[q + 1], acc1);
                    }
#pragma unroll
                    for (int q = 0; q < G; ++q) cur[q] = nx[q];
                    lp += G * RP;
                }
                constexpr int B = 6;
                static_assert(NREG % B == 0, "register rows come in blocks");
#pragma unroll
                for (int b = 0; b < NREG / B; ++b) {
                    double x[B], xn[B];
#pragma unroll
                    for (int q = 0; q < B; ++q) x[q] = rl64(a[B * b + q], t);
#pragma unroll
                    for (int q = 0; q < B; ++q) a[B * b + q] = fma(-x[q], w, a[B * b + q]);
#pragma unroll
                    for (int q = 0; q < B; ++q) xn[q] = rl64(a[B * b + q], t + 1);
#pragma unroll
                    for (int q = 0; q < B; q += 2) {
                        acc0 = fma(xn[q], a[B * b + q], acc0);
                        acc1 = fma(xn[q + 1], a[B * b + q + 1], acc1);
                    }
                }
                dd = acc0 + acc1;
            }
        }
        rr = rl64(dd, nc);                                            // r . r = |(Q^T c)(nc:)|^2
        }
        wave_lds_sync();

        NIN_MFW_STAMP(4);   // dense problem factored
        // ---- R y = (Q^T c)(0:nc) by columns: lane = row ------------------------------------------------------------------
        {
            const int li = lane < nc ? lane : 0;
            double ct = lane < nc ? Rm[li * RP + nc] : 0.0;
            const double ri = fast_rcp(Rm[li * RP + li]);
            // four columns per round, the next round's four loaded before this round's dependent chain (y_k: one product, one
            // broadcast, one FMA per column) starts: the loop was one LDS round trip per column (36 of them: 10 k cycles a node)
            const double *Rl = Rm + li * RP;
            int kb = (nc - 1) & ~3;
            double c4[4], n4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) c4[j] = (STRIPS || (lane < nc && kb + j < nc)) ? Rl[kb + j] : 0.0;   // (strip form: unguarded -- a column right of nc is a word of the row nobody uses, a lane beyond nc reads row 0 in vain; an exec region per load otherwise)
            for (; kb >= 0; kb -= 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) n4[j] = STRIPS ? Rl[(kb >= 4 ? kb - 4 : 0) + j] : (lane < nc && kb >= 4) ? Rl[kb - 4 + j] : 0.0;
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
        NIN_MFW_STAMP(5);   // back-substitution done
        // ---- residuals on the cell rows, weights ---------------------------------------------------------------------------
        {
            // r_e = 1 - d_e . y_e = 1 - z . b_e + u . y_dense: the c lane brings 1 - s, the three others their block of u . y
            const int sl = (myrec >> 6) & 15;
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
            if ((lane >> 2) < F && jq == 0) wbuf[pe] = we;
            if (lane < D) wbuf[po] = wo;
        }
        wave_lds_sync();
        {
            const int ne = F + D;
            // gls.pyx:470-472 (only if an interior node carries the Neumann flag): neumann_ws = the last cell's weight
            const double nwv = is_neu ? wbuf[ne - 1] : 0.0;
            const double addv = add_neumann ? nwv : 0.0;
            if (lane < ne) out[eb + lane] = wbuf[lane] + addv;
#ifndef NIN_MFW_STAMPS
            if (lane == 0) nws[p] = nwv;
#endif
        }
        wave_lds_sync();
        NIN_MFW_STAMP(6);   // weights stored
#ifdef NIN_MFW_STAMPS
        if (stamping && lane == 0)
            for (int i = 0; i < 7; ++i) nws[nodes[i]] = (double)(stamps[i] - stamps[0]);   // (this build leaves neumann_ws to the stamps)
        ++n_done;
#endif
    }
}

// ---- small nodes: the WHOLE system in one wavefront's registers, lane = row ---------------------------------------------
// What neither the cube-node kernel nor the fronts above take and is small -- in practice the BOUNDARY nodes that carry a
// Neumann flag: 4 cells, 4 internal + 4 boundary faces on a face of a hexahedron box (20 x 13), 2 cells on its edges, half a
// truncated octahedron on a tetrahedron boundary -- used to go to the block kernel: a workgroup, an LDS image of the system,
// a plan built with LDS atomics, for 20 rows -- 68 k SIMD cycles a node, more than a 96 x 37 Kuhn node takes here.  This kernel
// needs no plan: up to DM cells (3 DM + 1 columns) and up to 64 rows, lane r IS row r -- the cell rows, then three rows per
// internal face (in fsup order), then the Neumann rows (gls.pyx:269-281, 293-356, 394-416) -- built in the lane from a per-face
// record staged in LDS, the columns static register indices, and the row-lane Householder of the dense phase above (rows_step:
// one reduction per column, four columns at a time) does the rest.  Same zero-row rule as every GLS kernel (no internal face,
// fewer rows than unknowns, a zero pivot column through the NaN test).  A wave looks at 64 list entries at a time and gives
// Dirichlet boundary nodes (gls.pyx:165-166) their zero row right there; the others it computes one after the other.
constexpr int kSmallMaxFaces = 48;
template <int DM>
struct SmallDims {
    static constexpr int NP = 3 * DM, RP = NP + 1;
    static constexpr int LDS_Y = NP * RP, LDS_W = LDS_Y + NP + 4, LDS_F = LDS_W + ((DM + 1) & ~1),   // R rows | y | weights | face records
                         FREC = 14,                                                                   // per face: 12 values + (Ia, Ib) in one more + pad
                         LDS_PER_WAVE = LDS_F + FREC * kSmallMaxFaces;
};

template <int DM>
__global__ __launch_bounds__(256, (DM <= 4 ? 4 : DM <= 8 ? 3 : 2)) void nin_gls_small_kernel(GridView g, const int32_t *__restrict__ nodes,
                                                                                             int32_t count, int add_neumann,
                                                                                             double *__restrict__ out,
                                                                                             double *__restrict__ nws) {
    using Dm = SmallDims<DM>;
    constexpr int NP = Dm::NP, RP = Dm::RP;
    __shared__ double lds_all[4][Dm::LDS_PER_WAVE];
    const int lane = threadIdx.x & 63;
    double *const Rm = lds_all[threadIdx.x >> 6];
    double *const yb = Rm + Dm::LDS_Y, *const wbuf = Rm + Dm::LDS_W, *const frec = Rm + Dm::LDS_F;
    // a wave's 64 entries per round lie n_waves apart: boundary nodes that are computed come in runs (a Neumann plane is one
    // run of the list), and a run must spread over all waves instead of filling a few of them
    const int64_t n_waves = (int64_t)gridDim.x * 4, wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int64_t tile = wave; tile < count; tile += n_waves * 64) {
        unsigned long long todo;
        {
            const int64_t i = tile + lane * n_waves;
            const bool in = i < count;
            const int32_t pl = in ? (nodes ? nodes[i] : (int32_t)i) : 0;
            const int fll = in ? (int)g.flags[pl] : 0;
            const bool dirichlet = in && (fll & 1) && !(fll & 2);
            if (dirichlet) {
                const int32_t b0 = g.esup_ptr[pl], b1 = g.esup_ptr[pl + 1];
                for (int32_t j = b0; j < b1; ++j) out[j] = 0.0;
                nws[pl] = 0.0;
            }
            todo = __ballot(in && !dirichlet);
        }
        while (todo) {
            const int bit = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int64_t idx = tile + bit * n_waves;
            const int32_t p = __builtin_amdgcn_readfirstlane(nodes ? nodes[idx] : (int32_t)idx);
            const int32_t eb = __builtin_amdgcn_readfirstlane(g.esup_ptr[p]), ne = __builtin_amdgcn_readfirstlane(g.esup_ptr[p + 1]) - eb;
            const int32_t fb = __builtin_amdgcn_readfirstlane(g.fsup_ptr[p]), nf = __builtin_amdgcn_readfirstlane(g.fsup_ptr[p + 1]) - fb;
            const bool is_neu = (__builtin_amdgcn_readfirstlane((int)g.flags[p]) & 2) != 0;
            const double xv0 = g.coords[3 * (size_t)p + 0], xv1 = g.coords[3 * (size_t)p + 1], xv2 = g.coords[3 * (size_t)p + 2];
            // lane i < ne: cell i; lane f < nf: face f
            const int32_t mycell = lane < ne ? g.esup[eb + lane] : -1;
            const bool face_lane = lane < nf;
            const size_t f = face_lane ? (size_t)g.fsup[fb + lane] : 0;
            const int32_t ca = face_lane ? g.face_cells[2 * f] : -1, cb = face_lane ? g.face_cells[2 * f + 1] : -1;
            const bool internal = face_lane && cb >= 0;
            const unsigned long long m_int = __ballot(internal), m_bnd = __ballot(face_lane && !internal);
            const int n_if = __popcll(m_int), n_bf = __popcll(m_bnd);
            const int nc = 3 * ne, m = ne + 3 * n_if + (is_neu ? n_bf : 0);
            if (n_if == 0 || m < nc) {                            // (wave-uniform) outside the parity set: the zero row
                if (lane < ne) out[eb + lane] = 0.0;
                if (lane == 0) nws[p] = 0.0;
                continue;
            }
            // ---- a record per face: (Ia, Ib), -K_a N, +K_b N, T, tau U (internal) or -K_a N (boundary, Neumann nodes) ---------
            int Ia = 0, Ib = 0;                                  // positions of the face's two cells in the esup row
            for (int q = 0; q < ne; ++q) {                        // (every lane takes part: a shuffle reads active lanes only)
                const int32_t cq = __shfl(mycell, q);
                Ia = cq == ca ? q : Ia;
                Ib = cq == cb ? q : Ib;
            }
            if (face_lane) {
                const unsigned long long below = (1ull << lane) - 1ull;
                const int rec = internal ? __popcll(m_int & below) : n_if + __popcll(m_bnd & below);
                double *fr = frec + Dm::FREC * rec;
                const double N0 = (double)g.face_normal[3 * f + 0], N1 = (double)g.face_normal[3 * f + 1], N2 = (double)g.face_normal[3 * f + 2];
                const double *Ka = g.perm + 9 * (size_t)ca;
                fr[0] = -(Ka[0] * N0 + Ka[1] * N1 + Ka[2] * N2);
                fr[1] = -(Ka[3] * N0 + Ka[4] * N1 + Ka[5] * N2);
                fr[2] = -(Ka[6] * N0 + Ka[7] * N1 + Ka[8] * N2);
                if (internal) {
                    const double *Kb = g.perm + 9 * (size_t)cb;
                    fr[3] = Kb[0] * N0 + Kb[1] * N1 + Kb[2] * N2;
                    fr[4] = Kb[3] * N0 + Kb[4] * N1 + Kb[5] * N2;
                    fr[5] = Kb[6] * N0 + Kb[7] * N1 + Kb[8] * N2;
                    const double T0 = xv0 - g.face_center[3 * f + 0], T1 = xv1 - g.face_center[3 * f + 1], T2 = xv2 - g.face_center[3 * f + 2];
                    const double U0 = N1 * T2 - N2 * T1, U1 = N2 * T0 - N0 * T2, U2 = N0 * T1 - N1 * T0;
                    const double da = g.diff_mag[ca], db = g.diff_mag[cb];
                    double eta = 0.0;
                    eta = da > eta ? da : eta;
                    eta = db > eta ? db : eta;
                    const double tj = face_tau(sqrt(U0 * U0 + U1 * U1 + U2 * U2), eta);
                    fr[6] = T0; fr[7] = T1; fr[8] = T2;
                    fr[9] = tj * U0; fr[10] = tj * U1; fr[11] = tj * U2;
                }
                reinterpret_cast<int32_t *>(fr + 12)[0] = Ia;
                reinterpret_cast<int32_t *>(fr + 12)[1] = internal ? Ib : -1;
            }
            wave_lds_sync();
            // ---- lane r builds row r: two cell slots (Ia, Ib) carry three entries each, c = 1 on the cell rows ---------------
            double b[NP], a_unused[NP], ca_unused = 0.0, cbv = 0.0, dod[3] = {0.0, 0.0, 0.0};
            {
                int Ra = -1, Rb = -1;                             // the two cell slots of THIS lane's row
                double va[3] = {0.0, 0.0, 0.0}, vb[3] = {0.0, 0.0, 0.0};
                if (lane < ne) {                                   // gls.pyx:269-281
                    const size_t c = (size_t)mycell;
                    va[0] = g.centroids[3 * c + 0] - xv0; va[1] = g.centroids[3 * c + 1] - xv1; va[2] = g.centroids[3 * c + 2] - xv2;
                    dod[0] = va[0]; dod[1] = va[1]; dod[2] = va[2];
                    Ra = lane;
                    cbv = 1.0;
                } else if (lane < ne + 3 * n_if) {                 // gls.pyx:340-356: [-B_a | +B_b], B = [K N; T; tau U]
                    const int x = lane - ne, q = (x * 43) >> 7, k = x - 3 * q;
                    const double *fr = frec + Dm::FREC * q;
                    Ra = reinterpret_cast<const int32_t *>(fr + 12)[0];
                    Rb = reinterpret_cast<const int32_t *>(fr + 12)[1];
                    const int oa = k == 0 ? 0 : k == 1 ? 6 : 9, ob = k == 0 ? 3 : k == 1 ? 6 : 9;
                    const double sa = k == 0 ? 1.0 : -1.0;         // (record: -K_a N already negated; T, tau U stored once)
#pragma unroll
                    for (int t = 0; t < 3; ++t) { va[t] = sa * fr[oa + t]; vb[t] = fr[ob + t]; }
                } else if (lane < m) {                             // gls.pyx:394-416: -(K N) of the boundary face's cell
                    const double *fr = frec + Dm::FREC * (n_if + lane - ne - 3 * n_if);
                    Ra = reinterpret_cast<const int32_t *>(fr + 12)[0];
#pragma unroll
                    for (int t = 0; t < 3; ++t) va[t] = fr[t];
                }
#pragma unroll
                for (int sl = 0; sl < DM; ++sl) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) b[3 * sl + t] = sl == Ra ? va[t] : (sl == Rb ? vb[t] : 0.0);
                }
            }
            // ---- Householder, one reduction per column (rows_step), R through LDS ----------------------------------------------
            Reflector h = reflector(rl64(b[0], 0), wave_allsum(b[0] * b[0]));
            for (int k = 0; k < ne; ++k) rows_block<NP, false>(a_unused, b, ca_unused, cbv, h, k, nc, lane, Rm, RP);
            if (lane < nc) Rm[lane * RP + nc] = cbv;
            const double cbl = lane >= nc ? cbv : 0.0;
            const double rr = wave_allsum(cbl * cbl);
            wave_lds_sync();
            {
                const int li = lane < nc ? lane : 0;
                double ct = lane < nc ? Rm[li * RP + nc] : 0.0;
                const double ri = fast_rcp(Rm[li * RP + li]);
                const double *Rl = Rm + li * RP;
                int kb = (nc - 1) & ~3;
                double c4[4], n4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) c4[j] = (lane < nc && kb + j < nc) ? Rl[kb + j] : 0.0;
                for (; kb >= 0; kb -= 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) n4[j] = (lane < nc && kb >= 4) ? Rl[kb - 4 + j] : 0.0;
#pragma unroll
                    for (int j = 3; j >= 0; --j) {
                        const int k = kb + j;
                        if (k < nc) {
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
            {
                const int ld = lane < ne ? lane : 0;
                const double ro = 1.0 - fma(dod[2], yb[3 * ld + 2], fma(dod[1], yb[3 * ld + 1], dod[0] * yb[3 * ld]));
                double w = ro * fast_rcp(rr);
                w = (rr > 0.0 && __builtin_isfinite(w)) ? w : 0.0;
                if (lane < ne) wbuf[lane] = w;
            }
            wave_lds_sync();
            {
                const double nwv = is_neu ? wbuf[ne - 1] : 0.0;          // gls.pyx:470-472: the LAST cell's weight
                const double addv = add_neumann ? nwv : 0.0;
                if (lane < ne) out[eb + lane] = wbuf[lane] + addv;
                if (lane == 0) nws[p] = nwv;
            }
            wave_lds_sync();
        }
    }
}

__global__ void k_mfw_desc(GridView g, const int32_t *__restrict__ nodes, int32_t count, uint32_t *__restrict__ desc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t w[kMfwDescWords];
    if (!mfw_descriptor(g, nodes ? nodes[i] : (int32_t)i, w)) {   // (the list holds classified nodes only)
#pragma unroll
        for (int k = 0; k < kMfwDescWords; ++k) w[k] = 0u;
    }
#pragma unroll
    for (int k = 0; k < kMfwDescWords; ++k) desc[kMfwDescWords * i + k] = w[k];
}

}  // namespace

int launch_mfw_desc(const GridView &g, const int32_t *nodes, int32_t count, uint32_t *desc, hipStream_t stream) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(k_mfw_desc, dim3((unsigned)((count + 127) / 128)), dim3(128), 0, stream, g, nodes, count, desc);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_gls_mfw(const GridView &g, const int32_t *nodes, const uint32_t *desc, int32_t count, int kind, int add_neumann,
                   double *out, double *nws, int32_t *queue, hipStream_t stream) {
    if (count <= 0) return 0;
    int64_t blocks = ((int64_t)count + 3) / 4;
    const int64_t cap = (kind == 1 ? 3 : 2) * 256;   // persistent: the 4-wave workgroups that are resident (2 per CU at <= 256 registers, 3 at <= 168)
    if (blocks > cap) blocks = cap;
    const bool lane_columns = getenv("NIN_MFW_LANE_COLUMNS") != nullptr;   // A/B switch: the first form of the dense phase
    const bool no_strips = getenv("NIN_MFW_NO_STRIPS") != nullptr;         // A/B switch: the row-lane form instead of the strip form
#define NIN_MFW_LAUNCH(FMX, DMX, RL, GEN, ST)                                                                           \
    hipLaunchKernelGGL((nin_gls_mfw_kernel<FMX, DMX, RL, GEN, ST>), dim3((unsigned)blocks), dim3(256), 0, stream, g, nodes, desc, \
                       count, add_neumann, out, nws, queue)
    if (kind == 2) NIN_MFW_LAUNCH(kMfwMaxFronts, kMfwWideDense, true, true, false);
    else if (kind == 1 && lane_columns) NIN_MFW_LAUNCH(kMfwSmallFronts, kMfwSmallDense, false, false, false);
    // (the small instantiation keeps the row-lane form: with all 48 rows in ONE array a column costs it one product and one
    //  update, and the strip form's fixed cost per reflector -- the same for 18 columns as for 36 -- eats what the matrix unit
    //  saves: wedge60 2.66 ms either way, measured; NIN_MFW_SMALL_STRIPS=1 selects the strip form there)
    else if (kind == 1 && getenv("NIN_MFW_SMALL_STRIPS") != nullptr) NIN_MFW_LAUNCH(kMfwSmallFronts, kMfwSmallDense, true, false, true);
    else if (kind == 1) NIN_MFW_LAUNCH(kMfwSmallFronts, kMfwSmallDense, true, false, false);
    else if (lane_columns) NIN_MFW_LAUNCH(kMfwMaxFronts, kMfwMaxDense, false, false, false);
    else if (no_strips) NIN_MFW_LAUNCH(kMfwMaxFronts, kMfwMaxDense, true, false, false);
    else NIN_MFW_LAUNCH(kMfwMaxFronts, kMfwMaxDense, true, false, true);
#undef NIN_MFW_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// `kind`: 0 / 1 / 2 = at most 4 / 8 / 12 cells
int launch_gls_small(const GridView &g, const int32_t *nodes, int32_t count, int kind, int add_neumann, double *out, double *nws,
                     hipStream_t stream) {
    if (count <= 0) return 0;
    int64_t blocks = ((int64_t)count + 31) / 32;                // a wave looks at up to 64 list entries per round, n_waves apart: about
                                                                // 8 per wave while the chip has room (the nodes it computes go one after the other)
    const int64_t cap = 256 * (kind == 0 ? 4 : kind == 1 ? 3 : 2) * 2;
    if (blocks > cap) blocks = cap;
    if (kind == 0) hipLaunchKernelGGL(nin_gls_small_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, g, nodes, count, add_neumann, out, nws);
    else if (kind == 1) hipLaunchKernelGGL(nin_gls_small_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, stream, g, nodes, count, add_neumann, out, nws);
    else hipLaunchKernelGGL(nin_gls_small_kernel<12>, dim3((unsigned)blocks), dim3(256), 0, stream, g, nodes, count, add_neumann, out, nws);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

const char *kernel_name_gls_mfw() { return "nin_gls_mfw_kernel"; }

}  // namespace nin
