// kernels_gls_hex8mf.hip -- GLS weights of "cube" nodes (every interior node of a hexahedron mesh), gfx950:
// FOUR lanes per node, 16 nodes per wavefront, a multifrontal Householder QR that follows the sparsity of the
// system instead of sweeping a dense 44 x 24 matrix.
//
// The system (gls.pyx:252-356).  Unknowns: a gradient (3 columns) per cell plus the node value (the column c that
// the last-row identity turns into a right-hand side, see below).  Rows: one per cell -- (x_K - x_v) on its own 3
// columns, c = 1 -- and three per internal face -- [-B_a | +B_b] on the columns of its two cells, c = 0.  So the
// cells are the vertices and the faces the edges of a graph, and around an interior hexahedron node that graph is
// the CUBE: bipartite, 4 "even" cells E0..E3 that share no face, 4 "odd" cells O0..O3 (hex8_desc.hpp fixes the
// labelling at load time; E_l is adjacent to every odd cell but O_(3-l)).
//
// Column order [E0 E1 E2 E3 | O0 O1 O2 O3].  A Householder reflector only touches the rows that are non-zero in its
// pivot column, so
//   phase 1  the 12 steps on the even cells' columns are four INDEPENDENT fronts -- the cell row of E_l and the 9
//            rows of its 3 faces, 10 x (3 own + 12 odd + c) -- one per lane, no cross-lane traffic at all.  Each
//            leaves 3 rows of R (folded at once into what the weights need of them: z = R_ee^-T d_e, u = z^T R_eo,
//            s = z . b_e, so R itself is never stored) and 7 filled rows over the 12 odd columns;
//   phase 2  what is left is 32 x 12: per lane its 7 fill rows and the cell row of O_l, exactly where phase 1 left
//            them.  Row-distributed Householder: partial dots per lane, one quad reduction (2 DPP stages) per
//            column and step, pivot rows dealt round-robin (step k: row k / 4 of lane k % 4) so the lanes retire
//            rows evenly; the scalar chain of a step (beta, 1 / (beta (beta - alpha))) overlaps the partial dots,
//            which do not need it: v = x - beta e_p differs from x in the pivot lane's pivot entry only, a local fix;
//   then     R y = Q^T c by columns over the quad, r_i = 1 - d_i . y_i per cell, r . r from the 20 rows left over,
//            weights r_i / (r . r)  (the identity X[n-1, i] = r_i / (r.r), SURVEY 7.1(i), kernels_gls_block.hip).
// ~6 k FP64 FMAs per node against ~20 k for the dense sweep, none of them masked out, and no LDS traffic inside
// the factorisation (the dense kernel published every pivot column through LDS: 22 writes per step).
// Same mathematics as dgels on the reference's matrix -- a Householder QR, only in a column order that exposes the
// zeros -- so the weights agree with the reference to rounding (tools/proto_hex8_mf.py: 5e-15 against the oracle).
//
// Eligible nodes are binned at load time (k_classify + hex8_desc.hpp): 8 cells, 12 internal faces, cube graph.
// Everything else runs in kernels_gls_block.hip.  The launch is persistent and XCD-aware (per-XCD work counters),
// as the dense 16-lanes-per-node kernel of round 1 (41 ms, removed) was.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "device_grid.hpp"
#include "gls_device_math.hpp"
#include "hex8_desc.hpp"
#include "launch.hpp"

namespace nin {

namespace {

using namespace glsmath;

__device__ __forceinline__ double quad_sum(double v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    return v;
}
template <int L>
__device__ __forceinline__ double quad_bcast(double v) { return dpp_mov<L * 0x55>(v); }   // quad_perm [L,L,L,L]

constexpr int NPW = 16;          // nodes per wavefront pass (4 lanes each)
constexpr int NR = 8, NC = 13;   // phase 2, per lane: 8 rows x (12 odd columns + c)

// Back-substitution of the odd unknowns by columns: row K of R lives in lane K % 4, local row K / 4 (entries
// C[Q][j], j > K, right-hand side t[Q]).  y_K is formed in its lane and broadcast; every lane then takes
// R(K', K) y_K off the right-hand sides of its own rows K' < K.
template <int K>
struct BackLoop {
    static __device__ __forceinline__ void run(const double (&C)[NR][NC], const double (&rinvq)[3], double (&t)[3],
                                               double (&y)[12], int l) {
        constexpr int Q = K / 4, LAM = K % 4;
        const double yk = quad_bcast<LAM>(t[Q] * rinvq[Q]);
        y[K] = yk;
#pragma unroll
        for (int q = 0; q < Q; ++q) t[q] = fma(-C[q][K], yk, t[q]);          // rows 4 q + l < K in every lane
        if (LAM > 0) t[Q] = fma((l < LAM) ? -C[Q][K] : 0.0, yk, t[Q]);       // row 4 Q + l < K only below the pivot lane
        BackLoop<K - 1>::run(C, rinvq, t, y, l);
    }
};
template <>
struct BackLoop<-1> {
    static __device__ __forceinline__ void run(const double (&)[NR][NC], const double (&)[3], double (&)[3], double (&)[12], int) {}
};

// ---- the node on TWO wavefronts per SIMD (round 3; round 2's one-wave-per-SIMD kernel -- in-flight loads hidden from the compiler behind
//      inline-asm AGPR destinations and a manual s_waitcnt -- was deleted in round 4: correct, 25 % slower, and one compiler bump away from
//      spilling such a destination) --------
// A lone wavefront cannot keep the FP64 pipe busy (tools/micro_mfma64.hip: one wave issues a v_fma_f64 every 8.5 cycles, two
// waves together one every 5.9), and the kernel above is one wave per SIMD because its peak -- the end of phase 1, where the
// finished fill rows, the panel, a block in flight and the face records coexist -- is ~340 registers.  This one fits 256:
//   * phase 1 takes the columns of the odd cells ONE at a time (10 doubles in flight, not 30) and in FACE order (a lane's three
//     faces; the slot it has no face on is never computed), each through the reflectors' mutual products (w2_column); the
//     first 35 finished fill entries of a lane (c, face 0, the first column of face 1) wait in LDS -- lane-private slots,
//     [slot][lane], no conflicts, no synchronisation -- until the panel and the face records are gone;
//   * phase 2 forms each column's dot where it is used (the other wave fills the scalar chain's latency), with u and s in the
//     LDS slots the fill entries have left;
//   * the index levels of the next pass (list entry -> CSR row starts, coordinates -> cell / face ids) come in by LDS-DMA
//     during this one (nothing in flight that the compiler could move or spill); geometry and permeability by plain loads at
//     the top of the pass; a per-XCD work queue (the SIMD issues its older wave first: equal shares finish 4.0 / 5.85 ms apart).
// With two waves the SIMD is bound by instruction issue (FP64 at ~2.6 ns an instruction, everything else at ~1): every change
// since the first version that fitted was a cut in instructions -- 4 159 -> 4 096 a pass, of them FP64 2 926 -> 2 426; executed
// flops per node 20.5 k (kernel above) -> 16.7 k against 15.9 k algorithmic.  5.95 -> 4.8 .. 5.0 ms per launch at 216^3.
// Same arithmetic as the kernel above up to the order of a few sums: results differ by rounding only.
#ifdef NIN_W2_FENCE_COLUMNS
#define NIN_W2_COLUMN_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define NIN_W2_COLUMN_FENCE() do { } while (0)
#endif
constexpr int W2_SLOTS = 35;                                  // 64-lane slots of 8 B per wave
constexpr int W2_WAVE_DOUBLES = W2_SLOTS * 64 + NPW * 8 + 64;  // + the weights' staging + two dword rows (19 456 B per wave: two workgroups per CU)

template <int K>
__device__ __forceinline__ void p2_step_lean(double (&C)[NR][NC], double (&rinvq)[3], int l) {
    constexpr int Q = K / 4, LAM = K % 4;
    // (each column's dot is formed where it is used: ahead of the step's scalars -- the one-wave kernel's way of filling that
    //  chain's latency -- it costs one more FMA per column for the pivot lane's entry, and the other wave fills the latency
    //  here: 5.05 -> 4.92 ms, measured)
    const bool is_piv = (l == LAM);
    const double xq = (l > LAM) ? C[Q][K] : 0.0;              // row Q counts as part of x only above the pivot lane
    double ss = xq * xq;
#pragma unroll
    for (int r = Q + 1; r < NR; ++r) ss = fma(C[r][K], C[r][K], ss);
    ss = quad_sum(ss);
    const double alpha = quad_bcast<LAM>(C[Q][K]);
    const House h = house_unguarded(alpha, ss);
    rinvq[Q] = is_piv ? h.rinv : rinvq[Q];
    const double vq = is_piv ? h.vp : xq;                     // row Q's entry of v in this lane
#pragma unroll
    for (int j = K + 1; j < NC; ++j) {
        double a = vq * C[Q][j];
#pragma unroll
        for (int r = Q + 1; r < NR; ++r) a = fma(C[r][K], C[r][j], a);
        const double w = -(h.g * quad_sum(a));
        C[Q][j] = fma(w, vq, C[Q][j]);                        // the pivot lane's row Q becomes row K of R
#pragma unroll
        for (int r = Q + 1; r < NR; ++r) C[r][j] = fma(w, C[r][K], C[r][j]);
    }
}
template <int K, int KEND>
struct P2LeanLoop {
    static __device__ __forceinline__ void run(double (&C)[NR][NC], double (&rinvq)[3], int l) {
        p2_step_lean<K>(C, rinvq, l);
        P2LeanLoop<K + 1, KEND>::run(C, rinvq, l);
    }
};
template <int KEND>
struct P2LeanLoop<KEND, KEND> {
    static __device__ __forceinline__ void run(double (&)[NR][NC], double (&)[3], int) {}
};

// The three reflectors of a front's panel on a column that enters with ONE face's three entries (rows R0 .. R0 + 2: b[0 .. 2])
// and zeros elsewhere.  With the reflectors' mutual products cc = (v1 . v0, v2 . v0, v2 . v1) known (once per node), the dots
// with the filled column need the face's rows only -- H2 H1 H0 b = b + w0 v0 + w1 v1 + w2 v2,
//   w0 = -g0 (v0 . b),  w1 = -g1 (v1 . b + w0 v1 . v0),  w2 = -g2 (v2 . b + w0 v2 . v0 + w1 v2 . v1)
// -- 45 operations a column against 54 for the three reflectors one after the other.  v_k = P[k .. 9][k] (pivot entries
// included).  Out: u = z^T (rows 0 .. 2), the seven fill entries (rows 3 .. 9).  R0 < 0: the column c = e_0.
template <int R0>
__device__ __forceinline__ void w2_column(const double (&P)[10][3], const double (&g3)[3], const double (&cc)[3], const double (&z)[3],
                                          const double (&b)[3], double &u_out, double (&fill)[7]) {
    double w0, w1, w2;
    if (R0 < 0) {
        w0 = -(g3[0] * P[0][0]);
        w1 = -(g3[1] * (w0 * cc[0]));
        w2 = -(g3[2] * fma(w1, cc[2], w0 * cc[1]));
    } else {
        w0 = -(g3[0] * fma(P[R0 + 2][0], b[2], fma(P[R0 + 1][0], b[1], P[R0][0] * b[0])));
        w1 = -(g3[1] * fma(w0, cc[0], fma(P[R0 + 2][1], b[2], fma(P[R0 + 1][1], b[1], P[R0][1] * b[0]))));
        const double d2 = (R0 >= 2) ? fma(P[R0 + 2][2], b[2], fma(P[R0 + 1][2], b[1], P[R0 < 0 ? 0 : R0][2] * b[0]))
                                    : fma(P[R0 + 2][2], b[2], P[R0 + 1][2] * b[1]);     // (R0 = 1: row 1 lies above v2)
        w2 = -(g3[2] * fma(w1, cc[2], fma(w0, cc[1], d2)));
    }
    double B[10];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const bool own = R0 >= 0 && r >= R0 && r < R0 + 3;
        double v = own ? fma(w0, P[r][0], b[own ? r - R0 : 0]) : (R0 < 0 && r == 0) ? fma(w0, P[0][0], 1.0) : w0 * P[r][0];
        if (r >= 1) v = fma(w1, P[r][1], v);
        if (r >= 2) v = fma(w2, P[r][2], v);
        B[r] = v;
    }
    u_out = fma(z[2], B[2], fma(z[1], B[1], z[0] * B[0]));
#pragma unroll
    for (int r = 0; r < 7; ++r) fill[r] = B[3 + r];
}

// LDS-DMA of one dword per lane (lane i's word lands at row[i]); the compiler does not order LDS reads behind it: every
// consumer goes through w2_dma_wait()
__device__ __forceinline__ void w2_dma(const void *gptr, uint32_t *row) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gptr, (__attribute__((address_space(3))) void *)row, 4, 0, 0);
}
__device__ __forceinline__ void w2_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// level 0 of group `wg`: list entry and descriptor word (past the end: a clamped, valid entry -- never stored)
__device__ __forceinline__ void w2_stage_a(const int32_t *nodes, const int32_t *desc, int32_t wg, int32_t count, int nd, int l,
                                           uint32_t *row_p, uint32_t *row_dsc) {
    const int32_t idx = wg * NPW + nd;
    const uint32_t sel = (uint32_t)((idx >= 0 && idx < count) ? idx : count - 1);
    w2_dma(nodes + sel, row_p);
    w2_dma(desc + 4 * (size_t)sel + l, row_dsc);
}
// level 1: CSR row starts and the node's coordinates (as six dwords)
__device__ __forceinline__ void w2_stage_b(const GridView &g, uint32_t p, uint32_t *rows_n) {
    w2_dma(g.esup_ptr + p, rows_n);
    w2_dma(g.fsup_ptr + p, rows_n + 64);
    const uint32_t *xw = reinterpret_cast<const uint32_t *>(g.coords) + 6 * (size_t)p;
#pragma unroll
    for (int k = 0; k < 6; ++k) w2_dma(xw + k, rows_n + 64 * (2 + k));
}
// level 2: the lane's two cells, the cells across its three faces, the faces
__device__ __forceinline__ void w2_stage_c(const GridView &g, uint32_t eb, uint32_t fb, uint32_t dsc, uint32_t *rows_n) {
    w2_dma(g.esup + eb + (dsc & 7), rows_n + 64 * 8);
    w2_dma(g.esup + eb + ((dsc >> 3) & 7), rows_n + 64 * 9);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint32_t w = dsc >> (6 + 8 * i);
        w2_dma(g.esup + eb + ((w >> 4) & 7), rows_n + 64 * (10 + i));
        w2_dma(g.fsup + fb + (w & 15), rows_n + 64 * (13 + i));
    }
}

#ifdef NIN_W2_TRACE
__device__ unsigned long long nin_w2_trace[4096 * 4];   // per wave: start, end (s_memrealtime, 100 MHz), HW_ID, passes
#endif
// APPLY: instead of the weights, the node values W . u of `n_fields` cell fields (u: [n_fields][n_elems], values:
// [n_fields][n_points]) -- the 64 bytes of a node's row are never written, nor read again by an apply kernel
template <bool APPLY>
__global__ __launch_bounds__(256, 2) void nin_gls_hex8w2_kernel(GridView g, const int32_t *__restrict__ nodes,
                                                                const int32_t *__restrict__ desc, int32_t count,
                                                                int add_neumann, double *__restrict__ out,
                                                                double *__restrict__ nws, int32_t *__restrict__ queue,
                                                                const double *__restrict__ u_cells, int32_t n_fields,
                                                                double *__restrict__ values) {
    __shared__ double lds_all[4][W2_WAVE_DOUBLES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l = lane & 3, nd = lane >> 2;
    double *const L = lds_all[wave] + lane;                   // slot k of this lane: L[64 k]
    double *const wbuf = lds_all[wave] + W2_SLOTS * 64;
    // The index levels of the NEXT pass (list entry -> CSR row starts, coordinates -> cell / face ids) come in by LDS-DMA
    // (global_load_lds_dword: lane i's word lands at row[i]; no register is in flight, nothing the compiler could move or
    // spill): rows of 64 dwords.  (p, dsc) have rows of their own; the others lie in parking slots 13 .. 20, which are free
    // from the moment the tile is complete until the next pass parks again -- after it has read them.
    uint32_t *const row_p = reinterpret_cast<uint32_t *>(lds_all[wave] + W2_SLOTS * 64 + NPW * 8);
    uint32_t *const row_dsc = row_p + 64;
    uint32_t *const rows_n = reinterpret_cast<uint32_t *>(lds_all[wave] + 13 * 64);   // 0: eb, 1: fb, 2 .. 7: x_v, 8 .. 15: ids

    const int32_t n_groups = (count + NPW - 1) / NPW;
    // XCD x walks the x-th contiguous eighth of the node list; its waves pull consecutive 16-node groups off a per-XCD counter.
    // (Not round-robin: the SIMD issues its OLDER wave first, so the second workgroup of a CU runs at what the first leaves --
    // 5.85 against 4.0 ms for equal shares, measured.)  The ticket of pass n + 1 is drawn at the top of pass n.
    int32_t wg_lo, wg_end;
    int32_t *q;
    if ((gridDim.x & 7) == 0) {
        const int32_t xcd = blockIdx.x & 7, per = (n_groups + 7) / 8;
        wg_lo = xcd * per;
        wg_end = (xcd + 1) * per < n_groups ? (xcd + 1) * per : n_groups;
        q = queue + 16 * xcd;   // one counter per 64-byte line
    } else {
        wg_lo = 0;
        wg_end = n_groups;
        q = queue;
    }
    int32_t ticket = 0;
    if (lane == 0) ticket = atomicAdd(q, 1);
    int32_t wg = wg_lo + __builtin_amdgcn_readfirstlane(ticket);
    if (lane == 0) ticket = atomicAdd(q, 1);
    int32_t wg_next = wg_lo + __builtin_amdgcn_readfirstlane(ticket);
    // the first group's index levels, one after the other
    w2_stage_a(nodes, desc, wg, count, nd, l, row_p, row_dsc);
    w2_dma_wait();
    w2_stage_b(g, row_p[lane], rows_n);
    w2_dma_wait();
    w2_stage_c(g, rows_n[lane], rows_n[64 + lane], row_dsc[lane], rows_n);
#ifdef NIN_MF_STAMPS     // diagnostic build (tools/stamps_hex8mf.py): s_memtime at the phase boundaries of one pass
#ifndef NIN_MF_STAMP_PASS
#define NIN_MF_STAMP_PASS 8
#endif
    unsigned long long stamps[8];
    int n_stamp = 0, pass_no = 0;
#define NIN_MF_STAMP() do { if (n_stamp < 8) { __builtin_amdgcn_sched_barrier(0); stamps[n_stamp++] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define NIN_MF_STAMP() do { } while (0)
#endif
#ifdef NIN_W2_TRACE
    const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned trace_passes = 0;
#endif
    while (wg < wg_end) {
        if (lane == 0) ticket = atomicAdd(q, 1);              // names the group of the pass after the next
#ifdef NIN_W2_TRACE
        ++trace_passes;
#endif
#ifdef NIN_MF_STAMPS
        n_stamp = 0;
#endif
        NIN_MF_STAMP();                                   // 0: top of the pass
        const bool valid = wg * NPW + nd < count;
        w2_dma_wait();                                        // (the ids of this pass: requested in the middle of the last one)
        const uint32_t p = row_p[lane], dsc = row_dsc[lane];
        const uint32_t eb = rows_n[lane], fb = rows_n[64 + lane];
        const bool is_neu = (g.flags[p] & 2) != 0;
        double xv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) xv[k] = __hiloint2double((int)rows_n[64 * (3 + 2 * k) + lane], (int)rows_n[64 * (2 + 2 * k) + lane]);
        uint32_t id[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) id[k] = rows_n[64 * (8 + k) + lane];
        const uint32_t ce = id[0], co = id[1];

        // ---- the front of E_l: rows 0 = cell row, 1 + 3 i + r = row r of face i; own columns in P ------------
        double P[10][3], de[3], dod[3], nb0[3][3], sav[3][2][3];
        {
            double Ke[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) Ke[k] = g.perm[9 * (size_t)ce + k];
            const double dme = g.diff_mag[ce];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                de[t] = g.centroids[3 * (size_t)ce + t] - xv[t];      // gls.pyx:269-277
                dod[t] = g.centroids[3 * (size_t)co + t] - xv[t];
                P[0][t] = de[t];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const uint32_t w = dsc >> (6 + 8 * i);
                const uint32_t cn = id[2 + i], f = id[5 + i];
                // B = [K N; T1; tau T2] (gls.pyx:293-321), row = [-B_a | +B_b] (gls.pyx:340-356)
                const double N0 = (double)g.face_normal[3 * (size_t)f + 0], N1 = (double)g.face_normal[3 * (size_t)f + 1],
                             N2 = (double)g.face_normal[3 * (size_t)f + 2];
                const double T0 = xv[0] - g.face_center[3 * (size_t)f + 0], T1 = xv[1] - g.face_center[3 * (size_t)f + 1],
                             T2 = xv[2] - g.face_center[3 * (size_t)f + 2];
                const double U0 = N1 * T2 - N2 * T1, U1 = N2 * T0 - N0 * T2, U2 = N0 * T1 - N1 * T0;
                const double dmn = g.diff_mag[cn];
                double eta = 0.0;
                eta = dme > eta ? dme : eta;
                eta = dmn > eta ? dmn : eta;
                const double tj = face_tau_sq(U0 * U0 + U1 * U1 + U2 * U2, eta);   // |T_sj2|^(-eta) without the square root
                const bool side_a = ((w >> 7) & 1) != 0;
                const double sg = side_a ? -1.0 : 1.0;
                const double *Kn = g.perm + 9 * (size_t)cn;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    P[1 + 3 * i][t] = sg * (Ke[t * 3 + 0] * N0 + Ke[t * 3 + 1] * N1 + Ke[t * 3 + 2] * N2);
                    nb0[i][t] = -sg * (Kn[t * 3 + 0] * N0 + Kn[t * 3 + 1] * N1 + Kn[t * 3 + 2] * N2);
                }
                sav[i][0][0] = sg * T0; sav[i][0][1] = sg * T1; sav[i][0][2] = sg * T2;
                sav[i][1][0] = sg * (tj * U0); sav[i][1][1] = sg * (tj * U1); sav[i][1][2] = sg * (tj * U2);
#pragma unroll
                for (int t = 0; t < 3; ++t) { P[2 + 3 * i][t] = sav[i][0][t]; P[3 + 3 * i][t] = sav[i][1][t]; }
            }
        }
        NIN_MF_STAMP();                                   // 1: geometry in, face rows done
        w2_stage_a(nodes, desc, wg_next, count, nd, l, row_p, row_dsc);   // next pass: list entry and descriptor
        double C[NR][NC];
        double u[9], se, F1[7][2], F2[7][3];                  // u = z^T R_eo by face; fill entries of faces 1 and 2 (face 0: LDS)
        {
            double g3[3], z[3];
            front_panel(P, de, g3, z);
            pin(z[0]); pin(z[1]); pin(z[2]);
            // the reflectors' mutual products (w2_column)
            double cc[3];
            cc[0] = P[1][1] * P[1][0];
            cc[1] = P[2][2] * P[2][0];
            cc[2] = P[2][2] * P[2][1];
#pragma unroll
            for (int r = 2; r < 10; ++r) cc[0] = fma(P[r][1], P[r][0], cc[0]);
#pragma unroll
            for (int r = 3; r < 10; ++r) { cc[1] = fma(P[r][2], P[r][0], cc[1]); cc[2] = fma(P[r][2], P[r][1], cc[2]); }
            {
                const double none[3] = {0.0, 0.0, 0.0};
                double fill[7];
                w2_column<-1>(P, g3, cc, z, none, se, fill);   // c = e_0 on entry: only the cell row carries a 1
                pin(se);
#pragma unroll
                for (int r = 0; r < 7; ++r) L[64 * r] = fill[r];                            // slots 0 .. 6: column c
            }
            __builtin_amdgcn_sched_barrier(0);
            // The blocks in FACE order: face i of lane l belongs to odd slot i (i < 3 - l) or i + 1, and the slot a lane has
            // no face on (3 - l) is zero -- so every lane runs three one-face blocks, no role masks, no block of zeros, and
            // the fill entries F0, F1, F2 find their slots when the tile is put together (selects by lane role).
#pragma unroll
            for (int t = 0; t < 3; ++t) {                                                    // face 0: rows 1 .. 3
                const double b[3] = {nb0[0][t], -sav[0][0][t], -sav[0][1][t]};
                double fill[7];
                w2_column<1>(P, g3, cc, z, b, u[t], fill);
                pin(u[t]);
#pragma unroll
                for (int r = 0; r < 7; ++r) L[64 * (7 + 7 * t + r)] = fill[r];               // slots 7 .. 27: F0
                NIN_W2_COLUMN_FENCE();
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 3; ++t) {                                                    // face 1: rows 4 .. 6
                const double b[3] = {nb0[1][t], -sav[1][0][t], -sav[1][1][t]};
                double fill[7];
                w2_column<4>(P, g3, cc, z, b, u[3 + t], fill);
                pin(u[3 + t]);
                if (t == 0) {
#pragma unroll
                    for (int r = 0; r < 7; ++r) L[64 * (28 + r)] = fill[r];                  // slots 28 .. 34: F1, first column
                } else {
#pragma unroll
                    for (int r = 0; r < 7; ++r) F1[r][t - 1] = fill[r];
                }
                NIN_W2_COLUMN_FENCE();
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 3; ++t) {                                                    // face 2: rows 7 .. 9
                const double b[3] = {nb0[2][t], -sav[2][0][t], -sav[2][1][t]};
                double fill[7];
                w2_column<7>(P, g3, cc, z, b, u[6 + t], fill);
                pin(u[6 + t]);
#pragma unroll
                for (int r = 0; r < 7; ++r) F2[r][t] = fill[r];
                NIN_W2_COLUMN_FENCE();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        NIN_MF_STAMP();                                   // 2: phase 1 done
        // ---- the tile: odd slot 0 = F0 (lanes 0 .. 2); slot 1 = F1 (lanes 0, 1) or F0 (lane 3); slot 2 = F2 (lane 0) or F1
        //      (lanes 2, 3); slot 3 = F2 (lanes 1 .. 3); the slot a lane has no face on is zero.  The entries that waited in LDS
        //      come back; u and s take their place ------------------------------------------------------------------------------
        {
            const bool lt3 = l < 3, lt2 = l < 2, eq0 = l == 0, ge2 = l >= 2, ge1 = l >= 1, eq3 = l == 3;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
#pragma unroll
                for (int r = 0; r < 7; ++r) {
                    const double f1 = (t == 0) ? L[64 * (28 + r)] : F1[r][t == 0 ? 0 : t - 1];
                    const double f0 = L[64 * (7 + 7 * t + r)];
                    C[r][6 + t] = eq0 ? F2[r][t] : (ge2 ? f1 : 0.0);
                    C[r][9 + t] = ge1 ? F2[r][t] : 0.0;
                    C[r][3 + t] = lt2 ? f1 : (eq3 ? f0 : 0.0);
                    C[r][t] = lt3 ? f0 : 0.0;
                }
            }
#pragma unroll
            for (int r = 0; r < 7; ++r) C[r][12] = L[64 * r];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 9; ++j) L[64 * j] = u[j];
        L[64 * 9] = se;
        // row 7: the cell row of O_l, (x_K - x_v) on the columns of odd slot l, c = 1
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int t = 0; t < 3; ++t) C[7][3 * s + t] = (l == s) ? dod[t] : 0.0;
        }
        C[7][12] = 1.0;
        __builtin_amdgcn_sched_barrier(0);

        // ---- phase 2: 32 x 12 over the quad ------------------------------------------------------------------
        double rinvq[3] = {0.0, 0.0, 0.0};
        NIN_MF_STAMP();                                   // 3: tile complete
        w2_dma_wait();
        w2_stage_b(g, row_p[lane], rows_n);               // next pass: CSR row starts, node coordinates
        __builtin_amdgcn_sched_barrier(0);
        P2LeanLoop<0, 6>::run(C, rinvq, l);
        NIN_MF_STAMP();                                   // 4: phase 2, steps 0-5
        __builtin_amdgcn_sched_barrier(0);
        w2_dma_wait();
        w2_stage_c(g, rows_n[lane], rows_n[64 + lane], row_dsc[lane], rows_n);   // next pass: cell and face ids
        __builtin_amdgcn_sched_barrier(0);
        P2LeanLoop<6, 12>::run(C, rinvq, l);
        NIN_MF_STAMP();                                   // 5: phase 2 done
        double y[12], t3[3] = {C[0][12], C[1][12], C[2][12]};
        BackLoop<11>::run(C, rinvq, t3, y, l);
        NIN_MF_STAMP();                                   // 6: back-substitution done
        double tail = 0.0;
#pragma unroll
        for (int r = 3; r < NR; ++r) tail = fma(C[r][12], C[r][12], tail);
        const double rr = quad_sum(tail);                        // r . r = |(Q^T c)(24:44)|^2

        // ---- residuals on the two cell rows of this lane, weights ----------------------------------------------
        double re = 1.0 - L[64 * 9];                             // r_e = 1 - d_e . y_e = 1 - z . b_e + u . y_odd
#pragma unroll
        for (int i = 0; i < 3; ++i) {                            // face i sits on odd slot i (i < 3 - l) or i + 1
            const bool low = i < 3 - l;
#pragma unroll
            for (int t = 0; t < 3; ++t) re = fma(L[64 * (3 * i + t)], low ? y[3 * i + t] : y[3 * i + 3 + t], re);
        }
        double dots[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) dots[s] = fma(dod[2], y[3 * s + 2], fma(dod[1], y[3 * s + 1], dod[0] * y[3 * s]));
        const double dsel = (l == 0) ? dots[0] : (l == 1) ? dots[1] : (l == 2) ? dots[2] : dots[3];
        const double ro = 1.0 - dsel;
        const double rri = fast_rcp(rr);
        double we = re * rri, wo = ro * rri;
        const bool ok = rr > 0.0;                                // (as above: the zero row for a rank-deficient system)
        we = (ok && __builtin_isfinite(we)) ? we : 0.0;
        wo = (ok && __builtin_isfinite(wo)) ? wo : 0.0;

        wbuf[nd * 8 + (dsc & 7)] = we;
        wbuf[nd * 8 + ((dsc >> 3) & 7)] = wo;
        wave_lds_sync();
        const double nwv = is_neu ? wbuf[nd * 8 + 7] : 0.0;      // gls.pyx:470-472
        const double addv = add_neumann ? nwv : 0.0;
        const double o0 = wbuf[nd * 8 + 2 * l] + addv, o1 = wbuf[nd * 8 + 2 * l + 1] + addv;
        if (APPLY) {
            // this lane's two entries of the row (esup order) times u, summed over the quad: one value per node and field
            const size_t c0 = (size_t)g.esup[eb + 2 * l], c1 = (size_t)g.esup[eb + 2 * l + 1];
            for (int32_t f = 0; f < n_fields; ++f) {
                const double *uf = u_cells + (size_t)f * (size_t)g.n_elems;
                const double yv = quad_sum(fma(o1, uf[c1], o0 * uf[c0]));
                if (valid && l == 0) values[(size_t)f * (size_t)g.n_points + p] = yv;
            }
            if (valid && l == 0) nws[p] = nwv;
        } else if (valid) {
            out[eb + 2 * l] = o0;
            out[eb + 2 * l + 1] = o1;
            if (l == 0) nws[p] = nwv;
        }
        wave_lds_sync();
        NIN_MF_STAMP();                                   // 7: weights stored
#ifdef NIN_MF_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0 && ++pass_no == NIN_MF_STAMP_PASS)
            for (int i = 0; i < 8; ++i) nws[nodes[i]] = (double)(stamps[i] - stamps[0]);   // (diagnostic build: clobbers neumann_ws of the first 8 listed nodes)
#endif
        wg = wg_next;
        wg_next = wg_lo + __builtin_amdgcn_readfirstlane(ticket);
    }
    w2_dma_wait();   // nothing may still be on its way into this workgroup's LDS when the wave ends
#ifdef NIN_W2_TRACE
    if (lane == 0 && blockIdx.x < 1024) {
        unsigned long long *t = nin_w2_trace + 4 * ((size_t)blockIdx.x * 4 + wave);
        t[0] = trace_t0;
        t[1] = __builtin_amdgcn_s_memrealtime();
        t[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
        t[3] = trace_passes;
    }
#endif
}

__global__ void k_hex8_desc(GridView g, const int32_t *__restrict__ nodes, int32_t count, int32_t *__restrict__ desc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    int32_t d[4] = {0, 0, 0, 0};
    (void)hex8_descriptor(g, nodes ? nodes[i] : (int32_t)i, d);   // the list holds classified cube nodes only
#pragma unroll
    for (int l = 0; l < 4; ++l) desc[4 * i + l] = d[l];
}

}  // namespace

int launch_hex8_desc(const GridView &g, const int32_t *nodes, int32_t count, int32_t *desc, hipStream_t stream) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(k_hex8_desc, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, g, nodes, count, desc);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_gls_hex8mf(const GridView &g, const int32_t *nodes, const int32_t *desc, int32_t count, int add_neumann,
                      double *out, double *nws, int32_t *queue, hipStream_t stream) {
    if (count <= 0) return 0;
    int64_t blocks = ((int64_t)count + 4 * NPW - 1) / (4 * NPW);
    const char *cap_env = getenv("NIN_W2_BLOCKS");   // (experiments)
    const int64_t cap2 = cap_env ? atoll(cap_env) : 512;
    if (blocks > cap2) blocks = cap2;            // two 4-wave workgroups per CU are resident (256 registers, 74 KB of LDS each)
    if (blocks > 8) blocks &= ~(int64_t)7;
    if (getenv("NIN_DEBUG_OCCUPANCY") != nullptr) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nin_gls_hex8w2_kernel<false>, 256, 0);
        fprintf(stderr, "nin_gls_hex8w2_kernel: %d workgroups per CU, %lld launched\n", nb, (long long)blocks);
    }
    hipLaunchKernelGGL(nin_gls_hex8w2_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, g, nodes, desc, count, add_neumann, out, nws, queue,
                       nullptr, 0, nullptr);
#ifdef NIN_W2_TRACE
    if (getenv("NIN_W2_TRACE_FILE") != nullptr) {
        static int calls = 0;
        if (++calls == 6) {
            (void)hipStreamSynchronize(stream);
            static unsigned long long host[4096 * 4];
            (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(nin_w2_trace), sizeof(host));
            FILE *f = fopen(getenv("NIN_W2_TRACE_FILE"), "w");
            for (int64_t b = 0; b < blocks * 4 && b < 4096; ++b)
                fprintf(f, "%lld %llu %llu %llx %llu\n", (long long)b, host[4 * b], host[4 * b + 1], host[4 * b + 2], host[4 * b + 3]);
            fclose(f);
        }
    }
#endif
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// the two-wave kernel in its apply form: values[f][p] = (W . u_f)[p] for the listed cube nodes, neumann_ws as always
int launch_gls_hex8mf_apply(const GridView &g, const int32_t *nodes, const int32_t *desc, int32_t count, int add_neumann,
                            const double *u_cells, int32_t n_fields, double *values, double *nws, int32_t *queue, hipStream_t stream) {
    if (count <= 0) return 0;
    int64_t blocks = ((int64_t)count + 4 * NPW - 1) / (4 * NPW);
    if (blocks > 512) blocks = 512;
    if (blocks > 8) blocks &= ~(int64_t)7;
    hipLaunchKernelGGL(nin_gls_hex8w2_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, g, nodes, desc, count, add_neumann,
                       nullptr, nws, queue, u_cells, n_fields, values);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// (as rocprofv3 prints it: the weights form of the two-wave kernel is the instantiation <false>, the apply form <true>)
const char *kernel_name_gls_hex8mf() { return "nin_gls_hex8w2_kernel<false>"; }

}  // namespace nin
