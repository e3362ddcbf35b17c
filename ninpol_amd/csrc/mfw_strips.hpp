// mfw_strips.hpp -- cross-lane helpers and the STRIP form of the dense phase (the FP64 matrix unit as the cross-lane adder),
// shared by kernels_gls_mfw.hip (two-coloured nodes, two wavefronts per SIMD) and kernels_gls_mfx.hip (the wide general kind:
// unstructured tetrahedra, one wavefront per SIMD with the tiles in the accumulation registers).  Internal, device code only.
#pragma once
#include <hip/hip_runtime.h>

#include "gls_device_math.hpp"

namespace nin {
namespace mfwstrips {

using namespace glsmath;

__device__ __forceinline__ double rl64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t rl32(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

// ---- the dense problem with lane = ROW (large instantiation) -------------------------------------------------------------
// Cross-lane pieces.  v_permlane32_swap / v_permlane16_swap (gfx950) exchange halves / odd-even 16-lane rows between
// TWO registers: one instruction per dword moves two columns' partial sums towards each other.
__device__ __forceinline__ void swap32(double &x, double &y) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]);
    y = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void swap16(double &x, double &y) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]);
    y = __hiloint2double((int)hi[1], (int)lo[1]);
}
// sum over the 16 lanes of a row, in every lane of the row
__device__ __forceinline__ double row_allsum(double v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x124>(v);   // row_ror:4
    v += dpp_mov<0x128>(v);   // row_ror:8
    return v;
}
// sum over the 64 lanes, in every lane
__device__ __forceinline__ double wave_allsum(double v) {
    v = row_allsum(v);
    double x = v, y = v;
    swap16(x, y);             // x = [r0 r0 r2 r2], y = [r1 r1 r3 r3]
    v = x + y;
    x = v; y = v;
    swap32(x, y);             // x = [lo lo], y = [hi hi]
    return x + y;
}
// Four columns' per-lane partial sums -> one register whose 16-lane rows hold the four totals:
// row 0: p0, row 1: p2, row 2: p1, row 3: p3 (every lane of the row).
__device__ __forceinline__ double reduce4(double p0, double p1, double p2, double p3) {
    swap32(p0, p1);           // p0 = [p0.lo | p1.lo], p1 = [p0.hi | p1.hi]
    const double s01 = p0 + p1;
    swap32(p2, p3);
    const double s23 = p2 + p3;
    double x = s01, y = s23;
    swap16(x, y);             // x = [s01.r0, s23.r0, s01.r2, s23.r2], y = [s01.r1, s23.r1, s01.r3, s23.r3]
    return row_allsum(x + y);
}
constexpr int kReduce4Lane[4] = {0, 32, 16, 48};   // where column i of a reduce4 group is read back

// ---- the dense problem in STRIPS: the matrix unit does the cross-lane sums ------------------------------------------------
// v_mfma_f64_4x4x4_4b (tools/micro_mfma64.hip: 16 cycles, 512 flops -- the FP64 matrix peak of an MI355X equals its vector
// peak, so nothing is gained in arithmetic; what is gained is that the instruction SUMS ACROSS LANES): with a register read
// as a 4 x 16 strip S[k][c], k = lane >> 4, c = lane & 15 = 4 quad + j, it computes, quad by quad,
//         D_quad = (S1_quad)^T  S2_quad + C_quad          (4 x 4 blocks; D, S2, C in the same strip layout)
// i.e. a contraction over the strip's row index.  The dense (7 F + D) x (3 D + 1) problem is held as 16-row x 4-column tiles,
// one per register: lane (i = lane >> 4, quad, j = lane & 3) of tile [q][cb] = row 16 q + 4 quad + i, column 4 cb + j.  A panel of
// four reflectors (compact WY: H_0 .. H_3 = I - V T V^T, tools/proto_mfw.py dense_blocked) then needs, per trailing column
// block: W = V^T C (one instruction per row tile, the four quads' partial sums joined by two DPP adds), W' = T^T W (one
// instruction), C -= V W' (one per row tile) -- against one 64-lane reduction and two v_readlane per COLUMN AND REFLECTOR in the
// row-lane form.  Only the panel itself (4 columns) is factored by the vector unit, one reduction per step for all its columns.
// After a panel the column blocks move down by one register, so one body serves every panel.
#ifdef NIN_MFW_STAMPS
struct SubStamps { unsigned long long last, acc[6]; bool on; };
#define NIN_SUB(ST, J) do { if ((ST).on) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); (ST).acc[J] += t_ - (ST).last; (ST).last = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
struct SubStamps { };
#define NIN_SUB(ST, J) do { } while (0)
#endif
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
// sum over the four quads of a 16-lane row, in every quad
__device__ __forceinline__ double sum_quads(double v) {
    v += dpp_mov<0x124>(v);   // row_ror:4
    v += dpp_mov<0x128>(v);   // row_ror:8
    return v;
}
// sum over the four 16-lane rows, in every row
__device__ __forceinline__ double sum_rows(double v) {
    double x = v, y = v;
    swap16(x, y);
    v = x + y;
    x = v; y = v;
    swap32(x, y);
    return x + y;
}
template <int K>
__device__ __forceinline__ double quad_pick(double v) { return dpp_mov<K * 0x55>(v); }   // quad_perm [K,K,K,K]

// two sums over the four 16-lane rows at once: a and b each summed over the rows, in every row (8 instructions, not 12)
__device__ __forceinline__ void sum_rows2(double &a, double &b) {
    swap16(a, b);             // a = [a0 b0 a2 b2], b = [a1 b1 a3 b3]
    double s = a + b;         //     [a01 b01 a23 b23]
    double x = s, y = s;
    swap32(x, y);             // x = [a01 b01 a01 b01], y = [a23 b23 a23 b23]
    s = x + y;                //     [A B A B]
    a = s; b = s;
    swap16(a, b);             // a = [A A A A], b = [B B B B]
}

// One panel step K on the panel's tiles P[q], q >= Q0 (column block 0): pivot row rp = 4 p + K = (tile Q0, quad bp, i = K).
// Tiles above Q0 hold rows below every pivot of this panel: no masks there.  vp[K], gk[K]: the reflector's pivot entry and scalar.
// Tr[K]: on exit this lane's entry T[si][K] of the panel's triangular factor (H_0 .. H_3 = I - V T V^T: T[K][K] = g_K,
// T[0:K, K] = -g_K T[0:K, 0:K] (V^T v_K)).  The products v_l . v_K (l < K) are the SAME sums as the step's own dots -- column
// l below the pivot still holds v_l -- so they come out of the step's one reduction, in the lanes j = l; and since row si of T
// is zero left of its diagonal, sum_l Tr[l] G[l][K] over ALL l < K is the right sum in every lane (zero below the diagonal).
template <int NQ, int Q0, int K>
__device__ __forceinline__ void strip_panel_step(double (&P)[NQ], double (&vp)[4], double (&gk)[4], double (&Tr)[4], int bp, int si, int sb, int sj) {
    const bool in_piv_quad = sb == bp;
    const bool is_piv = in_piv_quad && si == K;                  // this lane's row of tile Q0 is the pivot row
    const bool below0 = sb > bp || (in_piv_quad && si > K);      // ... lies below it
    double xm[NQ], acc = 0.0;
#pragma unroll
    for (int q = Q0; q < NQ; ++q) {
        const double xk = quad_pick<K>(P[q]);
        xm[q] = (q == Q0 && !below0) ? 0.0 : xk;                 // the reflector's entries below the pivot
        acc = fma(xm[q], P[q], acc);                             // lane (.., j): sum over its rows of a[r][K] a[r][j]
    }
    // the pivot row's entry of column j, for every lane with that j: a shuffle from lane (K, bp, j) -- its latency hides behind the reduction
    const double ap = __shfl(P[Q0], 16 * K + 4 * bp + sj);
    const double d = sum_rows(sum_quads(acc));
    const House h = house_unguarded(quad_pick<K>(ap), quad_pick<K>(d));
    const double e = fma(h.vp, ap, d);                           // v_K . (column j): j > K the columns still to update, j < K v_j
    const double w = sj > K ? -(h.g * e) : 0.0;                  // w_j = -g (v . a_j), the panel's later columns only
    gk[K] = h.g;
    vp[K] = h.vp;
    {
        double t = 0.0;
        if (K >= 1) t = Tr[0] * quad_pick<0>(e);
        if (K >= 2) t = fma(Tr[1], quad_pick<1>(e), t);
        if (K >= 3) t = fma(Tr[2], quad_pick<2>(e), t);
        Tr[K] = si == K ? h.g : -(h.g * t);
    }
#pragma unroll
    for (int q = Q0; q < NQ; ++q) {
        double x = fma(w, q == Q0 && is_piv ? h.vp : xm[q], P[q]);
        if (q == Q0) x = (is_piv && sj == K) ? h.beta : x;       // R(rp, rp)
        P[q] = x;
    }
}

// Panel p (tiles Q0 .. NQ - 1; NT trailing column blocks at most): factor it, apply it, store its rows of R.
template <int NQ, int NCB, int Q0, int NT>
__device__ __forceinline__ void strip_panel(double (&C)[NQ][NCB], int p, int nc, int si, int sb, int sj, double eye, double *Rm, int RP, SubStamps &ST) {
    const int bp = p & 3, steps = nc - 4 * p < 4 ? nc - 4 * p : 4;
    double V[NQ], gk[4] = {0.0, 0.0, 0.0, 0.0}, vp[4] = {0.0, 0.0, 0.0, 0.0};
    double Tr[4] = {0.0, 0.0, 0.0, 0.0};   // this lane's row si of T (a step that did not run leaves its column zero)
    {
        double P[NQ];
#pragma unroll
        for (int q = Q0; q < NQ; ++q) P[q] = C[q][0];
        strip_panel_step<NQ, Q0, 0>(P, vp, gk, Tr, bp, si, sb, sj);
        if (steps > 1) strip_panel_step<NQ, Q0, 1>(P, vp, gk, Tr, bp, si, sb, sj);
        if (steps > 2) strip_panel_step<NQ, Q0, 2>(P, vp, gk, Tr, bp, si, sb, sj);
        if (steps > 3) strip_panel_step<NQ, Q0, 3>(P, vp, gk, Tr, bp, si, sb, sj);
#pragma unroll
        for (int q = Q0; q < NQ; ++q) { C[q][0] = P[q]; V[q] = P[q]; }
        // V: below the pivots the panel's columns ARE the reflectors; in the pivot quad the diagonal takes v's pivot entries,
        // everything above it (R) and every row above the quad (earlier panels' rows of R) is zero; no reflector, no column
        const double vdiag = sj == 0 ? vp[0] : sj == 1 ? vp[1] : sj == 2 ? vp[2] : vp[3];
        double v0 = V[Q0];
        v0 = (sb == bp && si == sj) ? vdiag : v0;
        v0 = (sb < bp || (sb == bp && si < sj)) ? 0.0 : v0;
        V[Q0] = v0;
        if (steps < 4) {
#pragma unroll
            for (int q = Q0; q < NQ; ++q) V[q] = sj < steps ? V[q] : 0.0;
        }
    }
    NIN_SUB(ST, 0);   // panel factored
    // rows 4 p .. 4 p + steps - 1 of R are final after this panel: tile Q0, quad bp.  The panel block's own entries now:
    // (every lane of the pivot rows stores: below the diagonal and right of c that is a reflector's entry or a zero in a word of R's row
    //  nobody reads -- one exec region per panel instead of one per store)
    const bool r_rows = sb == bp && si < steps;
    const int col0 = 4 * p + sj;
    double *dst = Rm + (4 * p + si) * RP + col0;
    if (r_rows) dst[0] = C[Q0][0];
    // (a panel with fewer than four pivots is the last one: c sits in its block, nothing lies to the right of it)
    if (NT > 0 && steps == 4) {
        // T came out of the steps, row si in this lane; as a strip, negated: -T[k][i] at lane (k = si, ., i = sj)
        const double Ts = -(sj == 0 ? Tr[0] : sj == 1 ? Tr[1] : sj == 2 ? Tr[2] : Tr[3]);
        NIN_SUB(ST, 1);   // T
        // W[cb] = V^T C[.][cb]: NT independent accumulation chains (blocks past the live ones hold zeros: harmless)
        double W[NT + 1];
#pragma unroll
        for (int cb = 1; cb <= NT; ++cb) W[cb] = 0.0;
#pragma unroll
        for (int q = Q0; q < NQ; ++q) {
#pragma unroll
            for (int cb = 1; cb <= NT; ++cb) W[cb] = mfma4(V[q], C[q][cb], W[cb]);
        }
#pragma unroll
        for (int cb = 1; cb <= NT; ++cb) W[cb] = sum_quads(W[cb]);
#pragma unroll
        for (int cb = 1; cb <= NT; ++cb) W[cb] = mfma4(Ts, W[cb], 0.0);        // -(T^T W), the same in every quad
        NIN_SUB(ST, 2);   // W, T^T W
        // C -= V W': the A operand is V^T per quad (one instruction with the identity transposes a tile).  The result goes
        // ONE BLOCK DOWN -- the next panel (or c) lands in block 0 without a single move; tiles above Q0 hold rows of R that
        // are already in LDS and are never read again
#pragma unroll
        for (int q = Q0; q < NQ; ++q) {
            const double VT = mfma4(V[q], eye, 0.0);
#pragma unroll
            for (int cb = 1; cb <= NT; ++cb) C[q][cb - 1] = mfma4(VT, W[cb], C[q][cb]);
            C[q][NT] = 0.0;
        }
        NIN_SUB(ST, 3);   // update
        if (r_rows) {
#pragma unroll
            for (int cb = 1; cb <= NT; ++cb) dst[4 * cb] = C[Q0][cb - 1];
        }
    }
    NIN_SUB(ST, 4);   // rows of R stored
}

// Panels 2 H and 2 H + 1: they pivot in tile H >> 1 and have NCB - 1 - 2 H trailing column blocks (the second of them sweeps one
// block of zeros).  Two instantiations per generation of four panels instead of one: -13 % MFMAs (tet40 1.35 -> 1.32 ms).
template <int NQ, int NCB, int H>
__device__ __forceinline__ void strip_halves(double (&C)[NQ][NCB], int n_panels, int nc, int si, int sb, int sj, double eye, double *Rm, int RP, SubStamps &ST) {
    constexpr int NT = NCB - 1 - 2 * H, Q0 = H >> 1;
    if constexpr (NT >= 0 && Q0 < NQ) {
        for (int p = 2 * H; p < (n_panels < 2 * H + 2 ? n_panels : 2 * H + 2); ++p) strip_panel<NQ, NCB, Q0, NT>(C, p, nc, si, sb, sj, eye, Rm, RP, ST);
        strip_halves<NQ, NCB, H + 1>(C, n_panels, nc, si, sb, sj, eye, Rm, RP, ST);
    }
}

// The whole dense factorisation.  On entry C[q][cb] = the tiles (c at column nc); on exit R's rows 0 .. nc - 1 (columns up to
// nc = Q^T c) are in LDS at Rm[row * RP + col] and the return value is r . r = |(Q^T c)(nc:)|^2.  Panels 4 Q0 .. 4 Q0 + 3 pivot in
// tile Q0; they have at most NCB - 1 - 4 Q0 trailing blocks (strip_halves).
template <int NQ, int NCB>
__device__ __forceinline__ double strip_factor(double (&C)[NQ][NCB], int nc, int lane, double *Rm, int RP, SubStamps &ST) {
    const int si = lane >> 4, sb = (lane >> 2) & 3, sj = lane & 3, rowbase = 4 * sb + si;
    const double eye = si == sj ? 1.0 : 0.0;                     // the 4 x 4 identity in every quad
    const int n_panels = (nc + 3) >> 2;
    static_assert(NQ >= 3 && NCB >= 2 && 4 * NQ >= NCB, "every panel pivots inside the tiles");
    strip_halves<NQ, NCB, 0>(C, n_panels, nc, si, sb, sj, eye, Rm, RP, ST);
    // c sits in block 0, column nc & 3; r . r over the rows that never were pivot rows
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const double x = (16 * q + rowbase >= nc && sj == (nc & 3)) ? C[q][0] : 0.0;
        t = fma(x, x, t);
    }
    return wave_allsum(t);
}

}  // namespace mfwstrips
}  // namespace nin
