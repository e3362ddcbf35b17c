// mfx_strips.hpp -- the dense phase of kernels_gls_mfx.hip: a blocked Householder QR of up to 160 x 64 held in 16-row x 4-column
// tiles (mfw_strips.hpp's strip form) with ONE body for every panel -- pivot tile, live row tiles and live column blocks are
// wave-uniform scalars.  Internal, device code only (its own header so that tools/test_xstrip.hip can drive it alone).
#pragma once
#include <hip/hip_runtime.h>

#include "gls_device_math.hpp"
#include "mfw_strips.hpp"

namespace nin {
namespace mfxstrips {

using namespace glsmath;
using namespace mfwstrips;

constexpr int XQ = 10, XCB = 16;                     // row tiles, column blocks: 160 x 64
constexpr int XRP = 65;                              // pitch of R in LDS (odd: lane = row reads are conflict-free)

// One step K of a panel whose pivot rows are quad bp of tile q0 (a scalar): PT = the pivot tile's panel block, the panel blocks
// of the tiles q0 < q < nq are C[q][0] in place.  As strip_panel_step (mfw_strips.hpp), the tile loops behind wave-uniform branches.
template <int K>
__device__ __forceinline__ void xpanel_step(double (&C)[XQ][XCB], double &PT, double (&xm)[XQ], double (&vp)[4], double (&gk)[4],
                                            double (&Tr)[4], int q0, int nq, int bp, int si, int sb, int sj) {
    const bool in_piv_quad = sb == bp;
    const bool is_piv = in_piv_quad && si == K;                  // this lane's row of the pivot tile is the pivot row
    const bool below0 = sb > bp || (in_piv_quad && si > K);      // ... lies below it
    const double xm0 = below0 ? quad_pick<K>(PT) : 0.0;          // the reflector's entries below the pivot
    double acc = xm0 * PT;
#pragma unroll
    for (int q = 1; q < XQ; ++q) {
        if (q > q0 && q < nq) {
            xm[q] = quad_pick<K>(C[q][0]);
            acc = fma(xm[q], C[q][0], acc);                      // lane (.., j): sum over its rows of a[r][K] a[r][j]
        }
    }
    const double ap = __shfl(PT, 16 * K + 4 * bp + sj);          // the pivot row's entry of column j
    const double d = sum_rows(sum_quads(acc));
    const House h = house_unguarded(quad_pick<K>(ap), quad_pick<K>(d));
    const double e = fma(h.vp, ap, d);                           // v_K . (column j)
    const double w = sj > K ? -(h.g * e) : 0.0;
    gk[K] = h.g;
    vp[K] = h.vp;
    {
        double t = 0.0;
        if (K >= 1) t = Tr[0] * quad_pick<0>(e);
        if (K >= 2) t = fma(Tr[1], quad_pick<1>(e), t);
        if (K >= 3) t = fma(Tr[2], quad_pick<2>(e), t);
        Tr[K] = si == K ? h.g : -(h.g * t);
    }
    {
        double x = fma(w, is_piv ? h.vp : xm0, PT);
        PT = (is_piv && sj == K) ? h.beta : x;                   // R(rp, rp)
    }
#pragma unroll
    for (int q = 1; q < XQ; ++q) {
        if (q > q0 && q < nq) C[q][0] = fma(w, xm[q], C[q][0]);
    }
}

// The dense factorisation of an nrows x (nc + 1) problem held in the tiles (c at column nc; rows and columns beyond: zeros).  On
// exit R's rows 0 .. nc - 1 (columns up to nc = Q^T c) are in LDS at Rm[row * XRP + col]; returns r . r = |(Q^T c)(nc:)|^2.
#ifdef NIN_MFX_STAMPS   // diagnostic build (tools/stamps_mfx.py): s_memtime between the pieces of a panel, summed over the panels
struct XStamps { unsigned long long last, acc[6]; bool on; };
#define NIN_XSUB(ST, J) do { if ((ST).on) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); (ST).acc[J] += t_ - (ST).last; (ST).last = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
struct XStamps { };
#define NIN_XSUB(ST, J) do { } while (0)
#endif
__device__ __forceinline__ double xstrip_factor(double (&C)[XQ][XCB], int nc, int nrows, int lane, double *Rm, XStamps &ST) {
    const int si = lane >> 4, sb = (lane >> 2) & 3, sj = lane & 3, rowbase = 4 * sb + si;
    const double eye = si == sj ? 1.0 : 0.0;                     // the 4 x 4 identity in every quad
    const int n_panels = (nc + 3) >> 2, nq = (nrows + 15) >> 4, ncb = (nc + 4) >> 2;
    for (int p = 0; p < n_panels; ++p) {
        const int q0 = p >> 2, bp = p & 3, steps = nc - 4 * p < 4 ? nc - 4 * p : 4;
        const int nt = ncb - 1 - p;                              // live column blocks right of the panel: 1 .. nt
        double V[XQ], xm[XQ], gk[4] = {0.0, 0.0, 0.0, 0.0}, vp[4] = {0.0, 0.0, 0.0, 0.0};
        double Tr[4] = {0.0, 0.0, 0.0, 0.0};                     // this lane's row si of T
        double PT = 0.0;
#pragma unroll
        for (int q = 0; q < XQ; ++q) PT = q == q0 ? C[q][0] : PT;
        xpanel_step<0>(C, PT, xm, vp, gk, Tr, q0, nq, bp, si, sb, sj);
        if (steps > 1) xpanel_step<1>(C, PT, xm, vp, gk, Tr, q0, nq, bp, si, sb, sj);
        if (steps > 2) xpanel_step<2>(C, PT, xm, vp, gk, Tr, q0, nq, bp, si, sb, sj);
        if (steps > 3) xpanel_step<3>(C, PT, xm, vp, gk, Tr, q0, nq, bp, si, sb, sj);
        // V: below the pivots the panel's columns ARE the reflectors; in the pivot quad the diagonal takes v's pivot entries,
        // everything above it (R) and every row above the quad (earlier panels' rows of R) is zero
        double v0 = PT;
        {
            const double vdiag = sj == 0 ? vp[0] : sj == 1 ? vp[1] : sj == 2 ? vp[2] : vp[3];
            v0 = (sb == bp && si == sj) ? vdiag : v0;
            v0 = (sb < bp || (sb == bp && si < sj)) ? 0.0 : v0;
            v0 = sj < steps ? v0 : 0.0;
        }
#pragma unroll
        for (int q = 0; q < XQ; ++q) V[q] = q == q0 ? v0 : (sj < steps ? C[q][0] : 0.0);
        // rows 4 p .. 4 p + steps - 1 of R are final after this panel: the pivot tile, quad bp.  The panel block's own entries now:
        const bool r_rows = sb == bp && si < steps;
        const int col0 = 4 * p + sj;
        double *dst = Rm + (4 * p + si) * XRP + col0;
        NIN_XSUB(ST, 0);   // panel factored
        if (r_rows && sj >= si && col0 <= nc) dst[0] = PT;
        // (a panel with fewer than four pivots is the last one: c sits in its block, nothing lies to the right of it; the rows of
        //  the pivot tile below its pivots count in r . r: the tile goes back to its place)
        if (steps < 4) {
#pragma unroll
            for (int q = 0; q < XQ; ++q) C[q][0] = q == q0 ? PT : C[q][0];
        }
        if (steps == 4 && nt > 0) {
            const double Ts = -(sj == 0 ? Tr[0] : sj == 1 ? Tr[1] : sj == 2 ? Tr[2] : Tr[3]);   // -T[k][i] at lane (k = si, ., i = sj)
            double W[XCB];
#pragma unroll
            for (int cb = 1; cb < XCB; ++cb) W[cb] = 0.0;
            NIN_XSUB(ST, 1);   // V, T, first rows of R
            // W[cb] = V^T C[.][cb]: four independent accumulation chains per group of column blocks
#pragma unroll
            for (int q = 0; q < XQ; ++q) {
                if (q >= q0 && q < nq) {
#pragma unroll
                    for (int g4 = 0; g4 < (XCB + 2) / 4; ++g4) {
                        if (4 * g4 + 1 <= nt) {
#pragma unroll
                            for (int cb = 4 * g4 + 1; cb < 4 * g4 + 5 && cb < XCB; ++cb) W[cb] = mfma4(V[q], C[q][cb], W[cb]);
                        }
                    }
                }
            }
            NIN_XSUB(ST, 2);   // W = V^T C
#pragma unroll
            for (int g4 = 0; g4 < (XCB + 2) / 4; ++g4) {
                if (4 * g4 + 1 <= nt) {
#pragma unroll
                    for (int cb = 4 * g4 + 1; cb < 4 * g4 + 5 && cb < XCB; ++cb) W[cb] = mfma4(Ts, sum_quads(W[cb]), 0.0);   // -(T^T W), the same in every quad
                }
            }
            NIN_XSUB(ST, 3);   // T^T W
            // C -= V W', written ONE BLOCK DOWN: the next panel (or c) lands in block 0 without a move.  Blocks beyond nt are
            // never read again (their contents are whatever the group's last sweep left there)
#pragma unroll
            for (int q = 0; q < XQ; ++q) {
                if (q >= q0 && q < nq) {
                    const double VT = mfma4(V[q], eye, 0.0);     // V^T per quad
#pragma unroll
                    for (int g4 = 0; g4 < (XCB + 2) / 4; ++g4) {
                        if (4 * g4 + 1 <= nt) {
#pragma unroll
                            for (int cb = 4 * g4 + 1; cb < 4 * g4 + 5 && cb < XCB; ++cb) C[q][cb - 1] = mfma4(VT, W[cb], C[q][cb]);
                        }
                    }
                    if (q == q0) {                               // the panel's rows of R, right of the panel block (a block that
                        if (r_rows) {                            //  holds column nc is stored whole: the pitch has room for it)
#pragma unroll
                            for (int cb = 1; cb < XCB; ++cb) {
                                if (cb <= nt) dst[4 * cb] = C[q][cb - 1];
                            }
                        }
                    }
                }
            }
            NIN_XSUB(ST, 4);   // update, rows of R stored
        }
    }
    // c sits in block 0, column nc & 3; r . r over the rows that never were pivot rows
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
        if (q < nq) {
            const double x = (16 * q + rowbase >= nc && sj == (nc & 3)) ? C[q][0] : 0.0;
            t = fma(x, x, t);
        }
    }
    return wave_allsum(t);
}

}  // namespace mfxstrips
}  // namespace nin
