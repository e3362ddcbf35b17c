// kernels_gls_quad4.hip -- GLS weights of "quad" nodes (every node inside a boundary face of a hexahedron mesh), gfx950:
// TWO lanes per node, 32 nodes per wavefront -- the cube-node kernel's scheme (kernels_gls_hex8mf.hip) on half a cube.
//
// Such a node has 4 cells, 4 internal and 4 boundary faces, and it is computed only when the variable flags it Neumann
// (gls.pyx:165-166) -- but then a whole boundary plane of them is: 47 k nodes at 216^3, which the one-wavefront kernel for small
// nodes (kernels_gls_mfw.hip, one node per wavefront, 20 of its 64 lanes busy) served at 6.4 ns a node, 0.3 ms per Neumann
// plane against 4.7 ms for the 9.9 M interior nodes.  The system (gls.pyx:252-416): a cell row per cell, three rows per internal
// face [-B_a | +B_b], one Neumann row -(K N) per boundary face on its cell's columns -- 20 x 12 (+ the column c).  The cells
// form a 4-cycle (quad4_desc.hpp): two even cells that share no face, two odd ones, each even cell adjacent to BOTH odd cells.
//   phase 1  lane e eliminates E_e in its front: cell row, 3 + 3 rows of its two internal faces, its Neumann row -- 8 rows x
//            (3 own | 3 + 3 odd | c); three reflectors, the columns through the reflectors' mutual products as in the
//            cube-node kernel; 3 rows of R folded into z, u, s, 5 fill rows left;
//   phase 2  14 x 6 over the pair: per lane its 5 fill rows, the cell row and the Neumann row of O_e; row-distributed
//            Householder, one DPP stage per reduction, pivot rows dealt round-robin;
//   then     back-substitution, r_i = 1 - d_i . y_i, weights r_i / (r . r); neumann_ws = the LAST cell's weight (gls.pyx:470-472).
// Dirichlet nodes of the list (gls.pyx:165-166) get their zero row; a pass without a computed node does nothing else.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device_grid.hpp"
#include "gls_device_math.hpp"
#include "launch.hpp"
#include "quad4_desc.hpp"

namespace nin {

namespace {

using namespace glsmath;

constexpr int QNPW = 32;         // nodes per wavefront pass (2 lanes each)
constexpr int QR = 7, QC = 7;    // phase 2, per lane: 7 rows x (6 odd columns + c)

__device__ __forceinline__ double pair_sum(double v) { return v + dpp_mov<0xB1>(v); }   // quad_perm [1,0,3,2]
template <int L>
__device__ __forceinline__ double pair_bcast(double v) { return dpp_mov<L | (L << 2) | ((2 + L) << 4) | ((2 + L) << 6)>(v); }

// The panel of an 8-row front (rows 0 = cell row, 1-3 / 4-6 = the two internal faces, 7 = the Neumann row): three Householder
// steps on the own columns; v_k stays in P[k..7][k] (pivot entries included); g[k] the reflectors' scalars; z = R_ee^-T d
__device__ __forceinline__ void quad_panel(double (&P)[8][3], const double (&d)[3], double (&g)[3], double (&z)[3]) {
    double rinv[3];
    {
        double ss = 0.0;
#pragma unroll
        for (int r = 1; r < 8; ++r) ss = fma(P[r][0], P[r][0], ss);
        const House h = house_unguarded(P[0][0], ss);
        g[0] = h.g; rinv[0] = h.rinv;
        double d1 = h.vp * P[0][1], d2 = h.vp * P[0][2];
#pragma unroll
        for (int r = 1; r < 8; ++r) { d1 = fma(P[r][0], P[r][1], d1); d2 = fma(P[r][0], P[r][2], d2); }
        const double w1 = -(h.g * d1), w2 = -(h.g * d2);
        P[0][0] = h.vp;
#pragma unroll
        for (int r = 0; r < 8; ++r) { P[r][1] = fma(w1, P[r][0], P[r][1]); P[r][2] = fma(w2, P[r][0], P[r][2]); }
    }
    {
        double ss = 0.0;
#pragma unroll
        for (int r = 2; r < 8; ++r) ss = fma(P[r][1], P[r][1], ss);
        const House h = house_unguarded(P[1][1], ss);
        g[1] = h.g; rinv[1] = h.rinv;
        double d2 = h.vp * P[1][2];
#pragma unroll
        for (int r = 2; r < 8; ++r) d2 = fma(P[r][1], P[r][2], d2);
        const double w2 = -(h.g * d2);
        P[1][1] = h.vp;
#pragma unroll
        for (int r = 1; r < 8; ++r) P[r][2] = fma(w2, P[r][1], P[r][2]);
    }
    {
        double ss = 0.0;
#pragma unroll
        for (int r = 3; r < 8; ++r) ss = fma(P[r][2], P[r][2], ss);
        const House h = house_unguarded(P[2][2], ss);
        g[2] = h.g; rinv[2] = h.rinv;
        P[2][2] = h.vp;
    }
    z[0] = d[0] * rinv[0];
    z[1] = fma(-P[0][1], z[0], d[1]) * rinv[1];
    z[2] = fma(-P[1][2], z[1], fma(-P[0][2], z[0], d[2])) * rinv[2];
}

// The three reflectors on a column that enters with one face's three entries (rows R0 .. R0 + 2; R0 < 0: the column c = e_0)
// and zeros elsewhere, through the reflectors' mutual products cc = (v1 . v0, v2 . v0, v2 . v1) -- w2_column of
// kernels_gls_hex8mf.hip on 8 rows.  Out: u = z^T (rows 0 .. 2), the five fill entries (rows 3 .. 7).
template <int R0>
__device__ __forceinline__ void quad_column(const double (&P)[8][3], const double (&g3)[3], const double (&cc)[3], const double (&z)[3],
                                            const double (&b)[3], double &u_out, double (&fill)[5]) {
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
    double B[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const bool own = R0 >= 0 && r >= R0 && r < R0 + 3;
        double v = own ? fma(w0, P[r][0], b[own ? r - R0 : 0]) : (R0 < 0 && r == 0) ? fma(w0, P[0][0], 1.0) : w0 * P[r][0];
        if (r >= 1) v = fma(w1, P[r][1], v);
        if (r >= 2) v = fma(w2, P[r][2], v);
        B[r] = v;
    }
    u_out = fma(z[2], B[2], fma(z[1], B[1], z[0] * B[0]));
#pragma unroll
    for (int r = 0; r < 5; ++r) fill[r] = B[3 + r];
}

// Phase 2, step K: pivot = local row Q = K / 2 of lane LAM = K % 2 (p2_step_lean of kernels_gls_hex8mf.hip on a pair)
template <int K>
__device__ __forceinline__ void quad_p2_step(double (&C)[QR][QC], double (&rinvq)[3], int e) {
    constexpr int Q = K / 2, LAM = K % 2;
    const bool is_piv = (e == LAM);
    const double xq = (e > LAM) ? C[Q][K] : 0.0;
    double ss = xq * xq;
#pragma unroll
    for (int r = Q + 1; r < QR; ++r) ss = fma(C[r][K], C[r][K], ss);
    ss = pair_sum(ss);
    const double alpha = pair_bcast<LAM>(C[Q][K]);
    const House h = house_unguarded(alpha, ss);
    rinvq[Q] = is_piv ? h.rinv : rinvq[Q];
    const double vq = is_piv ? h.vp : xq;
#pragma unroll
    for (int j = K + 1; j < QC; ++j) {
        double a = vq * C[Q][j];
#pragma unroll
        for (int r = Q + 1; r < QR; ++r) a = fma(C[r][K], C[r][j], a);
        const double w = -(h.g * pair_sum(a));
        C[Q][j] = fma(w, vq, C[Q][j]);
#pragma unroll
        for (int r = Q + 1; r < QR; ++r) C[r][j] = fma(w, C[r][K], C[r][j]);
    }
}
template <int K, int KEND>
struct QuadP2 {
    static __device__ __forceinline__ void run(double (&C)[QR][QC], double (&rinvq)[3], int e) {
        quad_p2_step<K>(C, rinvq, e);
        QuadP2<K + 1, KEND>::run(C, rinvq, e);
    }
};
template <int KEND>
struct QuadP2<KEND, KEND> {
    static __device__ __forceinline__ void run(double (&)[QR][QC], double (&)[3], int) {}
};
// back-substitution by columns: row K of R lives in lane K % 2, local row K / 2 (entries C[Q][j], j > K; right-hand side t[Q])
template <int K>
struct QuadBack {
    static __device__ __forceinline__ void run(const double (&C)[QR][QC], const double (&rinvq)[3], double (&t)[3], double (&y)[6], int e) {
        constexpr int Q = K / 2, LAM = K % 2;
        const double yk = pair_bcast<LAM>(t[Q] * rinvq[Q]);
        y[K] = yk;
#pragma unroll
        for (int q = 0; q < Q; ++q) t[q] = fma(-C[q][K], yk, t[q]);
        if (LAM > 0) t[Q] = fma((e < LAM) ? -C[Q][K] : 0.0, yk, t[Q]);
        QuadBack<K - 1>::run(C, rinvq, t, y, e);
    }
};
template <>
struct QuadBack<-1> {
    static __device__ __forceinline__ void run(const double (&)[QR][QC], const double (&)[3], double (&)[3], double (&)[6], int) {}
};

__global__ __launch_bounds__(256) void nin_gls_quad4_kernel(GridView g, const int32_t *__restrict__ nodes,
                                                            const int32_t *__restrict__ desc, int32_t count, int add_neumann,
                                                            double *__restrict__ out, double *__restrict__ nws) {
    __shared__ double wbuf_all[4][QNPW * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = lane & 1, nd = lane >> 1;
    double *const wbuf = wbuf_all[wave];
    const int32_t n_groups = (count + QNPW - 1) / QNPW;
    for (int32_t grp = (int32_t)blockIdx.x * 4 + wave; grp < n_groups; grp += (int32_t)gridDim.x * 4) {
        const int32_t idx = grp * QNPW + nd;
        const bool valid = idx < count;
        const uint32_t sel = (uint32_t)(valid ? idx : count - 1);   // past the end: a clamped (valid) entry, never stored
        const uint32_t p = (uint32_t)nodes[sel], dsc = (uint32_t)desc[2 * (size_t)sel + e];
        const uint32_t fl = g.flags[p];
        const bool is_neu = (fl & 2) != 0;
        const bool computed = !((fl & 1) && !is_neu);              // Dirichlet boundary node: the zero row (gls.pyx:165-166)
        const uint32_t eb = (uint32_t)g.esup_ptr[p], fb = (uint32_t)g.fsup_ptr[p];
        if (!__any(valid && computed)) {                            // (wave-uniform) nothing to compute in this pass
            if (valid) {
                out[eb + 2 * e] = 0.0;
                out[eb + 2 * e + 1] = 0.0;
                if (e == 0) nws[p] = 0.0;
            }
            continue;
        }
        double xv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) xv[k] = g.coords[3 * (size_t)p + k];
        const uint32_t ce = (uint32_t)g.esup[eb + (dsc & 3)], co = (uint32_t)g.esup[eb + ((dsc >> 2) & 3)];
        const uint32_t cn[2] = {(uint32_t)g.esup[eb + ((dsc >> 18) & 3)], (uint32_t)g.esup[eb + ((dsc >> 20) & 3)]};
        const uint32_t fbe = (uint32_t)g.fsup[fb + ((dsc >> 12) & 7)], fbo = (uint32_t)g.fsup[fb + ((dsc >> 15) & 7)];

        // ---- the front of E_e: rows 0 = cell row, 1 + 3 i + r = row r of face i (A: towards O0, B: towards O1), 7 = its Neumann row
        double P[8][3], de[3], dod[3], nb[2][3][3];   // nb[i][r][t]: row r of face i on the neighbour's columns
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
        for (int i = 0; i < 2; ++i) {
            const uint32_t w = dsc >> (4 + 4 * i);
            const uint32_t f = (uint32_t)g.fsup[fb + (w & 7)];
            // B = [K N; T1; tau T2] (gls.pyx:293-321), row = [-B_a | +B_b] (gls.pyx:340-356)
            const double N0 = (double)g.face_normal[3 * (size_t)f + 0], N1 = (double)g.face_normal[3 * (size_t)f + 1],
                         N2 = (double)g.face_normal[3 * (size_t)f + 2];
            const double T0 = xv[0] - g.face_center[3 * (size_t)f + 0], T1 = xv[1] - g.face_center[3 * (size_t)f + 1],
                         T2 = xv[2] - g.face_center[3 * (size_t)f + 2];
            const double U0 = N1 * T2 - N2 * T1, U1 = N2 * T0 - N0 * T2, U2 = N0 * T1 - N1 * T0;
            const double dmn = g.diff_mag[cn[i]];
            double eta = 0.0;
            eta = dme > eta ? dme : eta;
            eta = dmn > eta ? dmn : eta;
            const double tj = face_tau_sq(U0 * U0 + U1 * U1 + U2 * U2, eta);
            const double sg = ((w >> 3) & 1) ? -1.0 : 1.0;
            const double *Kn = g.perm + 9 * (size_t)cn[i];
            const double Tv[3] = {sg * T0, sg * T1, sg * T2}, Uv[3] = {sg * (tj * U0), sg * (tj * U1), sg * (tj * U2)};
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                P[1 + 3 * i][t] = sg * (Ke[t * 3 + 0] * N0 + Ke[t * 3 + 1] * N1 + Ke[t * 3 + 2] * N2);
                P[2 + 3 * i][t] = Tv[t];
                P[3 + 3 * i][t] = Uv[t];
                nb[i][0][t] = -sg * (Kn[t * 3 + 0] * N0 + Kn[t * 3 + 1] * N1 + Kn[t * 3 + 2] * N2);
                nb[i][1][t] = -Tv[t];
                nb[i][2][t] = -Uv[t];
            }
        }
        const double mneu = is_neu ? 1.0 : 0.0;                  // (the Neumann rows exist for flagged nodes only, gls.pyx:374-416)
        {
            const double N0 = (double)g.face_normal[3 * (size_t)fbe + 0], N1 = (double)g.face_normal[3 * (size_t)fbe + 1],
                         N2 = (double)g.face_normal[3 * (size_t)fbe + 2];
#pragma unroll
            for (int t = 0; t < 3; ++t) P[7][t] = -mneu * (Ke[t * 3 + 0] * N0 + Ke[t * 3 + 1] * N1 + Ke[t * 3 + 2] * N2);   // gls.pyx:394-416
        }
        double C[QR][QC], u[6], se;
        {
            double g3[3], z[3], cc[3];
            quad_panel(P, de, g3, z);
            cc[0] = P[1][1] * P[1][0];
            cc[1] = P[2][2] * P[2][0];
            cc[2] = P[2][2] * P[2][1];
#pragma unroll
            for (int r = 2; r < 8; ++r) cc[0] = fma(P[r][1], P[r][0], cc[0]);
#pragma unroll
            for (int r = 3; r < 8; ++r) { cc[1] = fma(P[r][2], P[r][0], cc[1]); cc[2] = fma(P[r][2], P[r][1], cc[2]); }
            {
                const double none[3] = {0.0, 0.0, 0.0};
                double fill[5];
                quad_column<-1>(P, g3, cc, z, none, se, fill);   // c = e_0 on entry
#pragma unroll
                for (int r = 0; r < 5; ++r) C[r][6] = fill[r];
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {                        // odd slot 0 <- face A: rows 1 .. 3
                const double b[3] = {nb[0][0][t], nb[0][1][t], nb[0][2][t]};
                double fill[5];
                quad_column<1>(P, g3, cc, z, b, u[t], fill);
#pragma unroll
                for (int r = 0; r < 5; ++r) C[r][t] = fill[r];
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {                        // odd slot 1 <- face B: rows 4 .. 6
                const double b[3] = {nb[1][0][t], nb[1][1][t], nb[1][2][t]};
                double fill[5];
                quad_column<4>(P, g3, cc, z, b, u[3 + t], fill);
#pragma unroll
                for (int r = 0; r < 5; ++r) C[r][3 + t] = fill[r];
            }
        }
        // rows 5, 6: the cell row of O_e (c = 1) and its Neumann row (c = 0), on the columns of odd slot e
        {
            const double *Ko = g.perm + 9 * (size_t)co;
            const double N0 = (double)g.face_normal[3 * (size_t)fbo + 0], N1 = (double)g.face_normal[3 * (size_t)fbo + 1],
                         N2 = (double)g.face_normal[3 * (size_t)fbo + 2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const double kn = -mneu * (Ko[t * 3 + 0] * N0 + Ko[t * 3 + 1] * N1 + Ko[t * 3 + 2] * N2);
                    C[5][3 * s + t] = (e == s) ? dod[t] : 0.0;
                    C[6][3 * s + t] = (e == s) ? kn : 0.0;
                }
            }
            C[5][6] = 1.0;
            C[6][6] = 0.0;
        }
        // ---- phase 2: 14 x 6 over the pair ---------------------------------------------------------------------------------
        double rinvq[3] = {0.0, 0.0, 0.0};
        QuadP2<0, 6>::run(C, rinvq, e);
        double y[6], t3[3] = {C[0][6], C[1][6], C[2][6]};
        QuadBack<5>::run(C, rinvq, t3, y, e);
        double tail = 0.0;
#pragma unroll
        for (int r = 3; r < QR; ++r) tail = fma(C[r][6], C[r][6], tail);
        const double rr = pair_sum(tail);                        // r . r = |(Q^T c)(12:20)|^2
        double re = 1.0 - se;                                    // r_e = 1 - d_e . y_e = 1 - z . b_e + u . y_odd
#pragma unroll
        for (int j = 0; j < 6; ++j) re = fma(u[j], y[j], re);
        const double d0 = fma(dod[2], y[2], fma(dod[1], y[1], dod[0] * y[0])), d1 = fma(dod[2], y[5], fma(dod[1], y[4], dod[0] * y[3]));
        const double ro = 1.0 - (e == 0 ? d0 : d1);
        const double rri = fast_rcp(rr);
        double we = re * rri, wo = ro * rri;
        const bool ok = computed && rr > 0.0;                    // (rank-deficient system or NaN from a zero column: the zero row)
        we = (ok && __builtin_isfinite(we)) ? we : 0.0;
        wo = (ok && __builtin_isfinite(wo)) ? wo : 0.0;
        wbuf[nd * 4 + (dsc & 3)] = we;
        wbuf[nd * 4 + ((dsc >> 2) & 3)] = wo;
        wave_lds_sync();
        const double nwv = (computed && is_neu) ? wbuf[nd * 4 + 3] : 0.0;   // gls.pyx:470-472: the LAST cell's weight
        const double addv = add_neumann ? nwv : 0.0;
        const double o0 = wbuf[nd * 4 + 2 * e] + addv, o1 = wbuf[nd * 4 + 2 * e + 1] + addv;
        if (valid) {
            out[eb + 2 * e] = o0;
            out[eb + 2 * e + 1] = o1;
            if (e == 0) nws[p] = nwv;
        }
        wave_lds_sync();
    }
}

__global__ void k_quad4_desc(GridView g, const int32_t *__restrict__ nodes, int32_t count, int32_t *__restrict__ desc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    int32_t d[2] = {0, 0};
    (void)quad4_descriptor(g, nodes ? nodes[i] : (int32_t)i, d);   // the list holds classified quad nodes only
    desc[2 * i] = d[0];
    desc[2 * i + 1] = d[1];
}

}  // namespace

int launch_quad4_desc(const GridView &g, const int32_t *nodes, int32_t count, int32_t *desc, hipStream_t stream) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(k_quad4_desc, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, g, nodes, count, desc);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_gls_quad4(const GridView &g, const int32_t *nodes, const int32_t *desc, int32_t count, int add_neumann, double *out,
                     double *nws, hipStream_t stream) {
    if (count <= 0) return 0;
    int64_t blocks = ((int64_t)count + 4 * QNPW - 1) / (4 * QNPW);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(nin_gls_quad4_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, g, nodes, desc, count, add_neumann, out, nws);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace nin
