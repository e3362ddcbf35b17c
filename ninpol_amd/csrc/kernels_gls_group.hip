// kernels_gls_group.hip -- GLS weights for homogeneous small-node classes, gfx950: a GROUP of 16 lanes
// per node (4 nodes per wavefront), the whole least-squares system in registers.
//
// Why a kernel of its own.  The first GLS kernel (kernels_gls.hip) maps one node to one wavefront with matrix ROWS
// across the lanes; every (reflector, column) pair then costs a full cross-lane reduction and the matrix lives in
// LDS.  Measured on the 10 M-cell hexahedron mesh (profiles/r01/v1_*): 261 ms, ~60 k issue cycles per
// node for ~5 k useful FP64 FMAs.  (The general kernel of today, kernels_gls_block.hip, keeps the matrix in LDS too.)  GLS is FP64-ALU-bound, so what matters is FMAs per
// issued instruction.  Here the M x NA matrix of a node is dealt out in 2-D over its 16 lanes:
//   columns  j -> lane l8 = j % 8, register slot q = j / 8          (3 slots for 24 columns)
//   rows     r -> half  h  = r % 2, local row     rl = r / 2        (22 local rows for 44 rows)
// so a lane holds 3 x 22 doubles with compile-time indices (132 VGPRs; VALU can address 256, which is
// why the rows are split over two lanes instead of 44 rows in one).  A Householder step is then:
//   * the two owner lanes of column k publish it through LDS (from inside step k - 1, right after its update:
//     look-ahead, see publish()),
//   * every lane reads back the half it needs and does its part of the norm and of the dot products
//     with its own columns -- plain FMA chains -- and ONE DPP exchange with the partner half finishes
//     them; beta / tau are formed redundantly in every lane;
//   * every lane updates its own 3 x 22 entries -- plain FMAs again.
// Interleaving the rows (r % 2) keeps both halves equally long as the active rows shrink.  The last
// column c is dealt by rows over the 16 lanes (a 4th register slot in some lane would cost a whole
// column application per step for every lane of the wavefront), as is the right-hand side of the
// back-substitution; those are the only 16-lane reductions (4 DPP steps).
//
// Same arithmetic as the reference in the sense that matters for parity: Householder QR of the same
// m x (n-1) matrix (gls.pyx:252-356 assembles it, dgels factors it) -- never normal equations -- then,
// instead of carrying the n_elem unit right-hand sides, the identity  X[n-1, i] = r_i / (r.r) with
// r = c - A y and R y = (Q^T c)(0:n-1): only the cell rows r_i = 1 - d_i . y_i are needed.
// The reflector is used in the unnormalised form H = I - v v^T / (beta (beta - alpha)),
// v = (alpha - beta, x): the same H as LAPACK's dlarfg, one division per step instead of two.
//
// Eligible nodes (binned on the host, abi.hip): exactly 8 cells and 12 faces, all internal -- every
// interior node of a hexahedron mesh (M = 44, 24 + 1 columns).  Everything else runs in kernels_gls_block.hip.
// The launch is persistent and XCD-aware (one workgroup per CU, per-XCD work counters): see the kernel body.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device_grid.hpp"
#include "launch.hpp"

namespace nin {

namespace {

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// value held by the partner lane (same column lane l8, other row half): lane i <-> i ^ 8 in its row of 16
__device__ __forceinline__ double partner(double v) { return dpp_mov<0x128>(v); }  // row_ror:8

// sum over the 16 lanes of a node group, result in all 16
__device__ __forceinline__ double group16_sum(double v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += partner(v);
    return v;
}

// LDS pointers are kept in their own address space and laundered once per node: a DS instruction takes
// base VGPR + immediate offset, but only small immediates fit the paired forms (ds_read2_b64: 255 x 8 B),
// and when the compiler sees the whole "node base + array offset + row" sum it hoists one base VGPR per
// PAIR of rows out of the node loop (measured: ~100 VGPRs of addresses, spilled).  A laundered pointer is
// an opaque base, so every access below is that base + a small compile-time offset.
typedef __attribute__((address_space(3))) double lds_f64;
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f64x2 lds_f64x2;
__device__ __forceinline__ lds_f64 *lds_base(double *p) {
    lds_f64 *q = (lds_f64 *)p;
    asm volatile("" : "+v"(q));
    return q;
}

// 1/d and 1/sqrt(s) from the hardware seeds (v_rcp_f64 / v_rsq_f64, ~2^-26) plus two Newton steps: full
// double precision to a few ulp, a third of the instructions and of the dependent latency of the IEEE
// division / sqrt expansions.  They only feed the reflector scalars, where an ulp-level error is an
// ulp-level departure of H from orthogonality (parity bar: 1e-10).
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double s) {
    double y = __builtin_amdgcn_rsq(s);
    double e = fma(-s * y, y, 1.0);            // 1 - s y^2
    y = fma(y * e, fma(e, 0.375, 0.5), y);     // y (1 + e/2 + 3 e^2/8)
    e = fma(-s * y, y, 1.0);
    y = fma(y * e, 0.5, y);
    return y;
}

__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// tau = |T_sj2|^(-eta) (gls.pyx:314); for a positive base pow(u, -eta) = exp(-eta log u).  The library exp / log are
// ~250 instructions per face record (3.5 % of the kernel); this pair is ~45: log via u = 2^e m, m in [sqrt(1/2),
// sqrt(2)), 2 atanh((m - 1) / (m + 1)) as an 11-term odd series (|s| <= 0.172: truncation 1e-17), exp via
// y = k ln2 + r, |r| <= 0.347, Taylor to r^14 (4e-18) and ldexp.  Against numpy's pow over u in [1e-6, 1e2],
// eta in (0, 1]: max relative error 1.8e-15, mean 1.6e-16 -- five orders below the 1e-10 weight tolerance.
// Kept out of line: inlined, the series coefficients are hoisted out of the node loop and stay live across the
// whole QR, which pushes the kernel into scratch.
__device__ __attribute__((noinline)) double face_tau(double un, double eta) {
    if (eta == 0.0) return 1.0;
    constexpr double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    double m = __builtin_amdgcn_frexp_mant(un);            // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(un);
    const bool low = m < 0.70710678118654752440;
    m = low ? 2.0 * m : m;
    e = low ? e - 1 : e;
    const double ef = (double)e;
    const double sden = m + 1.0;
    double r = __builtin_amdgcn_rcp(sden);                 // (m - 1) / (m + 1) with two Newton steps on the reciprocal
    r = fma(fma(-sden, r, 1.0), r, r);
    r = fma(fma(-sden, r, 1.0), r, r);
    const double sn = (m - 1.0) * r;
    const double z = sn * sn;
    double p = 1.0 / 23.0;
    p = fma(p, z, 1.0 / 21.0); p = fma(p, z, 1.0 / 19.0); p = fma(p, z, 1.0 / 17.0); p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0); p = fma(p, z, 1.0 / 11.0); p = fma(p, z, 1.0 / 9.0);  p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);  p = fma(p, z, 1.0 / 3.0);  p = fma(p, z, 1.0);
    const double lg = fma(ef, LN2_HI, fma(ef, LN2_LO, 2.0 * sn * p));   // log(un)
    const double y = -eta * lg;
    const double k = rint(y * 1.44269504088896340736);
    const double rr = fma(-k, LN2_LO, fma(-k, LN2_HI, y));
    double q = 1.0 / 87178291200.0;                        // 1 / 14!
    q = fma(q, rr, 1.0 / 6227020800.0); q = fma(q, rr, 1.0 / 479001600.0); q = fma(q, rr, 1.0 / 39916800.0);
    q = fma(q, rr, 1.0 / 3628800.0);    q = fma(q, rr, 1.0 / 362880.0);    q = fma(q, rr, 1.0 / 40320.0);
    q = fma(q, rr, 1.0 / 5040.0);       q = fma(q, rr, 1.0 / 720.0);       q = fma(q, rr, 1.0 / 120.0);
    q = fma(q, rr, 1.0 / 24.0);         q = fma(q, rr, 1.0 / 6.0);         q = fma(q, rr, 0.5);
    q = fma(q, rr, 1.0);                q = fma(q, rr, 1.0);
    return __builtin_amdgcn_ldexp(q, (int)k);
}

struct Hex8 {
    static constexpr int NE = 8, NIF = 12;
    static constexpr int M = NE + 3 * NIF;   // 44 rows
    static constexpr int NA = 3 * NE;        // 24 columns of A (+ the column c)
    static constexpr int SLOTS = NA / 8;     // 3 columns per lane
    static constexpr int HR = M / 2;         // 22 local rows per half
    static constexpr int CS = (M + 15) / 16; // 3 rows of c per lane
    static constexpr int STAGE = NA * M;     // staging buffer: the whole M x NA system, column-major (one wave per SIMD
                                             // leaves the LDS room: 4 waves x 4 nodes x 9.2 KB = 147 KB of the CU's 160)
    static constexpr int HRP = 24;           // padded half length of the published column (rows up to 47 exist, zero)
    static constexpr int XB = 2 * HRP + 2;   // published column by half + the pivot entry alpha
    static constexpr int NODE_DOUBLES = STAGE + XB + NE / 2 + NA;
    static constexpr int LANES = 16, NODES_PER_WAVE = 4;
};

// Everything a pass reads from HBM, fetched one pass ahead.  The reads form a dependent chain (node id ->
// CSR row starts -> cell / face ids -> geometry -> permeability); issued back to back at the top of a pass
// they cost ~12 k cycles of exposed latency (in-kernel stamps, tools/stamps_gls.py) of a ~57 k cycle pass, and
// with one wave per SIMD nothing else hides it.  So each level is issued at a different point of the
// PREVIOUS pass (top, after the faces, after the staging, Householder steps 6 and 12) and has long landed
// when the next level needs it.
struct NodeFetch {
    int32_t p, eb, fb, cell, ca, cb;
    size_t face;
    bool valid;
    double xv[3], cen[3], fcen[3], fn[3], da, db, Ka[9], Kb[9];

    __device__ __forceinline__ void level0(const int32_t *nodes, int32_t idx, int32_t count) {
        valid = idx < count;
        const int32_t sel = valid ? idx : count - 1;
        p = nodes ? nodes[sel] : sel;
    }
    __device__ __forceinline__ void level1(const GridView &g) {
        eb = g.esup_ptr[p];
        fb = g.fsup_ptr[p];
#pragma unroll
        for (int k = 0; k < 3; ++k) xv[k] = g.coords[3 * (size_t)p + k];
    }
    __device__ __forceinline__ void level2(const GridView &g, int l8, int sf) {
        cell = g.esup[eb + l8];
        face = (size_t)g.fsup[fb + sf];
    }
    __device__ __forceinline__ void level3(const GridView &g) {
        ca = g.face_cells[2 * face];
        cb = g.face_cells[2 * face + 1];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            cen[k] = g.centroids[3 * (size_t)cell + k];
            fcen[k] = g.face_center[3 * face + k];
            fn[k] = g.face_normal[3 * face + k];
        }
    }
    __device__ __forceinline__ void level4(const GridView &g) {
        da = g.diff_mag[ca];
        db = g.diff_mag[cb];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            Ka[k] = g.perm[9 * (size_t)ca + k];
            Kb[k] = g.perm[9 * (size_t)cb + k];
        }
    }
};

// One Householder step K (compile-time).  l8: column lane, h: row half, l16 = 8 h + l8.
//
// The published column lives in xb as two halves of HRP entries (row r -> half r % 2, slot r / 2).  The
// owner writes x = column K BELOW the pivot and ZERO at the pivot and at the row above it, so by
// induction every entry of the buffer at rows <= K is zero and readers need no masks: the pivot row drops
// out of the norm and of the dot products by itself.  The pivot entry alpha travels in its own slot.
// xh = xb + h * HRP (this lane's half), xc = xb + (l16 & 1) * HRP + (l16 >> 1) (rows l16 + 16 e): every LDS
// access is a laundered base + compile-time offset (see lds_base).
// Publish column K (owner lanes only): x = the column BELOW the pivot, zero at the pivot and at the row above it,
// alpha in its own slot.  LDS writes are slow to issue (~25 cycles each for one wave, tools/micro_isa.hip), so
// column K + 1 is published from INSIDE step K, right after its own update and interleaved with the update of the
// other columns: by the time step K + 1 reads it the data has long landed ("look-ahead" Householder).
template <int K>
__device__ __forceinline__ void publish(const double (&a)[Hex8::SLOTS][Hex8::HR], lds_f64 *xh, lds_f64 *xalpha, int l8, int h) {
    constexpr int QK = K / 8, LK = K % 8, HP = K % 2, PL = K / 2, HR = Hex8::HR;
    if (l8 == LK) {
        const bool pivot_half = (h == HP);
        // local row PL is the pivot row (pivot half), row K - 1 (K odd, other half) or row K + 1 (K even, other half)
        xh[PL] = (HP == 0 && !pivot_half) ? a[QK][PL] : 0.0;
#pragma unroll
        for (int rl = PL + 1; rl < HR; ++rl) xh[rl] = a[QK][rl];
        if (pivot_half) xalpha[0] = a[QK][PL];
    }
}

template <int K>
__device__ __forceinline__ void qr_step(double (&a)[Hex8::SLOTS][Hex8::HR], double (&cr)[Hex8::CS], double (&rinv)[Hex8::SLOTS],
                                        lds_f64 *xh, const lds_f64 *xc, lds_f64 *xalpha, int l8, int h, int l16,
                                        NodeFetch &nx, const GridView &g) {
    using C = Hex8;
    if (K == 6) nx.level3(g);    // next pass: geometry (its ids were fetched after this pass's staging)
    if (K == 12) nx.level4(g);   // next pass: permeability and diff_mag of the faces' cells
    constexpr int QK = K / 8, LK = K % 8, HP = K % 2, PL = K / 2, RL0 = (K + 1) / 2, HR = C::HR;
    constexpr int E0 = K / 16;                 // slots of c below E0 hold only rows above the pivot
    const bool owner = (l8 == LK);
    const bool pivot_half = (h == HP);
    lds_sync();                                // column K was published during step K - 1 (or before the loop)
    // pass 1 over this lane's half of the published column: |x|^2 and the dot products with the lane's own
    // columns.  x is read ONCE, as 16-byte ds_read_b128 where the pair is aligned, and kept in registers for pass 2.
    const double alpha = xalpha[0];
    double x[HR], xl[C::CS];
    {
        constexpr int RA = RL0 + (RL0 & 1);          // first even local row >= RL0
        if (RL0 & 1) x[RL0] = xh[RL0];
#pragma unroll
        for (int rl = RA; rl < HR; rl += 2) {
            const f64x2 v = *reinterpret_cast<const lds_f64x2 *>(xh + rl);
            x[rl] = v.x;
            x[rl + 1] = v.y;
        }
#pragma unroll
        for (int e = E0; e < C::CS; ++e) xl[e] = xc[8 * e];   // the rows of c this lane holds (rows l16 + 16 e)
    }
    double ss = 0.0, d[C::SLOTS];
#pragma unroll
    for (int q = 0; q < C::SLOTS; ++q) d[q] = 0.0;
#pragma unroll
    for (int rl = RL0; rl < HR; ++rl) {
        ss = fma(x[rl], x[rl], ss);
#pragma unroll
        for (int q = QK; q < C::SLOTS; ++q) d[q] = fma(x[rl], a[q][rl], d[q]);
    }
    ss += partner(ss);
    // beta = -sign(alpha) |(alpha, x)| (dlarfg);  H = I - g v v^T,  v = (alpha - beta, x),
    // g = 1 / (beta (beta - alpha)) = 1 / (S + |alpha| sqrt(S)),  S = alpha^2 + |x|^2;  1 / beta = g (beta - alpha)
    const bool live = ss != 0.0;
    const double S = fma(alpha, alpha, ss);
    const double rs = fast_rsqrt(S), sq = S * rs;                       // sqrt(S)
    const double beta = live ? -copysign(sq, alpha) : alpha;
    const double vk = alpha - beta;
    const double gden = fast_rcp(live ? fma(fabs(alpha), sq, S) : alpha);
    const double gk = live ? gden : 0.0;
    const double rinv_k = live ? -(gden * vk) : gden;                    // 1 / R(K,K), for the back-substitution
    if (owner) rinv[QK] = rinv_k;
    const double vkh = pivot_half ? vk : 0.0;                            // the pivot entry of v, in the half that holds row K
    double gw[C::SLOTS];
#pragma unroll
    for (int q = QK; q < C::SLOTS; ++q) {
        double w = fma(vkh, a[q][PL], d[q]);
        w += partner(w);
        w = -(gk * w);
        if (q == QK) w = (l8 > LK) ? w : 0.0;     // in slot QK only the columns to the right of K are updated
        gw[q] = w;
        a[q][PL] = fma(w, vkh, a[q][PL]);        // the pivot row (vkh = 0 elsewhere: no change)
    }
    // pass 2.  The slot that holds column K + 1 goes first and that column is published at once; the writes drain
    // while the other slots and the column c are updated.
    constexpr int QN = (K + 1 < C::NA) ? (K + 1) / 8 : QK;
#pragma unroll
    for (int rl = RL0; rl < HR; ++rl) a[QN][rl] = fma(gw[QN], x[rl], a[QN][rl]);
    if (K + 1 < C::NA) {
        lds_sync();                              // every lane has read column K (x, xl, alpha) by now
        __builtin_amdgcn_sched_barrier(0);       // keep the writes HERE: the scheduler otherwise sinks them below the
        publish<(K + 1 < C::NA) ? K + 1 : K>(a, xh, xalpha, l8, h);   // remaining updates, back onto the critical path
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int rl = RL0; rl < HR; ++rl) {
#pragma unroll
        for (int q = QK; q < C::SLOTS; ++q)
            if (q != QN) a[q][rl] = fma(gw[q], x[rl], a[q][rl]);
    }
    // the last column c, dealt by rows: lane l16 holds rows l16, l16 + 16, l16 + 32 (rows 44..47 are padding: 0)
    {
        const double vkc = (l16 == K % 16) ? vk : 0.0;
        double part = vkc * cr[E0];
#pragma unroll
        for (int e = E0; e < C::CS; ++e) part = fma(xl[e], cr[e], part);
        const double gwc = -(gk * group16_sum(part));
        cr[E0] = fma(gwc, vkc, cr[E0]);
#pragma unroll
        for (int e = E0; e < C::CS; ++e) cr[e] = fma(gwc, xl[e], cr[e]);
    }
}

template <int K, int KEND>
struct QrLoop {
    static __device__ __forceinline__ void run(double (&a)[Hex8::SLOTS][Hex8::HR], double (&cr)[Hex8::CS],
                                               double (&rinv)[Hex8::SLOTS], lds_f64 *xh, const lds_f64 *xc, lds_f64 *xalpha,
                                               int l8, int h, int l16, NodeFetch &nx, const GridView &g) {
        qr_step<K>(a, cr, rinv, xh, xc, xalpha, l8, h, l16, nx, g);
        QrLoop<K + 1, KEND>::run(a, cr, rinv, xh, xc, xalpha, l8, h, l16, nx, g);
    }
};
template <int KEND>
struct QrLoop<KEND, KEND> {
    static __device__ __forceinline__ void run(double (&)[Hex8::SLOTS][Hex8::HR], double (&)[Hex8::CS], double (&)[Hex8::SLOTS],
                                               lds_f64 *, const lds_f64 *, lds_f64 *, int, int, int, NodeFetch &, const GridView &) {}
};

// Back-substitution R y = b, row J (compile-time).  R(J, j) sits in lane (j % 8, J % 2) at local row J / 2,
// slot j / 8; b_J in cr[J / 16] of lane J % 16; y_j is kept in both lanes of column lane j % 8.
template <int J>
__device__ __forceinline__ void back_step(const double (&a)[Hex8::SLOTS][Hex8::HR], const double (&cr)[Hex8::CS],
                                          const double (&rinv)[Hex8::SLOTS], double (&y)[Hex8::SLOTS], int l8, int h,
                                          int l16) {
    using C = Hex8;
    constexpr int QJ = J / 8, LJ = J % 8, HJ = J % 2, PJ = J / 2;
    double part = (l16 == J % 16) ? -cr[J / 16] : 0.0;
    if (h == HJ) {
#pragma unroll
        for (int q = QJ; q < C::SLOTS; ++q) {
            const bool act = (q > QJ) || (l8 > LJ);
            part = fma(act ? a[q][PJ] : 0.0, y[q], part);
        }
    }
    const double tot = group16_sum(part);       // sum_{j > J} R(J,j) y_j - b_J
    const double yj = -tot * rinv[QJ];          // meaningful in the two lanes of column lane LJ
    y[QJ] = (l8 == LJ) ? yj : y[QJ];
}

template <int J>
struct BackLoop {
    static __device__ __forceinline__ void run(const double (&a)[Hex8::SLOTS][Hex8::HR], const double (&cr)[Hex8::CS],
                                               const double (&rinv)[Hex8::SLOTS], double (&y)[Hex8::SLOTS], int l8, int h,
                                               int l16) {
        back_step<J>(a, cr, rinv, y, l8, h, l16);
        BackLoop<J - 1>::run(a, cr, rinv, y, l8, h, l16);
    }
};
template <>
struct BackLoop<-1> {
    static __device__ __forceinline__ void run(const double (&)[Hex8::SLOTS][Hex8::HR], const double (&)[Hex8::CS],
                                               const double (&)[Hex8::SLOTS], double (&)[Hex8::SLOTS], int, int, int) {}
};

template <int DBG>
__global__ __launch_bounds__(256, 1) void nin_gls_group_kernel(GridView g, const int32_t *__restrict__ nodes,
                                                               int32_t count, int add_neumann,
                                                               double *__restrict__ out, double *__restrict__ nws,
                                                               int32_t *__restrict__ queue) {
    using C = Hex8;
    constexpr int NE = C::NE, NIF = C::NIF, M = C::M, NA = C::NA, SLOTS = C::SLOTS, HR = C::HR;
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int l16 = lane & 15, l8 = lane & 7, h = (lane >> 3) & 1, grp = lane >> 4;
    constexpr int GPW = C::NODES_PER_WAVE;
    double *node_lds = smem + ((size_t)wave * GPW + grp) * C::NODE_DOUBLES;
    double *stage = node_lds;                              // [8][M] columns of the slot being assembled
    double *xb = stage + C::STAGE;                         // [2][HRP] published column by half, then alpha
    int32_t *cells = reinterpret_cast<int32_t *>(xb + C::XB);  // [NE]
    double *prod = xb + C::XB + NE / 2;                    // [NA]

    const int32_t n_groups = (count + GPW - 1) / GPW;
    unsigned long long stamps[8];
    int n_stamp = 0;
#define NIN_STAMP() do { if (DBG == 3 && n_stamp < 8) { __builtin_amdgcn_sched_barrier(0); stamps[n_stamp++] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } } while (0)
    const int sf = l16 < NIF ? l16 : NIF - 1;   // lane's face (lanes 12..15 redo face 11 and write nothing)
    // Which 4-node groups this wave walks.  Workgroups go round-robin over the 8 XCDs (blockIdx % 8), each with its own
    // L2: with a plain grid stride the 8 faces / 8 cells a node shares with its neighbours in the next mesh row would
    // be fetched by a different XCD every time.  So XCD x takes the x-th CONTIGUOUS eighth of the node list, and its
    // waves pull consecutive groups off a per-XCD counter: neighbouring mesh rows meet in one L2 a few passes apart,
    // and a CU that starts late (another kernel -- the all-gather of the previous step -- holds it) simply takes
    // fewer groups instead of stretching the launch.  Every wave ends when its counter passes the range.
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
    auto grab = [&]() -> int32_t {
        int32_t v = 0;
        if (lane == 0) v = atomicAdd(q, 1);
        return wg_lo + __builtin_amdgcn_readfirstlane(v);
    };
    NodeFetch cur, nx;
    int32_t wg = grab();
    {
        cur.level0(nodes, wg * GPW + grp, count);
        cur.level1(g);
        cur.level2(g, l8, sf);
        cur.level3(g);
        cur.level4(g);
    }
    for (int32_t pass = 0; wg < wg_end; ++pass) {
        if (DBG == 3) n_stamp = 0;
        NIN_STAMP();
        const int32_t wg_next = grab();
        nx.level0(nodes, wg_next * GPW + grp, count);   // past the end: a clamped (valid) node, never used
        const bool valid = cur.valid;
        const int32_t p = cur.p, eb = cur.eb;
        const double xv0 = cur.xv[0], xv1 = cur.xv[1], xv2 = cur.xv[2];
        // lane l16 < 8 is cell l16 of the node: its row (x_K - x_v) (gls.pyx:269-277).  Lanes 8..15 redo
        // cell l16 - 8 (no zero-initialised merge values: those get hoisted out of the node loop as registers)
        double dc[3];
        {
            if (l16 < NE) cells[l16] = cur.cell;
            dc[0] = cur.cen[0] - xv0;
            dc[1] = cur.cen[1] - xv1;
            dc[2] = cur.cen[2] - xv2;
        }
        lds_sync();
        // lane l16 < 12 is face l16 of the node: B_a = [K_a N; T1; tau T2], B_b = [K_b N; T1; tau T2]
        // (gls.pyx:293-321), kept in registers until the system has been staged.  B_b differs from B_a only in
        // its first row.
        double Ba[3][3], Bb0[3];   // Ba[e][t]
        int Ia = 0, Ib = 0;
        {
            const int ca = cur.ca, cb = cur.cb;
            const double N0 = cur.fn[0], N1 = cur.fn[1], N2 = cur.fn[2];
            const double T0 = xv0 - cur.fcen[0], T1 = xv1 - cur.fcen[1], T2 = xv2 - cur.fcen[2];
            const double U0 = N1 * T2 - N2 * T1, U1 = N2 * T0 - N0 * T2, U2 = N0 * T1 - N1 * T0;
            const double da = cur.da, db = cur.db;
            double eta = 0.0;
            eta = da > eta ? da : eta;
            eta = db > eta ? db : eta;
            // tau = |T_sj2|^(-eta) (gls.pyx:314): for a positive base pow(u, -eta) = exp(-eta log u)
            const double un = sqrt(U0 * U0 + U1 * U1 + U2 * U2);
            const double tj = face_tau(un, eta);
            const double *Ka = cur.Ka, *Kb = cur.Kb;
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                const int cq = cells[q];
                Ia = cq == ca ? q : Ia;
                Ib = cq == cb ? q : Ib;
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                Ba[0][t] = Ka[t * 3 + 0] * N0 + Ka[t * 3 + 1] * N1 + Ka[t * 3 + 2] * N2;   // (K_a N)_t
                Bb0[t] = Kb[t * 3 + 0] * N0 + Kb[t * 3 + 1] * N1 + Kb[t * 3 + 2] * N2;
            }
            Ba[1][0] = T0; Ba[1][1] = T1; Ba[1][2] = T2;
            Ba[2][0] = tj * U0; Ba[2][1] = tj * U1; Ba[2][2] = tj * U2;
        }
        NIN_STAMP();
        // ---- deal the matrix into registers, one slot (8 columns) at a time through the staging buffer.
        //      Column j = 3 i + t (cell i, component t) -> slot j / 8, column lane j % 8; this lane then keeps
        //      rows 2 rl + h of columns l8, l8 + 8, l8 + 16.  [-B_a | +B_b] per face, gls.pyx:340-356. -----------
        nx.level1(g);   // next pass: CSR row starts + node coordinates
        double a[SLOTS][HR], cr[C::CS], dsave[SLOTS], y[SLOTS];
        lds_f64 *stage_l = lds_base(stage + l16);
        lds_f64 *stage_b = (lds_f64 *)stage;
        {
#pragma unroll
            for (int i = 0; i < C::STAGE / 16; ++i) stage_l[16 * i] = 0.0;
            lds_sync();
            NIN_STAMP();
            if (l16 < NE) {          // cell row l16: (x_K - x_v) on its own three columns
#pragma unroll
                for (int t = 0; t < 3; ++t) stage_b[(3 * l16 + t) * M + l16] = dc[t];
            }
            if (l16 < NIF) {         // face l16: rows NE + 3 l16 .. + 2 of the columns of its two cells
                const int row = NE + 3 * l16;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    lds_f64 *ca_ = stage_b + ((3 * Ia + t) * M + row), *cb_ = stage_b + ((3 * Ib + t) * M + row);
                    ca_[0] = -Ba[0][t]; ca_[1] = -Ba[1][t]; ca_[2] = -Ba[2][t];
                    cb_[0] = Bb0[t]; cb_[1] = Ba[1][t]; cb_[2] = Ba[2][t];
                }
            }
            lds_sync();
            NIN_STAMP();
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) {
                const lds_f64 *mine = lds_base(stage + (l8 + 8 * q) * M + h);
#pragma unroll
                for (int rl = 0; rl < HR; ++rl) a[q][rl] = mine[2 * rl];
                // the column's own cell-row entry d_i[t] (needed again for r_i = 1 - d_i . y_i): row i = j / 3
                const int j = l8 + 8 * q, i = (j * 43) >> 7;
                dsave[q] = stage[j * M + i];
                y[q] = 0.0;
            }
            lds_sync();
            NIN_STAMP();
        }
        nx.level2(g, l8, sf);   // next pass: this lane's cell and face ids
#pragma unroll
        for (int e = 0; e < C::CS; ++e) cr[e] = (l16 + 16 * e < NE) ? 1.0 : 0.0;   // c = 1 on the cell rows
#pragma unroll
        for (int e = 0; e < C::CS; ++e) xb[l16 + 16 * e] = 0.0;                    // published-column buffer starts all zero
        lds_sync();

        double rinv[SLOTS];
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) rinv[q] = 0.0;
        NIN_STAMP();
        if (DBG != 1) {
            lds_f64 *xh = lds_base(xb + h * C::HRP), *xalpha = lds_base(xb + 2 * C::HRP);
            publish<0>(a, xh, xalpha, l8, h);
            QrLoop<0, NA>::run(a, cr, rinv, xh, lds_base(xb + (l16 & 1) * C::HRP + (l16 >> 1)), xalpha, l8, h, l16, nx, g);
        }
        if (DBG != 1 && DBG != 2) BackLoop<NA - 1>::run(a, cr, rinv, y, l8, h, l16);
        NIN_STAMP();
        if (DBG == 1 || DBG == 2) { double acc = 0; for (int q = 0; q < SLOTS; ++q) for (int rl = 0; rl < HR; ++rl) acc += a[q][rl]; y[0] = acc; }

        // ---- r_i = 1 - d_i . y_i on the cell rows, r.r = |c~(NA:M)|^2, weights = r_i / (r.r) -------------
        if (h == 0) {
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) prod[l8 + 8 * q] = dsave[q] * y[q];
        }
        double tail = 0.0;
#pragma unroll
        for (int e = 0; e < C::CS; ++e) {
            const int r = l16 + 16 * e;
            tail = (r >= NA && r < M) ? fma(cr[e], cr[e], tail) : tail;
        }
        const double rr = group16_sum(tail);
        lds_sync();
        const int i3 = 3 * l8;
        const double ri = 1.0 - ((prod[i3 + 0] + prod[i3 + 1]) + prod[i3 + 2]);
        double w = ri / rr;
        w = (rr > 0.0 && w - w == 0.0) ? w : 0.0;   // rank-deficient system: undefined in the reference, 0 here
        lds_sync();
        // gls.pyx:470-472 (only if an interior node carries the Neumann flag): neumann_ws = last cell's weight
        const bool is_neu = (g.flags[p] & 2) != 0;
        if (DBG == 1) { nx.level3(g); nx.level4(g); }   // (diagnostic build without the QR: its prefetch hooks)
        if (h == 0) prod[l8] = w;
        lds_sync();
        const double nwv = is_neu ? prod[NE - 1] : 0.0;
        if (valid && h == 0) {
            out[eb + l8] = w + (add_neumann ? nwv : 0.0);
            if (l8 == 0) nws[p] = nwv;
        }
        lds_sync();
        NIN_STAMP();
        if (DBG == 3 && blockIdx.x == 0 && threadIdx.x == 0 && pass == 3) {
            const int32_t p0 = nodes ? nodes[0] : 0;   // debug build only: 4th pass of wave 0, into the row of its 1st node
            for (int i = 0; i < n_stamp; ++i) out[g.esup_ptr[p0] + i] = 1.0e6 + (double)(stamps[i] - stamps[0]);
        }
        cur = nx;
        wg = wg_next;
    }
}

}  // namespace

int launch_gls_hex8(const GridView &g, const int32_t *nodes, int32_t count, int add_neumann, double *out,
                    double *nws, int32_t *queue, hipStream_t stream) {
    if (count <= 0) return 0;
    using C = Hex8;
    constexpr int wpb = 4;
    const size_t dyn = (size_t)wpb * C::NODES_PER_WAVE * C::NODE_DOUBLES * sizeof(double);
    int64_t blocks = ((int64_t)count + wpb * C::NODES_PER_WAVE - 1) / (wpb * C::NODES_PER_WAVE);
    const int64_t cap = 256;   // one 145 KB workgroup per CU is resident anyway: persistent, and blockIdx % 8 is the XCD
    if (blocks > cap) blocks = cap;
    if (blocks > 8) blocks &= ~(int64_t)7;
    static const int max_blocks = getenv("NIN_GLS_MAX_BLOCKS") ? atoi(getenv("NIN_GLS_MAX_BLOCKS")) : 0;
    if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
    // 145 KB of dynamic LDS per block: above the default 64 KB limit.  The attribute belongs to the (function, device)
    // pair, and one process may drive several GPUs: remembered per device
    static bool attr_set_dev[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    bool &attr_set = attr_set_dev[dev_id & 63];
    if (!attr_set) {
        const void *ks[4] = {reinterpret_cast<const void *>(nin_gls_group_kernel<0>), reinterpret_cast<const void *>(nin_gls_group_kernel<1>),
                             reinterpret_cast<const void *>(nin_gls_group_kernel<2>), reinterpret_cast<const void *>(nin_gls_group_kernel<3>)};
        for (const void *k : ks)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) return -3;
        attr_set = true;
    }
    static const int dbg = getenv("NIN_GLS_DEBUG_MODE") ? atoi(getenv("NIN_GLS_DEBUG_MODE")) : 0;
    if (dbg == 3) hipLaunchKernelGGL(nin_gls_group_kernel<3>, dim3((unsigned)blocks), dim3(64 * wpb), dyn, stream, g, nodes, count, add_neumann, out, nws, queue);
    else if (dbg == 1) hipLaunchKernelGGL(nin_gls_group_kernel<1>, dim3((unsigned)blocks), dim3(64 * wpb), dyn, stream, g, nodes, count, add_neumann, out, nws, queue);
    else if (dbg == 2) hipLaunchKernelGGL(nin_gls_group_kernel<2>, dim3((unsigned)blocks), dim3(64 * wpb), dyn, stream, g, nodes, count, add_neumann, out, nws, queue);
    else
    hipLaunchKernelGGL(nin_gls_group_kernel<0>, dim3((unsigned)blocks), dim3(64 * wpb), dyn, stream, g, nodes, count,
                       add_neumann, out, nws, queue);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

const char *kernel_name_gls_hex8() { return "nin_gls_group_kernel"; }

}  // namespace nin
