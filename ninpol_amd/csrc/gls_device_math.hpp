// gls_device_math.hpp -- scalar helpers shared by the GLS kernels (internal, device code only)
#pragma once
#include <hip/hip_runtime.h>

namespace nin {
namespace glsmath {

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// 1/d and 1/sqrt(s) from the hardware seeds (v_rcp_f64 / v_rsq_f64: 2^-24.4, measured -- tools/micro_seed.hip) plus ONE cubic
// correction step each: r (1 + e + e^2), e = 1 - d r, and y (1 + e/2 + 3 e^2/8), e = 1 - s y^2 -- the error cubes, 1e-22
// before rounding; measured 1.0 / 1.24 ulp over 2^-20 .. 2^20.  (One NEWTON step on the reciprocal leaves 2.2e-15 = 20 ulp,
// which the FAN tensor's condition numbers turn into 3e-10 on the weights: tried in round 3, caught by the parity suite.)
// A third of the instructions and of the dependent latency of the IEEE division / sqrt expansions.
__device__ __forceinline__ double fast_rcp(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r, 1.0);
    return fma(r, fma(e, e, e), r);
}
__device__ __forceinline__ double fast_rsqrt(double s) {
    const double y = __builtin_amdgcn_rsq(s);
    const double e = fma(-s * y, y, 1.0);            // 1 - s y^2
    return fma(y * e, fma(e, 0.375, 0.5), y);          // y (1 + e/2 + 3 e^2/8)
}
// (the forms the cubic steps replace, kept for tools/micro_seed.hip)
__device__ __forceinline__ double rcp_newton(double d, int steps) {
    double r = __builtin_amdgcn_rcp(d);
    for (int i = 0; i < steps; ++i) r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// Householder scalars for the column (alpha, x), ss = |x|^2:  beta = -sign(alpha) |(alpha, x)| (dlarfg);
// H = I - g v v^T with v = (alpha - beta, x), g = 1 / (beta (beta - alpha)) = 1 / (S + |alpha| sqrt(S)),
// S = alpha^2 + ss;  rinv = 1 / beta = 1 / R(k,k).  x = 0: H = I, beta = alpha.
struct House {
    double beta, vp, g, rinv;
};
// The same without the x = 0 branch.  x = 0, alpha != 0 needs none: beta = -alpha, v = (2 alpha, 0), H flips the sign of
// the pivot row -- orthogonal all the same.  A column that is zero altogether (S = 0) turns into NaNs here, which the
// caller's final finiteness test maps to the zero row a rank-deficient system gets anyway.
__device__ __forceinline__ House house_unguarded(double alpha, double ss) {
    const double S = fma(alpha, alpha, ss);
    const double rs = fast_rsqrt(S), sq = S * rs;
    House h;
    h.beta = -copysign(sq, alpha);
    h.vp = alpha - h.beta;
    h.g = fast_rcp(fma(fabs(alpha), sq, S));
    h.rinv = -copysign(rs, alpha);                          // 1 / beta
    return h;
}
__device__ __forceinline__ House house(double alpha, double ss) {
    const bool live = ss != 0.0;
    const double S = fma(alpha, alpha, ss);
    const double rs = fast_rsqrt(S), sq = S * rs;                        // sqrt(S)
    House h;
    h.beta = live ? -copysign(sq, alpha) : alpha;
    h.vp = alpha - h.beta;
    const double gden = fast_rcp(live ? fma(fabs(alpha), sq, S) : alpha);
    h.g = live ? gden : 0.0;
    h.rinv = live ? -(gden * h.vp) : gden;
    return h;
}

// tau = |T_sj2|^(-eta) (gls.pyx:314); for a positive base pow(u, -eta) = exp(-eta log u).  The library exp / log are
// ~250 instructions per face record; this pair is ~45: log via u = 2^e m, m in [sqrt(1/2), sqrt(2)),
// 2 atanh((m - 1) / (m + 1)) as an 11-term odd series (|s| <= 0.172: truncation 1e-17), exp via y = k ln2 + r,
// |r| <= 0.347, Taylor to r^14 (4e-18) and ldexp.  Against numpy's pow over u in [1e-6, 1e2], eta in (0, 1]: max
// relative error 1.8e-15, mean 1.6e-16 -- five orders below the 1e-10 weight tolerance.  Kept out of line:
// inlined, the series coefficients are hoisted out of the node loop and stay live across the whole QR.
// SQUARED = true: the argument is |T_sj2|^2 and the result the same tau = (un2)^(-eta / 2) -- the caller skips the square root
template <bool SQUARED>
__device__ __attribute__((noinline)) static double face_tau_t(double un, double eta) {
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
    const double y = (SQUARED ? -0.5 * eta : -eta) * lg;
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
__device__ __forceinline__ double face_tau(double un, double eta) { return face_tau_t<false>(un, eta); }
__device__ __forceinline__ double face_tau_sq(double un2, double eta) { return face_tau_t<true>(un2, eta); }

// The panel of a front: three Householder steps on the front cell's own columns (10 rows: the cell row and the 3 x 3
// rows of its faces).  v_k stays in P[k..9][k] (pivot entries included), R's off-diagonal entries in P[0][1], P[0][2],
// P[1][2]; g[k] = the reflectors' scalars, rinv[k] = 1 / R(k, k); z = R_ee^-T d: all the weights need of the front
// cell's three rows of R.  Shared by kernels_gls_hex8mf.hip (a front per lane of the quad) and kernels_gls_mfw.hip.
__device__ __forceinline__ void front_panel(double (&P)[10][3], const double (&d)[3], double (&g)[3], double (&z)[3]) {
    double rinv[3];
    {
        double ss = 0.0;
#pragma unroll
        for (int r = 1; r < 10; ++r) ss = fma(P[r][0], P[r][0], ss);
        const House h = house_unguarded(P[0][0], ss);
        g[0] = h.g; rinv[0] = h.rinv;
        double d1 = h.vp * P[0][1], d2 = h.vp * P[0][2];
#pragma unroll
        for (int r = 1; r < 10; ++r) { d1 = fma(P[r][0], P[r][1], d1); d2 = fma(P[r][0], P[r][2], d2); }
        const double w1 = -(h.g * d1), w2 = -(h.g * d2);
        P[0][0] = h.vp;
#pragma unroll
        for (int r = 0; r < 10; ++r) { P[r][1] = fma(w1, P[r][0], P[r][1]); P[r][2] = fma(w2, P[r][0], P[r][2]); }
    }
    {
        double ss = 0.0;
#pragma unroll
        for (int r = 2; r < 10; ++r) ss = fma(P[r][1], P[r][1], ss);
        const House h = house_unguarded(P[1][1], ss);
        g[1] = h.g; rinv[1] = h.rinv;
        double d2 = h.vp * P[1][2];
#pragma unroll
        for (int r = 2; r < 10; ++r) d2 = fma(P[r][1], P[r][2], d2);
        const double w2 = -(h.g * d2);
        P[1][1] = h.vp;
#pragma unroll
        for (int r = 1; r < 10; ++r) P[r][2] = fma(w2, P[r][1], P[r][2]);
    }
    {
        double ss = 0.0;
#pragma unroll
        for (int r = 3; r < 10; ++r) ss = fma(P[r][2], P[r][2], ss);
        const House h = house_unguarded(P[2][2], ss);
        g[2] = h.g; rinv[2] = h.rinv;
        P[2][2] = h.vp;
    }
    z[0] = d[0] * rinv[0];
    z[1] = fma(-P[0][1], z[0], d[1]) * rinv[1];
    z[2] = fma(-P[1][2], z[1], fma(-P[0][2], z[0], d[2])) * rinv[2];
}

// Computed HERE: without it the optimiser sinks a value's whole computation down to its first use -- for u, s and z
// that is the end of the pass, with the rows of R they are made from parked in ~140 AGPRs across phase 2.
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x)); }

// Apply the three reflectors of the panel (v_k in P[k..9][k], pivot entries included; g[k]) to W columns of the
// front.  Z0 / ZA / ZB / ZC: which row groups of the columns can be non-zero on entry (row 0, the rows 1-3, 4-6, 7-9
// of face 0, 1, 2) -- the others are structural zeros that the first reflector fills.
template <int W, bool Z0, bool ZA, bool ZB, bool ZC>
__device__ __forceinline__ void apply_panel(const double (&P)[10][3], const double (&g)[3], double (&B)[10][W]) {
    double w[W];
    // reflector 0, rows 0..9
#pragma unroll
    for (int c = 0; c < W; ++c) {
        double d = 0.0;
        if (Z0) d = P[0][0] * B[0][c];
#pragma unroll
        for (int r = 1; r < 10; ++r) {
            const bool nz = (r <= 3) ? ZA : (r <= 6) ? ZB : ZC;
            if (nz) d = fma(P[r][0], B[r][c], d);
        }
        w[c] = -(g[0] * d);
    }
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const bool nz = (r == 0) ? Z0 : (r <= 3) ? ZA : (r <= 6) ? ZB : ZC;
#pragma unroll
        for (int c = 0; c < W; ++c) B[r][c] = nz ? fma(w[c], P[r][0], B[r][c]) : w[c] * P[r][0];
    }
    // reflectors 1 and 2, rows k..9 (dense by now)
#pragma unroll
    for (int k = 1; k < 3; ++k) {
#pragma unroll
        for (int c = 0; c < W; ++c) {
            double d = P[k][k] * B[k][c];
#pragma unroll
            for (int r = k + 1; r < 10; ++r) d = fma(P[r][k], B[r][c], d);
            w[c] = -(g[k] * d);
        }
#pragma unroll
        for (int r = k; r < 10; ++r) {
#pragma unroll
            for (int c = 0; c < W; ++c) B[r][c] = fma(w[c], P[r][k], B[r][c]);
        }
    }
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace glsmath
}  // namespace nin
