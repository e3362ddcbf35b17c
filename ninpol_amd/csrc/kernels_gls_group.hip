// kernels_gls_group.hip -- GLS weights for homogeneous small-node classes, gfx950: a GROUP of 16 lanes
// per node (4 nodes per wavefront), the whole least-squares system in registers.
//
// Why a second GLS kernel.  kernels_gls.hip maps one node to one wavefront with matrix ROWS across the
// lanes; every (reflector, column) pair then costs a full cross-lane reduction and the matrix lives in
// LDS.  Measured on the 10 M-cell hexahedron mesh (profiles/r01/v1_*): 261 ms, ~60 k issue cycles per
// node for ~5 k useful FP64 FMAs.  GLS is FP64-ALU-bound, so what matters is FMAs per
// issued instruction.  Here the M x NA matrix of a node is dealt out in 2-D over its 16 lanes:
//   columns  j -> lane l8 = j % 8, register slot q = j / 8          (3 slots for 24 columns)
//   rows     r -> half  h  = r % 2, local row     rl = r / 2        (22 local rows for 44 rows)
// so a lane holds 3 x 22 doubles with compile-time indices (132 VGPRs; VALU can address 256, which is
// why the rows are split over two lanes instead of 44 rows in one).  A Householder step is then:
//   * the two owner lanes of column k publish it through LDS,
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
// interior node of a hexahedron mesh (M = 44, 24 + 1 columns).  Everything else runs in kernels_gls.hip.
#include <hip/hip_runtime.h>

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

__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Hex8 {
    static constexpr int NE = 8, NIF = 12;
    static constexpr int M = NE + 3 * NIF;   // 44 rows
    static constexpr int NA = 3 * NE;        // 24 columns of A (+ the column c)
    static constexpr int SLOTS = NA / 8;     // 3 columns per lane
    static constexpr int HR = M / 2;         // 22 local rows per half
    static constexpr int CS = (M + 15) / 16; // 3 rows of c per lane
    static constexpr int FR = 20;            // doubles per face record in LDS
    static constexpr int NODE_DOUBLES = NIF * FR + M + NE / 2 + NA;
    static constexpr int LANES = 16, NODES_PER_WAVE = 4;
};

// One Householder step K (compile-time).  l8: column lane, h: row half, l16 = 8 h + l8.
template <int K>
__device__ __forceinline__ void qr_step(double (&a)[Hex8::SLOTS][Hex8::HR], double (&cr)[Hex8::CS], double *xb,
                                        int l8, int h, int l16) {
    using C = Hex8;
    constexpr int QK = K / 8, LK = K % 8, HP = K % 2, PL = K / 2, RL0 = (K + 1) / 2, HR = C::HR;
    const bool owner = (l8 == LK);
    const bool pivot_half = (h == HP);
    if (owner) {
#pragma unroll
        for (int rl = PL; rl < HR; ++rl) xb[h * HR + rl] = a[QK][rl];
    }
    lds_sync();
    // this lane's half of the published column, rows strictly below the pivot
    double x[HR];
#pragma unroll
    for (int rl = RL0; rl < HR; ++rl) x[rl] = xb[h * HR + rl];
    if (HP == 0) x[PL] = (h == 0) ? 0.0 : x[PL];   // K even: local row PL of half 0 IS the pivot row
    const double alpha = xb[HP * HR + PL];
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
    // beta = -sign(alpha) |(alpha, x)| (dlarfg);  H = I - g v v^T,  v = (alpha - beta, x),  g = 1 / (beta (beta - alpha))
    const bool live = ss != 0.0;
    const double beta = live ? -copysign(sqrt(fma(alpha, alpha, ss)), alpha) : alpha;
    const double vk = alpha - beta;
    const double gk = live ? 1.0 / (beta * (beta - alpha)) : 0.0;
    if (owner && pivot_half) a[QK][PL] = beta;   // R(K,K); the rest of this column is dead from here on
    double gw[C::SLOTS];
#pragma unroll
    for (int q = QK; q < C::SLOTS; ++q) {
        const bool act = (q > QK) || (l8 > LK);   // column j = l8 + 8 q is to the right of K
        double w = pivot_half ? fma(vk, a[q][PL], d[q]) : d[q];
        w += partner(w);
        gw[q] = act ? -(gk * w) : 0.0;
        if (pivot_half) a[q][PL] = fma(gw[q], vk, a[q][PL]);
    }
#pragma unroll
    for (int rl = RL0; rl < HR; ++rl) {
#pragma unroll
        for (int q = QK; q < C::SLOTS; ++q) a[q][rl] = fma(gw[q], x[rl], a[q][rl]);
    }
    // the last column c, dealt by rows: lane l16 holds rows l16, l16 + 16, l16 + 32
    {
        double vc[C::CS], part = 0.0;
#pragma unroll
        for (int e = 0; e < C::CS; ++e) {
            if (16 * e + 15 < K) { vc[e] = 0.0; continue; }   // every row of this slot is above the pivot
            const int r = l16 + 16 * e;
            const double xl = (r > K && r < C::M) ? xb[(r & 1) * HR + (r >> 1)] : 0.0;
            vc[e] = (r == K) ? vk : xl;
            part = fma(vc[e], cr[e], part);
        }
        const double gwc = -(gk * group16_sum(part));
#pragma unroll
        for (int e = 0; e < C::CS; ++e) cr[e] = fma(gwc, vc[e], cr[e]);
    }
    lds_sync();                          // the next step overwrites xb
    __builtin_amdgcn_sched_barrier(0);   // keep the steps apart (register pressure)
}

template <int K, int KEND>
struct QrLoop {
    static __device__ __forceinline__ void run(double (&a)[Hex8::SLOTS][Hex8::HR], double (&cr)[Hex8::CS], double *xb,
                                               int l8, int h, int l16) {
        qr_step<K>(a, cr, xb, l8, h, l16);
        QrLoop<K + 1, KEND>::run(a, cr, xb, l8, h, l16);
    }
};
template <int KEND>
struct QrLoop<KEND, KEND> {
    static __device__ __forceinline__ void run(double (&)[Hex8::SLOTS][Hex8::HR], double (&)[Hex8::CS], double *, int, int, int) {}
};

// Back-substitution R y = b, row J (compile-time).  R(J, j) sits in lane (j % 8, J % 2) at local row J / 2,
// slot j / 8; b_J in cr[J / 16] of lane J % 16; y_j is kept in both lanes of column lane j % 8.
template <int J>
__device__ __forceinline__ void back_step(const double (&a)[Hex8::SLOTS][Hex8::HR], const double (&cr)[Hex8::CS],
                                          double (&y)[Hex8::SLOTS], int l8, int h, int l16) {
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
    const double yj = -tot / a[QJ][PJ];         // meaningful in lane (LJ, HJ)
    const double yp = partner(yj);
    y[QJ] = (l8 == LJ) ? ((h == HJ) ? yj : yp) : y[QJ];
}

template <int J>
struct BackLoop {
    static __device__ __forceinline__ void run(const double (&a)[Hex8::SLOTS][Hex8::HR], const double (&cr)[Hex8::CS],
                                               double (&y)[Hex8::SLOTS], int l8, int h, int l16) {
        back_step<J>(a, cr, y, l8, h, l16);
        BackLoop<J - 1>::run(a, cr, y, l8, h, l16);
    }
};
template <>
struct BackLoop<-1> {
    static __device__ __forceinline__ void run(const double (&)[Hex8::SLOTS][Hex8::HR], const double (&)[Hex8::CS],
                                               double (&)[Hex8::SLOTS], int, int, int) {}
};

__global__ __launch_bounds__(256, 2) void nin_gls_group_kernel(GridView g, const int32_t *__restrict__ nodes,
                                                               int32_t count, int add_neumann,
                                                               double *__restrict__ out, double *__restrict__ nws) {
    using C = Hex8;
    constexpr int NE = C::NE, NIF = C::NIF, M = C::M, NA = C::NA, SLOTS = C::SLOTS, HR = C::HR, FR = C::FR;
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int l16 = lane & 15, l8 = lane & 7, h = (lane >> 3) & 1, grp = lane >> 4;
    constexpr int GPW = C::NODES_PER_WAVE;
    double *node_lds = smem + ((size_t)wave * GPW + grp) * C::NODE_DOUBLES;
    double *faces = node_lds;                              // [NIF][FR]
    double *xb = faces + NIF * FR;                         // [2][HR] published column, by half
    int32_t *cells = reinterpret_cast<int32_t *>(xb + M);  // [NE]
    double *prod = xb + M + NE / 2;                        // [NA]

    const int32_t n_groups = (count + GPW - 1) / GPW;
    for (int32_t wg = blockIdx.x * wpb + wave; wg < n_groups; wg += gridDim.x * wpb) {
        const int32_t idx = wg * GPW + grp;
        const bool valid = idx < count;
        const int32_t sel = valid ? idx : count - 1;
        const int32_t p = nodes ? nodes[sel] : sel;
        const int32_t eb = g.esup_ptr[p], fb = g.fsup_ptr[p];
        const double xv0 = g.coords[3 * (size_t)p + 0], xv1 = g.coords[3 * (size_t)p + 1],
                     xv2 = g.coords[3 * (size_t)p + 2];
        if (l16 < NE) cells[l16] = g.esup[eb + l16];
        lds_sync();
        // ---- face records: B_a = [K_a N; T1; tau T2], B_b = [K_b N; T1; tau T2]  (gls.pyx:293-321) ------
        if (l16 < NIF) {
            const int s = l16;
            const size_t f = (size_t)g.fsup[fb + s];
            const int ca = g.face_cells[2 * f], cb = g.face_cells[2 * f + 1];
            const double N0 = g.face_normal[3 * f + 0], N1 = g.face_normal[3 * f + 1], N2 = g.face_normal[3 * f + 2];
            const double T0 = xv0 - g.face_center[3 * f + 0], T1 = xv1 - g.face_center[3 * f + 1],
                         T2 = xv2 - g.face_center[3 * f + 2];
            const double U0 = N1 * T2 - N2 * T1, U1 = N2 * T0 - N0 * T2, U2 = N0 * T1 - N1 * T0;
            const double da = g.diff_mag[ca], db = g.diff_mag[cb];
            double eta = 0.0;
            eta = da > eta ? da : eta;
            eta = db > eta ? db : eta;
            const double tj = pow(sqrt(U0 * U0 + U1 * U1 + U2 * U2), -eta);
            const double *Ka = g.perm + 9 * (size_t)ca, *Kb = g.perm + 9 * (size_t)cb;
            int Ia = 0, Ib = 0;
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                const int cq = cells[q];
                Ia = cq == ca ? q : Ia;
                Ib = cq == cb ? q : Ib;
            }
            double *rec = faces + s * FR;
            reinterpret_cast<int32_t *>(rec)[0] = Ia;
            reinterpret_cast<int32_t *>(rec)[1] = Ib;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rec[2 + c] = Ka[c * 3 + 0] * N0 + Ka[c * 3 + 1] * N1 + Ka[c * 3 + 2] * N2;
                rec[11 + c] = Kb[c * 3 + 0] * N0 + Kb[c * 3 + 1] * N1 + Kb[c * 3 + 2] * N2;
            }
            rec[5] = T0; rec[6] = T1; rec[7] = T2;
            rec[14] = T0; rec[15] = T1; rec[16] = T2;
            rec[8] = tj * U0; rec[9] = tj * U1; rec[10] = tj * U2;
            rec[17] = tj * U0; rec[18] = tj * U1; rec[19] = tj * U2;
        }
        lds_sync();
        // ---- this lane's entries: columns j = l8 + 8 q <-> (cell i = j / 3, component t = j % 3),
        //      rows r = 2 rl + h -------------------------------------------------------------------------
        double a[SLOTS][HR], cr[C::CS], dsave[SLOTS], y[SLOTS];
        int ci[SLOTS], ct[SLOTS];
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            const int j = l8 + 8 * q;
            ci[q] = j / 3;
            ct[q] = j - 3 * ci[q];
            const double xvt = ct[q] == 0 ? xv0 : (ct[q] == 1 ? xv1 : xv2);
            dsave[q] = g.centroids[3 * (size_t)cells[ci[q]] + ct[q]] - xvt;   // (x_K - x_v)_t, gls.pyx:269-277
            y[q] = 0.0;
        }
#pragma unroll
        for (int rl = 0; rl < HR; ++rl) {
            const int r = 2 * rl + h;
            if (2 * rl + 1 < NE) {               // both halves of this local row are cell rows
#pragma unroll
                for (int q = 0; q < SLOTS; ++q) a[q][rl] = (r == ci[q]) ? dsave[q] : 0.0;
            } else {                             // face rows: r = NE + 3 s + e  ->  [-B_a | +B_b], gls.pyx:340-356
                const int fr = r - NE, s = (fr * 43) >> 7, e = fr - 3 * s;   // fr / 3, exact for fr < 100
                const double *rec = faces + s * FR;
                const int Ia = reinterpret_cast<const int32_t *>(rec)[0], Ib = reinterpret_cast<const int32_t *>(rec)[1];
#pragma unroll
                for (int q = 0; q < SLOTS; ++q) {
                    const bool isa = Ia == ci[q], isb = Ib == ci[q];
                    const double v = rec[(isa ? 2 : 11) + 3 * e + ct[q]];
                    a[q][rl] = isa ? -v : (isb ? v : 0.0);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < C::CS; ++e) cr[e] = (l16 + 16 * e < NE) ? 1.0 : 0.0;   // c = 1 on the cell rows
        lds_sync();

        QrLoop<0, NA>::run(a, cr, xb, l8, h, l16);
        BackLoop<NA - 1>::run(a, cr, y, l8, h, l16);

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
        if (h == 0) prod[l8] = w;
        lds_sync();
        const double nwv = is_neu ? prod[NE - 1] : 0.0;
        if (valid && h == 0) {
            out[eb + l8] = w + (add_neumann ? nwv : 0.0);
            if (l8 == 0) nws[p] = nwv;
        }
        lds_sync();
    }
}

}  // namespace

int launch_gls_hex8(const GridView &g, const int32_t *nodes, int32_t count, int add_neumann, double *out,
                    double *nws, hipStream_t stream) {
    if (count <= 0) return 0;
    using C = Hex8;
    constexpr int wpb = 4;
    const size_t dyn = (size_t)wpb * C::NODES_PER_WAVE * C::NODE_DOUBLES * sizeof(double);
    int64_t blocks = ((int64_t)count + wpb * C::NODES_PER_WAVE - 1) / (wpb * C::NODES_PER_WAVE);
    const int64_t cap = 256 * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(nin_gls_group_kernel, dim3((unsigned)blocks), dim3(64 * wpb), dyn, stream, g, nodes, count,
                       add_neumann, out, nws);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

const char *kernel_name_gls_hex8() { return "nin_gls_group_kernel"; }

}  // namespace nin
