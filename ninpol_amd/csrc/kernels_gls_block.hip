// kernels_gls_block.hip -- GLS weights for nodes of any shape, gfx950: one node per workgroup of NW wavefronts.
//
// Same mathematics as kernels_gls.hip (the reference's dense m x n system of gls.pyx:252-416, Householder QR as in
// dgels, only row n-1 of the solution kept, gls.pyx:466-472), laid out so that no dot product ever crosses lanes:
//
//   * a lane owns COLUMNS (column j sits in lane (n-1-j) % 64, slot (n-1-j) / 64, so the columns still alive at
//     step k always fill the low lanes) and wave w owns the ROWS i = w (mod NW).  The system lives in LDS
//     column-major with an odd column pitch: a wave's access to one row is conflict-free, and inside a lane's
//     column the rows of a wave sit at compile-time distances, so the row loop needs no address arithmetic
//     (ds_read / ds_write immediates);
//   * the reflector is kept unnormalised, H = I - v v^T / (beta (beta - alpha)), v = x - beta e_k, so everything
//     step k needs are the dots d_j = x . a_j of the raw pivot column with every live column (d_k is its squared
//     norm): beta = -sign(alpha) sqrt(d_k), w_j = (d_j - beta a_kj) / (d_k + |alpha| sqrt(d_k)), a_ij -= x_i w_j;
//   * one sweep per step: a wave walks its rows once, reads each element from LDS, updates it, writes it back and
//     in the same breath accumulates the dots step k+1 will need.  x_i and the next pivot-column entry are picked
//     out of the lane that owns them with v_readlane (the pivot lane is wave-uniform), so the row loop is
//     ds_read, fma, ds_write, fma per element with no reduction and no broadcast traffic;
//   * the per-wave partial dots meet in LDS (double-buffered), one workgroup barrier per step.
//
// R is left in the rows 0..n-2 of the LDS array; the weights are r_i / (r.r) with r the residual of the last
// column c against the others (kernels_gls.hip): back-substitute R y = (Q^T c)(0:n-1), then
// r_i = 1 - (x_Ki - x_v) . y_i for the n_elem cell rows and r.r = |(Q^T c)(n-1:m)|^2.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device_grid.hpp"
#include "gls_device_math.hpp"
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

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

__device__ __forceinline__ int ufirst(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double s) {
    double y = __builtin_amdgcn_rsq(s);
    double e = fma(-s * y, y, 1.0);
    y = fma(y * e, fma(e, 0.375, 0.5), y);
    e = fma(-s * y, y, 1.0);
    y = fma(y * e, 0.5, y);
    return y;
}

// Orders LDS traffic between the waves of the workgroup (or inside the single wave when NW == 1).
template <int NW>
__device__ __forceinline__ void group_sync() {
    if (NW == 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// Diagnostic build (-DNIN_BLOCK_STAMPS, tools/stamps_block.py): block 0 records s_memtime at four points of every
// step into the neumann_ws array instead of the result.
#ifdef NIN_BLOCK_STAMPS
#define NIN_STAMP(S, K, J)                                                                        \
    do {                                                                                          \
        if ((S).stamps && (S).lane == 0) {                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            (S).stamps[(((S).wave * 256) + (K)) * 4 + (J)] = (double)__builtin_amdgcn_s_memtime(); \
            __builtin_amdgcn_sched_barrier(0);                                                    \
        }                                                                                         \
    } while (0)
#else
#define NIN_STAMP(S, K, J) do { } while (0)
#endif

struct Sys {       // one node's system, wave-uniform
    double *A;     // [n + 1][ld] column-major, ld odd >= m; column n stays zero (parking column for dead lanes)
    double *aux;   // partial dots [2][NW][n]; later y[n] and the weight row
    int n, m, ld, lane, wave;
    int pstride;   // entries per wave in a partial-dot buffer: n, or -- packed -- the live columns of the dense phase
    bool db;       // partial dots double-buffered: one workgroup barrier per step instead of two
    double *stamps;
};

// This lane's column in slot q (nullptr-free: lanes without a column get the parking column).
__device__ __forceinline__ double *lane_column(const Sys &s, int q, bool has) {
    return s.A + (has ? (s.n - 1 - (s.lane + 64 * q)) : s.n) * s.ld;
}

// Dots of column k0 with every column, over this wave's rows from k0 on: what step k0 (the first dense step) starts from.
template <int NW, int CS, int TOP>
__device__ __forceinline__ void first_dots(const Sys &s, int k0, double (&dn)[CS]) {
    const int l0 = (s.n - 1 - k0) & 63;
    const double *col[TOP + 1];
#pragma unroll
    for (int q = 0; q <= TOP; ++q) col[q] = lane_column(s, q, s.lane + 64 * q < s.n);
    // four rows at a time: their loads go out together (a row at a time this pass is a chain of LDS latencies)
    int i = k0 + ((s.wave - k0) % NW + NW) % NW;
    for (; i + 3 * NW < s.m; i += 4 * NW) {
        double own[4][TOP + 1];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q <= TOP; ++q) own[u][q] = col[q][i + u * NW];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double x = readlane_f64(own[u][TOP], l0);
#pragma unroll
            for (int q = 0; q <= TOP; ++q) dn[q] = fma(x, own[u][q], dn[q]);
        }
    }
    for (; i < s.m; i += NW) {
        double own[TOP + 1];
#pragma unroll
        for (int q = 0; q <= TOP; ++q) own[q] = col[q][i];
        const double x = readlane_f64(own[TOP], l0);
#pragma unroll
        for (int q = 0; q <= TOP; ++q) dn[q] = fma(x, own[q], dn[q]);
    }
}

// Householder step k.  TOP = slot of the pivot column (the highest slot with a live column).
template <int NW, int CS, int TOP>
__device__ __forceinline__ void qr_step(const Sys &s, int k, int buf, double (&dn)[CS], double (&rkeep)[CS],
                                        bool &singular) {
    const int n = s.n, m = s.m, lane = s.lane;
    const int jjk = n - 1 - k, lk = jjk & 63;          // the pivot column's reversed index and lane
    const int jjn = jjk - 1, ln = jjn & 63;            // column k + 1: same slot, or lane 63 of the slot below
    const bool next_in_top = (jjn >> 6) == TOP;
    double d[TOP + 1], rowk[TOP + 1], w[TOP + 1];
    double *col[TOP + 1];
    bool live[TOP + 1], upd[TOP + 1];
#pragma unroll
    for (int q = 0; q <= TOP; ++q) {
        const int jj = lane + 64 * q;
        live[q] = jj <= jjk;
        upd[q] = jj < jjk;
        if (NW == 1) {
            d[q] = dn[q];
        } else {
            const int ps = s.pstride;
            const double *P = s.aux + ((s.db ? buf : 0) * NW * ps + (jj < ps ? jj : ps - 1));   // dead lanes: any valid word
            double part[NW];
#pragma unroll
            for (int u = 0; u < NW; ++u) part[u] = P[u * ps];
            double acc = part[0];
#pragma unroll
            for (int u = 1; u < NW; ++u) acc += part[u];
            d[q] = acc;
        }
        col[q] = lane_column(s, q, live[q]);
        rowk[q] = col[q][k];          // dead lanes read the parking column: 0
        dn[q] = 0.0;
    }
    if (NW > 1 && !s.db) group_sync<NW>();   // one partial buffer: nobody may publish before everybody has read
    NIN_STAMP(s, k, 0);
    const double dk = readlane_f64(d[TOP], lk), alpha = readlane_f64(rowk[TOP], lk);
    if (!(dk != 0.0)) singular = true;                  // an all-zero pivot column (or NaN): no solution row
    const double sq = dk * fast_rsqrt(dk);              // sqrt(d_k) = |(alpha, x)|
    const double beta = -copysign(sq, alpha);
    const double inv = fast_rcp(fma(fabs(alpha), sq, dk));   // 1 / (beta (beta - alpha))
    const double vk = alpha - beta;
#pragma unroll
    for (int q = 0; q <= TOP; ++q) {
        w[q] = upd[q] ? (d[q] - beta * rowk[q]) * inv : 0.0;
        rkeep[q] = fma(-vk, w[q], rowk[q]);             // row k of R (written once no wave can still be reading row k)
        if (q == TOP && lane == lk) rkeep[q] = beta;
    }
#pragma unroll
    for (int q = TOP + 1; q < CS; ++q) rkeep[q] = 0.0;

    NIN_STAMP(s, k, 1);
    // The sweep: rows i0, i0 + NW, ... of this wave, G at a time with the next G loads in flight.  Inside a
    // lane's column those rows are NW doubles apart: every access is base + immediate.  Dead lanes of the top
    // slot work on the parking column (0 in, w = 0, 0 out); with a single slot they are simply masked off.
    constexpr int G = TOP == 0 ? 5 : 4;   // (5, not 8: the partial last group runs unpipelined -- fewer rows end up in it; A/B: -4 %)
    const int i0 = k + 1 + ((s.wave - (k + 1)) % NW + NW) % NW;
    const int rows = i0 < m ? (m - i0 + NW - 1) / NW : 0;
    const int full = rows / G, rem = rows % G;
    if (TOP > 0 || live[0]) {
        double acc[2][TOP + 1];
        double *p[TOP + 1];
#pragma unroll
        for (int q = 0; q <= TOP; ++q) {
            acc[0][q] = acc[1][q] = 0.0;
            p[q] = col[q] + i0;
        }
        auto row = [&](int u, double (&cur)[G][TOP + 1]) {
            const double x = readlane_f64(cur[u][TOP], lk);
            double nv[TOP + 1];
#pragma unroll
            for (int q = 0; q <= TOP; ++q) {
                nv[q] = fma(-x, w[q], cur[u][q]);
                p[q][u * NW] = nv[q];
            }
            const double xn = readlane_f64(next_in_top ? nv[TOP] : nv[TOP > 0 ? TOP - 1 : 0], ln);
#pragma unroll
            for (int q = 0; q <= TOP; ++q) acc[u & 1][q] = fma(xn, nv[q], acc[u & 1][q]);
        };
        // a full group, stage by stage across its G rows: a row's chain (broadcast x -> update -> broadcast the new
        // entry of column k + 1 -> dot) is four dependent instructions deep, the G chains side by side fill each
        // other's latencies (the compiler, left alone, runs them one after the other through one scalar register pair)
        auto rows = [&](double (&cur)[G][TOP + 1]) {
            double x[G], xn[G], nv[G][TOP + 1];
#pragma unroll
            for (int u = 0; u < G; ++u) x[u] = readlane_f64(cur[u][TOP], lk);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < G; ++u) {
#pragma unroll
                for (int q = 0; q <= TOP; ++q) nv[u][q] = fma(-x[u], w[q], cur[u][q]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < G; ++u) {
#pragma unroll
                for (int q = 0; q <= TOP; ++q) p[q][u * NW] = nv[u][q];
                xn[u] = readlane_f64(next_in_top ? nv[u][TOP] : nv[u][TOP > 0 ? TOP - 1 : 0], ln);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < G; ++u) {
#pragma unroll
                for (int q = 0; q <= TOP; ++q) acc[u & 1][q] = fma(xn[u], nv[u][q], acc[u & 1][q]);
            }
        };
        if (full > 0) {
            double cur[G][TOP + 1];
#pragma unroll
            for (int u = 0; u < G; ++u) {
#pragma unroll
                for (int q = 0; q <= TOP; ++q) cur[u][q] = p[q][u * NW];
            }
            for (int g = 1; g < full; ++g) {
                double nx[G][TOP + 1];
#pragma unroll
                for (int u = 0; u < G; ++u) {
#pragma unroll
                    for (int q = 0; q <= TOP; ++q) nx[u][q] = p[q][(G + u) * NW];
                }
                rows(cur);
#pragma unroll
                for (int u = 0; u < G; ++u) {
#pragma unroll
                    for (int q = 0; q <= TOP; ++q) cur[u][q] = nx[u][q];
                }
#pragma unroll
                for (int q = 0; q <= TOP; ++q) p[q] += G * NW;
            }
            rows(cur);
#pragma unroll
            for (int q = 0; q <= TOP; ++q) p[q] += G * NW;
        }
        if (rem > 0) {   // the last, partial group (wave-uniform guards)
            double cur[G][TOP + 1];
#pragma unroll
            for (int u = 0; u < G - 1; ++u)
                if (u < rem) {
#pragma unroll
                    for (int q = 0; q <= TOP; ++q) cur[u][q] = p[q][u * NW];
                }
#pragma unroll
            for (int u = 0; u < G - 1; ++u)
                if (u < rem) row(u, cur);
        }
#pragma unroll
        for (int q = 0; q <= TOP; ++q) dn[q] = acc[0][q] + acc[1][q];
    }
    NIN_STAMP(s, k, 2);
    if (NW > 1) {
        double *P = s.aux + ((size_t)(s.db ? (buf ^ 1) : 0) * NW + s.wave) * s.pstride;
#pragma unroll
        for (int q = 0; q <= TOP; ++q)
            if (upd[q]) P[lane + 64 * q] = dn[q];
    }
}

// ---- sparse first phase (NW > 1) ---------------------------------------------------------------------------------
// The cells around the node are the vertices and its internal faces the edges of a graph (a face row couples exactly
// its two cells).  Cells that share no face are INDEPENDENT: the column of one has no non-zero in the rows of another,
// so the Householder steps on their columns are independent too, each confined to its "front" -- the cell's own row
// and the rows of its faces, 10 rows for a cell with 3 faces at the node -- instead of sweeping all m rows.  The
// kernel picks a maximal independent set C (greedy in esup order; the cube graph of a hexahedron node and the
// truncated octahedron of a Kuhn-tetrahedra node are bipartite: half the cells), numbers C's columns first and lays
// the rows out so that the dense convention "step k pivots on row k" still holds:
//     rows 3 r .. 3 r + 2       the three pivot rows of front r (cell row, first two rows of its first face)
//     rows L0_r .. L0_r + nL_r  the rest of front r (3 nfi - 2 rows), all L blocks behind the pivot rows
//     then the rows of the other cells, of the faces between two of them, and the Neumann rows.
// Phase 1: wave w factors the fronts r = w (mod NW) on its own -- different fronts touch different rows, so there is
// not one workgroup barrier in it.  What is left is the dense problem on the columns >= 3 |C| and the rows >= 3 |C|,
// which the sweep of qr_step takes from k = 3 |C| on.  Kuhn tetrahedra: 36 steps of ~10 rows + 36 dense steps on <= 96
// rows instead of 72 steps on <= 132.  Same mathematics as before (a Householder QR with a column / row permutation).
struct Plan {               // per node, in the (not yet used) partial-dot buffers
    unsigned long long *adj;   // [ne] neighbour masks
    unsigned long long *blocked;   // cells that own a Neumann row (a non-zero of their columns outside their front)
    int32_t *nfi;              // [ne] internal faces per cell; later the per-cell face counter
    int16_t *crow;             // [ne] row of the cell row
    int16_t *lst;              // [ne] per front rank: first row of its L block
    int16_t *nl;               // [ne] per front rank: rows in its L block
    int16_t *frow;             // [nf][3] rows of a face (a Neumann face: [0] only)
    uint8_t *fi;               // [nf][2] positions of the face's cells in the esup row (0xFF: none)
};

// One front, one wave: its rows (3 pivot rows + the L block, at most 13 for a cell with 4 faces at the node) are
// taken into registers -- lane = column, as everywhere in this file -- factored there with the pivot column's entries
// broadcast by v_readlane, and written back once.
template <int CS>
__device__ __forceinline__ bool front_qr(const Sys &s, int rk, int L0, int nL) {
    constexpr int MAXR = 13;
    const int n = s.n, lane = s.lane, nr = 3 + nL;
    bool bad = false;
    double *col[CS];
    int cidx[CS];
#pragma unroll
    for (int q = 0; q < CS; ++q) {
        const int jj = lane + 64 * q;
        cidx[q] = jj < n ? n - 1 - jj : -1;              // this lane's column in slot q
        col[q] = lane_column(s, q, jj < n);
    }
    double a[MAXR][CS];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const int i = r < 3 ? 3 * rk + r : L0 + (r - 3);
#pragma unroll
        for (int q = 0; q < CS; ++q) a[r][q] = 0.0;
        if (r < nr) {
#pragma unroll
            for (int q = 0; q < CS; ++q) a[r][q] = col[q][i];
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int k = 3 * rk + t, jjk = n - 1 - k, lk = jjk & 63, qk = jjk >> 6;   // pivot column: lane lk of slot qk
        double x[MAXR], d[CS], w[CS];
        double dk = 0.0;
#pragma unroll
        for (int q = 0; q < CS; ++q) d[q] = 0.0;
#pragma unroll
        for (int r = t; r < MAXR; ++r) {
            double v = a[r][0];
#pragma unroll
            for (int q = 1; q < CS; ++q) v = qk == q ? a[r][q] : v;
            x[r] = readlane_f64(v, lk);
            dk = fma(x[r], x[r], dk);
#pragma unroll
            for (int q = 0; q < CS; ++q) d[q] = fma(x[r], a[r][q], d[q]);
        }
        const double alpha = x[t];
        bad = bad || !(dk != 0.0);                          // an all-zero pivot column (or NaN): no solution row
        const double sq = dk * fast_rsqrt(dk);
        const double beta = -copysign(sq, alpha);
        const double inv = fast_rcp(fma(fabs(alpha), sq, dk));
        const double vk = alpha - beta;
#pragma unroll
        for (int q = 0; q < CS; ++q) w[q] = cidx[q] > k ? (d[q] - beta * a[t][q]) * inv : 0.0;
#pragma unroll
        for (int r = t + 1; r < MAXR; ++r) {
#pragma unroll
            for (int q = 0; q < CS; ++q) a[r][q] = fma(-x[r], w[q], a[r][q]);
        }
#pragma unroll
        for (int q = 0; q < CS; ++q) a[t][q] = cidx[q] == k ? beta : fma(-vk, w[q], a[t][q]);   // row k of R
    }
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        const int i = r < 3 ? 3 * rk + r : L0 + (r - 3);
        if (r < nr) {
#pragma unroll
            for (int q = 0; q < CS; ++q)
                if (cidx[q] >= 3 * rk) col[q][i] = a[r][q];
        }
    }
    return bad;
}

template <int NW, int CS>
__global__ __launch_bounds__(64 * NW) void nin_gls_block_kernel(GridView g, const int32_t *__restrict__ nodes,
                                                                 int32_t count, int add_neumann,
                                                                 double *__restrict__ out, double *__restrict__ nws,
                                                                 int32_t *__restrict__ queue, int dbg) {
    extern __shared__ double smem[];
    (void)dbg;
    constexpr bool DB = NW != 4;   // partial dots double-buffered (one barrier per step) except where LDS is tight
    const int tid = threadIdx.x, nthr = 64 * NW;
    const int lane = tid & 63;
    const int wave = ufirst(tid >> 6);

    // NW > 1: nodes come off an atomic counter, CH at a time (Dirichlet nodes cost nothing, interior ones up to ~100 us:
    // a static stride leaves a ~10 % tail).  The NEXT chunk is fetched while this one is worked on; its index travels
    // through the first LDS word, ordered by the workgroup barriers that are there anyway.  NW == 1 (small systems,
    // mostly boundary nodes that are skipped at once): a plain grid stride -- there the counter itself would be the
    // bottleneck and there is no tail to speak of.
    constexpr int CH = NW == 2 ? 4 : 1;
    volatile int32_t *slot = reinterpret_cast<volatile int32_t *>(smem);
    double *const sys_lds = smem + 2;
    int within = 0;
    // NW == 1: the wave looks at 64 list entries at a time, one per lane -- Dirichlet boundary nodes (gls.pyx:165-166; on
    // an all-Dirichlet hexahedron mesh every node of this class: 280 k of them at 216^3) get their zero row right there, the
    // others are taken one after the other.  (One node per wave and round cost 0.14 ms of every launch for nothing but skips.)
    int64_t tile = (int64_t)blockIdx.x * 64;
    unsigned long long todo = 0;
    auto next_listed = [&]() -> int32_t {
        for (;;) {
            if (todo) {
                const int b = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                return (int32_t)(tile - (int64_t)gridDim.x * 64) + b;
            }
            if (tile >= count) return count;
            const int64_t i = tile + lane;
            const bool in = i < count;
            const int32_t pl = in ? (nodes ? nodes[i] : (int32_t)i) : 0;
            const int fll = in ? (int)g.flags[pl] : 0;
            const bool dirichlet = in && (fll & 1) && !(fll & 2);
            if (dirichlet) {
                const int32_t b0 = g.esup_ptr[pl], b1 = g.esup_ptr[pl + 1];
                for (int32_t j = b0; j < b1; ++j) out[j] = 0.0;
#ifdef NIN_BLOCK_STAMPS
                if ((dbg >> 8) == 0) nws[pl] = 0.0;
#else
                nws[pl] = 0.0;
#endif
            }
            todo = __ballot(in && !dirichlet);
            tile += (int64_t)gridDim.x * 64;
        }
    };
    auto next_node = [&](int32_t idx) -> int32_t {
        group_sync<NW>();          // the node is done (its LDS may be reused) and the prefetched chunk index has landed
        if (NW == 1) return next_listed();
        if (++within < CH) return idx + 1;
        within = 0;
        return ufirst(*slot) * CH;
    };
    if (NW > 1) {
        if (tid == 0) *slot = atomicAdd(queue, 1);
        group_sync<NW>();
    }
    for (int32_t idx = NW == 1 ? next_listed() : ufirst(*slot) * CH; idx < count; idx = next_node(idx)) {
        if (NW > 1 && within == 0) {
            group_sync<NW>();      // every wave holds idx: the slot may be rewritten
            if (tid == 0) *slot = atomicAdd(queue, 1);
        }
        const int32_t p = ufirst(nodes ? nodes[idx] : idx);
        const int32_t eb = ufirst(g.esup_ptr[p]), ne = ufirst(g.esup_ptr[p + 1]) - eb;
        const int32_t fb = ufirst(g.fsup_ptr[p]), nf = ufirst(g.fsup_ptr[p + 1]) - fb;
        const int fl = ufirst((int)g.flags[p]);
        const bool is_neu = (fl & 2) != 0;

        int n_if = 0;   // internal faces of the node (n_esuf == 2, gls.pyx:293-296); every wave counts them itself
        for (int f0 = 0; f0 < nf; f0 += 64) {
            const int fi = f0 + lane;
            const bool internal = fi < nf && g.face_cells[2 * (size_t)g.fsup[fb + fi] + 1] >= 0;
            n_if += __popcll(__ballot(internal));
        }
        const int n_bf = nf - n_if;
        const int n = 3 * ne + 1;
        const int m = ne + 3 * n_if + (is_neu ? n_bf : 0);
        // Dirichlet boundary node (gls.pyx:165-166), n_bface >= n_face (gls.pyx:266-267), or fewer rows than
        // unknowns next to the node value: zero row.
        if (((fl & 1) && !is_neu) || n_if == 0 || m < n - 1) {
            for (int i = tid; i < ne; i += nthr) out[eb + i] = 0.0;
#ifdef NIN_BLOCK_STAMPS
            if (tid == 0 && (dbg >> 8) == 0) nws[p] = 0.0;
#else
            if (tid == 0) nws[p] = 0.0;
#endif
            continue;
        }
        Sys s;
        const int ld = m | 1;   // odd pitch: the 64 lanes of a row access hit 64 different 8-byte bank pairs
        s.A = sys_lds;
        s.aux = sys_lds + (size_t)(n + 1) * ld;
        s.n = n; s.m = m; s.ld = ld; s.lane = lane; s.wave = wave;
        s.pstride = n; s.db = DB;
#ifdef NIN_BLOCK_STAMPS
        s.stamps = (dbg >> 8) == p ? nws : nullptr;   // NIN_GLS_BLOCK_DEBUG = node << 8: that node's block records
#else
        s.stamps = nullptr;
#endif
        NIN_STAMP(s, 200, 0);                                    // coarse timeline of the node: K = 200, 201
        constexpr int AUXW = NW == 1 ? 2 : (DB ? 2 : 1) * NW;   // the partial-dot buffers: AUXW * n doubles
        int32_t *cells = reinterpret_cast<int32_t *>(s.aux + AUXW * n);
        uint8_t *cpos = reinterpret_cast<uint8_t *>(cells + 2 * ((ne + 1) >> 1));   // [ne] column block of a cell (SPARSE)
        constexpr bool SPARSE = NW > 1;
        volatile int32_t *sing = slot + 2, *n1_word = slot + 3;

        // the plan lives in the partial-dot buffers, which nothing uses before the first dense step
        Plan pl;
        bool plan_fits = false;
        if (SPARSE) {
            char *pa = reinterpret_cast<char *>(s.aux);
            pl.adj = reinterpret_cast<unsigned long long *>(pa); pa += 8 * (size_t)ne;
            pl.blocked = reinterpret_cast<unsigned long long *>(pa); pa += 8;
            pl.nfi = reinterpret_cast<int32_t *>(pa); pa += 4 * (size_t)((ne + 1) & ~1);
            pl.crow = reinterpret_cast<int16_t *>(pa); pa += 2 * (size_t)((ne + 3) & ~3);
            pl.lst = reinterpret_cast<int16_t *>(pa); pa += 2 * (size_t)((ne + 3) & ~3);
            pl.nl = reinterpret_cast<int16_t *>(pa); pa += 2 * (size_t)((ne + 3) & ~3);
            pl.frow = reinterpret_cast<int16_t *>(pa); pa += 6 * (size_t)((nf + 3) & ~3);
            pl.fi = reinterpret_cast<uint8_t *>(pa); pa += 2 * (size_t)nf;
            plan_fits = ne <= 64 && nf <= 64 && (size_t)(pa - reinterpret_cast<char *>(s.aux)) <= (size_t)AUXW * n * 8;
        }

        for (int i = tid; i < (n + 1) * ld; i += nthr) s.A[i] = 0.0;
        for (int i = tid; i < ne; i += nthr) cells[i] = g.esup[eb + i];
        if (SPARSE) {
            if (plan_fits)
                for (int i = tid; i < ne; i += nthr) { pl.adj[i] = 0ull; pl.nfi[i] = 0; }
            if (tid == 0) { *sing = 0; *n1_word = 0; if (plan_fits) *pl.blocked = 0ull; }
        }
        group_sync<NW>();

        const double xv0 = g.coords[3 * (size_t)p + 0], xv1 = g.coords[3 * (size_t)p + 1],
                     xv2 = g.coords[3 * (size_t)p + 2];
        if (SPARSE && plan_fits) {
            // ---- the plan, wave 0: a lane per face builds the cell graph, then a lane per cell and a lane per face place
            //      columns and rows (ballots, one scan, shuffles: no serial pass over LDS) -----------------------------------
            if (wave == 0) {
                const bool face_lane = lane < nf;
                int Ia = 0, Ib = 0xFF;
                if (face_lane) {
                    const size_t f = (size_t)g.fsup[fb + lane];
                    const int ca = g.face_cells[2 * f], cb = g.face_cells[2 * f + 1];
                    int ib = 0;
                    for (int q = 0; q < ne; ++q) {
                        const int cq = cells[q];
                        Ia = cq == ca ? q : Ia;
                        ib = cq == cb ? q : ib;
                    }
                    Ib = cb >= 0 ? ib : 0xFF;
                    pl.fi[2 * lane] = (uint8_t)Ia;
                    pl.fi[2 * lane + 1] = (uint8_t)Ib;
                    if (cb < 0 && is_neu) atomicOr(pl.blocked, 1ull << Ia);
                    if (cb >= 0) {
                        atomicOr(&pl.adj[Ia], 1ull << Ib);
                        atomicOr(&pl.adj[Ib], 1ull << Ia);
                        atomicAdd(&pl.nfi[Ia], 1);
                        atomicAdd(&pl.nfi[Ib], 1);
                    }
                }
                group_sync<1>();
                // lane c = cell c
                const bool cell_lane = lane < ne;
                const unsigned long long my_adj = cell_lane ? pl.adj[lane] : 0ull;
                const int my_nfi = cell_lane ? pl.nfi[lane] : 0;
                const unsigned long long blocked = *pl.blocked;
                const unsigned long long elig = __ballot(cell_lane && my_nfi >= 1 && my_nfi <= 4 && !((blocked >> lane) & 1ull));
                unsigned long long chosen = 0ull;
                for (int c = 0; c < ne; ++c) {            // greedy maximal independent set, esup order (scalar work)
                    const unsigned long long ac = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(my_adj >> 32), c) << 32) |
                                                  (unsigned)__builtin_amdgcn_readlane((int)my_adj, c);
                    if (((elig >> c) & 1ull) && !(ac & chosen)) chosen |= 1ull << c;
                }
                const unsigned long long below = (1ull << lane) - 1ull;
                const unsigned long long cellmask = ne >= 64 ? ~0ull : ((1ull << ne) - 1ull);
                const int n1 = __popcll(chosen), R0p = 3 * n1;
                const bool is_ch = ((chosen >> lane) & 1ull) != 0;
                const int rank = __popcll(chosen & below), other = __popcll(~chosen & cellmask & below);
                const int rest = is_ch ? 3 * my_nfi - 2 : 0;
                int scan = rest;                           // inclusive scan of the L-block lengths over the cell lanes
#pragma unroll
                for (int sh = 1; sh < 64; sh <<= 1) {
                    const int up = __shfl_up(scan, sh);
                    if (lane >= sh) scan += up;
                }
                const int total_rest = __builtin_amdgcn_readlane(scan, 63);
                const int my_lst = R0p + scan - rest;       // first row of this cell's L block (chosen lanes)
                const int n_other = ne - n1;
                if (cell_lane) {
                    pl.crow[lane] = (int16_t)(is_ch ? 3 * rank : R0p + total_rest + other);
                    cpos[lane] = (uint8_t)(is_ch ? rank : n1 + other);
                    if (is_ch) { pl.lst[rank] = (int16_t)my_lst; pl.nl[rank] = (int16_t)rest; }
                }
                // lane fi = face fi: the front it belongs to (its chosen cell), its slot there, its rows
                const bool internal = face_lane && Ib != 0xFF;
                const int key = !internal ? -1 : (((chosen >> Ia) & 1ull) ? Ia : (((chosen >> Ib) & 1ull) ? Ib : -1));
                int slot_in_front = 0;
                for (unsigned long long rem = chosen; rem; rem &= rem - 1ull) {
                    const int c = __ffsll((long long)rem) - 1;
                    const unsigned long long mm = __ballot(key == c);
                    if (key == c) slot_in_front = __popcll(mm & below);
                }
                const int key_lst = __shfl(my_lst, key >= 0 ? key : 0), key_rank = __shfl(rank, key >= 0 ? key : 0);
                const unsigned long long m_free = __ballot(internal && key < 0), m_neu = __ballot(face_lane && !internal && is_neu);
                const int base_free = R0p + total_rest + n_other, base_neu = base_free + 3 * __popcll(m_free);
                if (face_lane) {
                    int16_t *fr = pl.frow + 3 * lane;
                    if (internal && key >= 0) {
                        if (slot_in_front == 0) {
                            fr[0] = (int16_t)(3 * key_rank + 1); fr[1] = (int16_t)(3 * key_rank + 2); fr[2] = (int16_t)key_lst;
                        } else {
                            const int base = key_lst + 1 + 3 * (slot_in_front - 1);
                            fr[0] = (int16_t)base; fr[1] = (int16_t)(base + 1); fr[2] = (int16_t)(base + 2);
                        }
                    } else if (internal) {
                        const int base = base_free + 3 * __popcll(m_free & below);
                        fr[0] = (int16_t)base; fr[1] = (int16_t)(base + 1); fr[2] = (int16_t)(base + 2);
                    } else if (is_neu) {
                        fr[0] = (int16_t)(base_neu + __popcll(m_neu & below));
                    }
                }
                if (lane == 0) *n1_word = n1;
            }
            group_sync<NW>();
            // ---- assembly through the plan's row / column maps ----------------------------------------------------------
            if (wave == 1) {
                // cell rows: [x_K - x_v] on the cell's own block, 1 in the last column (gls.pyx:269-281)
                for (int i = lane; i < ne; i += 64) {
                    const size_t c = (size_t)cells[i];
                    double *row = s.A + pl.crow[i];
                    const int cb3 = 3 * cpos[i];
                    row[(cb3 + 0) * ld] = g.centroids[3 * c + 0] - xv0;
                    row[(cb3 + 1) * ld] = g.centroids[3 * c + 1] - xv1;
                    row[(cb3 + 2) * ld] = g.centroids[3 * c + 2] - xv2;
                    row[(n - 1) * ld] = 1.0;
                }
            }
            if (wave == 0) {
                for (int f0 = 0; f0 < nf; f0 += 64) {
                    const int fi = f0 + lane;
                    if (fi >= nf) continue;
                    const size_t f = (size_t)g.fsup[fb + fi];
                    const int Ia = pl.fi[2 * fi], Ib = pl.fi[2 * fi + 1];
                    const int16_t *fr = pl.frow + 3 * fi;
                    const double N0 = g.face_normal[3 * f + 0], N1 = g.face_normal[3 * f + 1], N2 = g.face_normal[3 * f + 2];
                    if (Ib != 0xFF) {
                        const int ca = cells[Ia], cb = cells[Ib];
                        const double T0 = xv0 - g.face_center[3 * f + 0], T1 = xv1 - g.face_center[3 * f + 1],
                                     T2 = xv2 - g.face_center[3 * f + 2];
                        // T_sj2 = N x T_sj1, tau = |T_sj2|^(-eta), eta = max diff_mag of the two cells (gls.pyx:304-318)
                        const double U0 = N1 * T2 - N2 * T1, U1 = N2 * T0 - N0 * T2, U2 = N0 * T1 - N1 * T0;
                        const double da = g.diff_mag[ca], db = g.diff_mag[cb];
                        double eta = 0.0;
                        eta = da > eta ? da : eta;
                        eta = db > eta ? db : eta;
                        const double tj = glsmath::face_tau(sqrt(U0 * U0 + U1 * U1 + U2 * U2), eta);
                        const double *Ka = g.perm + 9 * (size_t)ca, *Kb = g.perm + 9 * (size_t)cb;
                        double *ra = s.A + (size_t)(3 * cpos[Ia]) * ld, *rb = s.A + (size_t)(3 * cpos[Ib]) * ld;
                        const int r0 = fr[0], r1 = fr[1], r2 = fr[2];   // rows: K N, T1, tau T2
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const double nLa = Ka[c * 3 + 0] * N0 + Ka[c * 3 + 1] * N1 + Ka[c * 3 + 2] * N2;  // row c of K . N
                            const double nLb = Kb[c * 3 + 0] * N0 + Kb[c * 3 + 1] * N1 + Kb[c * 3 + 2] * N2;
                            const double t1 = c == 0 ? T0 : (c == 1 ? T1 : T2);
                            const double u = tj * (c == 0 ? U0 : (c == 1 ? U1 : U2));
                            ra[c * ld + r0] = -nLa; rb[c * ld + r0] = nLb;
                            ra[c * ld + r1] = -t1;  rb[c * ld + r1] = t1;
                            ra[c * ld + r2] = -u;   rb[c * ld + r2] = u;
                        }
                    } else if (is_neu) {  // set_neumann_rows, gls.pyx:394-416 (its RHS column is never read back)
                        const double *Ka = g.perm + 9 * (size_t)cells[Ia];
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            s.A[(size_t)(3 * cpos[Ia] + c) * ld + fr[0]] = -(Ka[c * 3 + 0] * N0 + Ka[c * 3 + 1] * N1 + Ka[c * 3 + 2] * N2);
                    }
                }
            }
        } else {
        if (SPARSE)
            for (int i = tid; i < ne; i += nthr) cpos[i] = (uint8_t)i;
        if (wave == (NW > 1 ? 1 : 0)) {
            // cell rows: [x_K - x_v] on the cell's own block, 1 in the last column (gls.pyx:269-281)
            for (int i = lane; i < ne; i += 64) {
                const size_t c = (size_t)cells[i];
                double *row = s.A + i;
                row[(3 * i + 0) * ld] = g.centroids[3 * c + 0] - xv0;
                row[(3 * i + 1) * ld] = g.centroids[3 * c + 1] - xv1;
                row[(3 * i + 2) * ld] = g.centroids[3 * c + 2] - xv2;
                row[(n - 1) * ld] = 1.0;
            }
        }
        if (wave == 0) {
            int if_base = 0, bf_base = 0;
            for (int f0 = 0; f0 < nf; f0 += 64) {
                const int fi = f0 + lane;
                const bool valid = fi < nf;
                const size_t f = valid ? (size_t)g.fsup[fb + fi] : 0;
                const int ca = valid ? g.face_cells[2 * f] : 0, cb = valid ? g.face_cells[2 * f + 1] : -1;
                const bool internal = valid && cb >= 0;
                const bool bface = valid && cb < 0;
                const unsigned long long mi = __ballot(internal), mb = __ballot(bface);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (internal) {
                    const int row = ne + 3 * (if_base + __popcll(mi & below));
                    const double N0 = g.face_normal[3 * f + 0], N1 = g.face_normal[3 * f + 1], N2 = g.face_normal[3 * f + 2];
                    const double T0 = xv0 - g.face_center[3 * f + 0], T1 = xv1 - g.face_center[3 * f + 1],
                                 T2 = xv2 - g.face_center[3 * f + 2];
                    // T_sj2 = N x T_sj1, tau = |T_sj2|^(-eta), eta = max diff_mag of the two cells (gls.pyx:304-318)
                    const double U0 = N1 * T2 - N2 * T1, U1 = N2 * T0 - N0 * T2, U2 = N0 * T1 - N1 * T0;
                    const double da = g.diff_mag[ca], db = g.diff_mag[cb];
                    double eta = 0.0;
                    eta = da > eta ? da : eta;
                    eta = db > eta ? db : eta;
                    const double tj = glsmath::face_tau(sqrt(U0 * U0 + U1 * U1 + U2 * U2), eta);
                    const double *Ka = g.perm + 9 * (size_t)ca, *Kb = g.perm + 9 * (size_t)cb;
                    int Ia = 0, Ib = 0;
                    for (int q = 0; q < ne; ++q) {
                        const int cq = cells[q];
                        Ia = cq == ca ? q : Ia;
                        Ib = cq == cb ? q : Ib;
                    }
                    double *ra = s.A + (3 * Ia) * ld + row, *rb = s.A + (3 * Ib) * ld + row;   // rows row.. row+2 = K N, T1, tau T2
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const double nLa = Ka[c * 3 + 0] * N0 + Ka[c * 3 + 1] * N1 + Ka[c * 3 + 2] * N2;  // row c of K . N
                        const double nLb = Kb[c * 3 + 0] * N0 + Kb[c * 3 + 1] * N1 + Kb[c * 3 + 2] * N2;
                        const double t1 = c == 0 ? T0 : (c == 1 ? T1 : T2);
                        const double u = tj * (c == 0 ? U0 : (c == 1 ? U1 : U2));
                        ra[c * ld + 0] = -nLa; rb[c * ld + 0] = nLb;
                        ra[c * ld + 1] = -t1;  rb[c * ld + 1] = t1;
                        ra[c * ld + 2] = -u;   rb[c * ld + 2] = u;
                    }
                }
                if (bface && is_neu) {  // set_neumann_rows, gls.pyx:394-416 (its RHS column is never read back)
                    const int row = ne + 3 * n_if + bf_base + __popcll(mb & below);
                    const double N0 = g.face_normal[3 * f + 0], N1 = g.face_normal[3 * f + 1], N2 = g.face_normal[3 * f + 2];
                    const double *Ka = g.perm + 9 * (size_t)ca;
                    int Ia = 0;
                    for (int q = 0; q < ne; ++q) Ia = cells[q] == ca ? q : Ia;
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        s.A[(3 * Ia + c) * ld + row] = -(Ka[c * 3 + 0] * N0 + Ka[c * 3 + 1] * N1 + Ka[c * 3 + 2] * N2);
                }
                if_base += __popcll(mi);
                bf_base += __popcll(mb);
            }
        }
        }
        group_sync<NW>();
        NIN_STAMP(s, 200, 1);                                    // plan + assembly done

        // ---- phase 1 (SPARSE): the fronts of the independent cells, a wave each, no barrier between them -------------
        int R0 = 0;
        if (SPARSE) {
            const int n1 = ufirst(*n1_word);
            R0 = 3 * n1;
            for (int rk = wave; rk < n1; rk += NW)
                if (front_qr<CS>(s, rk, pl.lst[rk], pl.nl[rk])) *sing = 1;
            group_sync<NW>();
            // the dense phase only ever publishes the live columns (reversed indices 0 .. n - 1 - R0): packed to that
            // length, two buffers may fit where one of length n did -- one barrier per step instead of two
            const int nlive = n - R0;
            if (!DB && 2 * NW * nlive <= AUXW * n) { s.db = true; s.pstride = nlive; }
        }
        NIN_STAMP(s, 200, 2);                                    // fronts done

        // ---- Householder QR of the remaining columns R0 .. n-2, the last one carried along ---------------------------
        bool singular = false;
        double dn[CS], rkeep[CS];
#pragma unroll
        for (int q = 0; q < CS; ++q) { dn[q] = 0.0; rkeep[q] = 0.0; }
        {
            const int top0 = (n - 1 - R0) >> 6;
#define NIN_TOP(T) if (CS > T && top0 == T) first_dots<NW, CS, T>(s, R0, dn)
            NIN_TOP(0); NIN_TOP(1); NIN_TOP(2); NIN_TOP(3);
#undef NIN_TOP
        }
        int buf = 0;
        if (NW > 1) {
            double *P = s.aux + (size_t)wave * s.pstride;
#pragma unroll
            for (int q = 0; q < CS; ++q)
                if (lane + 64 * q < s.pstride) P[lane + 64 * q] = dn[q];
            group_sync<NW>();
        }
        NIN_STAMP(s, 200, 3);                                    // first dots published
        for (int k = R0; k < n - 1; ++k) {
            if (k > R0 && wave == (k - 1) % NW) {   // row k-1 of R: every wave has finished reading it as a pivot row
#pragma unroll
                for (int q = 0; q < CS; ++q) {
                    const int jj = lane + 64 * q;
                    if (jj <= n - k) s.A[(n - 1 - jj) * ld + (k - 1)] = rkeep[q];
                }
            }
            const int top = (n - 1 - k) >> 6;
#define NIN_TOP(T) if (CS > T && top == T) qr_step<NW, CS, T>(s, k, buf, dn, rkeep, singular)
            NIN_TOP(0); NIN_TOP(1); NIN_TOP(2); NIN_TOP(3);
#undef NIN_TOP
            if (NW > 1) {
                group_sync<NW>();
                if (s.db) buf ^= 1;
            }
            NIN_STAMP(s, k, 3);
        }
        if (wave == (n - 2) % NW) {
            if (lane <= 1) s.A[(n - 1 - lane) * ld + (n - 2)] = rkeep[0];
        }
        group_sync<NW>();
        NIN_STAMP(s, 201, 0);                                    // dense steps done

        // ---- tail, wave 0: R y = c~(0:n-1) by columns (lane = row), then the residual ------------------------
        if (wave == 0) {
            double ct[CS];
#pragma unroll
            for (int q = 0; q < CS; ++q) {
                const int i = lane + 64 * q;
                ct[q] = i < n - 1 ? s.A[(n - 1) * ld + i] : 0.0;
            }
            double *y = s.aux, *wrow = s.aux + n;
            // column k of R is read one step ahead of its use: the loop is a chain of LDS latencies otherwise
            double coln[CS];
            auto load_col = [&](int k, double (&c)[CS]) {
                const int ks = k >> 6;
#pragma unroll
                for (int q = 0; q < CS; ++q) {
                    const int i = lane + 64 * q;
                    c[q] = (q <= ks && i <= k) ? s.A[k * ld + i] : 0.0;
                }
            };
            load_col(n - 2, coln);
            for (int k = n - 2; k >= R0; --k) {
                const int ks = k >> 6, kl = k & 63;
                double col[CS];
#pragma unroll
                for (int q = 0; q < CS; ++q) col[q] = coln[q];
                if (k > R0) load_col(k - 1, coln);
                double rkk = 1.0, ck = 0.0;
#pragma unroll
                for (int q = 0; q < CS; ++q) {
                    if (q == ks) {
                        rkk = readlane_f64(col[q], kl);
                        ck = readlane_f64(ct[q], kl);
                    }
                }
                const double yk = ck * fast_rcp(rkk);
#pragma unroll
                for (int q = 0; q < CS; ++q) ct[q] = fma(-yk, col[q], ct[q]);
                if (lane == 0) y[k] = yk;
            }
            if (SPARSE && R0 > 0) {
                // the columns of the phase-1 fronts: R couples a front's three rows only to each other (and to the
                // columns >= R0, already taken off the right-hand side above) -- all fronts at once, a lane each
                double *ctl = s.aux + 2 * n;       // (AUXW >= 4 for NW > 1: room behind y and the weight row)
#pragma unroll
                for (int q = 0; q < CS; ++q) {
                    const int i = lane + 64 * q;
                    if (i < R0) ctl[i] = ct[q];
                }
                group_sync<1>();
                if (3 * lane < R0) {
                    const int r0 = 3 * lane;
                    const double *A0 = s.A + (size_t)r0 * ld, *A1 = A0 + ld, *A2 = A1 + ld;   // columns r0, r0 + 1, r0 + 2
                    const double y2 = ctl[r0 + 2] * fast_rcp(A2[r0 + 2]);
                    const double y1 = fma(-A2[r0 + 1], y2, ctl[r0 + 1]) * fast_rcp(A1[r0 + 1]);
                    const double y0 = fma(-A2[r0], y2, fma(-A1[r0], y1, ctl[r0])) * fast_rcp(A0[r0]);
                    y[r0] = y0; y[r0 + 1] = y1; y[r0 + 2] = y2;
                }
            }
            double rr = 0.0;
            for (int i = n - 1 + lane; i < m; i += 64) {
                const double c = s.A[(n - 1) * ld + i];
                rr = fma(c, c, rr);
            }
            rr = wave_sum(rr);
            group_sync<1>();
            singular = singular || !(rr > 0.0) || (SPARSE && *sing != 0);
            const double irr = 1.0 / rr;
            for (int i = lane; i < ne; i += 64) {
                const double *row = s.A;   // the cell rows were consumed by the QR: rebuild x_K - x_v as assembled
                (void)row;
                const size_t c = (size_t)cells[i];
                const double d0 = g.centroids[3 * c + 0] - xv0, d1 = g.centroids[3 * c + 1] - xv1,
                             d2 = g.centroids[3 * c + 2] - xv2;
                const int ci = SPARSE ? 3 * (int)cpos[i] : 3 * i;   // the cell's column block
                const double r = 1.0 - (d0 * y[ci + 0] + d1 * y[ci + 1] + d2 * y[ci + 2]);
                wrow[i] = singular ? 0.0 : r * irr;
            }
            group_sync<1>();
            // gls.pyx:470-472: neumann_ws = solution entry (n-1, n_elem-1), i.e. the LAST cell's weight
            const double nwv = is_neu ? wrow[ne - 1] : 0.0;
            const double add = add_neumann ? nwv : 0.0;
            for (int i = lane; i < ne; i += 64) out[eb + i] = wrow[i] + add;
            NIN_STAMP(s, 201, 1);                                // weights stored
#ifdef NIN_BLOCK_STAMPS
            if (lane == 0 && (dbg >> 8) == 0) nws[p] = nwv;
#else
            if (lane == 0) nws[p] = nwv;
#endif
        }
    }
}

template <int NW, int CS>
int launch_block(const GridView &g, const int32_t *nodes, int32_t count, int32_t lds_bytes, int add_neumann,
                 double *out, double *nws, int32_t *queue, hipStream_t stream) {
    auto kern = nin_gls_block_kernel<NW, CS>;
    static const int dbg = getenv("NIN_GLS_BLOCK_DEBUG") ? atoi(getenv("NIN_GLS_BLOCK_DEBUG")) : 0;
    // per instantiation AND per device (the attribute belongs to the (function, device) pair and one process may
    // drive several GPUs): raise the dynamic-LDS limit once per new maximum
    static int lds_allowed_dev[64] = {};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    int &lds_allowed = lds_allowed_dev[dev_id & 63];
    if (lds_allowed == 0) lds_allowed = 48 * 1024;
    if (lds_bytes > lds_allowed) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return -3;
        lds_allowed = lds_bytes;
    }
    int per_cu = (160 * 1024) / (lds_bytes < 1024 ? 1024 : lds_bytes);
    const int wave_cap = 32 / NW;
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = (int64_t)256 * per_cu * 2;
    if (blocks > count) blocks = count;
    if (NW == 1 && blocks > ((int64_t)count + 63) / 64) blocks = ((int64_t)count + 63) / 64;   // (a wave looks at 64 list entries per round)
    static const int max_blocks = getenv("NIN_GLS_BLOCK_MAX_BLOCKS") ? atoi(getenv("NIN_GLS_BLOCK_MAX_BLOCKS")) : 0;
    if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;   // diagnostic: occupancy experiments
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * NW), (size_t)lds_bytes, stream, g, nodes, count, add_neumann,
                       out, nws, queue, dbg);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int NW>
int launch_block_cs(int cs, const GridView &g, const int32_t *nodes, int32_t count, int32_t lds_bytes, int add_neumann,
                    double *out, double *nws, int32_t *queue, hipStream_t stream) {
    switch (cs) {
        case 1: return launch_block<NW, 1>(g, nodes, count, lds_bytes, add_neumann, out, nws, queue, stream);
        case 2: return launch_block<NW, 2>(g, nodes, count, lds_bytes, add_neumann, out, nws, queue, stream);
        case 3: return launch_block<NW, 3>(g, nodes, count, lds_bytes, add_neumann, out, nws, queue, stream);
        case 4: return launch_block<NW, 4>(g, nodes, count, lds_bytes, add_neumann, out, nws, queue, stream);
    }
    return -5;
}

}  // namespace

int launch_gls_block(const GridView &g, const int32_t *nodes, int32_t count, int32_t waves, int32_t col_slots,
                     int32_t lds_bytes, int add_neumann, double *out, double *nws, int32_t *queue, hipStream_t stream) {
    if (count <= 0) return 0;
    switch (waves) {
        case 1: return launch_block_cs<1>(col_slots, g, nodes, count, lds_bytes, add_neumann, out, nws, queue, stream);
        case 2: return launch_block_cs<2>(col_slots, g, nodes, count, lds_bytes, add_neumann, out, nws, queue, stream);
        case 4: return launch_block_cs<4>(col_slots, g, nodes, count, lds_bytes, add_neumann, out, nws, queue, stream);
        case 8: return launch_block_cs<8>(col_slots, g, nodes, count, lds_bytes, add_neumann, out, nws, queue, stream);
    }
    return -1;
}

const char *kernel_name_gls_block() { return "nin_gls_block_kernel"; }

}  // namespace nin
