// kernels_gls.hip -- GLS weights, gfx950: the fallback for nodes no other kernel takes (more cells than kernels_gls_mfg.hip's 64, a
// system that does not fit the LDS of one CU: a handful of nodes of a large unstructured mesh, none of a structured one): one node per
// WORKGROUP of 8 or 16 wavefronts, the dense system in a global-memory scratch slot.
//
// What the reference does per node (gls.pyx:161-219): assemble the dense m x n system
//   M = [ d_i^T on block i | 1 ]        n_elem rows    (x_K - x_v, gls.pyx:269-281)
//       [ -B_f,a  ... +B_f,b | 0 ]      3 rows per internal face: K N, T1, tau T2 (gls.pyx:293-356)
//       [ -K N on the owner  | 0 ]      1 row per boundary face of a Neumann node (gls.pyx:394-416)
// with n = 3 n_elem + 1 unknowns (a gradient per cell + the node value), solve it for n_elem (+1)
// right-hand sides with LAPACK dgels and keep ONLY row n-1 of the solution (gls.pyx:466-472).
//
// What this kernel does: the same Householder QR of the same matrix (so the same conditioning and
// the same kappa*eps-level agreement with dgels -- never normal equations), but it exploits that
// row n-1 of the pseudo-inverse against the unit right-hand sides e_i is r_i / (r.r), where r is
// the least-squares residual of M's last column against its first n-1 columns:
// one QR, no right-hand sides, one back-application of Q.  Zero rows (the empty row triples the
// reference leaves for boundary faces) are dropped: they change nothing in a Householder QR.
//
// Mapping: rows across the 64 lanes (row r lives in lane r%64, slot r/64), the matrix column-major.  EVERY wavefront of the team reads
// the pivot column and forms the reflector (the same arithmetic on the same words: the same bits), then takes every TW-th group of four
// trailing columns -- the reflector stays in registers while it is applied, dot products are DPP row reductions plus four readlanes --,
// one workgroup barrier per reflector.  (Until round 4 one WAVEFRONT ran a node alone: a 66-cell node -- 363 x 199 -- took 12.4 ms,
// the long pole of a whole 2 M-cell launch; the team takes ~1 ms.)  Nodes are pre-binned by system size (device_grid.hpp) so each
// launch has one rows-per-lane count.
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

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes, result in every lane.  All 64 lanes must be active.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror  -> every lane holds its 16-lane row sum
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// Orders this wave's LDS / scratch traffic: a lane may read what another lane of the wave wrote.
template <bool LDS>
__device__ __forceinline__ void wave_sync() {
    if (LDS) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

__device__ __forceinline__ int ufirst(int v) { return __builtin_amdgcn_readfirstlane(v); }

constexpr int JB = 4;  // columns updated together (independent reductions in flight)

template <int RPL, int TW>
__global__ __launch_bounds__(64 * TW) void nin_gls_team_kernel(GridView g, const int32_t *__restrict__ nodes, int32_t count, int add_neumann,
                                                              double *__restrict__ out, double *__restrict__ nws, double *scratch,
                                                              long long scratch_stride) {
    constexpr bool LDS = false;
    const int tid = threadIdx.x, nthr = 64 * TW;
    const int lane = tid & 63;
    const int wave = ufirst(tid >> 6);
    double *base = scratch + (size_t)blockIdx.x * scratch_stride;

    for (int32_t idx = blockIdx.x; idx < count; idx += gridDim.x) {
        const int32_t p = ufirst(nodes ? nodes[idx] : idx);
        const int32_t eb = ufirst(g.esup_ptr[p]), ne = ufirst(g.esup_ptr[p + 1]) - eb;
        const int32_t fb = ufirst(g.fsup_ptr[p]), nf = ufirst(g.fsup_ptr[p + 1]) - fb;
        const int fl = ufirst((int)g.flags[p]);
        const bool is_neu = (fl & 2) != 0;

        // internal faces of the node (n_esuf == 2, gls.pyx:293-296)
        int n_if = 0;
        for (int f0 = 0; f0 < nf; f0 += 64) {
            const int fi = f0 + lane;
            const bool internal = fi < nf && g.face_cells[2 * (size_t)g.fsup[fb + fi] + 1] >= 0;
            n_if += __popcll(__ballot(internal));
        }
        const int n_bf = nf - n_if;
        const int n = 3 * ne + 1;                              // columns, the last one is c
        const int m = ne + 3 * n_if + (is_neu ? n_bf : 0);     // rows actually populated
        // Dirichlet boundary node (gls.pyx:165-166), n_bface >= n_face (gls.pyx:266-267: the system stays empty and
        // dgels returns an all-zero row n-1), or fewer rows than unknowns next to the node value: zero row -- the
        // same rule as kernels_gls_block.hip, so a node gets the same answer whichever kernel its size class runs on.
        if (((fl & 1) && !is_neu) || n_if == 0 || m < n - 1) {
            for (int i = tid; i < ne; i += nthr) out[eb + i] = 0.0;
            if (tid == 0) nws[p] = 0.0;
            continue;
        }
        const int ld = m;
        int32_t *cells = reinterpret_cast<int32_t *>(base);   // [ne] (padded to an even count)
        double *tau = base + ((ne + 1) >> 1);                  // [n]
        double *A = tau + n;                                   // [ld * n] column-major

        for (int i = tid; i < ne; i += nthr) cells[i] = g.esup[eb + i];
        for (int i = tid; i < ld * n; i += nthr) A[i] = 0.0;
        __syncthreads();
        if (wave == 0) {   // assembly: one wavefront (a lane per cell / face: little work against the factorisation)

        const double xv0 = g.coords[3 * (size_t)p + 0], xv1 = g.coords[3 * (size_t)p + 1],
                     xv2 = g.coords[3 * (size_t)p + 2];
        // cell rows: [x_K - x_v] on the cell's own block, 1 in the last column
        for (int i = lane; i < ne; i += 64) {
            const size_t c = (size_t)cells[i];
            A[i + (3 * i + 0) * ld] = g.centroids[3 * c + 0] - xv0;
            A[i + (3 * i + 1) * ld] = g.centroids[3 * c + 1] - xv1;
            A[i + (3 * i + 2) * ld] = g.centroids[3 * c + 2] - xv2;
            A[i + (n - 1) * ld] = 1.0;
        }
        // face rows
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
                const double tj = pow(sqrt(U0 * U0 + U1 * U1 + U2 * U2), -eta);
                const double *Ka = g.perm + 9 * (size_t)ca, *Kb = g.perm + 9 * (size_t)cb;
                int Ia = 0, Ib = 0;
                for (int q = 0; q < ne; ++q) {
                    const int cq = cells[q];
                    Ia = cq == ca ? q : Ia;
                    Ib = cq == cb ? q : Ib;
                }
                double *ra = A + row + (size_t)(3 * Ia) * ld, *rb = A + row + (size_t)(3 * Ib) * ld;
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
                    A[row + (size_t)(3 * Ia + c) * ld] = -(Ka[c * 3 + 0] * N0 + Ka[c * 3 + 1] * N1 + Ka[c * 3 + 2] * N2);
            }
            if_base += __popcll(mi);
            bf_base += __popcll(mb);
        }
        }
        __syncthreads();

        // ---- Householder QR of the first n-1 columns, applied to the last one as it goes: every wavefront forms reflector k from the
        //      pivot column (nobody writes it during the step), wavefront w applies it to the column groups w, w + TW, ..; one barrier;
        //      then wavefront 0 leaves v_k in the column for the back-application (R itself is never read again) ----------
        bool singular = false;
        for (int k = 0; k < n - 1; ++k) {
            double v[RPL];
            double ss = 0.0;
#pragma unroll
            for (int c = 0; c < RPL; ++c) {
                const int r = lane + 64 * c;
                v[c] = (r >= k && r < m) ? A[r + (size_t)k * ld] : 0.0;
                ss += (r > k) ? v[c] * v[c] : 0.0;
            }
            ss = wave_sum(ss);
            const double alpha = A[k + (size_t)k * ld];
            double tk = 0.0;
            if (ss != 0.0) {  // dlarfg: beta = -sign(alpha) |(alpha, x)|, tau = (beta-alpha)/beta, v = x/(alpha-beta)
                const double beta = -copysign(sqrt(alpha * alpha + ss), alpha);
                tk = (beta - alpha) / beta;
                const double sc = 1.0 / (alpha - beta);
#pragma unroll
                for (int c = 0; c < RPL; ++c) {
                    const int r = lane + 64 * c;
                    v[c] = r > k ? v[c] * sc : (r == k ? 1.0 : 0.0);
                }
            } else {
                singular = singular || (alpha == 0.0);
            }
            if (tid == 0) tau[k] = tk;
            if (tk != 0.0) {
                for (int j0 = k + 1 + JB * wave; j0 < n; j0 += JB * TW) {
                    double a[JB][RPL], s[JB];
#pragma unroll
                    for (int jj = 0; jj < JB; ++jj) {
                        s[jj] = 0.0;
                        const int j = j0 + jj;
#pragma unroll
                        for (int c = 0; c < RPL; ++c) {
                            const int r = lane + 64 * c;
                            a[jj][c] = (j < n && r >= k && r < m) ? A[r + (size_t)j * ld] : 0.0;
                            s[jj] += v[c] * a[jj][c];
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < JB; ++jj) s[jj] = wave_sum(s[jj]) * tk;
#pragma unroll
                    for (int jj = 0; jj < JB; ++jj) {
                        const int j = j0 + jj;
#pragma unroll
                        for (int c = 0; c < RPL; ++c) {
                            const int r = lane + 64 * c;
                            if (j < n && r >= k && r < m) A[r + (size_t)j * ld] = a[jj][c] - s[jj] * v[c];
                        }
                    }
                }
            }
            __syncthreads();
            if (wave == 0 && tk != 0.0) {
#pragma unroll
                for (int c = 0; c < RPL; ++c) {
                    const int r = lane + 64 * c;
                    if (r > k && r < m) A[r + (size_t)k * ld] = v[c];
                }
            }
        }
        __syncthreads();
        if (wave == 0) {
        // ---- residual of the last column: r = Q [0; c~(n-1:m)], weights = r(0:ne) / (r.r) ----------
        double z[RPL];
        double rr = 0.0;
#pragma unroll
        for (int c = 0; c < RPL; ++c) {
            const int r = lane + 64 * c;
            z[c] = (r >= n - 1 && r < m) ? A[r + (size_t)(n - 1) * ld] : 0.0;
            rr += z[c] * z[c];
        }
        rr = wave_sum(rr);
        for (int k = n - 2; k >= 0; --k) {
            const double tk = tau[k];
            if (tk == 0.0) continue;  // uniform: every lane reads the same word
            double vv[RPL], s = 0.0;
#pragma unroll
            for (int c = 0; c < RPL; ++c) {
                const int r = lane + 64 * c;
                vv[c] = (r > k && r < m) ? A[r + (size_t)k * ld] : (r == k ? 1.0 : 0.0);
                s += vv[c] * z[c];
            }
            s = wave_sum(s) * tk;
#pragma unroll
            for (int c = 0; c < RPL; ++c) z[c] -= s * vv[c];
        }
        singular = singular || !(rr > 0.0);
        wave_sync<LDS>();
#pragma unroll
        for (int c = 0; c < RPL; ++c) {
            const int r = lane + 64 * c;
            if (r < ne) tau[r] = singular ? 0.0 : z[c] / rr;  // tau[] reused as the weight row (ne <= n)
        }
        wave_sync<LDS>();
        // gls.pyx:470-472: neumann_ws = solution entry (n-1, n_elem-1), i.e. the LAST cell's weight
        const double nwv = is_neu ? tau[ne - 1] : 0.0;
        const double add = add_neumann ? nwv : 0.0;
        for (int i = lane; i < ne; i += 64) out[eb + i] = tau[i] + add;
        if (lane == 0) nws[p] = nwv;
        }
        __syncthreads();   // (the slot is the next node's)
    }
}

template <int RPL>
int launch_rpl(const GridView &g, const int32_t *nodes, int32_t count, int add_neumann, double *out, double *nws,
               double *scratch, int64_t scratch_stride, int32_t scratch_slots, hipStream_t stream) {
    const int64_t blocks = count < scratch_slots ? count : scratch_slots;   // one team, one scratch slot per workgroup
    constexpr int TW = (RPL == 4 || RPL == 8) ? 16 : 8;                    // (256 rows and more: ~100 columns and more, 25 column groups a step;
                                                                           //  16 rows per lane: 160 registers, eight wavefronts)
    hipLaunchKernelGGL((nin_gls_team_kernel<RPL, TW>), dim3((unsigned)blocks), dim3(64 * TW), 0, stream, g, nodes, count, add_neumann, out, nws,
                       scratch, (long long)scratch_stride);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

int launch_gls_class(const GridView &g, const int32_t *nodes, int32_t count, int32_t lds_bytes,
                     int32_t rows_per_lane, int add_neumann, double *out, double *nws, double *scratch,
                     int64_t scratch_stride, int32_t scratch_slots, hipStream_t stream) {
    if (count <= 0) return 0;
    (void)lds_bytes;   // systems that fit LDS go to kernels_gls_block.hip; this kernel serves the global-scratch class
#define NIN_RPL(R) return launch_rpl<R>(g, nodes, count, add_neumann, out, nws, scratch, scratch_stride, scratch_slots, stream)
    if (rows_per_lane <= 1) { NIN_RPL(1); }
    if (rows_per_lane <= 2) { NIN_RPL(2); }
    if (rows_per_lane <= 4) { NIN_RPL(4); }
    if (rows_per_lane <= 8) { NIN_RPL(8); }
    if (rows_per_lane <= 16) { NIN_RPL(16); }
#undef NIN_RPL
    return -5;  // more than 1024 rows in one node's system
}

const char *kernel_name_gls() { return "nin_gls_team_kernel"; }

}  // namespace nin
