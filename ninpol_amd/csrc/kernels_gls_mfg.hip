// kernels_gls_mfg.hip -- GLS weights of interior nodes with MORE cells than a wavefront's registers hold (mfg_desc.hpp: up to 32 fronts +
// 40 dense cells; a Poisson-Delaunay cloud has 6 % of its nodes beyond kernels_gls_mfx.hip's 16 + 21), gfx950: one wavefront per node,
// the dense problem -- up to 256 x 121 -- as 16-row x 4-column tiles in a GLOBAL-memory slot of the wavefront.
//
// What these nodes had before: the block kernel while the FULL m x n system fits a CU's LDS (up to ~36 cells), then the wave kernel of
// kernels_gls.hip on a global-memory copy of the full system: one Householder reflector per sweep of the whole trailing matrix --
// ~60 MB of traffic and ~7 ms a node; 1 770 such nodes of a 186 k-cell random cloud took 13.7 of its 13.8 ms.  Here
//   * the fronts are eliminated first, in the lanes (kernels_gls_mfx.hip's phase 1, in passes of 16 fronts): (7 F + D + 3 free) x
//     (3 D + 1) is left -- a quarter of the entries;
//   * the dense problem is factored in PANELS OF FOUR reflectors (compact WY, mfw_strips.hpp's strip form: the FP64 matrix unit as the
//     cross-lane adder), LEFT-LOOKING by groups of four column blocks: a group's tiles are read once into registers, take the reflectors
//     of every finished panel left of the group (read back from where those panels were factored, the next panel's tiles on their way
//     while one is applied), are factored and written once: ~0.6 MB a node, every access a whole 512-byte tile (lane = element), no
//     load behind a store.  (Measured on the way, 1 770 nodes of the random cloud: right-looking a panel at a time -- every trailing
//     tile read and written per panel, 2.5 MB a node -- 1.43 ms; panels in pairs 1.00 ms; left-looking 0.64 ms; the tile loops
//     entered by chunk, the reflectors' pivot tile stored ready-made 0.51 ms.)
//   * a lane only ever touches its own element of a tile, so the tiles need no synchronisation between the panels; the loops over panels
//     and groups are ordinary run-time loops, the loops over a group's row tiles are unrolled behind wave-uniform guards (for_tiles_down);
//   * R stays where the factorisation leaves it and R y = Q^T c is solved block column by block column straight from the tiles.
// One wavefront per SIMD (443 registers: the group's 64 tiles, two panels of reflectors); a second wavefront buys nothing -- the
// kernel is bound by instruction issue, not by latency (two wavefronts at 256 registers each: the same nodes per second).
// tools/stamps_mfg.py (a -DNIN_MFG_STAMPS build) gives one wavefront's cycles by phase: of ~435 k per node, 137 k apply finished panels
// to the groups, 125 k are the panels' own steps (vector work), 72 k phase 1, 41 k the back substitution.
// Same mathematics as dgels on the reference's matrix (gls.pyx:252-474): a Householder QR under a column / row order that exposes the zeros.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device_grid.hpp"
#include "gls_device_math.hpp"
#include "launch.hpp"
#include "mfg_desc.hpp"
#include "mfw_strips.hpp"

namespace nin {

namespace {

using namespace glsmath;
using namespace mfwstrips;

constexpr int GQ = kMfgRowTiles, GCB = kMfgColBlocks;
constexpr int G_Y = 0, G_W = 128, G_DESC = G_W + 64, G_PER_WAVE = G_DESC + kMfgDescWords / 2;   // LDS (doubles): y | weights | descriptor

// Element (row, col) of the dense problem in the wavefront's slot: row tiles are counted FROM THE BOTTOM -- kk = nq - 1 - (row >> 4), nq =
// the node's row tiles -- so that the tiles a panel still touches are always kk = 0 .. top (top falls by one every four panels): every tile loop
// is "kk = top down to 0", entered by ONE switch and straight-line from there (for_tiles_down) -- a lone wavefront pays ~50 cycles per taken
// branch, and a guard per tile was a third of this kernel's time.  Tile (kk, cb) lies at ((cb * 16 + kk) * 64 (the tiles of a column block lie
// behind one another), element = lane 16 (row & 3) + 4 ((row >> 2) & 3) + (col & 3).
__device__ __forceinline__ int tile_index(int row, int col, int nq) {
    return ((((col >> 2) * GQ) + (nq - 1 - (row >> 4))) << 6) + 16 * (row & 3) + 4 * ((row >> 2) & 3) + (col & 3);
}

__device__ __forceinline__ void wave_global_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

#ifndef NIN_MFG_WAVES
#define NIN_MFG_WAVES 1   // wavefronts per SIMD
#endif
#ifdef NIN_MFG_STAMPS   // measurement build (tools/stamps_mfg.py): one wavefront's cycles by phase, written over the node's row of weights
struct GStamps { unsigned long long last, acc[8]; };
#define NIN_GST(ST, J) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); (ST).acc[J] += t_ - (ST).last; (ST).last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
struct GStamps { };
#define NIN_GST(ST, J) do { } while (0)
#endif
constexpr int GW = 4;   // column blocks of a group (16 columns: four panels, all pivoting in the same row tile)

template <int KK>
struct TileNo { static constexpr int value = KK; };
// f(TileNo<kk>) for kk = top, top - 1, .., 0 (top < 0: nothing; top <= 15)
// ... in chunks of four: a chunk above `top` is skipped by one branch, inside the chunk that holds `top` at most three tiles are (a lone
// wavefront pays ~50 cycles per TAKEN branch: 2.5 on average this way, 5.5 with a guard per tile)
template <int K0, class F>
__device__ __forceinline__ void tiles_chunk_down(int top, F &&f) {
    if (top >= K0) {
        if (top >= K0 + 3) f(TileNo<K0 + 3>{});
        if (top >= K0 + 2) f(TileNo<K0 + 2>{});
        if (top >= K0 + 1) f(TileNo<K0 + 1>{});
        f(TileNo<K0>{});
    }
}
template <class F>
__device__ __forceinline__ void for_tiles_down(int top, F &&f) {
    static_assert(GQ == 16, "four chunks of four row tiles");
    tiles_chunk_down<12>(top, f);
    tiles_chunk_down<8>(top, f);
    tiles_chunk_down<4>(top, f);
    tiles_chunk_down<0>(top, f);
}

// One step K of panel J of a group: C[kk][J] = the panel block of row tile kk; PT = the pivot tile (kk = top), the pivot rows are its quad bp.
// As strip_panel_step (mfw_strips.hpp).
template <int K, int J>
__device__ __forceinline__ void gpanel_step(double (&C)[GQ][GW], double &PT, double (&vp)[4], double (&Tr)[4], int top, int bp, int si, int sb, int sj) {
    const bool in_piv_quad = sb == bp;
    const bool is_piv = in_piv_quad && si == K;
    const bool below0 = sb > bp || (in_piv_quad && si > K);
    const double xm0 = below0 ? quad_pick<K>(PT) : 0.0;
    double acc = xm0 * PT;
    for_tiles_down(top - 1, [&](auto t) __attribute__((always_inline)) {
        constexpr int kk = decltype(t)::value;
        acc = fma(quad_pick<K>(C[kk][J]), C[kk][J], acc);
    });
    const double ap = __shfl(PT, 16 * K + 4 * bp + sj);          // the pivot row's entry of column j
    const double d = sum_rows(sum_quads(acc));
    const House h = house_unguarded(quad_pick<K>(ap), quad_pick<K>(d));
    const double e = fma(h.vp, ap, d);
    const double w = sj > K ? -(h.g * e) : 0.0;
    vp[K] = h.vp;
    {
        double t = 0.0;
        if (K >= 1) t = Tr[0] * quad_pick<0>(e);
        if (K >= 2) t = fma(Tr[1], quad_pick<1>(e), t);
        if (K >= 3) t = fma(Tr[2], quad_pick<2>(e), t);
        Tr[K] = si == K ? h.g : -(h.g * t);
    }
    {
        const double x = fma(w, is_piv ? h.vp : xm0, PT);
        PT = (is_piv && sj == K) ? h.beta : x;
    }
    for_tiles_down(top - 1, [&](auto t) __attribute__((always_inline)) {
        constexpr int kk = decltype(t)::value;
        C[kk][J] = fma(w, quad_pick<K>(C[kk][J]), C[kk][J]);     // (the reflector's entry again: two moves against a register per tile)
    });
}

// a factored pivot tile -> the reflectors' entries in it: zero above the diagonal of the pivot quad and in the rows above the quad (R lives
// there), the diagonal = v's pivot entries
__device__ __forceinline__ double pivot_tile_v(double t, double vdiag, int bp, int si, int sb, int sj) {
    t = (sb == bp && si == sj) ? vdiag : t;
    return (sb < bp || (sb == bp && si < sj)) ? 0.0 : t;
}

// C(:, block j) -= V T^T V^T C(:, block j) for the blocks J0 .. GW - 1 of the group; V[kk] = the reflectors' tile kk <= top
template <int J0>
__device__ __forceinline__ void group_apply(const double (&V)[GQ], double Ts, double (&C)[GQ][GW], int top, double eye) {
    double W[GW];
#pragma unroll
    for (int j = J0; j < GW; ++j) W[j] = 0.0;
    for_tiles_down(top, [&](auto t) __attribute__((always_inline)) {
        constexpr int kk = decltype(t)::value;
#pragma unroll
        for (int j = J0; j < GW; ++j) W[j] = mfma4(V[kk], C[kk][j], W[j]);
    });
#pragma unroll
    for (int j = J0; j < GW; ++j) W[j] = mfma4(Ts, sum_quads(W[j]), 0.0);   // -(T^T V^T C), the same in every quad
    for_tiles_down(top, [&](auto t) __attribute__((always_inline)) {
        constexpr int kk = decltype(t)::value;
        const double VT = mfma4(V[kk], eye, 0.0);                // V^T per quad: the A operand of C -= V W'
#pragma unroll
        for (int j = J0; j < GW; ++j) C[kk][j] = mfma4(VT, W[j], C[kk][j]);
    });
}

// Panel J of group g (block 4 g + J; its pivot tile is kk = top): factor it in place, leave -T and the reflectors' pivot tile in the panel's
// two auxiliary tiles, apply it to the group's blocks right of it.  false: this was the last panel (fewer than four pivots: c sits in its block).
template <int J>
__device__ __forceinline__ bool group_panel(double (&C)[GQ][GW], double *mine, int g, int nc, int top, int si, int sb, int sj, double eye, GStamps &ST) {
    const int p = GW * g + J, steps = nc - 4 * p < 4 ? nc - 4 * p : 4;
    if (steps <= 0) return false;
    double vp[4] = {0.0, 0.0, 0.0, 0.0}, Tr[4] = {0.0, 0.0, 0.0, 0.0};
    double PT = 0.0;
#pragma unroll
    for (int k = 0; k < GQ; ++k) PT = k == top ? C[k][J] : PT;
    gpanel_step<0, J>(C, PT, vp, Tr, top, J, si, sb, sj);
    if (steps > 1) gpanel_step<1, J>(C, PT, vp, Tr, top, J, si, sb, sj);
    if (steps > 2) gpanel_step<2, J>(C, PT, vp, Tr, top, J, si, sb, sj);
    if (steps > 3) gpanel_step<3, J>(C, PT, vp, Tr, top, J, si, sb, sj);
#pragma unroll
    for (int k = 0; k < GQ; ++k) C[k][J] = k == top ? PT : C[k][J];
    NIN_GST(ST, 3);
    if (steps < 4) return false;
    const double Ts = -(sj == 0 ? Tr[0] : sj == 1 ? Tr[1] : sj == 2 ? Tr[2] : Tr[3]);
    const double vdiag = sj == 0 ? vp[0] : sj == 1 ? vp[1] : sj == 2 ? vp[2] : vp[3];
    const double vpiv = pivot_tile_v(PT, vdiag, J, si, sb, sj);
    mine[(GQ * GCB + 2 * p) << 6] = Ts;
    mine[(GQ * GCB + 2 * p + 1) << 6] = vpiv;
    if constexpr (J + 1 < GW) {
        double V[GQ];
#pragma unroll
        for (int k = 0; k < GQ; ++k) V[k] = k == top ? vpiv : C[k][J];
        group_apply<J + 1>(V, Ts, C, top, eye);
    }
    NIN_GST(ST, 4);
    return true;
}

// the tiles of panel p's reflectors -> V: tile kk < top from the panel's block, the pivot tile (kk = top) from its auxiliary tile
__device__ __forceinline__ void load_reflectors(const double *mine, int p, int top, double (&V)[GQ]) {
    const double *const blk = mine + ((p * GQ) << 6), *const piv = mine + ((GQ * GCB + 2 * p + 1) << 6);
    for_tiles_down(top, [&](auto t) __attribute__((always_inline)) {
        constexpr int kk = decltype(t)::value;
        V[kk] = *(kk == top ? piv : blk + (kk << 6));
    });
}

// Factor the nrows x (nc + 1) problem held in the slot (c at column nc), solve R y = Q^T c: y -> yb (LDS), returns r . r.
// LEFT-LOOKING by groups of four column blocks: a group's 4 x nq tiles are read once, take the reflectors of every panel left of the group
// (read back from where the panels were factored: V below R in the panel's own tiles; -T and the reflectors' pivot tile in two auxiliary
// tiles per panel), are factored -- four panels, each applied to the blocks right of it in the registers -- and written once.  Nothing but
// finished panels is ever re-read: per node ~0.6 MB of traffic against 2.5 MB of the right-looking form a panel at a time, and no load
// waits for a store.
__device__ __forceinline__ double mfg_factor_solve(double *__restrict__ slot, int nc, int nrows, int lane, double *yb, GStamps &ST) {
    const int si = lane >> 4, sb = (lane >> 2) & 3, sj = lane & 3, rowbase = 4 * sb + si;
    const double eye = si == sj ? 1.0 : 0.0;
    const int nq = (nrows + 15) >> 4, ncb = (nc + 4) >> 2, n_groups = (ncb + GW - 1) / GW;
    double *const mine = slot + lane;
    for (int g = 0; g < n_groups; ++g) {
        double C[GQ][GW];
        double *const gp = mine + ((GW * g * GQ) << 6);
        for_tiles_down(nq - 1, [&](auto t) __attribute__((always_inline)) {
            constexpr int kk = decltype(t)::value;
#pragma unroll
            for (int j = 0; j < GW; ++j) C[kk][j] = gp[(j * GQ + kk) << 6];   // (blocks beyond the node's: zero-filled by the caller)
        });
        NIN_GST(ST, 1);
        // the panels left of the group (every one of them has four reflectors), two per round: while one is applied the next one's tiles
        // are on their way (they are final: no store of this loop touches them)
        double VA[GQ], VB[GQ], TsA = 0.0, TsB = 0.0;
        if (g > 0) {
            load_reflectors(mine, 0, nq - 1, VA);
            TsA = mine[(GQ * GCB) << 6];
        }
        for (int p = 0; p < GW * g; p += 2) {                    // (GW * g is even)
            load_reflectors(mine, p + 1, nq - 1 - ((p + 1) >> 2), VB);
            TsB = mine[(GQ * GCB + 2 * (p + 1)) << 6];
            group_apply<0>(VA, TsA, C, nq - 1 - (p >> 2), eye);
            if (p + 2 < GW * g) {
                load_reflectors(mine, p + 2, nq - 1 - ((p + 2) >> 2), VA);
                TsA = mine[(GQ * GCB + 2 * (p + 2)) << 6];
            }
            group_apply<0>(VB, TsB, C, nq - 1 - ((p + 1) >> 2), eye);
        }
        NIN_GST(ST, 2);
        // the group's own panels
        const int top = nq - 1 - g;
        if (group_panel<0>(C, mine, g, nc, top, si, sb, sj, eye, ST))
            if (group_panel<1>(C, mine, g, nc, top, si, sb, sj, eye, ST))
                if (group_panel<2>(C, mine, g, nc, top, si, sb, sj, eye, ST)) (void)group_panel<3>(C, mine, g, nc, top, si, sb, sj, eye, ST);
        NIN_GST(ST, 3);
        for_tiles_down(nq - 1, [&](auto t) __attribute__((always_inline)) {
            constexpr int kk = decltype(t)::value;
#pragma unroll
            for (int j = 0; j < GW; ++j) gp[(j * GQ + kk) << 6] = C[kk][j];
        });
    }
    // c sits in block nc >> 2, column nc & 3
    const int cbc = nc >> 2, jc = nc & 3;
    double *const cc = mine + ((cbc * GQ + nq - 1) << 6);        // (row tile q from the top: cc[-(q << 6)])
    NIN_GST(ST, 1);
    double rr = 0.0;
    {
        double t = 0.0;
        for (int q = 0; q < nq; ++q) {
            const double x = cc[-(q << 6)];
            const double xr = (16 * q + rowbase >= nc && sj == jc) ? x : 0.0;
            t = fma(xr, xr, t);
        }
        rr = wave_allsum(t);
    }
    // ---- R y = (Q^T c)(0 : nc), block column by block column from the last: b[k] = this lane's ROW of the right-hand side in row tile k
    //      (from the top) ----
    constexpr int BQ = (kMfgMaxDense * 3 + 15) / 16;           // row tiles that hold pivot rows
    double b[BQ];
#pragma unroll
    for (int k = 0; k < BQ; ++k) {
        b[k] = 0.0;
        if (16 * k < nc) b[k] = __shfl(cc[-(k << 6)], (lane & ~3) | jc);
    }
    double Tn[BQ];                                               // the tiles of the next block column: on their way while this one is solved
#pragma unroll
    for (int k = 0; k < BQ; ++k) {
        Tn[k] = 0.0;
        if (k <= ((nc - 1) >> 4)) Tn[k] = mine[((((nc - 1) >> 2) * GQ + nq - 1 - k) << 6)];
    }
    for (int cb = (nc - 1) >> 2; cb >= 0; --cb) {
        const int q = cb >> 2, quad = cb & 3, live = nc - 4 * cb < 4 ? nc - 4 * cb : 4;   // (a last block shares its columns with c)
        double T[BQ], Dt = 0.0, bq = 0.0;
#pragma unroll
        for (int k = 0; k < BQ; ++k) {
            T[k] = Tn[k];
            Dt = k == q ? Tn[k] : Dt;
            bq = k == q ? b[k] : bq;
        }
        if (cb > 0) {
            const double *const cn = mine + (((cb - 1) * GQ + nq - 1) << 6);
#pragma unroll
            for (int k = 0; k < BQ; ++k)
                if (k <= ((cb - 1) >> 2)) Tn[k] = cn[-(k << 6)];
        }
        double y[4];
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            double s = rl64(bq, 16 * i + 4 * quad);               // (wave-uniform lanes: v_readlane, the block's entries as scalars)
#pragma unroll
            for (int j = 3; j > i; --j) s = fma(-rl64(Dt, 16 * i + 4 * quad + j), y[j], s);
            const double yi = s * fast_rcp(rl64(Dt, 16 * i + 4 * quad + i));
            y[i] = i < live ? yi : 0.0;
        }
        if (lane < live) yb[4 * cb + lane] = lane == 0 ? y[0] : lane == 1 ? y[1] : lane == 2 ? y[2] : y[3];
        const double ysel = sj == 0 ? y[0] : sj == 1 ? y[1] : sj == 2 ? y[2] : y[3];
#pragma unroll
        for (int k = 0; k < BQ; ++k) {
            if (k <= q) {                                        // the rows above the block: b -= R(:, block) y
                double part = T[k] * ysel;
                part += dpp_mov<0xB1>(part);                     // sum over the quad's four columns
                part += dpp_mov<0x4E>(part);
                const bool above = k < q || sb < quad;
                b[k] = above ? b[k] - part : b[k];
            }
        }
    }
    NIN_GST(ST, 5);
    return rr;
}

__global__ __launch_bounds__(64, NIN_MFG_WAVES) void nin_gls_mfg_kernel(GridView g, const int32_t *__restrict__ nodes, const uint32_t *__restrict__ desc,
                                                           int32_t count, int add_neumann, double *__restrict__ out, double *__restrict__ nws,
                                                           int32_t *__restrict__ queue, double *__restrict__ tiles) {
    __shared__ double Lm[G_PER_WAVE];
    const int lane = threadIdx.x;
    double *const yb = Lm + G_Y, *const wbuf = Lm + G_W;
    uint32_t *const dl = reinterpret_cast<uint32_t *>(Lm + G_DESC);
    const uint8_t *const slotpos = reinterpret_cast<const uint8_t *>(dl + kMfgSlotTable);
    double *const slot = tiles + (size_t)blockIdx.x * kMfgSlotDoubles;

    auto ticket = [&]() -> int32_t {
        int32_t v = 0;
        if (lane == 0) v = atomicAdd(queue, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    for (int32_t idx = ticket(); idx < count; idx = ticket()) {
        GStamps ST;
#ifdef NIN_MFG_STAMPS
        for (int k = 0; k < 8; ++k) ST.acc[k] = 0ull;
        ST.last = __builtin_amdgcn_s_memtime();
#endif
        const int32_t p = __builtin_amdgcn_readfirstlane(nodes ? nodes[idx] : idx);
        const uint32_t *dw = desc + (size_t)kMfgDescWords * idx;
        for (int k = lane; k < kMfgDescWords; k += 64) dl[k] = dw[k];
        const uint32_t fd = (uint32_t)__builtin_amdgcn_readfirstlane((int)dw[0]);
        const int F = fd & 255, D = (fd >> 8) & 255, nfree = (fd >> 16) & 255, ne = F + D;
        const uint32_t eb = (uint32_t)__builtin_amdgcn_readfirstlane(g.esup_ptr[p]);
        const uint32_t fb = (uint32_t)__builtin_amdgcn_readfirstlane(g.fsup_ptr[p]);
        const bool is_neu = (__builtin_amdgcn_readfirstlane((int)g.flags[p]) & 2) != 0;
        const double xv0 = g.coords[3 * (size_t)p + 0], xv1 = g.coords[3 * (size_t)p + 1], xv2 = g.coords[3 * (size_t)p + 2];
        const int nc = 3 * D, nrows = 7 * F + D + 3 * nfree, nq = (nrows + 15) >> 4, ncb = (nc + 4) >> 2;
        // the tiles of this node: zero (a lane clears its own element of every tile)
        for (int cb = 0; cb < ((ncb + GW - 1) / GW) * GW; ++cb)    // (whole groups of column blocks)
            for (int q = 0; q < nq; ++q) slot[((cb * GQ + q) << 6) + lane] = 0.0;
        wave_lds_sync();
        wave_global_sync();
        // ---- phase 1: the fronts in passes of 16, FOUR lanes per front (lane 4 f + j: c for j = 0, the columns of neighbour j - 1 otherwise);
        //      the 7 fill rows of a front go straight to their elements of the tiles ------------------------------------------------
        const int fq = lane >> 2, jq = lane & 3, my = jq > 0 ? jq - 1 : 0;
        double u[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
        uint32_t pe_[2] = {0u, 0u}, slot_[2] = {0u, 0u};
#pragma unroll 1                                                 // (one body: two copies of it cost 100 registers more)
        for (int pass = 0; pass < 2; ++pass) {
            if (16 * pass < F) {                                 // (wave-uniform)
                const int f = 16 * pass + fq;
                const uint32_t wa = dl[kMfgW0 + f], wb = dl[kMfgW1 + f];
                const uint32_t pe = wa & 63u, myslot = (wb >> (6 * my)) & 63u;
                pe_[0] = pass == 0 ? pe : pe_[0];
                pe_[1] = pass == 1 ? pe : pe_[1];
                slot_[0] = pass == 0 ? myslot : slot_[0];
                slot_[1] = pass == 1 ? myslot : slot_[1];
                const uint32_t ce = (uint32_t)g.esup[eb + pe], cm = (uint32_t)g.esup[eb + slotpos[myslot]];
                double P[10][3], B[10][3], de[3];
                double Ke[9], Km[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) { Ke[k] = g.perm[9 * (size_t)ce + k]; Km[k] = g.perm[9 * (size_t)cm + k]; }
                const double dme = g.diff_mag[ce];
                de[0] = g.centroids[3 * (size_t)ce + 0] - xv0;  // gls.pyx:269-277
                de[1] = g.centroids[3 * (size_t)ce + 1] - xv1;
                de[2] = g.centroids[3 * (size_t)ce + 2] - xv2;
#pragma unroll
                for (int t = 0; t < 3; ++t) { P[0][t] = de[t]; B[0][t] = (jq == 0 && t == 0) ? 1.0 : 0.0; }   // c = e_0 on entry
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    // B = [K N; T1; tau T2] (gls.pyx:293-321), row = [-B_a | +B_b] (gls.pyx:340-356)
                    const uint32_t fc = (uint32_t)g.fsup[fb + ((wa >> (6 + 7 * i)) & 127u)];
                    const uint32_t cn = (uint32_t)g.esup[eb + slotpos[(wb >> (6 * i)) & 63u]];
                    const double N0 = (double)g.face_normal[3 * (size_t)fc + 0], N1 = (double)g.face_normal[3 * (size_t)fc + 1],
                                 N2 = (double)g.face_normal[3 * (size_t)fc + 2];
                    const double T0 = xv0 - g.face_center[3 * (size_t)fc + 0], T1 = xv1 - g.face_center[3 * (size_t)fc + 1],
                                 T2 = xv2 - g.face_center[3 * (size_t)fc + 2];
                    const double U0 = N1 * T2 - N2 * T1, U1 = N2 * T0 - N0 * T2, U2 = N0 * T1 - N1 * T0;
                    const double dmn = g.diff_mag[cn];
                    double eta = 0.0;
                    eta = dme > eta ? dme : eta;
                    eta = dmn > eta ? dmn : eta;
                    const double tj = face_tau(sqrt(U0 * U0 + U1 * U1 + U2 * U2), eta);
                    const double sg = ((wa >> (27 + i)) & 1u) ? -1.0 : 1.0;
                    const bool mine_ = jq > 0 && my == i;
                    const double s0[3] = {sg * T0, sg * T1, sg * T2}, s1[3] = {sg * (tj * U0), sg * (tj * U1), sg * (tj * U2)};
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        P[1 + 3 * i][t] = sg * (Ke[t * 3 + 0] * N0 + Ke[t * 3 + 1] * N1 + Ke[t * 3 + 2] * N2);
                        P[2 + 3 * i][t] = s0[t];
                        P[3 + 3 * i][t] = s1[t];
                        const double nb = -sg * (Km[t * 3 + 0] * N0 + Km[t * 3 + 1] * N1 + Km[t * 3 + 2] * N2);
                        B[1 + 3 * i][t] = mine_ ? nb : 0.0;
                        B[2 + 3 * i][t] = mine_ ? -s0[t] : 0.0;
                        B[3 + 3 * i][t] = mine_ ? -s1[t] : 0.0;
                    }
                }
                double g3[3], z[3];
                front_panel(P, de, g3, z);
                __builtin_amdgcn_sched_barrier(0);
                apply_panel<3, true, true, true, true>(P, g3, B);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const double ut = fma(z[2], B[2][t], fma(z[1], B[1][t], z[0] * B[0][t]));
                    u[0][t] = pass == 0 ? ut : u[0][t];
                    u[1][t] = pass == 1 ? ut : u[1][t];
                }
                if (f < F) {
#pragma unroll
                    for (int r = 0; r < 7; ++r) {
                        const int row = 7 * f + r;
                        if (jq == 0) slot[tile_index(row, nc, nq)] = B[3 + r][0];
                        else {
#pragma unroll
                            for (int t = 0; t < 3; ++t) slot[tile_index(row, 3 * (int)myslot + t, nq)] = B[3 + r][t];
                        }
                    }
                }
            }
        }
        // the dense cells' rows: (x_K - x_v) on the cell's own columns, c = 1 (lane d = dense slot d)
        const uint32_t po = slotpos[lane < kMfgMaxDense ? lane : 0];
        double dod[3];
        {
            const uint32_t co = (uint32_t)g.esup[eb + po];
            dod[0] = g.centroids[3 * (size_t)co + 0] - xv0;
            dod[1] = g.centroids[3 * (size_t)co + 1] - xv1;
            dod[2] = g.centroids[3 * (size_t)co + 2] - xv2;
            if (lane < D) {
                const int row = 7 * F + lane;
#pragma unroll
                for (int t = 0; t < 3; ++t) slot[tile_index(row, 3 * lane + t, nq)] = dod[t];
                slot[tile_index(row, nc, nq)] = 1.0;
            }
        }
        if (lane < nfree) {
            // a free face (both its cells dense): its three rows [-B_a | +B_b] (gls.pyx:293-356)
            const uint32_t fw = dl[kMfgFree0 + lane];
            const uint32_t fc = (uint32_t)g.fsup[fb + (fw & 127u)];
            const int sa = (fw >> 7) & 63u, sbb = (fw >> 13) & 63u;
            const uint32_t ca_ = (uint32_t)g.esup[eb + slotpos[sa]], cb_ = (uint32_t)g.esup[eb + slotpos[sbb]];
            const double N0 = (double)g.face_normal[3 * (size_t)fc + 0], N1 = (double)g.face_normal[3 * (size_t)fc + 1],
                         N2 = (double)g.face_normal[3 * (size_t)fc + 2];
            const double T0 = xv0 - g.face_center[3 * (size_t)fc + 0], T1 = xv1 - g.face_center[3 * (size_t)fc + 1],
                         T2 = xv2 - g.face_center[3 * (size_t)fc + 2];
            const double U0 = N1 * T2 - N2 * T1, U1 = N2 * T0 - N0 * T2, U2 = N0 * T1 - N1 * T0;
            const double da = g.diff_mag[ca_], db = g.diff_mag[cb_];
            double eta = 0.0;
            eta = da > eta ? da : eta;
            eta = db > eta ? db : eta;
            const double tj = face_tau(sqrt(U0 * U0 + U1 * U1 + U2 * U2), eta);
            const double *Ka = g.perm + 9 * (size_t)ca_, *Kb = g.perm + 9 * (size_t)cb_;
            const double Tv[3] = {T0, T1, T2}, Uv[3] = {tj * U0, tj * U1, tj * U2};
            const int row = 7 * F + D + 3 * lane;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                slot[tile_index(row + 0, 3 * sa + t, nq)] = -(Ka[t * 3 + 0] * N0 + Ka[t * 3 + 1] * N1 + Ka[t * 3 + 2] * N2);
                slot[tile_index(row + 0, 3 * sbb + t, nq)] = Kb[t * 3 + 0] * N0 + Kb[t * 3 + 1] * N1 + Kb[t * 3 + 2] * N2;
                slot[tile_index(row + 1, 3 * sa + t, nq)] = -Tv[t];
                slot[tile_index(row + 1, 3 * sbb + t, nq)] = Tv[t];
                slot[tile_index(row + 2, 3 * sa + t, nq)] = -Uv[t];
                slot[tile_index(row + 2, 3 * sbb + t, nq)] = Uv[t];
            }
        }
        wave_global_sync();

        NIN_GST(ST, 0);
        const double rr = mfg_factor_solve(slot, nc, nrows, lane, yb, ST);
        wave_lds_sync();
        // ---- residuals on the cell rows, weights ---------------------------------------------------------------------------
        const double rri = fast_rcp(rr);
        const bool ok = rr > 0.0;                                // rank-deficient system (or NaN from a zero column): the zero row
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (16 * pass < F) {
                // r_e = 1 - d_e . y_e = 1 - z . b_e + u . y_dense: the c lane brings 1 - s, the three others their block of u . y
                const int sl = (int)slot_[pass];
                double part = fma(u[pass][2], yb[3 * sl + 2], fma(u[pass][1], yb[3 * sl + 1], u[pass][0] * yb[3 * sl]));
                part = jq == 0 ? 1.0 - u[pass][0] : part;
                part += dpp_mov<0xB1>(part);
                part += dpp_mov<0x4E>(part);
                double we = part * rri;
                we = (ok && __builtin_isfinite(we)) ? we : 0.0;
                if (16 * pass + fq < F && jq == 0) wbuf[pe_[pass]] = we;
            }
        }
        {
            const int ld = lane < D ? lane : 0;
            const double ro = 1.0 - fma(dod[2], yb[3 * ld + 2], fma(dod[1], yb[3 * ld + 1], dod[0] * yb[3 * ld]));
            double wo = ro * rri;
            wo = (ok && __builtin_isfinite(wo)) ? wo : 0.0;
            if (lane < D) wbuf[po] = wo;
        }
        wave_lds_sync();
        {
            // gls.pyx:470-472 (only if an interior node carries the Neumann flag): neumann_ws = the last cell's weight
            const double nwv = is_neu ? wbuf[ne - 1] : 0.0;
            const double addv = add_neumann ? nwv : 0.0;
            if (lane < ne) out[eb + lane] = wbuf[lane] + addv;
#ifdef NIN_MFG_STAMPS
            NIN_GST(ST, 6);
            if (lane < 8) out[eb + lane] = lane == 7 ? (double)(nq * 1000 + ncb) : (double)ST.acc[lane];
#endif
            if (lane == 0) nws[p] = nwv;
        }
        wave_lds_sync();
    }
}

__global__ void k_mfg_desc(GridView g, const int32_t *__restrict__ nodes, int32_t count, uint32_t *__restrict__ desc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t w[kMfgDescWords];
    if (!mfg_descriptor(g, nodes ? nodes[i] : (int32_t)i, w)) {   // (the list holds classified nodes only)
        for (int k = 0; k < kMfgDescWords; ++k) w[k] = 0u;
    }
    for (int k = 0; k < kMfgDescWords; ++k) desc[(size_t)kMfgDescWords * i + k] = w[k];
}

}  // namespace

int launch_mfg_desc(const GridView &g, const int32_t *nodes, int32_t count, uint32_t *desc, hipStream_t stream) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(k_mfg_desc, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, stream, g, nodes, count, desc);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// `tiles`: n_slots slots of kMfgSlotDoubles doubles (one per resident wavefront); at most n_slots workgroups are launched
int launch_gls_mfg(const GridView &g, const int32_t *nodes, const uint32_t *desc, int32_t count, int add_neumann, double *out, double *nws,
                   int32_t *queue, double *tiles, int32_t n_slots, hipStream_t stream) {
    if (count <= 0) return 0;
    if (!tiles || n_slots <= 0) return -1;
    const int64_t blocks = count < n_slots ? count : n_slots;
    hipLaunchKernelGGL(nin_gls_mfg_kernel, dim3((unsigned)blocks), dim3(64), 0, stream, g, nodes, desc, count, add_neumann, out, nws, queue, tiles);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

const char *kernel_name_gls_mfg() { return "nin_gls_mfg_kernel"; }

}  // namespace nin
