// tools/test_xstrip.hip -- the dense phase of kernels_gls_mfx.hip alone, against a host Householder QR: random nrows x (nc + 1)
// problems through (a) mfw_strips.hpp's unrolled strip_factor<TQ, TCB> in the five size classes the kernel instantiates and
// (b) mfx_strips.hpp's single-body xstrip_factor (wave-uniform branches; round 4's first form, kept as the A/B baseline: it is
// 1.9 x slower, see the timing lines).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ninpol_amd/csrc tools/test_xstrip.hip -o tools/_bin/test_xstrip && tools/_bin/test_xstrip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include "mfx_strips.hpp"

using namespace nin::mfxstrips;

__global__ __launch_bounds__(64) void k_factor(const double *A, int lda, int nc, int nrows, double *Rout, double *rr_out) {
    __shared__ double Rm[64 * XRP];
    const int lane = threadIdx.x, si = lane >> 4, sb = (lane >> 2) & 3, sj = lane & 3;
    for (int i = lane; i < 64 * XRP; i += 64) Rm[i] = 0.0;
    double C[XQ][XCB];
#pragma unroll
    for (int q = 0; q < XQ; ++q)
#pragma unroll
        for (int cb = 0; cb < XCB; ++cb) {
            const int row = 16 * q + 4 * sb + si, col = 4 * cb + sj;
            C[q][cb] = (row < nrows && col <= nc) ? A[row * lda + col] : 0.0;
        }
    __syncthreads();
    XStamps ST;
    const double rr = xstrip_factor(C, nc, nrows, lane, Rm, ST);
    __syncthreads();
    for (int i = lane; i < 64 * XRP; i += 64) Rout[i] = Rm[i];
    if (lane == 0) *rr_out = rr;
}

__global__ __launch_bounds__(64) void k_time(const double *A, int lda, int nc, int nrows, double *Rout, int reps) {
    __shared__ double Rm[64 * XRP];
    const int lane = threadIdx.x, si = lane >> 4, sb = (lane >> 2) & 3, sj = lane & 3;
    double acc = 0.0;
    for (int it = 0; it < reps; ++it) {
        double C[XQ][XCB];
#pragma unroll
        for (int q = 0; q < XQ; ++q)
#pragma unroll
            for (int cb = 0; cb < XCB; ++cb) {
                const int row = 16 * q + 4 * sb + si, col = 4 * cb + sj;
                C[q][cb] = (row < nrows && col <= nc) ? A[row * lda + col] + acc : 0.0;
            }
        XStamps ST;
        acc += 1e-300 * xstrip_factor(C, nc, nrows, lane, Rm, ST);
    }
    Rout[(size_t)blockIdx.x * 64 + lane] = acc + Rm[lane];
}

// the same through mfw_strips.hpp's unrolled strip_factor (one body per half-generation, no branches): the A/B baseline
template <int TQ, int TCB>
__global__ __launch_bounds__(64) void k_time_unrolled(const double *A, int lda, int nc, int nrows, double *Rout, int reps) {
    __shared__ double Rm[64 * XRP];
    const int lane = threadIdx.x, si = lane >> 4, sb = (lane >> 2) & 3, sj = lane & 3;
    double acc = 0.0;
    for (int it = 0; it < reps; ++it) {
        double C[TQ][TCB];
#pragma unroll
        for (int q = 0; q < TQ; ++q)
#pragma unroll
            for (int cb = 0; cb < TCB; ++cb) {
                const int row = 16 * q + 4 * sb + si, col = 4 * cb + sj;
                C[q][cb] = (row < nrows && col <= nc) ? A[row * lda + col] + acc : 0.0;
            }
        nin::mfwstrips::SubStamps ST;
        acc += 1e-300 * nin::mfwstrips::strip_factor<TQ, TCB>(C, nc, lane, Rm, XRP, ST);
    }
    Rout[(size_t)blockIdx.x * 64 + lane] = acc + Rm[lane];
}

template <int TQ, int TCB>
__global__ __launch_bounds__(64) void k_factor_class(const double *A, int lda, int nc, int nrows, double *Rout, double *rr_out) {
    __shared__ double Rm[64 * XRP];
    const int lane = threadIdx.x, si = lane >> 4, sb = (lane >> 2) & 3, sj = lane & 3;
    for (int i = lane; i < 64 * XRP; i += 64) Rm[i] = 0.0;
    double C[TQ][TCB];
#pragma unroll
    for (int q = 0; q < TQ; ++q)
#pragma unroll
        for (int cb = 0; cb < TCB; ++cb) {
            const int row = 16 * q + 4 * sb + si, col = 4 * cb + sj;
            C[q][cb] = (row < nrows && col <= nc) ? A[row * lda + col] : 0.0;
        }
    __syncthreads();
    nin::mfwstrips::SubStamps ST;
    const double rr = nin::mfwstrips::strip_factor<TQ, TCB>(C, nc, lane, Rm, XRP, ST);
    __syncthreads();
    for (int i = lane; i < 64 * XRP; i += 64) Rout[i] = Rm[i];
    if (lane == 0) *rr_out = rr;
}

// host: Householder QR of the nrows x (nc + 1) matrix on its first nc columns; returns R (nc x (nc + 1)) and |(Q^T c)(nc:)|^2
static void host_qr(std::vector<double> a, int lda, int nrows, int nc, std::vector<double> &R, double &rr) {
    for (int k = 0; k < nc; ++k) {
        double ss = 0;
        for (int r = k + 1; r < nrows; ++r) ss += a[r * lda + k] * a[r * lda + k];
        const double alpha = a[k * lda + k], S = alpha * alpha + ss, sq = std::sqrt(S);
        const double beta = alpha >= 0 ? -sq : sq, vp = alpha - beta, g = 1.0 / (S + std::fabs(alpha) * sq);
        for (int j = k + 1; j <= nc; ++j) {
            double d = vp * a[k * lda + j];
            for (int r = k + 1; r < nrows; ++r) d += a[r * lda + k] * a[r * lda + j];
            const double w = -g * d;
            a[k * lda + j] += w * vp;
            for (int r = k + 1; r < nrows; ++r) a[r * lda + j] += w * a[r * lda + k];
        }
        a[k * lda + k] = beta;
        for (int r = k + 1; r < nrows; ++r) a[r * lda + k] = 0.0 * 0 + a[r * lda + k];
    }
    R.assign((size_t)nc * (nc + 1), 0.0);
    for (int k = 0; k < nc; ++k)
        for (int j = k; j <= nc; ++j) R[k * (nc + 1) + j] = a[k * lda + j];
    rr = 0;
    for (int r = nc; r < nrows; ++r) rr += a[r * lda + nc] * a[r * lda + nc];
}

int main(int argc, char **argv) {
    const bool timing = !(argc > 1 && std::string(argv[1]) == "--no-timing");
    const int lda = 64;
    std::mt19937_64 gen(1);
    std::normal_distribution<double> nd;
    double *dA, *dR, *drr;
    hipMalloc(&dA, 160 * lda * 8); hipMalloc(&dR, 64 * XRP * 8); hipMalloc(&drr, 8);
    const int cases[][2] = {{20, 9}, {44, 13}, {96, 36}, {101, 42}, {113, 48}, {118, 49}, {127, 52}, {135, 57}, {144, 60}, {160, 63},
                            {64, 63}, {100, 3}, {17, 16}, {33, 30}, {160, 12}};
    int bad = 0;
    for (auto &cs : cases) {
        const int nrows = cs[0], nc = cs[1];
        std::vector<double> A((size_t)160 * lda, 0.0);
        for (int r = 0; r < nrows; ++r)
            for (int c = 0; c <= nc; ++c) A[r * lda + c] = (gen() % 4 == 0) ? 0.0 : nd(gen);   // some structural zeros
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        std::vector<double> R;
        double rr;
        host_qr(A, lda, nrows, nc, R, rr);
        for (int form = 0; form < 2; ++form) {
            const char *what = "single body";
            if (form == 0) hipLaunchKernelGGL(k_factor, dim3(1), dim3(64), 0, 0, dA, lda, nc, nrows, dR, drr);
            else {   // the smallest class that holds the problem, as the kernel picks it
                if (nrows <= 96 && nc < 40) { hipLaunchKernelGGL((k_factor_class<6, 10>), dim3(1), dim3(64), 0, 0, dA, lda, nc, nrows, dR, drr); what = "class 6 x 10"; }
                else if (nrows <= 112 && nc < 44) { hipLaunchKernelGGL((k_factor_class<7, 11>), dim3(1), dim3(64), 0, 0, dA, lda, nc, nrows, dR, drr); what = "class 7 x 11"; }
                else if (nrows <= 128 && nc < 52) { hipLaunchKernelGGL((k_factor_class<8, 13>), dim3(1), dim3(64), 0, 0, dA, lda, nc, nrows, dR, drr); what = "class 8 x 13"; }
                else if (nrows <= 144 && nc < 60) { hipLaunchKernelGGL((k_factor_class<9, 15>), dim3(1), dim3(64), 0, 0, dA, lda, nc, nrows, dR, drr); what = "class 9 x 15"; }
                else { hipLaunchKernelGGL((k_factor_class<10, 16>), dim3(1), dim3(64), 0, 0, dA, lda, nc, nrows, dR, drr); what = "class 10 x 16"; }
            }
            std::vector<double> Rg(64 * XRP);
            double rrg = 0;
            if (hipMemcpy(Rg.data(), dR, Rg.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 2; }
            hipMemcpy(&rrg, drr, 8, hipMemcpyDeviceToHost);
            double err = 0, scale = 0;
            int wi = -1, wj = -1;
            for (int k = 0; k < nc; ++k)
                for (int j = k; j <= nc; ++j) {
                    const double e = std::fabs(Rg[k * XRP + j] - R[k * (nc + 1) + j]);
                    if (!(e <= err)) { err = e; wi = k; wj = j; }
                    scale = std::fmax(scale, std::fabs(R[k * (nc + 1) + j]));
                }
            const double erel = err / scale, err_rr = std::fabs(rrg - rr) / rr;
            const bool ok = erel < 1e-12 && err_rr < 1e-11;
            bad += !ok;
            printf("%3d x %2d %-13s: max |R - R_host| / max|R| = %.2e at (%d, %d)   rr %.6e vs %.6e (%.1e)  %s\n", nrows, nc, what, erel, wi, wj, rrg, rr,
                   err_rr, ok ? "ok" : "FAIL");
        }
    }
    // timing: the same 118 x 48 problem on 1 / 256 / 1024 / 2048 wavefronts (blocks of one wave), 20 factorisations each
    if (timing) {
        const int nrows = 118, nc = 48;
        std::vector<double> A((size_t)160 * lda, 0.0);
        for (int r = 0; r < nrows; ++r)
            for (int c = 0; c <= nc; ++c) A[r * lda + c] = nd(gen);
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        double *dRb;
        hipMalloc(&dRb, (size_t)4096 * 64 * XRP * 8);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        auto time_unrolled = [&](auto kern, const char *what, int blocks) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, dA, lda, nc, nrows, dRb, 2);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, dA, lda, nc, nrows, dRb, 20);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("timing %d x %d, UNROLLED strip_factor%s: %5d waves: %.1f us each per wave\n", nrows, nc, what, blocks, ms * 1e3 / 20);
        };
        for (int blocks : {1, 1024}) {
            time_unrolled(k_time_unrolled<10, 16>, "<10,16> (sweeps all 160 x 64)", blocks);
            time_unrolled(k_time_unrolled<8, 13>, "<8,13> (128 x 52: the exact class)", blocks);
        }
        for (int blocks : {1, 256, 1024, 2048, 4096}) {
            hipLaunchKernelGGL(k_time, dim3(blocks), dim3(64), 0, 0, dA, lda, nc, nrows, dRb, 2);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_time, dim3(blocks), dim3(64), 0, 0, dA, lda, nc, nrows, dRb, 20);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("timing %d x %d: %5d waves x 20 factorisations: %.3f ms = %.1f us each per wave = %.1f ns per factorisation chip-wide\n", nrows, nc, blocks, ms,
                   ms * 1e3 / 20, ms * 1e6 / 20 / blocks);
        }
    }
    printf(bad ? "FAILED %d cases\n" : "all ok\n", bad);
    return bad ? 1 : 0;
}
