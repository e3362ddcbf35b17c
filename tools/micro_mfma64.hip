// micro_mfma64.hip -- FP64 matrix instructions on gfx950: operand / result lane maps (found by probing with one-hot
// operands) and issue cost (independent and dependent chains, 1 and 2 waves per SIMD), next to the FP64 FMA.
// Background: DESIGN.md 4.2a -- could the dense phase of the one-wavefront multifrontal kernel let the matrix unit do
// its cross-lane reductions?   hipcc --offload-arch=gfx950 -O3 tools/micro_mfma64.hip -o tools/_bin/micro_mfma64
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));

// ---- layout probes: a[lane], b[lane] given, d[lane] (4x4x4: 1 value, 16x16x4: 4 values) returned ----
__global__ void probe_4x4x4(const double *a, const double *b, double *d, int cbsz, int abid) {
    const int l = threadIdx.x;
    double r = 0.0;
    if (cbsz == 0) r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
    else if (abid == 0) r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 2, 0, 0);
    else if (abid == 1) r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 2, 1, 0);
    else if (abid == 2) r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 2, 2, 0);
    else r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 2, 3, 0);
    d[l] = r;
}
__global__ void probe_16x16x4(const double *a, const double *b, double *d) {
    const int l = threadIdx.x;
    v4d c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[4 * l + r] = c[r];
}

// ---- issue cost: N instructions per wave in chains of length `dep` (1 = fully dependent) ----
template <int KIND, int CHAINS>
__global__ __launch_bounds__(256) void issue(double *sink, int iters, unsigned long long *cycles) {
    double acc[CHAINS];
    v4d acc4[CHAINS];
    const double x = 1.0 + 1e-9 * threadIdx.x, y = 1.0 - 1e-9 * threadIdx.x;
    for (int c = 0; c < CHAINS; ++c) { acc[c] = c; acc4[c] = v4d{(double)c, 1.0, 2.0, 3.0}; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (KIND == 0) acc[c] = __builtin_fma(acc[c], x, y);
            if (KIND == 1) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[c], 0, 0, 0);
            if (KIND == 2) acc4[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc4[c], 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c] + acc4[c][0] + acc4[c][1] + acc4[c][2] + acc4[c][3];
    if (s == 1.2345e-300) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int KIND, int CHAINS>
static void time_issue(const char *name, int waves_per_simd) {
    double *sink;
    unsigned long long *cyc, h = 0;
    CHECK(hipMalloc((void **)&sink, 64));
    CHECK(hipMalloc((void **)&cyc, 8));
    const int iters = 20000;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const int blocks = 256 * waves_per_simd;   // 4 waves per block: one per SIMD and block
    hipLaunchKernelGGL((issue<KIND, CHAINS>), blocks, 256, 0, 0, sink, 100, cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((issue<KIND, CHAINS>), blocks, 256, 0, 0, sink, iters, cyc);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    const double n = (double)iters * CHAINS;
    const double flop = KIND == 0 ? 128.0 : KIND == 1 ? 512.0 : 2048.0;
    printf("%-14s chains %2d, %d wave(s)/SIMD: %6.2f ns per wave instruction (%.1f ns per SIMD slot), s_memtime %.1f ticks/instr; %.1f TFLOP/s chip-wide\n", name,
           CHAINS, waves_per_simd, ms * 1e6 / n, ms * 1e6 / n / waves_per_simd, (double)h / n, flop * n * blocks * 4 / (ms * 1e-3) / 1e12);
}

int main() {
    double *a, *b, *d;
    CHECK(hipMalloc((void **)&a, 64 * 8));
    CHECK(hipMalloc((void **)&b, 64 * 8));
    CHECK(hipMalloc((void **)&d, 256 * 8));
    std::vector<double> ha(64), hb(64), hd(256);
    // 4x4x4_4B: A one-hot at lane la, B all-ones -> D non-zero where (block, i) match: which lanes?
    printf("== v_mfma_f64_4x4x4_4b: D[blk][i][j] = sum_k A[blk][i][k] B[blk][k][j]; one value per lane ==\n");
    for (int la : {0, 1, 4, 5, 16, 21}) {
        for (int l = 0; l < 64; ++l) { ha[l] = l == la ? 1.0 : 0.0; hb[l] = 1.0; }
        CHECK(hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe_4x4x4, 1, 64, 0, 0, a, b, d, 0, 0);
        CHECK(hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost));
        printf("  A one-hot at lane %2d, B = 1: D non-zero at lanes", la);
        for (int l = 0; l < 64; ++l) if (hd[l] != 0.0) printf(" %d", l);
        printf("\n");
    }
    for (int lb : {0, 1, 4, 5, 16, 21}) {
        for (int l = 0; l < 64; ++l) { hb[l] = l == lb ? 1.0 : 0.0; ha[l] = 1.0; }
        CHECK(hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe_4x4x4, 1, 64, 0, 0, a, b, d, 0, 0);
        CHECK(hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost));
        printf("  B one-hot at lane %2d, A = 1: D non-zero at lanes", lb);
        for (int l = 0; l < 64; ++l) if (hd[l] != 0.0) printf(" %d", l);
        printf("\n");
    }
    // pairing of k: A one-hot at la, B one-hot at lb: non-zero iff same block and same k
    printf("  k-pairing (A one-hot lane la, B one-hot lane lb -> lanes of D):\n");
    for (int la : {0, 1, 4}) for (int lb : {0, 1, 4, 5}) {
        for (int l = 0; l < 64; ++l) { ha[l] = l == la ? 1.0 : 0.0; hb[l] = l == lb ? 1.0 : 0.0; }
        CHECK(hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe_4x4x4, 1, 64, 0, 0, a, b, d, 0, 0);
        CHECK(hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost));
        printf("    la %d lb %d:", la, lb);
        for (int l = 0; l < 64; ++l) if (hd[l] != 0.0) printf(" %d", l);
        printf("\n");
    }
    // broadcast of A block `abid` to all four blocks (cbsz = 2)
    for (int abid = 0; abid < 4; ++abid) {
        for (int l = 0; l < 64; ++l) { ha[l] = (l / 16 == 1 && l % 16 == 0) ? 1.0 : 0.0; hb[l] = 1.0; }   // one-hot in block 1
        CHECK(hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe_4x4x4, 1, 64, 0, 0, a, b, d, 2, abid);
        CHECK(hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost));
        printf("  cbsz 2 abid %d, A one-hot at lane 16 (block 1): D non-zero at lanes", abid);
        for (int l = 0; l < 64; ++l) if (hd[l] != 0.0) printf(" %d", l);
        printf("\n");
    }
    printf("== v_mfma_f64_16x16x4: 4 values per lane ==\n");
    for (int la : {0, 1, 16, 17}) {
        for (int l = 0; l < 64; ++l) { ha[l] = l == la ? 1.0 : 0.0; hb[l] = 1.0; }
        CHECK(hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe_16x16x4, 1, 64, 0, 0, a, b, d);
        CHECK(hipMemcpy(hd.data(), d, 2048, hipMemcpyDeviceToHost));
        printf("  A one-hot at lane %2d: D non-zero at (lane, reg)", la);
        int n = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hd[4 * l + r] != 0.0 && n++ < 6) printf(" (%d,%d)", l, r);
        printf(" ... %d in all\n", n);
    }
    printf("== issue cost ==\n");
    for (int w : {1, 2}) {
        time_issue<0, 1>("v_fma_f64", w); time_issue<0, 4>("v_fma_f64", w); time_issue<0, 8>("v_fma_f64", w);
        time_issue<1, 1>("mfma_4x4x4", w); time_issue<1, 2>("mfma_4x4x4", w); time_issue<1, 4>("mfma_4x4x4", w); time_issue<1, 8>("mfma_4x4x4", w);
        time_issue<2, 1>("mfma_16x16x4", w); time_issue<2, 2>("mfma_16x16x4", w); time_issue<2, 4>("mfma_16x16x4", w);
    }
    return 0;
}
