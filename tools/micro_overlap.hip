// tools/micro_overlap.hip -- can a lone wavefront hide FP64 MFMAs in the latency shadow of a DEPENDENT FP64 chain?
// (the question behind look-ahead in the wide kernel's panel factorisation: one wave per SIMD, nobody else to fill the gaps)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro_overlap.hip -o tools/_bin/micro_overlap && tools/_bin/micro_overlap
#include <hip/hip_runtime.h>

#include <cstdio>

__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// MODE 0: dependent fma chain only; 1: independent MFMAs only (8 accumulators); 2: one MFMA per chain link; 3: one MFMA per two links;
// 4: dependent DPP-move + add chain (the reductions); 5: that with one MFMA per link
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, int iters, unsigned long long *ticks) {
    double x = threadIdx.x * 1e-3 + 1.0, y = 0.999999;
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i;
    const double a = x, b = y;
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0 || MODE == 2 || MODE == 3) x = fma(x, y, 1e-9);
            if (MODE == 4 || MODE == 5) {
                int lo = __double2loint(x), hi = __double2hiint(x);
                lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
                hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
                x = x + __hiloint2double(hi, lo) * 1e-9;
            }
            if (MODE == 1 || MODE == 2 || MODE == 5 || (MODE == 3 && (u & 1))) acc[u] = mfma4(a, b, acc[u]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = x;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

template <int MODE>
void run(const char *what, double *out, unsigned long long *dt, int blocks) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 10, dt);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, dt);
    unsigned long long t = 0;
    hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
    printf("%-64s %5d waves: %6.1f ticks per link\n", what, blocks, (double)t / (iters * 8.0));
}

int main() {
    double *out;
    unsigned long long *dt;
    hipMalloc(&out, 4096 * 64 * 8);
    hipMalloc(&dt, 8);
    for (int blocks : {1, 1024}) {
        run<0>("dependent v_fma_f64 chain", out, dt, blocks);
        run<1>("independent v_mfma_f64_4x4x4 (8 accumulators)", out, dt, blocks);
        run<2>("chain link + one independent MFMA", out, dt, blocks);
        run<3>("chain link + one MFMA every second link", out, dt, blocks);
        run<4>("dependent DPP move + mul + add chain", out, dt, blocks);
        run<5>("DPP chain link + one independent MFMA", out, dt, blocks);
    }
    return 0;
}
