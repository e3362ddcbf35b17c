// Issue cost of the mfw kernel's row operation and of its parts on gfx950 (cycles per instruction, s_memtime):
//   hipcc --offload-arch=gfx950 -O3 tools/micro_readlane.hip -o tools/_bin/micro_readlane && tools/_bin/micro_readlane
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, long long *cyc, int iters, int lane_sel) {
    double a[8], w = out[threadIdx.x & 63], acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = out[64 + q * 64 + (threadIdx.x & 63)];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // 2 independent fma per row
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = fma(-w, 1.0000001, a[q]);
#pragma unroll
            for (int q = 0; q < 8; q += 2) { acc0 = fma(w, a[q], acc0); acc1 = fma(w, a[q + 1], acc1); }
        } else if (MODE == 1) {     // the row op, staged over 8 rows: rl rl | fma | rl rl | fma
            double x[8], xn[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[q]), lane_sel), __builtin_amdgcn_readlane(__double2loint(a[q]), lane_sel));
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = fma(-x[q], w, a[q]);
#pragma unroll
            for (int q = 0; q < 8; ++q) xn[q] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[q]), lane_sel + 1), __builtin_amdgcn_readlane(__double2loint(a[q]), lane_sel + 1));
#pragma unroll
            for (int q = 0; q < 8; q += 2) { acc0 = fma(xn[q], a[q], acc0); acc1 = fma(xn[q + 1], a[q + 1], acc1); }
        } else if (MODE == 2) {     // half the broadcasts: rl rl | fma | fma (x carried)
            double x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[q]), lane_sel), __builtin_amdgcn_readlane(__double2loint(a[q]), lane_sel));
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = fma(-x[q], w, a[q]);
#pragma unroll
            for (int q = 0; q < 8; q += 2) { acc0 = fma(x[q], a[q], acc0); acc1 = fma(x[q + 1], a[q + 1], acc1); }
        } else if (MODE == 3) {     // dpp instead of readlane (row_bcast-like quad perm; wrong data, same issue pattern)
            double x[8], xn[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(a[q]), 0x55, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, __double2loint(a[q]), 0x55, 0xF, 0xF, true));
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = fma(-x[q], w, a[q]);
#pragma unroll
            for (int q = 0; q < 8; ++q) xn[q] = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(a[q]), 0xAA, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, __double2loint(a[q]), 0xAA, 0xF, 0xF, true));
#pragma unroll
            for (int q = 0; q < 8; q += 2) { acc0 = fma(xn[q], a[q], acc0); acc1 = fma(xn[q + 1], a[q + 1], acc1); }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = acc0 + acc1;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += a[q];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    double *out; long long *cyc;
    (void)hipMalloc(&out, 8 << 20); (void)hipMemset(out, 0, 8 << 20); (void)hipMalloc(&cyc, 8);
    const int iters = 2000;
    const char *names[] = {"2 fma per row", "row op: 4 readlane + 2 fma", "x carried: 2 readlane + 2 fma", "dpp row op: 4 dpp mov + 2 fma"};
    const int per_row[] = {2, 6, 4, 6};
    for (int wps : {1, 2, 4, 6, 8}) {          // waves per SIMD: wps workgroups of 4 waves on each of the 256 CUs
        for (int mode = 0; mode < 4; ++mode) {
            long long h = 0;
            float ms = 0.f;
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wps), dim3(256), 0, 0, out, cyc, iters, 3);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256 * wps), dim3(256), 0, 0, out, cyc, iters, 3);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256 * wps), dim3(256), 0, 0, out, cyc, iters, 3);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256 * wps), dim3(256), 0, 0, out, cyc, iters, 3);
                (void)hipEventRecord(e1, 0);
                (void)hipDeviceSynchronize();
                (void)hipEventElapsedTime(&ms, e0, e1);
                (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            }
            const double rows = (double)iters * 8;
            printf("%d waves/SIMD  %-32s %6.2f ticks/row/wave  kernel %7.1f us -> %6.2f ns per row per SIMD = %5.2f ns/instr\n", wps, names[mode], h / rows,
                   ms * 1e3, ms * 1e6 / rows / wps, ms * 1e6 / rows / wps / per_row[mode]);
        }
    }
    return 0;
}
