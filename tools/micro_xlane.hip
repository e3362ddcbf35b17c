// Issue cost (ns and ~cycles per instruction at saturation) of the cross-lane operations on gfx950:
//   hipcc --offload-arch=gfx950 -O3 tools/micro_xlane.hip -o tools/_bin/micro_xlane && tools/_bin/micro_xlane
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters) {
    int v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = out[q * 64 + (threadIdx.x & 63)];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            if (MODE == 0) {          // v_permlane32_swap: 1 instruction per pair
                auto r = __builtin_amdgcn_permlane32_swap((unsigned)v[q], (unsigned)v[q + 1], false, false);
                v[q] = (int)r[0]; v[q + 1] = (int)r[1];
            } else if (MODE == 1) {   // v_permlane16_swap
                auto r = __builtin_amdgcn_permlane16_swap((unsigned)v[q], (unsigned)v[q + 1], false, false);
                v[q] = (int)r[0]; v[q + 1] = (int)r[1];
            } else if (MODE == 2) {   // dpp row_ror:4 moves, 2 per pair
                v[q] = __builtin_amdgcn_update_dpp(0, v[q], 0x124, 0xF, 0xF, true);
                v[q + 1] = __builtin_amdgcn_update_dpp(0, v[q + 1], 0x124, 0xF, 0xF, true);
            } else if (MODE == 3) {   // v_readlane + v_mov back (2 + 2 per pair)
                v[q] = __builtin_amdgcn_readlane(v[q], 5) + 1;
                v[q + 1] = __builtin_amdgcn_readlane(v[q + 1], 7) + 1;
            } else if (MODE == 4) {   // plain v_add_u32, 2 per pair
                v[q] = v[q] * 3 + 1;
                v[q + 1] = v[q + 1] * 5 + 1;
            } else if (MODE == 5) {   // ds_bpermute, 2 per pair
                v[q] = __builtin_amdgcn_ds_bpermute(v[q + 1] & 0xFC, v[q]);
                v[q + 1] = __builtin_amdgcn_ds_bpermute(v[q] & 0xFC, v[q + 1]);
            }
        }
    }
    int s = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
}

int main() {
    int *out;
    (void)hipMalloc(&out, 8 << 20); (void)hipMemset(out, 0, 8 << 20);
    const int iters = 4000;
    const char *names[] = {"v_permlane32_swap", "v_permlane16_swap", "v_mov_dpp row_ror", "v_readlane + v_add", "v_mad + v_add (2 valu)", "ds_bpermute"};
    const double per_iter[] = {4, 4, 8, 16, 16, 8};   // instructions per loop iteration
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps : {2, 4}) {
        for (int mode = 0; mode < 6; ++mode) {
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
                if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
                (void)hipEventRecord(e1, 0);
                (void)hipDeviceSynchronize();
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double ns = ms * 1e6 / iters / wps / per_iter[mode];
            printf("%d waves/SIMD  %-24s %6.2f ns per instruction per SIMD (~%4.1f cycles at 2.1 GHz)\n", wps, names[mode], ns, ns * 2.1);
        }
    }
    return 0;
}
