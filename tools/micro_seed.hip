// Accuracy of the hardware seeds v_rcp_f64 / v_rsq_f64 and of the correction steps behind them (gls_device_math.hpp):
//   hipcc --offload-arch=gfx950 -O3 -I ninpol_amd/csrc tools/micro_seed.hip -o tools/_bin/micro_seed && tools/_bin/micro_seed
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "gls_device_math.hpp"
using namespace nin::glsmath;
constexpr int NV = 6;
__global__ void k(const double *x, double *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    o[NV * i + 0] = __builtin_amdgcn_rcp(d);
    o[NV * i + 1] = rcp_newton(d, 1);
    o[NV * i + 2] = rcp_newton(d, 2);
    o[NV * i + 3] = fast_rcp(d);
    o[NV * i + 4] = __builtin_amdgcn_rsq(d);
    o[NV * i + 5] = fast_rsqrt(d);
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), o(NV * (size_t)n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = std::ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 41) - 20); }
    double *dx, *dout;
    if (hipMalloc(&dx, n * 8) != hipSuccess || hipMalloc(&dout, NV * (size_t)n * 8) != hipSuccess) return 1;
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    (void)hipMemcpy(o.data(), dout, NV * (size_t)n * 8, hipMemcpyDeviceToHost);
    const char *nm[NV] = {"v_rcp_f64", "  + one Newton step", "  + two Newton steps", "  + one cubic step (fast_rcp)", "v_rsq_f64", "  + one cubic step (fast_rsqrt)"};
    for (int c = 0; c < NV; ++c) {
        long double worst = 0;
        for (int i = 0; i < n; ++i) {
            long double ex = c < 4 ? 1.0L / (long double)x[i] : 1.0L / sqrtl((long double)x[i]);
            long double e = fabsl(((long double)o[NV * (size_t)i + c] - ex) / ex);
            if (e > worst) worst = e;
        }
        printf("%-34s max relative error %.3Le (%.2Lf ulp of double)\n", nm[c], worst, worst / 1.1102230246251565e-16L);
    }
    return 0;
}
