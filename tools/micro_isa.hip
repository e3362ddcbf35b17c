// Issue-rate microbenchmarks for the GLS kernels' building blocks (gfx950): cycles per wave-instruction with
// 1, 2 and 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/micro_isa.hip -o /tmp/micro_isa && /tmp/micro_isa
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int MODE>
__global__ void bench(double *out, long long *cyc, int iters, int lanesel) {
    __shared__ double lds[8192];
    const int tid = threadIdx.x;
    double a = tid * 0.5 + 1.0, b = 1.0000001, c = 0.5, d = 0.25;
    int s0 = 0, s1 = 0;
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    double *p = lds + (tid & 63) + (tid >> 6) * 512;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // dependent DP FMA chain
            REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
        } else if (MODE == 1) {   // 4 independent DP FMA chains
            REP8(REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1.0000001), "v"(0.5));))
        } else if (MODE == 2) {   // v_readlane with an SGPR lane select, independent
            REP64(asm volatile("v_readlane_b32 %0, %2, %3\n v_readlane_b32 %1, %2, %3" : "=s"(s0), "=s"(s1) : "v"(tid), "s"(lanesel));)
        } else if (MODE == 3) {   // readlane pair -> DP FMA with the SGPR pair (the sweep's pattern)
            REP64(asm volatile("v_readlane_b32 s20, %1, %2\n v_readlane_b32 s21, %3, %2\n s_nop 1\n v_fma_f64 %0, s[20:21], %4, %0"
                               : "+v"(a) : "v"(tid), "s"(lanesel), "v"(tid), "v"(d) : "s20", "s21");)
        } else if (MODE == 4) {   // ds_read_b64 + ds_write_b64, conflict-free rows
            REP8(REP8({ double t = p[0]; asm volatile("" : "+v"(t)); p[64] = t; }))
        } else if (MODE == 5) {   // v_mov_b32 (plain 32-bit VALU), dependent
            REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(s0) : "v"(tid));)
        } else if (MODE == 6) {   // the whole row pattern: read, 2 rl, fma, write, 2 rl, fmac
            REP8(REP8({
                double t = p[0];
                asm volatile("s_waitcnt lgkmcnt(0)\n v_readlane_b32 s20, %0, %2\n v_readlane_b32 s21, %1, %2\n s_nop 1" :: "v"(__double2loint(t)), "v"(__double2hiint(t)), "s"(lanesel) : "s20", "s21");
                asm volatile("v_fma_f64 %0, -s[20:21], %1, %0" : "+v"(t) : "v"(c));
                p[0] = t;
                asm volatile("v_readlane_b32 s22, %0, %2\n v_readlane_b32 s23, %1, %2\n s_nop 1" :: "v"(__double2loint(t)), "v"(__double2hiint(t)), "s"(lanesel) : "s22", "s23");
                asm volatile("v_fma_f64 %0, s[22:23], %1, %0" : "+v"(a) : "v"(t));
                p += 0; }))
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + tid] = a + b + c + d + s0 + s1 + p[0];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int per_iter) {
    double *out; long long *cyc;
    hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&cyc, 4096 * 8);
    for (int waves : {1, 2, 4}) {   // waves per SIMD: one block per CU of 4 * waves wavefronts
        const int threads = 64 * 4 * waves, blocks = 256;
        const int iters = 200;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        bench<MODE><<<blocks, threads>>>(out, cyc, iters, 5);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        bench<MODE><<<blocks, threads>>>(out, cyc, iters, 5);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        printf("%-28s waves/SIMD=%d: %.3f ms, memtime ticks/instr/wave = %.2f, ns per instr per wave = %.3f\n", name, waves,
               ms, (double)h[0] / (iters * per_iter), ms * 1e6 / (iters * (double)per_iter));
    }
}

int main() {
    run<0>("dp fma dependent", 64);
    run<1>("dp fma 4 chains", 256);
    run<2>("readlane x2 indep", 128);
    run<3>("rl,rl,nop,fma(sgpr)", 256);
    run<4>("ds_read+ds_write b64", 128);
    run<5>("v_add_u32 dependent", 64);
    run<6>("row pattern (9 instr)", 64 * 9);
    return 0;
}
