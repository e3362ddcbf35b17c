// Issue-rate microbenchmarks for the GLS kernels' building blocks (gfx950): cycles per wave-instruction with
// 1, 2 and 4 waves per SIMD.   mkdir -p tools/_bin && hipcc --offload-arch=gfx950 -O3 -w tools/micro_isa.hip -o tools/_bin/micro_isa  (git-ignored; travels to the
// GPU box with the tree), then on the box: tools/_bin/micro_isa
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
        } else if (MODE == 7) {   // ds_read_b64, every lane the same address (broadcast)
            double *pb = lds + (tid >> 6) * 512 + (it & 7);
            REP8(REP8({ double t = pb[0]; asm volatile("" : "+v"(t)); a += t; pb += 8; }))
        } else if (MODE == 8) {   // ds_read_b64, one address per lane (conflict-free)
            double *pb = lds + (tid >> 6) * 512 + (tid & 63);
            REP8(REP8({ double t = pb[0]; asm volatile("" : "+v"(t)); a += t; pb += 1; }))
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

// Replica of the block kernel's sweep: rows of `stride8` bytes, 8 loads in flight, read - fma - (write) - fmac.
template <int STORE, int RL>
__global__ void sweep(double *out, long long *cyc, int iters, int stride8, int rows, int lanesel, int rev, int live) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    for (int i = tid; i < 9600; i += blockDim.x) lds[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    char *col = reinterpret_cast<char *>(lds) + (rev ? (63 - lane) : lane) * 8;
    double acc0 = 0.0, acc1 = 0.0;
    const double w = 1e-12 * lane;
    for (int it = 0; it < iters; ++it) if (lane <= live) {
        double cur[8]; double *ca[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { ca[u] = reinterpret_cast<double *>(col + (wave + u * nw) * stride8); cur[u] = *ca[u]; }
        for (int g = 0; g < rows / 8; ++g) {
            double nx[8]; double *na[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int r = wave + ((g + 1) * 8 + u) * nw; r = r < rows * nw ? r : 0;
                na[u] = reinterpret_cast<double *>(col + r * stride8); nx[u] = *na[u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                double x = 1e-3 * u;
                if (RL) { int lo = __builtin_amdgcn_readlane(__double2loint(cur[u]), lanesel), hi = __builtin_amdgcn_readlane(__double2hiint(cur[u]), lanesel); x = __hiloint2double(hi, lo); }
                const double nv = fma(-x, w, cur[u]);
                if (STORE) *ca[u] = nv;
                double xn = 1e-3 * (u + 1);
                if (RL) { int lo = __builtin_amdgcn_readlane(__double2loint(nv), lanesel + 1), hi = __builtin_amdgcn_readlane(__double2hiint(nv), lanesel + 1); xn = __hiloint2double(hi, lo); }
                if (u & 1) acc1 = fma(xn, nv, acc1); else acc0 = fma(xn, nv, acc0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { cur[u] = nx[u]; ca[u] = na[u]; }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + tid] = acc0 + acc1;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int STORE, int RL>
void run_sweep(const char *name, int stride8, int rev, int blocks_per_cu, int live = 63) {
    double *out; long long *cyc;
    hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&cyc, 4096 * 8);
    const int nw = 4, rows = 32, iters = 200;
    const size_t lds = blocks_per_cu == 2 ? 80000 : 40000;
    hipFuncSetAttribute(reinterpret_cast<const void *>(sweep<STORE, RL>), hipFuncAttributeMaxDynamicSharedMemorySize, 80000);
    sweep<STORE, RL><<<256 * blocks_per_cu, 64 * nw, lds>>>(out, cyc, iters, stride8, rows, 5, rev, live);
    hipDeviceSynchronize();
    std::vector<long long> h(1); hipMemcpy(h.data(), cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s live %2d stride %4d B rev %d blocks/CU %d: ticks per row per wave = %.1f\n", name, live, stride8, rev, blocks_per_cu,
           (double)h[0] / (iters * rows));
}

// LDS throughput by lane stride: 8 independent ds_read_b64 (+ optional ds_write_b64) per iteration
template <int WRITE>
__global__ void lds_stride(double *out, long long *cyc, int iters, int lane_stride, int row_step) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 9800; i += blockDim.x) lds[i] = 1.0 + i;
    __syncthreads();
    double *p = lds + lane * lane_stride + wave;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[u * row_step];
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u]));
        if (WRITE) {
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u * row_step] = v[u] + 1.0;
        }
        a0 += v[0] + v[4]; a1 += v[1] + v[5]; a2 += v[2] + v[6]; a3 += v[3] + v[7];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + tid] = a0 + a1 + a2 + a3;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int WRITE>
void run_stride(int lane_stride, int row_step) {
    double *out; long long *cyc;
    hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&cyc, 4096 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void *>(lds_stride<WRITE>), hipFuncAttributeMaxDynamicSharedMemorySize, 80000);
    const int iters = 400;
    lds_stride<WRITE><<<512, 256, 80000>>>(out, cyc, iters, lane_stride, row_step);
    hipDeviceSynchronize();
    std::vector<long long> h(1); hipMemcpy(h.data(), cyc, 8, hipMemcpyDeviceToHost);
    printf("lds %s lane stride %4d doubles, row step %d: %.1f ticks per b64 access per wave (8 waves/CU) -> %.1f clk CU-wide\n",
           WRITE ? "read+write" : "read", lane_stride, row_step, (double)h[0] / (iters * 8 * (WRITE ? 2 : 1)),
           (double)h[0] / (iters * 8 * (WRITE ? 2 : 1)) / 8);
}

// LDS write cost by instruction flavour and exec mask: the hex8 kernel's "publish" (few active lanes)
template <int KIND>
__global__ void lds_write(double *out, long long *cyc, int iters, int active) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *p = lds + wave * 2048 + lane * 17;
    const double v0 = tid, v1 = tid + 1;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if ((lane & 7) < active) {
        for (int it = 0; it < iters; ++it) {
            if (KIND == 0) {   // 16 x ds_write_b64
#pragma unroll
                for (int u = 0; u < 16; ++u) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"((unsigned)(size_t)p), "v"(v0), "n"(u * 8) : "memory");
            } else if (KIND == 1) {   // 8 x ds_write2_b64
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" :: "v"((unsigned)(size_t)p), "v"(v0), "v"(v1), "n"(2 * u), "n"(2 * u + 1) : "memory");
            } else {   // 8 x ds_write_b128 (needs 16-byte alignment: lane * 17 doubles is odd -> use lane * 18)
                double *q = lds + wave * 2048 + lane * 18;
                typedef double d2 __attribute__((ext_vector_type(2)));
                d2 vv; vv.x = v0; vv.y = v1;
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"((unsigned)(size_t)q), "v"(vv), "n"(u * 16) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + tid] = lds[tid];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run_write(const char *name, int active) {
    double *out; long long *cyc;
    hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&cyc, 4096 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void *>(lds_write<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 80000);
    const int iters = 400;
    lds_write<KIND><<<256, 256, 80000>>>(out, cyc, iters, active);
    hipDeviceSynchronize();
    std::vector<long long> h(1); hipMemcpy(h.data(), cyc, 8, hipMemcpyDeviceToHost);
    printf("ldsw %-14s active lanes %2d/64: %.1f ticks per 16 doubles per wave (4 waves/CU) -> %.2f clk per double-per-lane CU-wide\n", name,
           active * 8, (double)h[0] / iters, (double)h[0] / iters / 16 / 4);
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
    for (int act : {8, 1}) { run_write<0>("ds_write_b64", act); run_write<1>("ds_write2_b64", act); run_write<2>("ds_write_b128", act); }
    for (int st : {1, 133, 129, 131, 135, 137, 141, 143, 145, 128, 67, 45}) run_stride<0>(st, 4);
    for (int st : {1, 133, 129, 137}) run_stride<1>(st, 4);
    run_sweep<1, 1>("sweep: read+write+2fma+4rl", 584, 1, 2, 61);
    run_sweep<1, 1>("sweep: read+write+2fma+4rl", 584, 1, 2, 30);
    for (int bpc : {2, 4}) {
        run_sweep<0, 0>("sweep: read+2fma", 584, 1, bpc);
        run_sweep<0, 0>("sweep: read+2fma", 512, 0, bpc);
        run_sweep<1, 0>("sweep: read+write+2fma", 584, 1, bpc);
        run_sweep<1, 1>("sweep: read+write+2fma+4rl", 584, 1, bpc);
        run_sweep<1, 1>("sweep: read+write+2fma+4rl", 512, 0, bpc);
    }
    run<0>("dp fma dependent", 64);
    run<1>("dp fma 4 chains", 256);
    run<2>("readlane x2 indep", 128);
    run<3>("rl,rl,nop,fma(sgpr)", 256);
    run<4>("ds_read+ds_write b64", 128);
    run<5>("v_add_u32 dependent", 64);
    run<7>("ds_read_b64 broadcast + add", 128);
    run<8>("ds_read_b64 per-lane + add", 128);
    run<6>("row pattern (9 instr)", 64 * 9);
    return 0;
}
