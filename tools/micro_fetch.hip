// micro_fetch.hip -- what does rocprofv3's FETCH_SIZE count on gfx950 for the access shapes of OUR kernels?
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ...
//  other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".)
// Every kernel reads a KNOWN number of bytes / distinct cache lines out of a 2 GiB buffer (8 x the 256 MiB Infinity Cache,
// each byte touched at most once per kernel, so nothing is served on-die):
//   stream16 / stream8 / stream4   coalesced, 16 / 8 / 4 B per lane                      -> bytes = N
//   gather8_l64, gather8_l128      8 B per lane, every lane its own 64-B / 128-B line     -> lines = lanes
//   gather24                       3 doubles (one 24-B centroid) per lane at a random cell of a packed [E][3] table
//   gather16_rows                  16 B per lane, 4 lanes share one 64-B record at a random position (descriptor-like)
// Build + run + counters: bash tools/micro_fetch.sh   (hipcc --offload-arch=gfx950; rocprofv3 --pmc FETCH_SIZE)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// a permutation of [0, 2^k) for every k: multiplication by an odd constant is a bijection modulo any power of two (the
// caller masks), and the xor-shift folds the high bits of the product down so neighbours do not land in neighbouring lines
__device__ __forceinline__ uint32_t mix(uint32_t x, uint32_t mask) {
    x = (x * 0x9E3779B1u) & mask;
    x ^= x >> 7;            // (bijective on the masked range: a unit upper-triangular GF(2) map)
    x = (x * 0x85EBCA6Bu) & mask;
    return x;
}

template <class V>
__global__ __launch_bounds__(256) void stream(const V *__restrict__ src, size_t n, double *__restrict__ sink) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const V v = src[i];
        acc += (double)((const uint32_t *)&v)[0];
    }
    if (acc == 1.2345e-300) sink[0] = acc;
}

// lane t reads 8 B at the start of line perm(t) (line = `line_bytes`); n_lines a power of two
__global__ __launch_bounds__(256) void gather8(const char *__restrict__ src, uint32_t n_lines, int line_shift, size_t lanes,
                                                double *__restrict__ sink) {
    double acc = 0.0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < lanes; t += (size_t)gridDim.x * blockDim.x) {
        const uint32_t line = mix((uint32_t)t, n_lines - 1);
        acc += *(const double *)(src + ((size_t)line << line_shift));
    }
    if (acc == 1.2345e-300) sink[0] = acc;
}

// lane t reads the 3 doubles of cell perm(t) of a packed [E][3] table (24-B records: 2.67 per 64-B line)
__global__ __launch_bounds__(256) void gather24(const double *__restrict__ src, uint32_t n_cells_pow2, size_t lanes,
                                                 double *__restrict__ sink) {
    double acc = 0.0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < lanes; t += (size_t)gridDim.x * blockDim.x) {
        const size_t c = mix((uint32_t)t, n_cells_pow2 - 1);
        acc += src[3 * c] + src[3 * c + 1] + src[3 * c + 2];
    }
    if (acc == 1.2345e-300) sink[0] = acc;
}

// 4 consecutive lanes read the four 16-B quarters of the 64-B record perm(t / 4)
__global__ __launch_bounds__(256) void gather16_rows(const char *__restrict__ src, uint32_t n_lines, size_t lanes,
                                                      double *__restrict__ sink) {
    double acc = 0.0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < lanes; t += (size_t)gridDim.x * blockDim.x) {
        const uint32_t line = mix((uint32_t)(t >> 2), n_lines - 1);
        const uint4 v = *(const uint4 *)(src + ((size_t)line << 6) + ((t & 3) << 4));
        acc += (double)v.x;
    }
    if (acc == 1.2345e-300) sink[0] = acc;
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    char *buf;
    double *sink;
    CHECK(hipMalloc((void **)&buf, bytes));
    CHECK(hipMalloc((void **)&sink, 64));
    CHECK(hipMemset(buf, 1, bytes));
    const int grid = 256 * 16;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    auto timed = [&](const char *name, double known_bytes, auto launch) {
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        launch();
        CHECK(hipEventRecord(b));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        printf("%-16s known %10.1f MiB  %.3f ms  %.0f GB/s\n", name, known_bytes / 1048576.0, ms, known_bytes / ms / 1e6);
    };
    timed("(warm-up)", (double)bytes, [&] { hipLaunchKernelGGL(stream<uint2>, grid, 256, 0, 0, (const uint2 *)buf, bytes / 8, sink); });
    timed("stream16", (double)bytes, [&] { hipLaunchKernelGGL(stream<uint4>, grid, 256, 0, 0, (const uint4 *)buf, bytes / 16, sink); });
    timed("stream8", (double)bytes, [&] { hipLaunchKernelGGL(stream<uint2>, grid, 256, 0, 0, (const uint2 *)buf, bytes / 8, sink); });
    timed("stream4", (double)bytes, [&] { hipLaunchKernelGGL(stream<uint32_t>, grid, 256, 0, 0, (const uint32_t *)buf, bytes / 4, sink); });
    // gathers: every line of the buffer touched exactly once (lanes == lines, perm is a bijection on the power-of-two range)
    const uint32_t l64 = (uint32_t)(bytes >> 6), l128 = (uint32_t)(bytes >> 7);
    timed("gather8_l64", (double)l64 * 64, [&] { hipLaunchKernelGGL(gather8, grid, 256, 0, 0, buf, l64, 6, (size_t)l64, sink); });
    timed("gather8_l128", (double)l128 * 128, [&] { hipLaunchKernelGGL(gather8, grid, 256, 0, 0, buf, l128, 7, (size_t)l128, sink); });
    const uint32_t cells = 1u << 26;   // 64 Mi cells x 24 B = 1.5 GiB, each read once
    timed("gather24", (double)cells * 24, [&] { hipLaunchKernelGGL(gather24, grid, 256, 0, 0, (const double *)buf, cells, (size_t)cells, sink); });
    timed("gather16_rows", (double)l64 * 64, [&] { hipLaunchKernelGGL(gather16_rows, grid, 256, 0, 0, buf, l64, (size_t)l64 * 4, sink); });
    timed("stream16 again", (double)bytes, [&] { hipLaunchKernelGGL(stream<uint4>, grid, 256, 0, 0, (const uint4 *)buf, bytes / 16, sink); });
    return 0;
}
