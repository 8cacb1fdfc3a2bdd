// store_bench.hip — standalone microbenchmark (not part of the backend): what a kernel that only WRITES records gets
// out of the memory system of an MI355X, for the store patterns considered for the 192-byte vertex records
// (csrc/paths.h) and the 128-byte carry records (csrc/wavefront.h):
//   coalesced   lane i of the grid stores 16 bytes at 16 * i (a plain streaming store: the ceiling)
//   record      every lane stores its own record of R bytes with R / 16 stores of 16 bytes (what storeVertex does):
//               one store instruction touches 64 different lines
//   transposed  a wave stages its 64 records in LDS and writes them out so that consecutive lanes store consecutive
//               16-byte pieces (every store instruction is coalesced)
//   record8     the record pattern with 8-byte stores
// Usage: store_bench [records (default 2^24)] [record bytes (default 192)]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double v2d __attribute__((ext_vector_type(2)));

template <int PIECES>
__global__ __launch_bounds__(256) void k_record(double* out, unsigned long long n) {
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double* r = out + i * (2 * PIECES);
    double x = (double)i;
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
        v2d v; v.x = x + k; v.y = x - k;
        *reinterpret_cast<v2d*>(r + 2 * k) = v;
    }
}
template <int PIECES>
__global__ __launch_bounds__(256) void k_record8(double* out, unsigned long long n) {
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double* r = out + i * (2 * PIECES);
    double x = (double)i;
#pragma unroll
    for (int k = 0; k < 2 * PIECES; k++) r[k] = x + k;
}
__global__ __launch_bounds__(256) void k_coalesced(double* out, unsigned long long pieces) {
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= pieces) return;
    v2d v; v.x = (double)i; v.y = -(double)i;
    *reinterpret_cast<v2d*>(out + 2 * i) = v;
}
template <int PIECES>
__global__ __launch_bounds__(256) void k_transposed(double* out, unsigned long long n) {
    __shared__ v2d stage[4][64 * PIECES + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    double x = (double)i;
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
        v2d v; v.x = x + k; v.y = x - k;
        stage[wave][lane * PIECES + k] = v;
    }
    // (a wave reads back only what it wrote: no barrier needed)
    const unsigned long long base = ((unsigned long long)blockIdx.x * 256 + wave * 64) * PIECES;   // in 16-byte pieces
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
        const unsigned long long p = base + (unsigned long long)k * 64 + lane;
        if (p < n * PIECES) *reinterpret_cast<v2d*>(out + 2 * p) = stage[wave][k * 64 + lane];
    }
}

template <class F>
static float best_of(int reps, F&& launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(a);
        launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return best;
}

template <int PIECES>
static void run(double* out, unsigned long long n) {
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const double gb = (double)n * 16 * PIECES * 1e-9;
    float t;
    t = best_of(5, [&] { hipLaunchKernelGGL(k_coalesced, dim3((unsigned)((n * PIECES + 255) / 256)), dim3(256), 0, 0, out, n * PIECES); });
    printf("%3d-byte records  coalesced   %.3f ms  %.0f GB/s\n", 16 * PIECES, t, gb / t * 1e3);
    t = best_of(5, [&] { hipLaunchKernelGGL(k_record<PIECES>, dim3(blocks), dim3(256), 0, 0, out, n); });
    printf("%3d-byte records  record      %.3f ms  %.0f GB/s\n", 16 * PIECES, t, gb / t * 1e3);
    t = best_of(5, [&] { hipLaunchKernelGGL(k_record8<PIECES>, dim3(blocks), dim3(256), 0, 0, out, n); });
    printf("%3d-byte records  record8     %.3f ms  %.0f GB/s\n", 16 * PIECES, t, gb / t * 1e3);
    t = best_of(5, [&] { hipLaunchKernelGGL(k_transposed<PIECES>, dim3(blocks), dim3(256), 0, 0, out, n); });
    printf("%3d-byte records  transposed  %.3f ms  %.0f GB/s\n", 16 * PIECES, t, gb / t * 1e3);
}

int main(int argc, char** argv) {
    unsigned long long n = argc > 1 ? strtoull(argv[1], nullptr, 0) : (1ull << 24);
    n = (n + 255) / 256 * 256;
    double* out;
    if (hipMalloc(&out, n * 192 + 4096) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(out, 0, n * 192);
    printf("%llu records\n", n);
    run<4>(out, n);    // 64-byte line
    run<8>(out, n);    // 128-byte carry record
    run<12>(out, n);   // 192-byte vertex record
    hipFree(out);
    return 0;
}
