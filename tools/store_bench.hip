// store_bench.hip — standalone microbenchmark (not part of the backend): what a kernel that only WRITES records gets
// out of the memory system of an MI355X, for the store patterns considered for the 192-byte vertex records
// (csrc/paths.h) and the 128-byte carry records (csrc/wavefront.h):
//   coalesced   lane i of the grid stores 16 bytes at 16 * i (a plain streaming store: the ceiling)
//   record      every lane stores its own record of R bytes with R / 16 stores of 16 bytes (what storeVertex does):
//               one store instruction touches 64 different lines
//   transposed  a wave stages its 64 records in LDS and writes them out so that consecutive lanes store consecutive
//               16-byte pieces (every store instruction is coalesced)
//   record8     the record pattern with 8-byte stores
//   permuted    as record, but the records of a workgroup's 256 lanes are permuted among the lanes (what a kernel sees
//               whose lanes were sorted by material) and every third record is left out (subpaths that ended)
//   permuted-coop  the same records, staged in LDS together with their addresses; consecutive lanes then store
//               consecutive pieces of ONE record (a store instruction touches 64 / pieces records instead of 64)
// and the same questions for kernels that only READ records (sum of what they read, so that nothing is optimised away):
//   read-record / read-permuted / read-permuted-coop (the lanes of a wave fetch the pieces of their 64 records so that
//   consecutive lanes load consecutive pieces of ONE record, through LDS) / read-coalesced
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

__device__ int g_ordered = 0;   // 1: the sparse patterns keep the lanes in record order (no permutation)
__device__ __forceinline__ unsigned permute256(unsigned t) { return g_ordered ? t : ((t * 101u + 37u) & 255u); }   // a bijection on 0..255
template <int PIECES>
__global__ __launch_bounds__(256) void k_permuted(double* out, unsigned long long n) {
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + permute256(threadIdx.x);
    if (i >= n || i % 3 == 2) return;
    double* r = out + i * (2 * PIECES);
    double x = (double)i;
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
        v2d v; v.x = x + k; v.y = x - k;
        *reinterpret_cast<v2d*>(r + 2 * k) = v;
    }
}
template <int PIECES>
__global__ __launch_bounds__(256) void k_permuted_coop(double* out, unsigned long long n) {
    __shared__ v2d stage[4][64 * PIECES];
    __shared__ double* dst[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + permute256(threadIdx.x);
    const bool on = i < n && i % 3 != 2;
    double x = (double)i;
    dst[wave][lane] = on ? out + i * (2 * PIECES) : nullptr;
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
        v2d v; v.x = x + k; v.y = x - k;
        stage[wave][lane * PIECES + k] = v;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
        const int p = k * 64 + lane;
        const int r = p / PIECES, q = p - r * PIECES;
        double* d = dst[wave][r];
        if (d) *reinterpret_cast<v2d*>(d + 2 * q) = stage[wave][p];
    }
}

template <int PIECES, bool PERMUTED>
__global__ __launch_bounds__(256) void k_read_record(const double* in, unsigned long long n, double* sink) {
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + (PERMUTED ? permute256(threadIdx.x) : threadIdx.x);
    if (i >= n || (PERMUTED && i % 3 == 2)) return;
    const double* r = in + i * (2 * PIECES);
    double acc = 0;
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
        v2d v = *reinterpret_cast<const v2d*>(r + 2 * k);
        acc += v.x + v.y;
    }
    if (acc == 12345.678) *sink = acc;
}
template <int PIECES>
__global__ __launch_bounds__(256) void k_read_coop(const double* in, unsigned long long n, double* sink) {
    __shared__ v2d stage[4][64 * PIECES];
    __shared__ const double* src[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + permute256(threadIdx.x);
    const bool on = i < n && i % 3 != 2;
    src[wave][lane] = on ? in + i * (2 * PIECES) : nullptr;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
        const int p = k * 64 + lane;
        const int r = p / PIECES, q = p - r * PIECES;
        const double* d = src[wave][r];
        v2d v; v.x = 0; v.y = 0;
        if (d) v = *reinterpret_cast<const v2d*>(d + 2 * q);
        stage[wave][p] = v;
    }
    __builtin_amdgcn_wave_barrier();
    double acc = 0;
#pragma unroll
    for (int k = 0; k < PIECES; k++) { v2d v = stage[wave][lane * PIECES + k]; acc += v.x + v.y; }
    if (acc == 12345.678) *sink = acc;
}
__global__ __launch_bounds__(256) void k_read_coalesced(const double* in, unsigned long long pieces, double* sink) {
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= pieces) return;
    v2d v = *reinterpret_cast<const v2d*>(in + 2 * i);
    if (v.x + v.y == 12345.678) *sink = v.x;
}

// every lane reads one record at a pseudo-random place of the array (a gather of whole records: what the connection
// kernels do with vertex records and the ray tracing kernels with tree nodes)
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
template <int PIECES>
__global__ __launch_bounds__(256) void k_read_random(const double* in, unsigned long long n, double* sink, int rounds) {
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    double acc = 0;
    for (int r = 0; r < rounds; r++) {
        const unsigned long long j = mix64(i * rounds + r + 77) % n;
        const double* rec = in + j * (2 * PIECES);
#pragma unroll
        for (int k = 0; k < PIECES; k++) {
            v2d v = *reinterpret_cast<const v2d*>(rec + 2 * k);
            acc += v.x + v.y;
        }
    }
    if (acc == 12345.678) *sink = acc;
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
    const double gb23 = gb * 2 / 3;
    t = best_of(5, [&] { hipLaunchKernelGGL(k_permuted<PIECES>, dim3(blocks), dim3(256), 0, 0, out, n); });
    printf("%3d-byte records  permuted       %.3f ms  %.0f GB/s\n", 16 * PIECES, t, gb23 / t * 1e3);
    t = best_of(5, [&] { hipLaunchKernelGGL(k_permuted_coop<PIECES>, dim3(blocks), dim3(256), 0, 0, out, n); });
    printf("%3d-byte records  permuted-coop  %.3f ms  %.0f GB/s\n", 16 * PIECES, t, gb23 / t * 1e3);
    double* sink = out + n * 24;
    t = best_of(5, [&] { hipLaunchKernelGGL(k_read_coalesced, dim3((unsigned)((n * PIECES + 255) / 256)), dim3(256), 0, 0, out, n * PIECES, sink); });
    printf("%3d-byte records  read-coalesced      %.3f ms  %.0f GB/s\n", 16 * PIECES, t, gb / t * 1e3);
    t = best_of(5, [&] { hipLaunchKernelGGL((k_read_record<PIECES, false>), dim3(blocks), dim3(256), 0, 0, out, n, sink); });
    printf("%3d-byte records  read-record         %.3f ms  %.0f GB/s\n", 16 * PIECES, t, gb / t * 1e3);
    t = best_of(5, [&] { hipLaunchKernelGGL((k_read_record<PIECES, true>), dim3(blocks), dim3(256), 0, 0, out, n, sink); });
    printf("%3d-byte records  read-permuted       %.3f ms  %.0f GB/s\n", 16 * PIECES, t, gb23 / t * 1e3);
    t = best_of(5, [&] { hipLaunchKernelGGL(k_read_coop<PIECES>, dim3(blocks), dim3(256), 0, 0, out, n, sink); });
    printf("%3d-byte records  read-permuted-coop  %.3f ms  %.0f GB/s\n", 16 * PIECES, t, gb23 / t * 1e3);
    // 2^22 lanes x 4 rounds = as many records as the array holds, each at a random place
    t = best_of(5, [&] { hipLaunchKernelGGL(k_read_random<PIECES>, dim3((unsigned)(n / 4 / 256)), dim3(256), 0, 0, out, n, sink, 4); });
    printf("%3d-byte records  read-random         %.3f ms  %.0f GB/s  (%.1f G records/s)\n", 16 * PIECES, t, gb / t * 1e3, n / t * 1e-6);
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
    int one = 1;
    hipMemcpyToSymbol(HIP_SYMBOL(g_ordered), &one, sizeof(int));
    printf("the permuted patterns once more with the lanes in record order (every third record still left out):\n");
    run<8>(out, n);
    run<12>(out, n);
    hipFree(out);
    return 0;
}
