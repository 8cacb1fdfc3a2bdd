// atomic_bench.hip — standalone microbenchmark (not part of the backend): throughput of scattered
// hardware f64 atomic adds into a W x H RGB sum buffer on MI355X, for the layouts considered for the
// BDPT framebuffer:  planar (R[], G[], B[]),  interleaved stride 3,  interleaved stride 4 (32-byte pixels).
// Usage: atomic_bench [width height n_updates]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// mode 0 planar, 1 stride-3, 2 stride-4; coherent != 0: lane i updates pixel (base + i) (one pixel-sample per lane)
__global__ void k_atomics(double* fb, unsigned long long npix, unsigned long long n, int mode, int coherent) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long p = coherent ? (i % npix) : (mix64(i + 12345) % npix);
    double v = 1.0 + (double)(i & 7);
    if (mode == 0) {
        unsafeAtomicAdd(fb + p, v); unsafeAtomicAdd(fb + npix + p, v); unsafeAtomicAdd(fb + 2 * npix + p, v);
    } else if (mode == 1) {
        unsafeAtomicAdd(fb + 3 * p, v); unsafeAtomicAdd(fb + 3 * p + 1, v); unsafeAtomicAdd(fb + 3 * p + 2, v);
    } else {
        unsafeAtomicAdd(fb + 4 * p, v); unsafeAtomicAdd(fb + 4 * p + 1, v); unsafeAtomicAdd(fb + 4 * p + 2, v);
    }
}

int main(int argc, char** argv) {
    unsigned long long w = argc > 1 ? atoll(argv[1]) : 512, h = argc > 2 ? atoll(argv[2]) : 512;
    unsigned long long n = argc > 3 ? atoll(argv[3]) : (1ull << 24);
    unsigned long long npix = w * h;
    double* fb;
    hipMalloc(&fb, sizeof(double) * 4 * npix);
    hipMemset(fb, 0, sizeof(double) * 4 * npix);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const char* names[3] = {"planar", "stride3", "stride4"};
    for (int coherent = 0; coherent < 2; coherent++)
        for (int mode = 0; mode < 3; mode++) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(a);
                hipLaunchKernelGGL(k_atomics, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, fb, npix, n, mode, coherent);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("%-8s %-9s %llux%llu  %llu updates (3 f64 atomics each): %.3f ms  -> %.2f G atomics/s\n", names[mode],
                   coherent ? "coherent" : "scattered", w, h, n, best, 3.0 * n / best * 1e-6);
        }
    return 0;
}
