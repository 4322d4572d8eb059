// Micro-benchmark: the matcher's inner loop (8 xor, 8 popcount, 3 add3, key, max, min, min) with register operands only.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed) {
    unsigned a[8];
    for (int j = 0; j < 8; j++) a[j] = threadIdx.x * 2654435761u + j * 40503u;
    unsigned k1 = ~0u, k2 = ~0u;
    unsigned b0 = seed, b1 = seed * 3u, b2 = seed * 5u, b3 = seed * 7u, b4 = seed * 11u, b5 = seed * 13u, b6 = seed * 17u, b7 = seed * 19u;
    for (int i = 0; i < iters; i++) {
        unsigned d = __builtin_popcount(a[0] ^ b0) + __builtin_popcount(a[1] ^ b1) + __builtin_popcount(a[2] ^ b2) + __builtin_popcount(a[3] ^ b3) +
                     __builtin_popcount(a[4] ^ b4) + __builtin_popcount(a[5] ^ b5) + __builtin_popcount(a[6] ^ b6) + __builtin_popcount(a[7] ^ b7);
        unsigned key = (d << 23) | (unsigned)i;
        k2 = min(k2, max(k1, key));
        k1 = min(k1, key);
        b0 = b0 * 1664525u + 1013904223u; b1 += b0; b2 ^= b1; b3 += b2; b4 ^= b3; b5 += b4; b6 ^= b5; b7 += b6;  // scalar (uniform) updates
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = k1 + k2;
}
int main() {
    unsigned* d; hipMalloc(&d, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 2048, 2048, 1024}) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 7u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 16000, 7u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double waves_per_simd = blocks * 4.0 / 1024.0;
        printf("%d blocks (%.0f waves/SIMD): %.3f ms -> %.1f cycles per iteration per SIMD-wave-slot, %.1f per wave\n", blocks, waves_per_simd, ms,
               ms * 1e-3 * 2.4e9 / 16000 / waves_per_simd, ms * 1e-3 * 2.4e9 / 16000);
    }
    return 0;
}
