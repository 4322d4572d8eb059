// Micro-benchmark: ds_bpermute_b32 vs ds_read_u16 (random LDS gather) throughput per CU on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (unsigned short)(i * 7);
    __syncthreads();
    unsigned v = threadIdx.x * 2654435761u, acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (MODE == 0) v = (unsigned)__shfl((int)v, (int)((v >> 3) & 63)) + 1u;
            else v = lds[(v >> 3) & 4095] + v * 3u;
            acc += v;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE> void run(const char* name) {
    unsigned* d; hipMalloc(&d, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, d, 10); hipDeviceSynchronize();
    const int iters = 2000;
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = 2048.0 * 4 * iters * 16;
    printf("%-16s %8.3f ms  %.3e wave-ops/s = %.1f cycles per op per CU at 2.4 GHz\n", name, ms, ops / (ms * 1e-3), 2.4e9 * 256 / (ops / (ms * 1e-3)));
}
int main() { run<0>("ds_bpermute_b32"); run<1>("ds_read_u16 rand"); return 0; }
