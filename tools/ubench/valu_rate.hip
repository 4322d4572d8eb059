// Micro-benchmark: VALU issue rate on gfx950 (wave-instructions per second, chip-wide) for plain f32 FMA,
// packed f32 FMA, f16->f32 conversion and packed u16 min.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2_t p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    ushort2_t u0 = {(unsigned short)threadIdx.x, 3}, u1 = {5, 7}, u2 = {9, 11}, u3 = {13, 15};
    const float b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
                a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                p0 = __builtin_elementwise_fma(p0, float2_t{b, b}, float2_t{c, c}); p1 = __builtin_elementwise_fma(p1, float2_t{b, b}, float2_t{c, c});
                p2 = __builtin_elementwise_fma(p2, float2_t{b, b}, float2_t{c, c}); p3 = __builtin_elementwise_fma(p3, float2_t{b, b}, float2_t{c, c});
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                u0 = __builtin_elementwise_min(u0, u1) + u2; u1 = __builtin_elementwise_max(u1, u2) + u3;
                u2 = __builtin_elementwise_min(u2, u3) + u0; u3 = __builtin_elementwise_max(u3, u0) + u1;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                a0 = a0 + c; a1 = a1 * b; a2 = a2 + c; a3 = a3 * b; a4 = a4 + c; a5 = a5 * b; a6 = a6 + c; a7 = a7 * b;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                                 (float)(u0.x + u1.x + u2.y + u3.y);
}
template <int MODE>
void run(const char* name, double instr_per_iter) {
    float* d; hipMalloc(&d, 2048 * 512 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(512), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(512), 0, 0, d, iters);   // 4 blocks/CU x 8 waves = 8 waves/SIMD
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves = 1024.0 * 8, total = waves * iters * instr_per_iter;
    printf("%-22s %8.3f ms  %.3e wave-instr/s  = %.2f cycles/instr/SIMD at 2.4 GHz\n", name, ms, total / (ms * 1e-3),
           2.4e9 * 1024 / (total / (ms * 1e-3)));
    hipFree(d);
}
int main() {
    run<0>("v_fma_f32", 64); run<1>("v_pk_fma_f32", 64); run<2>("v_pk_min/max/add_u16", 128); run<3>("v_add/mul_f32", 64);
    return 0;
}
