// Issue cost of the packed binary32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two operations per lane on a
// register pair) against their scalar forms, cycles per wave64 instruction per SIMD, 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_t __attribute__((ext_vector_type(2)));
#define KERNEL(NAME, ASM)                                                                              \
    __global__ __launch_bounds__(512) void NAME(float* out, int iters) {                               \
        float2_t a[8];                                                                                 \
        for (int j = 0; j < 8; j++) a[j] = float2_t{(float)threadIdx.x * 1e-3f + j, 1.0f + j};        \
        float2_t b = {1.0001f, 0.9999f}, c = {1e-7f, -1e-7f};                                          \
        for (int i = 0; i < iters; i++) {                                                              \
            _Pragma("unroll") for (int r = 0; r < 8; r++) {                                           \
                _Pragma("unroll") for (int j = 0; j < 8; j++) asm volatile(ASM : "+v"(a[j]) : "v"(b), "v"(c)); \
            }                                                                                          \
        }                                                                                              \
        float s = 0;                                                                                   \
        for (int j = 0; j < 8; j++) s += a[j].x + a[j].y;                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                \
    }
KERNEL(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
KERNEL(k_pk_add, "v_pk_add_f32 %0, %0, %1")
KERNEL(k_mov64, "v_mov_b64 %0, %1")
typedef void (*kern_t)(float*, int);
void run(const char* name, kern_t k) {
    float* d; hipMalloc(&d, 1024 * 512 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(k, dim3(1024), dim3(512), 0, 0, d, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(1024), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double total = 1024.0 * 8 * iters * 64;
    printf("%-12s %8.3f ms  %.2f cycles/instr/SIMD at 2.4 GHz\n", name, ms, 2.4e9 * 1024 / (total / (ms * 1e-3)));
    hipFree(d);
}
int main() {
    run("v_pk_fma_f32", k_pk_fma); run("v_pk_mul_f32", k_pk_mul); run("v_pk_add_f32", k_pk_add); run("v_mov_b64", k_mov64);
    return 0;
}
