// Micro-benchmark (round 5, for the matcher's (best, runner-up) update): issue cost of the integer / float min, max and median
// instructions on gfx950 (cycles per wave64 instruction per SIMD), eight independent chains per thread, 8 waves per SIMD -- the frame of
// op_rate.hip.  hipcc --offload-arch=gfx950 -O3 minmax_rate.hip -o minmax_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define KERNEL(NAME, ASM)                                                                              \
    __global__ __launch_bounds__(512) void NAME(unsigned* out, int iters) {                            \
        unsigned a[8];                                                                                 \
        for (int j = 0; j < 8; j++) a[j] = threadIdx.x * 2654435761u + j * 40503u;                     \
        unsigned b = threadIdx.x | 0x3c003c00u, c = 0x00010203u + blockIdx.x;                          \
        for (int i = 0; i < iters; i++) {                                                              \
            _Pragma("unroll") for (int r = 0; r < 8; r++) {                                           \
                _Pragma("unroll") for (int j = 0; j < 8; j++) asm volatile(ASM : "+v"(a[j]) : "v"(b), "v"(c) : "vcc"); \
            }                                                                                          \
        }                                                                                              \
        unsigned s = 0;                                                                                \
        for (int j = 0; j < 8; j++) s += a[j];                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                \
    }
KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL(k_max_u32, "v_max_u32 %0, %0, %1")
KERNEL(k_min_u32, "v_min_u32 %0, %0, %1")
KERNEL(k_max_i32, "v_max_i32 %0, %0, %1")
KERNEL(k_med3_u32, "v_med3_u32 %0, %0, %1, %2")
KERNEL(k_max3_u32, "v_max3_u32 %0, %0, %1, %2")
KERNEL(k_max_f32, "v_max_f32 %0, %0, %1")
KERNEL(k_med3_f32, "v_med3_f32 %0, %0, %1, %2")
KERNEL(k_max_u16, "v_max_u16 %0, %0, %1")
KERNEL(k_pk_max_u16, "v_pk_max_u16 %0, %0, %1")
KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
KERNEL(k_sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cmp_gt, "v_cmp_gt_u32 vcc, %0, %1")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_max_f16, "v_max_f16 %0, %0, %1")
KERNEL(k_pk_max_f16, "v_pk_max_f16 %0, %0, %1")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_med3_u16, "v_med3_u16 %0, %0, %1, %2")
KERNEL(k_med3_i16, "v_med3_i16 %0, %0, %1, %2")
KERNEL(k_med3_f16, "v_med3_f16 %0, %0, %1, %2")
KERNEL(k_min_u16, "v_min_u16 %0, %0, %1")
KERNEL(k_max3_u16, "v_max3_u16 %0, %0, %1, %2")
KERNEL(k_max_i16, "v_max_i16 %0, %0, %1")
KERNEL(k_sub_u16, "v_sub_u16 %0, %0, %1")
KERNEL(k_mov_sgpr, "v_mov_b32 %0, s4")

typedef void (*kern_t)(unsigned*, int);
void run(const char* name, kern_t k) {
    unsigned* d; hipMalloc(&d, 1024 * 512 * sizeof(unsigned));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(k, dim3(1024), dim3(512), 0, 0, d, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(1024), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double total = 1024.0 * 8 * iters * 64;
    printf("%-14s %8.3f ms  %.2f cycles/instr/SIMD at 2.4 GHz\n", name, ms, 2.4e9 * 1024 / (total / (ms * 1e-3)));
    hipFree(d);
}
int main() {
#define R(k) run(#k, k);
    R(k_add_u32) R(k_max_u32) R(k_min_u32) R(k_max_i32) R(k_med3_u32) R(k_max3_u32) R(k_max_f32) R(k_med3_f32) R(k_max_u16) R(k_pk_max_u16)
    R(k_pk_max_i16) R(k_sub_u32) R(k_xor) R(k_cndmask) R(k_cmp_gt) R(k_add3) R(k_max_f16) R(k_pk_max_f16) R(k_fma_f32) R(k_med3_u16) R(k_med3_i16) R(k_med3_f16) R(k_min_u16) R(k_max3_u16) R(k_max_i16) R(k_sub_u16) R(k_mov_sgpr)
    return 0;
}
