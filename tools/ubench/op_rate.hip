// Micro-benchmark: issue cost of individual gfx950 VALU instructions (cycles per wave64 instruction per SIMD),
// eight independent chains per thread, 8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 op_rate.hip -o op_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
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
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_add, "v_add_f32 %0, %0, %1")
KERNEL(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL(k_min, "v_min_f32 %0, %0, %1")
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL(k_cvt_ub, "v_cvt_f32_ubyte1 %0, %0")
KERNEL(k_cvt_f16, "v_cvt_f16_f32 %0, %0")
KERNEL(k_cvt_f32, "v_cvt_f32_f16 %0, %0")
KERNEL(k_cvt_i, "v_cvt_i32_f32 %0, %0")
KERNEL(k_pkmin, "v_pk_min_u16 %0, %0, %1")
KERNEL(k_pkaddh, "v_pk_add_f16 %0, %0, %1")
KERNEL(k_align, "v_alignbit_b32 %0, %0, %1, 16")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_addu, "v_add_u32 %0, %0, %1")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL(k_andor, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_lshlor, "v_lshl_or_b32 %0, %0, 3, %1")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_cmp, "v_cmp_gt_f32 vcc, %0, %1")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_fmamix, "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,1]")
KERNEL(k_mulf16, "v_mul_f16 %0, %0, %1")
KERNEL(k_min3, "v_min3_f32 %0, %0, %1, %2")
KERNEL(k_minu16, "v_min_u16 %0, %0, %1")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL(k_sad, "v_sad_u8 %0, %0, %1, %2")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_sdwa, "v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")

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
    printf("%-12s %8.3f ms  %.2f cycles/instr/SIMD at 2.4 GHz\n", name, ms, 2.4e9 * 1024 / (total / (ms * 1e-3)));
    hipFree(d);
}
int main() {
#define R(k) run(#k, k);
    R(k_fma) R(k_add) R(k_mul) R(k_min) R(k_med3) R(k_min3) R(k_cvt_ub) R(k_cvt_f16) R(k_cvt_f32) R(k_cvt_i) R(k_pkmin) R(k_pkaddh)
    R(k_align) R(k_and) R(k_lshl) R(k_bfe) R(k_mad24) R(k_addu) R(k_or3) R(k_andor) R(k_lshlor) R(k_perm) R(k_cmp) R(k_cndmask)
    R(k_fmamix) R(k_mulf16) R(k_minu16) R(k_mul_lo) R(k_rcp) R(k_bcnt) R(k_sad) R(k_mov) R(k_dpp) R(k_sdwa)
    return 0;
}
