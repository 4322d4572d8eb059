// Is the hardware's UNORM8 -> f32 conversion of buffer_load_format_xyzw (DATA_FORMAT 8_8_8_8, NUM_FORMAT UNORM)
// exactly fl32(byte / 255.0f) for all 256 bytes?  hipcc --offload-arch=gfx950 -O3 unorm_fmt.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef int int4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));
__global__ void k(const uint8_t* src, float4_t* dst, uint32_t nbytes) {
    const uint64_t base = (uint64_t)src;
    int4_t rsrc;
    rsrc.x = (int)(uint32_t)base;
    rsrc.y = (int)(uint32_t)(base >> 32) & 0xffff;  // stride 0
    rsrc.z = (int)nbytes;
    rsrc.w = 0x50FAC;  // DST_SEL xyzw, NUM_FORMAT UNORM(0), DATA_FORMAT 8_8_8_8 (10)
    rsrc.x = __builtin_amdgcn_readfirstlane(rsrc.x);
    rsrc.y = __builtin_amdgcn_readfirstlane(rsrc.y);
    rsrc.z = __builtin_amdgcn_readfirstlane(rsrc.z);
    rsrc.w = __builtin_amdgcn_readfirstlane(rsrc.w);
    const uint32_t off = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    float4_t v;
    asm volatile("buffer_load_format_xyzw %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(rsrc) : "memory");
    dst[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main() {
    uint8_t h[1024];
    for (int i = 0; i < 256; i++) { h[4 * i] = i; h[4 * i + 1] = 255 - i; h[4 * i + 2] = (i * 7) & 255; h[4 * i + 3] = (i * 13 + 5) & 255; }
    uint8_t* d; float4_t* o; hipMalloc(&d, 1024); hipMalloc(&o, 256 * 16);
    hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(4), dim3(64), 0, 0, d, o, 1024u);
    float out[1024]; hipMemcpy(out, o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; i++) {
        float want = (float)h[i] / 255.0f;
        if (memcmp(&want, &out[i], 4)) { if (bad < 10) printf("byte %u: got %.9g want %.9g\n", h[i], out[i], want); bad++; }
    }
    printf("unorm8 format load: %d of 1024 differ from byte/255.0f\n", bad);
    return 0;
}
