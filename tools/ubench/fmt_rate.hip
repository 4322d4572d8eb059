// Micro-benchmark: phase A of k_front (RGBA8 -> f16 luminance into LDS, 22 staged rows per 16-row band of a 1280x720 frame,
// 256 frames) with the byte -> f32 conversion done (a) by the vector unit, as the kernel does it today, or (b) by the
// texture path: buffer_load_format_xyz on a typed buffer (DATA_FORMAT 8_8_8_8, NUM_FORMAT UNORM) hands every lane
// r/255, g/255, b/255 as binary32.  Also checks that (b) is bit-identical to (a) for all 2^24 (r, g, b).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fmt_rate.hip -o fmt_rate && ./fmt_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float3_t __attribute__((ext_vector_type(3)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef int int4_t __attribute__((ext_vector_type(4)));
__device__ float3_t fmt_load3(int4_t rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v3f32");
__device__ float4_t fmt_load4(int4_t rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
__device__ uint4_t raw_load4(int4_t rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");

constexpr int W = 1280, H = 720, R = 16, NT = 1024, LS = 1288;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t lum_pair_valu(uint32_t rgba0, uint32_t rgba1) {
    const float2_t rc_hi = {0x1.010102p-8f, 0x1.010102p-8f}, rc_lo = {-0x1.fdfdfep-33f, -0x1.fdfdfep-33f};
    const float2_t Rr = {(float)(rgba0 & 255u), (float)(rgba1 & 255u)};
    const float2_t G = {(float)((rgba0 >> 8) & 255u), (float)((rgba1 >> 8) & 255u)};
    const float2_t B = {(float)((rgba0 >> 16) & 255u), (float)((rgba1 >> 16) & 255u)};
    const float2_t tr = Rr * rc_lo, tg = G * rc_lo, tb = B * rc_lo;
    const float2_t r = __builtin_elementwise_fma(Rr, rc_hi, tr);
    const float2_t g = __builtin_elementwise_fma(G, rc_hi, tg);
    const float2_t b = __builtin_elementwise_fma(B, rc_hi, tb);
    const float2_t pr = r * 0.229f, pg = g * 0.587f, pb = b * 0.114f;
    const float2_t s = pr + pg;
    const float2_t l = s + pb;
    uint32_t d;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(l.x), "v"(l.y));
    return d;
}
__device__ __forceinline__ uint32_t lum_pair_fmt(float3_t a, float3_t b) {
    const float pr0 = a.x * 0.229f, pg0 = a.y * 0.587f, pb0 = a.z * 0.114f;
    const float pr1 = b.x * 0.229f, pg1 = b.y * 0.587f, pb1 = b.z * 0.114f;
    const float l0 = (pr0 + pg0) + pb0, l1 = (pr1 + pg1) + pb1;
    uint32_t d;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(l0), "v"(l1));
    return d;
}
__device__ __forceinline__ int4_t make_rsrc(const void* p, uint32_t bytes, uint32_t word3) {
    const uint64_t base = (uint64_t)p;
    int4_t r = {(int)(uint32_t)base, (int)((uint32_t)(base >> 32) & 0xffffu), (int)bytes, (int)word3};
    r.x = __builtin_amdgcn_readfirstlane(r.x);
    r.y = __builtin_amdgcn_readfirstlane(r.y);
    r.z = __builtin_amdgcn_readfirstlane(r.z);
    r.w = __builtin_amdgcn_readfirstlane(r.w);
    return r;
}
constexpr uint32_t kRawWord3 = 0x00020000u;  // raw dword buffer
constexpr uint32_t kFmtWord3 = 0x00050FACu;  // DST_SEL xyzw, NUM_FORMAT UNORM, DATA_FORMAT 8_8_8_8

// MODE 0: dwordx4 + vector-unit conversion (today).  1: format_xyz, a quad per lane (4 loads of stride 16 B).
// 2: format_xyzw, same.  3: format_xyz, a texel pair per lane (2 loads of stride 8 B, ds_write_b32).
// 4: raw loads only (no conversion): the memory floor of the phase.
template <int MODE>
__global__ __launch_bounds__(NT, 8) void k_stage(const uint8_t* frames, uint32_t* out, int n_bands) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint16_t* const grey = reinterpret_cast<uint16_t*>(lds_raw);
    const uint32_t L = blockIdx.x, xcd = L & 7u, slot = L >> 3;
    const uint32_t frame = (slot / n_bands) * 8u + xcd, band = slot % n_bands;
    const int tid = threadIdx.x, y0 = (int)band * R;
    const uint8_t* src0 = frames + (size_t)frame * (W * H * 4);
    const int4_t rs = make_rsrc(src0, W * H * 4, MODE == 0 || MODE == 4 ? kRawWord3 : kFmtWord3);
    if (MODE == 3) {
        constexpr int per_row = W / 2, n_items = (R + 6) * per_row;  // 14080 pairs
        for (int i0 = tid; i0 < n_items; i0 += NT * 4) {
            float3_t a[4], b[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u * NT;
                const int ly = (int)(((float)i + 0.5f) * (1.0f / (float)per_row)), tx = i - ly * per_row;
                const int gy = y0 - 3 + ly;
                const int off = ((H - 1 - gy) * W + tx * 2) * 4;
                dst[u] = i < n_items ? ly * LS + 8 + tx * 2 : -1;
                a[u] = fmt_load3(rs, off, 0, 0);
                b[u] = fmt_load3(rs, off + 4, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (dst[u] >= 0) *reinterpret_cast<uint32_t*>(grey + dst[u]) = lum_pair_fmt(a[u], b[u]);
        }
    } else {
        constexpr int per_row = W / 4, rpp = NT / per_row;
        const int ty = tid / per_row, tx = tid - ty * per_row;
        const bool lane_ok = ty < rpp;
        constexpr int U = (MODE == 0 || MODE == 4) ? 8 : 2;
        for (int lyb = ty; lyb < R + 6; lyb += rpp * U) {
            uint4_t v[U];
            float3_t f3[U][4];
            float4_t f4[U][4];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int ly = lyb + u * rpp, gy = y0 - 3 + ly;
                const int off = ((H - 1 - gy) * W + (lane_ok ? tx : 0) * 4) * 4;
                dst[u] = (lane_ok && ly < R + 6) ? ly * LS + 8 + tx * 4 : -1;
                if (MODE == 0 || MODE == 4) v[u] = raw_load4(rs, off, 0, 0);
                if (MODE == 1)
#pragma unroll
                    for (int k = 0; k < 4; k++) f3[u][k] = fmt_load3(rs, off + 4 * k, 0, 0);
                if (MODE == 2)
#pragma unroll
                    for (int k = 0; k < 4; k++) f4[u][k] = fmt_load4(rs, off + 4 * k, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (dst[u] >= 0) {
                    uint2 o;
                    if (MODE == 0) o = make_uint2(lum_pair_valu(v[u].x, v[u].y), lum_pair_valu(v[u].z, v[u].w));
                    if (MODE == 4) o = make_uint2(v[u].x ^ v[u].y, v[u].z ^ v[u].w);
                    if (MODE == 1) o = make_uint2(lum_pair_fmt(f3[u][0], f3[u][1]), lum_pair_fmt(f3[u][2], f3[u][3]));
                    if (MODE == 2) {
                        float3_t q[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) q[k] = float3_t{f4[u][k].x, f4[u][k].y, f4[u][k].z};
                        o = make_uint2(lum_pair_fmt(q[0], q[1]), lum_pair_fmt(q[2], q[3]));
                    }
                    *reinterpret_cast<uint2*>(grey + dst[u]) = o;
                }
            }
        }
    }
    __syncthreads();
    const uint32_t probe = reinterpret_cast<uint32_t*>(grey)[(tid * 37 + 4) % ((R + 6) * LS / 2)];
    if (probe == 0x12345679u) out[L] = probe;  // keeps the staging alive; never true for f16 pairs <= 1.0
}

// exactness: every (r, g, b), both conversions -> the same f16?
__global__ void k_fill_all(uint32_t* rgba) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    rgba[i] = i | 0xff000000u;
}
__global__ void k_check(const uint8_t* rgba, uint32_t n_texels, uint32_t* n_bad, uint32_t* first_bad) {
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) * 2u;
    const int4_t rs = make_rsrc(rgba, n_texels * 4u, kFmtWord3);
    const float3_t a = fmt_load3(rs, (int)(i * 4u), 0, 0), b = fmt_load3(rs, (int)(i * 4u + 4u), 0, 0);
    const uint32_t* p = reinterpret_cast<const uint32_t*>(rgba);
    const uint32_t want = lum_pair_valu(p[i], p[i + 1]), got = lum_pair_fmt(a, b);
    if (want != got) {
        if (atomicAdd(n_bad, 1u) == 0u) *first_bad = i;
    }
}
__global__ void k_noise(uint32_t* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        uint32_t a = (uint32_t)i * 2654435761u;
        a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16;
        p[i] = a;
    }
}

template <int MODE>
int run(const char* name, const uint8_t* frames, uint32_t* out, int n_frames) {
    const int n_bands = H / R;
    const size_t lds = 80 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stage<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_stage<MODE>, dim3(n_frames * n_bands), dim3(NT), lds, 0, frames, out, n_bands);
    CHECK(hipDeviceSynchronize());
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_stage<MODE>, dim3(n_frames * n_bands), dim3(NT), lds, 0, frames, out, n_bands);
    hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = (double)n_frames * W * H * 4;
    printf("%-44s %7.4f ms per %d frames  %.2f TB/s of RGBA\n", name, ms, n_frames, bytes / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    const int n_frames = 256;
    uint8_t* frames; uint32_t* out;
    const size_t nb = (size_t)n_frames * W * H * 4;
    CHECK(hipMalloc(&frames, nb)); CHECK(hipMalloc(&out, 1 << 20));
    hipLaunchKernelGGL(k_noise, dim3((unsigned)((nb / 4 + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<uint32_t*>(frames), nb / 4);
    CHECK(hipDeviceSynchronize());
    {   // exactness over all 2^24 colours
        uint32_t* all; uint32_t *n_bad, *first_bad;
        CHECK(hipMalloc(&all, (size_t)(1u << 24) * 4)); CHECK(hipMalloc(&n_bad, 8)); first_bad = n_bad + 1;
        CHECK(hipMemset(n_bad, 0, 8));
        hipLaunchKernelGGL(k_fill_all, dim3((1u << 24) / 256), dim3(256), 0, 0, all);
        hipLaunchKernelGGL(k_check, dim3((1u << 23) / 256), dim3(256), 0, 0, reinterpret_cast<const uint8_t*>(all), 1u << 24, n_bad, first_bad);
        uint32_t h[2]; CHECK(hipMemcpy(h, n_bad, 8, hipMemcpyDeviceToHost));
        printf("format-load luminance vs vector-unit luminance over all 2^24 (r,g,b): %u pairs differ (first at texel %u)\n", h[0], h[1]);
        hipFree(all); hipFree(n_bad);
    }
    if (run<4>("raw dwordx4 loads, no conversion", frames, out, n_frames)) return 1;
    if (run<0>("dwordx4 + vector-unit conversion (today)", frames, out, n_frames)) return 1;
    if (run<1>("format_xyz, quad per lane", frames, out, n_frames)) return 1;
    if (run<2>("format_xyzw, quad per lane", frames, out, n_frames)) return 1;
    if (run<3>("format_xyz, texel pair per lane", frames, out, n_frames)) return 1;
    return 0;
}
