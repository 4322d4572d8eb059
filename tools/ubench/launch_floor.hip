// Launch-latency floors of one blocking call (the reference's call shape): what a hipStreamSynchronize costs against
// polling a flag the kernel writes into pinned host memory, one launch against two dependent ones, two streams.
// build: hipcc --offload-arch=gfx950 -O2 -o launch_floor launch_floor.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void k_null(volatile uint32_t* flag, uint32_t seq) {
    if (flag && threadIdx.x == 0 && blockIdx.x == 0) {
        __threadfence_system();
        *flag = seq;
    }
}
__global__ void k_spin(uint32_t ns) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ns / 10) {}  // 100 MHz
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s, s2;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    uint32_t* h;
    hipHostMalloc(&h, 64, hipHostMallocDefault);
    uint32_t* d;
    hipHostGetDevicePointer((void**)&d, h, 0);
    *h = 0;
    const int N = 2000;
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    for (int mode = 0; mode < 9; mode++) {
        double t = 0;
        uint32_t seq = 0;
        for (int i = 0; i < N + 50; i++) {
            if (i == 50) t = now();
            seq++;
            switch (mode) {
                case 0: hipLaunchKernelGGL(k_null, dim3(64), dim3(256), 0, s, nullptr, 0u); hipStreamSynchronize(s); break;
                case 1: hipLaunchKernelGGL(k_null, dim3(64), dim3(256), 0, s, d, seq); while (*(volatile uint32_t*)h != seq) {} break;
                case 2: hipLaunchKernelGGL(k_null, dim3(64), dim3(256), 0, s, nullptr, 0u); hipLaunchKernelGGL(k_null, dim3(64), dim3(256), 0, s, nullptr, 0u); hipStreamSynchronize(s); break;
                case 3: hipLaunchKernelGGL(k_null, dim3(64), dim3(256), 0, s, nullptr, 0u); hipLaunchKernelGGL(k_null, dim3(64), dim3(256), 0, s, d, seq); while (*(volatile uint32_t*)h != seq) {} break;
                case 4: hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, 10000u); hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, 10000u); hipStreamSynchronize(s); break;
                case 6: hipLaunchKernelGGL(k_null, dim3(64), dim3(256), 0, s, nullptr, 0u); while (hipStreamQuery(s) == hipErrorNotReady) {} break;
                case 7: hipLaunchKernelGGL(k_null, dim3(64), dim3(256), 0, s, nullptr, 0u); hipEventRecord(ev, s); while (hipEventQuery(ev) == hipErrorNotReady) {} break;
                case 8: hipLaunchKernelGGL(k_null, dim3(64), dim3(256), 0, s, nullptr, 0u); hipLaunchKernelGGL(k_null, dim3(64), dim3(256), 0, s, nullptr, 0u); while (hipStreamQuery(s) == hipErrorNotReady) {} break;
                case 5: hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, 10000u); hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s2, 10000u); hipStreamSynchronize(s); hipStreamSynchronize(s2); break;
            }
        }
        const char* names[] = {"1 launch + hipStreamSynchronize", "1 launch + host flag polled", "2 launches + hipStreamSynchronize", "2 launches + host flag polled",
                               "2 x 10 us kernels, one stream, sync", "2 x 10 us kernels, two streams, 2 syncs",
                               "1 launch + hipStreamQuery spin", "1 launch + event + hipEventQuery spin", "2 launches + hipStreamQuery spin"};
        printf("%-44s %7.2f us\n", names[mode], (now() - t) / N);
    }
    return 0;
}
