// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  Two spin kernels of ~100 us on a handful of blocks each:
// in-order they take the sum, overlapped the maximum.   hipcc --offload-arch=gfx950 -O2 -o anyorder anyorder.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long ticks, int* out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] += 1;
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    int* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    const long long ticks = 10000;      // wall_clock64: 100 MHz -> 100 us
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            for (int it = 0; it < 20; it++) {
                hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, ticks, d);
                if (mode == 0) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, ticks, d + 8);
                else hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, nullptr, nullptr, mode == 1 ? hipExtAnyOrderLaunch : 0, ticks, d + 8);
            }
            hipStreamSynchronize(s);
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20;
            printf("mode %d (%s): %.1f us per pair of 100-us kernels\n", mode, mode == 0 ? "hipLaunchKernelGGL x2" : (mode == 1 ? "second with hipExtAnyOrderLaunch" : "second with hipExtLaunchKernelGGL flags 0"), us);
        }
    }
    int h[16]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("counts %d %d  err %s\n", h[0], h[8], hipGetErrorString(hipGetLastError()));
    return 0;
}
