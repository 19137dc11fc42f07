// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 (wave64), 1..2 waves per SIMD, dependent chains of
// length 1 (8 independent accumulators) so that the measurement is issue-bound, not latency-bound.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(64) void k(float* out, int iters, float a, float b) {
    f2 x[16];
    for (int i = 0; i < 16; i++) x[i] = f2{(float)threadIdx.x + i, (float)i};
    const f2 aa = {a, a}, bb = {b, b};
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (PK) x[u] = __builtin_elementwise_fma(x[u], aa, bb);
            else { x[u].x = __builtin_fmaf(x[u].x, a, b); x[u].y = __builtin_fmaf(x[u].y, a, b); }
        }
    }
    long long t1 = clock64();
    float s = 0; for (int i = 0; i < 16; i++) s += x[i].x + x[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0);
}
int main() {
    float* d; hipMalloc(&d, ((1 << 20) + 16) * 4);
    const int iters = 20000;
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        int blocks = 256 * 4 * waves_per_simd;
        for (int pk = 0; pk < 2; pk++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // instructions per wave: pk: 8 per iter (16 fma lanes-ops); scalar: 16 per iter
            double instr = (double)iters * (pk ? 16 : 32);
            printf("waves/SIMD %d  %s: %.3f ms, %.2f ns per instr per wave -> %.2f cycles @2.4GHz per wave-instr slot (x%d waves)\n", waves_per_simd,
                   pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms, ms * 1e6 / instr, ms * 1e6 / instr * 2.4 / waves_per_simd, waves_per_simd);
        }
    }
    return 0;
}
