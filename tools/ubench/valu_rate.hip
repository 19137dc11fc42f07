// Micro-benchmark (round 6): how many cycles does a SIMD of gfx950 need per wave64 VALU instruction?
// The guide (MI355X_MICROARCH.md "Per-instruction cycle constants") says v_fma_f32 = 2 cycles (SIMD-32); tools/ubench/pkfma.hip
// measured 4.3 with 64-thread workgroups and WALL time at an assumed 2.4 GHz.  This one removes both assumptions:
//   * the instruction stream is inline asm (32 or 64 independent accumulators, back to back; the ISA excerpt is dumped by
//     tools/ubench/valu_rate.sh with llvm-objdump);
//   * every wave records s_memtime (shader clock) and s_memrealtime (100 MHz) around its loop and its HW_ID / XCC_ID, so the
//     report has cycles per instruction in the wave's OWN clock, the clock frequency under this load, and the PLACEMENT:
//     how many waves shared a SIMD (the r5 verdict's hypothesis: one-wave workgroups double up);
//   * workgroups of 64 / 256 / 512 / 1024 threads at 1, 2, 4, 8 waves per SIMD;
//   * operand variants: VOP3 fma with SGPR / VGPR multiplicand, VOP2 mul / add, packed fma / mul, and a dependent chain (latency).
// Output: one line per (variant, workgroup, waves per SIMD).  "cyc/instr/SIMD" = median over waves of (wave's cycles / its
// instructions) divided by the waves resident on its SIMD (max over the kernel, from HW_ID).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define HWREG(id) (((32 - 1) << 11) | (0 << 6) | (id))
constexpr int HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20;

struct Rec { unsigned long long c0, c1, r0, r1; unsigned hw, xcc; };

enum Variant { FMA_S = 0, FMA_V, MUL_S, ADD_V, PKFMA_V, PKFMA_S, PKMUL_V, FMA_DEP, FMA_BANK, MUL_V, ADD_S, FMAC_V, FMA_ACC, MAX_V, PKADD_V, MOV_V, NVAR };
static const char* vname[NVAR] = {"v_fma_f32 x,x,s,v", "v_fma_f32 x,x,v,v", "v_mul_f32 x,s,x", "v_add_f32 x,v,x", "v_pk_fma_f32 x,x,v,v",
                                  "v_pk_fma_f32 x,x,s,v", "v_pk_mul_f32 x,x,v", "v_fma_f32 dependent", "v_fma_f32 x,x,x+1,x+2",
                                  "v_mul_f32 x,v,x", "v_add_f32 x,s,x", "v_fmac_f32 x,v,v", "v_fma_f32 x,v,v,x", "v_max_f32 x,v,x", "v_pk_add_f32 x,x,v", "v_mov_b32 x,v"};
static const int vflops[NVAR] = {2, 2, 1, 1, 4, 4, 2, 2, 2, 1, 1, 2, 2, 1, 2, 0};

#define REP32(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19) M(20) M(21) \
    M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ void k(Rec* rec, float* sink, int iters, float a, float b) {
    float x[32];
    f2 y[32];
    for (int i = 0; i < 32; i++) { x[i] = (float)threadIdx.x + i; y[i] = f2{x[i], (float)i}; }
    float va = a + 0.0f * threadIdx.x, vb = b + 0.0f * threadIdx.x;           // VGPR copies
    asm volatile("" : "+v"(va), "+v"(vb));
    f2 va2 = {va, va}, vb2 = {vb, vb};
    asm volatile("" : "+v"(va2), "+v"(vb2));
    unsigned long long sa2 = ((unsigned long long)__float_as_uint(a) << 32) | __float_as_uint(a);
    __builtin_amdgcn_s_barrier();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (V == FMA_S) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(vb));
            REP32(M) REP32(M)
#undef M
        } else if (V == FMA_V) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(va), "v"(vb));
            REP32(M) REP32(M)
#undef M
        } else if (V == MUL_S) {
#define M(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "s"(a));
            REP32(M) REP32(M)
#undef M
        } else if (V == ADD_V) {
#define M(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "v"(vb));
            REP32(M) REP32(M)
#undef M
        } else if (V == PKFMA_V) {
#define M(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(va2), "v"(vb2));
            REP32(M) REP32(M)
#undef M
        } else if (V == PKFMA_S) {
#define M(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "s"(sa2), "v"(vb2));
            REP32(M) REP32(M)
#undef M
        } else if (V == PKMUL_V) {
#define M(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(va2));
            REP32(M) REP32(M)
#undef M
        } else if (V == MUL_V) {
#define M(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "v"(va));
            REP32(M) REP32(M)
#undef M
        } else if (V == ADD_S) {
#define M(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "s"(b));
            REP32(M) REP32(M)
#undef M
        } else if (V == FMAC_V) {
#define M(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(va), "v"(vb));
            REP32(M) REP32(M)
#undef M
        } else if (V == FMA_ACC) {
#define M(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(va), "v"(vb));
            REP32(M) REP32(M)
#undef M
        } else if (V == MAX_V) {
#define M(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(x[i]) : "v"(vb));
            REP32(M) REP32(M)
#undef M
        } else if (V == PKADD_V) {
#define M(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(vb2));
            REP32(M) REP32(M)
#undef M
        } else if (V == MOV_V) {
#define M(i) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(vb));
            REP32(M) REP32(M)
#undef M
        } else if (V == FMA_DEP) {                       // one chain: issue-to-issue latency of a dependent fma
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "s"(a), "v"(vb));
            REP32(M) REP32(M)
#undef M
        } else if (V == FMA_BANK) {                      // three different VGPR sources per instruction (register-bank pressure)
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) & 31]), "v"(x[(i + 2) & 31]));
            REP32(M) REP32(M)
#undef M
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    if (V == PKFMA_V || V == PKFMA_S || V == PKMUL_V || V == PKADD_V) { for (int i = 0; i < 32; i++) s += y[i].x + y[i].y; }      // (the unused array is dead: 8 waves per SIMD need <= 64 VGPRs)
    else { for (int i = 0; i < 32; i++) s += x[i]; }
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        rec[w] = Rec{c0, c1, r0, r1, (unsigned)__builtin_amdgcn_s_getreg(HWREG(HW_REG_HW_ID)), (unsigned)__builtin_amdgcn_s_getreg(HWREG(HW_REG_XCC_ID))};
    }
}

typedef void (*kern_t)(Rec*, float*, int, float, float);
static kern_t kerns[NVAR] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>};

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int n_cu = pr.multiProcessorCount, n_simd = n_cu * 4;
    printf("# device %s, %d CUs (%d SIMDs), clockRate %.0f MHz; %d iterations x 64 instructions per wave\n", pr.name, n_cu, n_simd, pr.clockRate / 1e3, iters);
    printf("# columns: variant | wg threads | waves/SIMD asked | wall ms | wave cycles/instr (median) | sclk MHz (s_memtime / s_memrealtime) |"
           " SIMDs used | waves per SIMD over the kernel max / median | waves CONCURRENTLY on a SIMD max / median | first wave's start to last wave's end |"
           " SIMD cycles per wave-instruction = span x sclk / (instructions per wave x waves per SIMD) | TFLOP/s from the HIP-event time\n");
    Rec* d; float* sink;
    const int max_waves = n_simd * 8;
    hipMalloc(&d, sizeof(Rec) * max_waves); hipMalloc(&sink, 64);
    std::vector<Rec> h(max_waves);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs[4] = {64, 256, 512, 1024};
    for (int v = 0; v < NVAR; v++)
        for (int wi = 0; wi < 4; wi++)
            for (int wps = 1; wps <= 8; wps *= 2) {
                const int wg = wgs[wi], waves = n_simd * wps;
                if (wg / 64 > 4 * wps) continue;                          // the workgroup alone would exceed the asked occupancy
                if ((v == PKFMA_V || v == PKFMA_S || v == PKMUL_V || v == PKADD_V) && wps == 8) continue;                   // 64 + 6 VGPR pairs: the packed variants do not fit 8 waves per SIMD
                const int blocks = waves * 64 / wg;
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(kerns[v], dim3(blocks), dim3(wg), 0, 0, d, sink, iters, 1.0001f, 0.5f);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                if (hipGetLastError() != hipSuccess) { printf("%s wg %d: launch failed\n", vname[v], wg); continue; }
                hipMemcpy(h.data(), d, sizeof(Rec) * waves, hipMemcpyDeviceToHost);
                // placement: waves per (xcc, se, sh, cu, simd); only waves whose time windows overlap the median window count as co-resident
                std::map<unsigned, int> per_simd;
                auto key = [](const Rec& r) { return ((r.xcc & 0xf) << 16) | (r.hw & 0xfff0) ; };      // hw[15:4]: simd, pipe, cu, sh, se
                for (int w = 0; w < waves; w++) per_simd[key(h[w])]++;
                std::vector<double> cpi(waves), mhz(waves);
                const double instr = (double)iters * 64;
                for (int w = 0; w < waves; w++) {
                    cpi[w] = (double)(h[w].c1 - h[w].c0) / instr;
                    mhz[w] = (double)(h[w].c1 - h[w].c0) / ((double)(h[w].r1 - h[w].r0) / 100.0);
                }
                // waves CONCURRENTLY on a SIMD: the largest number of [r0, r1] intervals (100 MHz counter, common to the chip) of one SIMD's waves that
                // overlap -- the count above is over the whole kernel, and a grid the chip cannot hold at once runs in rounds
                std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;
                unsigned long long t_lo = ~0ull, t_hi = 0;
                for (int w = 0; w < waves; w++) { ev[key(h[w])].push_back({h[w].r0, +1}); ev[key(h[w])].push_back({h[w].r1, -1}); t_lo = std::min(t_lo, h[w].r0); t_hi = std::max(t_hi, h[w].r1); }
                std::vector<int> conc;
                for (auto& kv : ev) { std::sort(kv.second.begin(), kv.second.end()); int c = 0, m = 0; for (auto& e : kv.second) { c += e.second; m = std::max(m, c); } conc.push_back(m); }
                std::sort(conc.begin(), conc.end());
                std::vector<int> occ; for (auto& kv : per_simd) occ.push_back(kv.second);
                std::sort(occ.begin(), occ.end()); std::sort(cpi.begin(), cpi.end()); std::sort(mhz.begin(), mhz.end());
                const double tflops = (double)waves * instr * 64 * vflops[v] / (ms * 1e-3) / 1e12;
                const double span_us = (double)(t_hi - t_lo) / 100.0;
                // SIMD cycles per wave-instruction from the chip-wide span: span x clock / (instructions per wave x waves per SIMD)
                const double cyc_simd = span_us * mhz[waves / 2] / (instr * (double)waves / per_simd.size());
                printf("%-24s | %4d | %d | %8.3f | %6.2f | %5.0f | %4zu | %d / %d | conc %d / %d | span %8.1f us | %5.2f | %6.1f\n", vname[v], wg, wps, ms, cpi[waves / 2], mhz[waves / 2],
                       per_simd.size(), occ.back(), occ[occ.size() / 2], conc.back(), conc[conc.size() / 2], span_us, cyc_simd, tflops);
                fflush(stdout);
            }
    return 0;
}
