// pbre_step_inst.hip -- one explicit instantiation of the Panda step launcher (pbre_panda.hpp), compiled once per (MODE, RT):
//   hipcc ... -DPBRE_INST_MODE=<0..5> -DPBRE_INST_RT=<false|true> -c -o obj/pbre_step_<m>_<rt>.o pbre_step_inst.hip        (build.sh)
#include "pbre_panda.hpp"
PBRE_STEP_INST(, PBRE_INST_MODE, PBRE_INST_RT)

#ifdef PBRE_WAVE_TRACE
// (tools/wave_trace.py; not part of include/pbre.h) the row waves' records of the launches since the last reset of the counter
extern "C" int pbre_debug_wave_trace(unsigned long long* out, int max_records, int reset) {
    unsigned int n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_wtrace_n), sizeof n) != hipSuccess) return -1;
    if (n > 16384u) n = 16384u;
    if ((int)n > max_records) n = (unsigned)max_records;
    if (out && n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wtrace), (size_t)n * 3 * sizeof(unsigned long long)) != hipSuccess) return -1;
    const unsigned int z = 0;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_wtrace_n), &z, sizeof z) != hipSuccess) return -1;
    return (int)n;
}
extern "C" int pbre_debug_wave_diag(int mode) {      // (see g_wave_diag)
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wave_diag), &mode, sizeof mode) == hipSuccess ? 0 : -1;
}
#endif
