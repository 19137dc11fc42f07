// lanes_host.hpp -- TEST INFRASTRUCTURE: CPU emulation of one 16-lane env group, so the
// lane-generic step in csrc/pbre_core.hpp can be executed (slowly) and checked against the
// oracle in the GPU-less dev container.  Never loaded by the product.
#pragma once
#include <cmath>
#include <cstring>

namespace pbre_emu {

constexpr int W = 16;
struct VF { float v[W]; VF() {} VF(float s) { for (int i = 0; i < W; i++) v[i] = s; } };
struct VI { int v[W];   VI() {} VI(int s)   { for (int i = 0; i < W; i++) v[i] = s; } };
struct VB { bool v[W];  VB() {} VB(bool s)  { for (int i = 0; i < W; i++) v[i] = s; } };

#define PBRE_EMU_BIN(op) \
    inline VF operator op(const VF& a, const VF& b) { VF r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
PBRE_EMU_BIN(+) PBRE_EMU_BIN(-) PBRE_EMU_BIN(*) PBRE_EMU_BIN(/)
#undef PBRE_EMU_BIN

struct HostLanes {
    using F = VF; using I = VI; using B = VB;
    static F c(float x) { return VF(x); }
    static I ci(int x) { return VI(x); }
    static I lane() { VI r; for (int i = 0; i < W; i++) r.v[i] = i; return r; }
    static F load(const float* p) { VF r; for (int i = 0; i < W; i++) r.v[i] = p[i]; return r; }
    static I loadI(const int* p) { VI r; for (int i = 0; i < W; i++) r.v[i] = p[i]; return r; }
    static F loadm(const float* p, const B& m) { VF r; for (int i = 0; i < W; i++) r.v[i] = m.v[i] ? p[i] : 0.f; return r; }
    static void store(float* p, const F& x) { for (int i = 0; i < W; i++) p[i] = x.v[i]; }
    static void storem(float* p, const F& x, const B& m) { for (int i = 0; i < W; i++) if (m.v[i]) p[i] = x.v[i]; }
#define U1(name, expr) static F name(const F& a) { VF r; for (int i = 0; i < W; i++) { float x = a.v[i]; r.v[i] = (expr); } return r; }
    U1(abs, std::fabs(x)) U1(sqrt, std::sqrt(x)) U1(sin, std::sin(x)) U1(cos, std::cos(x)) U1(asin, std::asin(x))
#undef U1
    static F fma(const F& a, const F& b, const F& c_) { VF r; for (int i = 0; i < W; i++) r.v[i] = std::fma(a.v[i], b.v[i], c_.v[i]); return r; }
    static F min(const F& a, const F& b) { VF r; for (int i = 0; i < W; i++) r.v[i] = std::fmin(a.v[i], b.v[i]); return r; }
    static F max(const F& a, const F& b) { VF r; for (int i = 0; i < W; i++) r.v[i] = std::fmax(a.v[i], b.v[i]); return r; }
    static F atan2(const F& a, const F& b) { VF r; for (int i = 0; i < W; i++) r.v[i] = std::atan2(a.v[i], b.v[i]); return r; }
    static F med3(const F& x, const F& lo, const F& hi) { return min(max(x, lo), hi); }
#define CMP(name, op) static B name(const F& a, const F& b) { VB r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
    CMP(lt, <) CMP(le, <=) CMP(gt, >) CMP(ge, >=) CMP(eq, ==) CMP(ne, !=)
#undef CMP
#define CMPI(name, op) static B name(const I& a, const I& b) { VB r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
    CMPI(eqi, ==) CMPI(nei, !=) CMPI(lti, <) CMPI(gei, >=)
#undef CMPI
    static B band(const B& a, const B& b) { VB r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] && b.v[i]; return r; }
    static B bor(const B& a, const B& b) { VB r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] || b.v[i]; return r; }
    static B bnot(const B& a) { VB r; for (int i = 0; i < W; i++) r.v[i] = !a.v[i]; return r; }
    static B bfalse() { return VB(false); }
    static bool any(const B& a) { for (int i = 0; i < W; i++) if (a.v[i]) return true; return false; }
    static F sel(const B& m, const F& a, const F& b) { VF r; for (int i = 0; i < W; i++) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
    static I seli(const B& m, const I& a, const I& b) { VI r; for (int i = 0; i < W; i++) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
    static B bit(const I& m, int k) { VB r; for (int i = 0; i < W; i++) r.v[i] = (m.v[i] >> k) & 1; return r; }
    static B biti(const I& m, const I& k) { VB r; for (int i = 0; i < W; i++) r.v[i] = (m.v[i] >> k.v[i]) & 1; return r; }
    static I maxi(const I& a, int b) { VI r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] > b ? a.v[i] : b; return r; }
    static F itof(const I& a) { VF r; for (int i = 0; i < W; i++) r.v[i] = (float)a.v[i]; return r; }
    static I ftoi(const F& a) { VI r; for (int i = 0; i < W; i++) r.v[i] = (int)a.v[i]; return r; }
    // cross-lane
    static F bcast(const F& a, int k) { return VF(a.v[k]); }
    static F gather(const F& a, const I& idx) { VF r; for (int i = 0; i < W; i++) r.v[i] = a.v[idx.v[i] & (W - 1)]; return r; }
    static I gatherI(const I& a, const I& idx) { VI r; for (int i = 0; i < W; i++) r.v[i] = a.v[idx.v[i] & (W - 1)]; return r; }
    // butterfly order identical to the device DPP all-reduce (xor 1, xor 2, half mirror, mirror)
    static F sum(const F& a) {
        VF x = a, y;
        for (int i = 0; i < W; i++) y.v[i] = x.v[i] + x.v[i ^ 1]; x = y;
        for (int i = 0; i < W; i++) y.v[i] = x.v[i] + x.v[i ^ 2]; x = y;
        for (int i = 0; i < W; i++) y.v[i] = x.v[i] + x.v[(i & 8) | (7 - (i & 7))]; x = y;
        for (int i = 0; i < W; i++) y.v[i] = x.v[i] + x.v[15 - i]; x = y;
        return x;
    }
    static F vmin(const F& a) { float m = a.v[0]; for (int i = 1; i < W; i++) m = std::fmin(m, a.v[i]); return VF(m); }
};

}  // namespace pbre_emu
