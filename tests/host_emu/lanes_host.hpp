// lanes_host.hpp -- TEST INFRASTRUCTURE: CPU emulation of one env lane group (16 lanes: Panda row, 64 lanes: iCub
// wave), so the lane-generic step in csrc/pbre_core.hpp can be executed (slowly) and checked against the oracle in the
// GPU-less dev container.  Never loaded by the product.
#pragma once
#include <cmath>
#include <cstring>
#include "../../pybullet-robot-envs_amd/csrc/pbre_math.hpp"

namespace pbre_emu {

template <int W> struct VF { float v[W]; VF() {} VF(float s) { for (int i = 0; i < W; i++) v[i] = s; } };
template <int W> struct VI { int v[W];   VI() {} VI(int s)   { for (int i = 0; i < W; i++) v[i] = s; } };
template <int W> struct VB { bool v[W];  VB() {} VB(bool s)  { for (int i = 0; i < W; i++) v[i] = s; } };

#define PBRE_EMU_BIN(op) \
    template <int W> inline VF<W> operator op(const VF<W>& a, const VF<W>& b) { VF<W> r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
PBRE_EMU_BIN(+) PBRE_EMU_BIN(-) PBRE_EMU_BIN(*) PBRE_EMU_BIN(/)
#undef PBRE_EMU_BIN

template <int W>
struct HostLanesT {
    using F = VF<W>; using I = VI<W>; using B = VB<W>;
    using Robot = HostLanesT<W>;
    static F lo(const F& x) { return x; }
    static F wide(const F& x) { return x; }
    static F fma_lo(const F& s, const F& m, const F& acc) { return fma(s, m, acc); }
    static F uni(const F& x) { return x; }
    static unsigned long long lanebits(const B& b) { unsigned long long m = 0; for (int i = 0; i < W && i < 64; i++) if (b.v[i]) m |= 1ull << i; return m; }
    static F setlane(const F& x, int j, const F& src) { F r = x; r.v[j] = src.v[j]; return r; }
    // (device: one asm block -- compare, select, v_fmac with a DPP row broadcast; the same operations)
    template <int J>
    static F row_tail(F& rec, const F& recval, const F& bval, const F& m, const F& acc) { rec = setlane(rec, J, recval); return fma(bcast(bval, J), m, acc); }
    template <int N> struct RowStore {
        F v[N];
        void init() {}
        const F& get(int i) const { return v[i]; }
        void put(int i, const F& x) { v[i] = x; }
    };
    static F c(float x) { return F(x); }
    static I ci(int x) { return I(x); }
    static I lane() { I r; for (int i = 0; i < W; i++) r.v[i] = i; return r; }
    static F load(const float* p) { F r; for (int i = 0; i < W; i++) r.v[i] = p[i]; return r; }
    static I loadI(const int* p) { I r; for (int i = 0; i < W; i++) r.v[i] = p[i]; return r; }
    static F loadm(const float* p, const B& m) { F r; for (int i = 0; i < W; i++) r.v[i] = m.v[i] ? p[i] : 0.f; return r; }
    static F loadu(const float* p) { return F(*p); }
    static float first(const F& x) { return x.v[0]; }
    static bool lane0() { return true; }
    static void fence() {}
    static F loadx(const float* p, const I& idx, const B& m) { F r; for (int i = 0; i < W; i++) r.v[i] = m.v[i] ? p[idx.v[i]] : 0.f; return r; }
    static void storex(float* p, const I& idx, const F& x, const B& m) { for (int i = 0; i < W; i++) if (m.v[i]) p[idx.v[i]] = x.v[i]; }
    static void store(float* p, const F& x) { for (int i = 0; i < W; i++) p[i] = x.v[i]; }
    static void storem(float* p, const F& x, const B& m) { for (int i = 0; i < W; i++) if (m.v[i]) p[i] = x.v[i]; }
#define U1(name, expr) static F name(const F& a) { F r; for (int i = 0; i < W; i++) { float x = a.v[i]; r.v[i] = (expr); } return r; }
    U1(abs, std::fabs(x)) U1(sqrt, std::sqrt(x)) U1(sin, std::sin(x)) U1(cos, std::cos(x)) U1(asin, std::asin(x))
#undef U1
    static void sincos(const F& a, F& s, F& c) { for (int i = 0; i < W; i++) pbre::sincos_f(a.v[i], s.v[i], c.v[i]); }      // same arithmetic as the device
    static F fma(const F& a, const F& b, const F& c_) { F r; for (int i = 0; i < W; i++) r.v[i] = std::fma(a.v[i], b.v[i], c_.v[i]); return r; }
    static F min(const F& a, const F& b) { F r; for (int i = 0; i < W; i++) r.v[i] = std::fmin(a.v[i], b.v[i]); return r; }
    static F max(const F& a, const F& b) { F r; for (int i = 0; i < W; i++) r.v[i] = std::fmax(a.v[i], b.v[i]); return r; }
    static F atan2(const F& a, const F& b) { F r; for (int i = 0; i < W; i++) r.v[i] = std::atan2(a.v[i], b.v[i]); return r; }
    static F med3(const F& x, const F& lo, const F& hi) { return min(max(x, lo), hi); }
#define CMP(name, op) static B name(const F& a, const F& b) { B r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
    CMP(lt, <) CMP(le, <=) CMP(gt, >) CMP(ge, >=) CMP(eq, ==) CMP(ne, !=)
#undef CMP
#define CMPI(name, op) static B name(const I& a, const I& b) { B r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
    CMPI(eqi, ==) CMPI(nei, !=) CMPI(lti, <) CMPI(gei, >=)
#undef CMPI
    static B band(const B& a, const B& b) { B r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] && b.v[i]; return r; }
    static B bor(const B& a, const B& b) { B r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] || b.v[i]; return r; }
    static B bnot(const B& a) { B r; for (int i = 0; i < W; i++) r.v[i] = !a.v[i]; return r; }
    static B bfalse() { return B(false); }
    static bool any(const B& a) { for (int i = 0; i < W; i++) if (a.v[i]) return true; return false; }
    static F sel(const B& m, const F& a, const F& b) { F r; for (int i = 0; i < W; i++) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
    static I seli(const B& m, const I& a, const I& b) { I r; for (int i = 0; i < W; i++) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
    static B bit(const I& m, int k) { B r; for (int i = 0; i < W; i++) r.v[i] = ((unsigned)m.v[i] >> k) & 1u; return r; }
    static B biti(const I& m, const I& k) { B r; for (int i = 0; i < W; i++) r.v[i] = ((unsigned)m.v[i] >> (k.v[i] & 31)) & 1u; return r; }
    static I maxi(const I& a, int b) { I r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] > b ? a.v[i] : b; return r; }
    static F itof(const I& a) { F r; for (int i = 0; i < W; i++) r.v[i] = (float)a.v[i]; return r; }
    static I ftoi(const F& a) { I r; for (int i = 0; i < W; i++) r.v[i] = (int)a.v[i]; return r; }
    // cross-lane
    static F bcast(const F& a, int k) { return F(a.v[k]); }
    static F bcast_row(const F& a, int k) { return bcast(a, k); }
    static F gather(const F& a, const I& idx) { F r; for (int i = 0; i < W; i++) r.v[i] = a.v[idx.v[i] & (W - 1)]; return r; }
    static I gatherI(const I& a, const I& idx) { I r; for (int i = 0; i < W; i++) r.v[i] = a.v[idx.v[i] & (W - 1)]; return r; }
    // summation order identical to the device: inside each 16-lane row the DPP butterfly (xor 1, xor 2, half mirror,
    // mirror); a 64-lane group then combines its four row sums as (r3 + r2) + (r1 + r0) (row_bcast15 / row_bcast31)
    static F sum(const F& a) {
        F x = a, y;
        for (int i = 0; i < W; i++) y.v[i] = x.v[i] + x.v[i ^ 1]; x = y;
        for (int i = 0; i < W; i++) y.v[i] = x.v[i] + x.v[i ^ 2]; x = y;
        for (int i = 0; i < W; i++) y.v[i] = x.v[i] + x.v[(i & ~7) | (7 - (i & 7))]; x = y;
        for (int i = 0; i < W; i++) y.v[i] = x.v[i] + x.v[(i & ~15) | (15 - (i & 15))]; x = y;
        if (W == 128) {   // two virtual lanes per physical lane: lane i and lane i + 64 are added first, then the 64-lane order
            F h = a;
            for (int i = 0; i < 64; i++) h.v[i] = a.v[i] + a.v[i + 64];
            x = h;
            for (int i = 0; i < 64; i++) y.v[i] = x.v[i] + x.v[i ^ 1]; x = y;
            for (int i = 0; i < 64; i++) y.v[i] = x.v[i] + x.v[i ^ 2]; x = y;
            for (int i = 0; i < 64; i++) y.v[i] = x.v[i] + x.v[(i & ~7) | (7 - (i & 7))]; x = y;
            for (int i = 0; i < 64; i++) y.v[i] = x.v[i] + x.v[(i & ~15) | (15 - (i & 15))]; x = y;
            return F((x.v[48] + x.v[32]) + (x.v[16] + x.v[0]));
        }
        if (W == 64) return F((x.v[48] + x.v[32]) + (x.v[16] + x.v[0]));
        if (W == 32) return F(x.v[0] + x.v[16]);
        return x;
    }
    static F sum_obj(const F& a) { return sum(a); }
    static F sum_step(const F& x, int k) {      // one stage of the 16-lane butterfly of sum()
        F y;
        for (int i = 0; i < W; i++) {
            const int o = k == 0 ? (i ^ 1) : (k == 1 ? (i ^ 2) : (k == 2 ? ((i & ~7) | (7 - (i & 7))) : ((i & ~15) | (15 - (i & 15)))));
            y.v[i] = x.v[i] + x.v[o];
        }
        return y;
    }
    static F vmin(const F& a) { float m = a.v[0]; for (int i = 1; i < W; i++) m = std::fmin(m, a.v[i]); return F(m); }
};
using HostLanes = HostLanesT<16>;

}  // namespace pbre_emu
