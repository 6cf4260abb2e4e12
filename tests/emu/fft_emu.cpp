// Host emulation of the gfx950 FFT line transform (tests only).
// Compiles fast_solver_lippmann_schwinger_amd/csrc/fft_core.hpp with g++ by giving
// the few HIP spellings it uses a host meaning, then executes the stage / LDS
// phases thread by thread and checks them against a naive long-double DFT.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <complex>
#include <cstring>
struct double2 { double x, y; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
#define __device__
#define __forceinline__ inline
#define __restrict__
#define LSFC_FFT_HOST_EMULATION 1
#include "../../fast_solver_lippmann_schwinger_amd/csrc/fft_core.hpp"
#include "../../fast_solver_lippmann_schwinger_amd/csrc/fft_configs.hpp"

using namespace lsfc::fft;
typedef std::complex<long double> lc;
static long g_local_exchanges = 0;      // exchanges emulated wave group by wave group

template <class C, class LL, int SA, int SB>
static void emu_exchange(std::vector<std::vector<cplx>>& regs, std::vector<char>& smem, int nlines, int LSTRmode) {
    // lines either interleaved (LSTR>1: off=0, xi=line) or separate regions (LSTR==1)
    auto offxi = [&](int line, int& off, int& xi) {
        if (LL::LSTR == 1) { off = line * LL::line_elems(C::L); xi = 0; } else { off = 0; xi = line; }
    };
    const int comps = LL::SPLIT ? 2 : 1;
    // An exchange the kernels treat as WAVE-LOCAL (fft_core.hpp: exchange_wave_local) is emulated the way the device
    // runs it: wave group by wave group, each group writing and immediately reading with every other group's data
    // absent (the buffer is poisoned first).  If the locality claim were wrong the reads would pick up poison and the
    // comparison with the naive DFT would fail.  Other exchanges: all writes, (barrier), all reads.
    const bool local = exchange_wave_local<C, SA, SB>(LL::wave_group()) && exchange_wave_local<C, SB, SA>(LL::wave_group());
    const int G = local ? (LL::wave_group() < C::T ? LL::wave_group() : C::T) : C::T;
    g_local_exchanges += local ? 1 : 0;
    for (int c = 0; c < comps; ++c) {
        for (size_t i = 0; i + sizeof(double) <= smem.size(); i += sizeof(double)) { const double nan = NAN; memcpy(&smem[i], &nan, sizeof nan); }
        for (int g0 = 0; g0 < C::T; g0 += G) {
            for (int line = 0; line < nlines; ++line) for (int t = g0; t < g0 + G; ++t) {
                int off, xi; offxi(line, off, xi);
                cplx (&v)[C::E] = *reinterpret_cast<cplx(*)[C::E]>(regs[line * C::T + t].data());
                if (LL::SPLIT) { if (c == 0) lds_write<C, SA, LL, 0>(v, t, smem.data(), off, xi); else lds_write<C, SA, LL, 1>(v, t, smem.data(), off, xi); }
                else lds_write<C, SA, LL, 2>(v, t, smem.data(), off, xi);
            }
            for (int line = 0; line < nlines; ++line) for (int t = g0; t < g0 + G; ++t) {
                int off, xi; offxi(line, off, xi);
                cplx (&v)[C::E] = *reinterpret_cast<cplx(*)[C::E]>(regs[line * C::T + t].data());
                if (LL::SPLIT) { if (c == 0) lds_read<C, SB, LL, 0>(v, t, smem.data(), off, xi); else lds_read<C, SB, LL, 1>(v, t, smem.data(), off, xi); }
                else lds_read<C, SB, LL, 2>(v, t, smem.data(), off, xi);
            }
        }
    }
}

static bool g_twfull = false;        // second sweep: stage twiddles read from the full table instead of computed
template <class C, int S, int DIR, int PRUNE>
static void emu_stage(std::vector<std::vector<cplx>>& regs, int nlines, const cplx* tw) {
    std::vector<cplx> full((size_t)C::TWLEN + 1);
    twfull_table<C>(full.data(), tw);
    for (int line = 0; line < nlines; ++line) for (int t = 0; t < C::T; ++t) {
        cplx (&v)[C::E] = *reinterpret_cast<cplx(*)[C::E]>(regs[line * C::T + t].data());
        if (g_twfull) stage<C, S, DIR, PRUNE, true>(v, t, full.data());
        else stage<C, S, DIR, PRUNE>(v, t, tw);
    }
}

template <class C, class LL> static double run_cfg(const char* name, bool prune) {
    const int L = C::L, n = L / 2, nlines = (LL::LSTR == 1) ? 3 : LL::LSTR;
    std::vector<cplx> tw(L);
    for (int j = 0; j < L; ++j) { long double a = -2.0L * M_PIl * j / L; tw[j] = make_double2((double)cosl(a), (double)sinl(a)); }
    std::vector<std::vector<lc>> xin(nlines, std::vector<lc>(L, lc(0, 0)));
    srand(1234 + L);
    for (int l = 0; l < nlines; ++l) for (int j = 0; j < (prune ? n : L); ++j) xin[l][j] = lc(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
    std::vector<std::vector<cplx>> regs(nlines * C::T, std::vector<cplx>(C::E));
    for (int l = 0; l < nlines; ++l) for (int t = 0; t < C::T; ++t) for (int e = 0; e < C::E; ++e) {
        lc v = xin[l][t + C::T * e]; regs[l * C::T + t][e] = make_double2((double)v.real(), (double)v.imag());
    }
    size_t lds_elems = (size_t)LL::line_elems(L) * nlines + 64;
    std::vector<char> smem(lds_elems * 16, 0);
    // forward
    if (prune) emu_stage<C, 0, +1, 1>(regs, nlines, tw.data()); else emu_stage<C, 0, +1, 0>(regs, nlines, tw.data());
    emu_exchange<C, LL, 0, 1>(regs, smem, nlines, 0);
    emu_stage<C, 1, +1, 0>(regs, nlines, tw.data());
    if constexpr (C::NS >= 3) { emu_exchange<C, LL, 1, 2>(regs, smem, nlines, 0); emu_stage<C, 2, +1, 0>(regs, nlines, tw.data()); }
    if constexpr (C::NS >= 4) { emu_exchange<C, LL, 2, 3>(regs, smem, nlines, 0); emu_stage<C, 3, +1, 0>(regs, nlines, tw.data()); }
    // compare with naive DFT through perm_table
    std::vector<int> perm(L); perm_table<C>(perm.data());
    std::vector<char> seen(L, 0); for (int s = 0; s < L; ++s) { if (perm[s] < 0 || perm[s] >= L || seen[perm[s]]) { printf("%s: perm not a bijection\n", name); return 1; } seen[perm[s]] = 1; }
    long double err = 0, nrm = 0;
    std::vector<lc> wl(L);                            // exp(-2 pi i j / L) in long double, for the naive reference DFT
    for (int j = 0; j < L; ++j) { long double a = -2.0L * M_PIl * j / L; wl[j] = lc(cosl(a), sinl(a)); }
    for (int l = 0; l < nlines; ++l) {
        std::vector<lc> X(L);
        for (int k = 0; k < L; ++k) { lc acc(0, 0); for (int j = 0; j < L; ++j) acc += xin[l][j] * wl[(long long)j * k % L]; X[k] = acc; }
        for (int t = 0; t < C::T; ++t) for (int e = 0; e < C::E; ++e) {
            cplx g = regs[l * C::T + t][e]; lc ref = X[perm[t + C::T * e]];
            err += std::norm(lc(g.x, g.y) - ref); nrm += std::norm(ref);
        }
    }
    double fwd_err = (double)sqrtl(err / nrm);
    // inverse
    if constexpr (C::NS >= 4) { emu_stage<C, 3, -1, 0>(regs, nlines, tw.data()); emu_exchange<C, LL, 3, 2>(regs, smem, nlines, 0); }
    if constexpr (C::NS >= 3) { emu_stage<C, 2, -1, 0>(regs, nlines, tw.data()); emu_exchange<C, LL, 2, 1>(regs, smem, nlines, 0); }
    emu_stage<C, 1, -1, 0>(regs, nlines, tw.data());
    emu_exchange<C, LL, 1, 0>(regs, smem, nlines, 0);
    if (prune) emu_stage<C, 0, -1, 2>(regs, nlines, tw.data()); else emu_stage<C, 0, -1, 0>(regs, nlines, tw.data());
    err = 0; nrm = 0;
    for (int l = 0; l < nlines; ++l) for (int t = 0; t < C::T; ++t) for (int e = 0; e < (prune ? C::E / 2 : C::E); ++e) {
        cplx g = regs[l * C::T + t][e]; lc ref = xin[l][t + C::T * e] * (long double)L;
        err += std::norm(lc(g.x, g.y) - ref); nrm += std::norm(ref);
    }
    double inv_err = (double)sqrtl(err / nrm);
    printf("%-34s prune=%d fwd_err=%.2e roundtrip_err=%.2e\n", name, (int)prune, fwd_err, inv_err);
    return fwd_err > inv_err ? fwd_err : inv_err;
}

int main() {
    double worst = 0;
#define RUN(CFG) do { \
    worst = fmax(worst, run_cfg<CFG, LdsLayout<1, 3, true>>(#CFG " contig split", true)); \
    worst = fmax(worst, run_cfg<CFG, LdsLayout<1, 3, false>>(#CFG " contig full", false)); \
    worst = fmax(worst, run_cfg<CFG, LdsLayout<8, 3, true>>(#CFG " strided split", true)); \
    worst = fmax(worst, run_cfg<CFG, LdsLayout<8, 3, false>>(#CFG " strided full", true)); } while (0)
    for (int pass = 0; pass < 2; ++pass) {
    g_twfull = pass == 1;
    printf("---- stage twiddles: %s\n", g_twfull ? "full table" : "product tree");
    RUN(Cfg32); RUN(Cfg64); RUN(Cfg128); RUN(Cfg256); RUN(Cfg512); RUN(Cfg1024); RUN(Cfg2048); RUN(Cfg1024S);
    RUN(Cfg48); RUN(Cfg96); RUN(Cfg192); RUN(Cfg384); RUN(Cfg768); RUN(Cfg1536);
    RUN(Cfg80); RUN(Cfg160); RUN(Cfg320); RUN(Cfg640); RUN(Cfg1280);
    }
    // the 1024-point line in 8 interleaved lines (the z pass at 512^3) must have its radix-8 <-> radix-8 exchange local
    static_assert(exchange_wave_local<Cfg1024, 1, 2>(8) && exchange_wave_local<Cfg1024, 2, 1>(8) && !exchange_wave_local<Cfg1024, 0, 1>(8), "Cfg1024 locality");
    static_assert(exchange_wave_local<Cfg1024, 0, 1>(64) && exchange_wave_local<Cfg512, 0, 1>(64), "one wave per contiguous line");
    printf("wave-local exchanges emulated group by group: %ld\n", g_local_exchanges);
    printf("worst=%.3e\n", worst);
    return worst < 1e-13 ? 0 : 1;
}
