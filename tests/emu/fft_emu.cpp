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

// The fft_forward_ws / fft_inverse_ws form of an exchange: stage<..., LLW> stores every output at its exchange position from
// inside the stage (all threads), then exchange_read loads the stage-SB slots (all threads).  Whole-complex layouts only.
static bool g_ws = false;            // third sweep: exchanges issued from inside the stages
template <class C, class LL, int S, int SB, int DIR, int PRUNE>
static void emu_stage_ws(std::vector<std::vector<cplx>>& regs, std::vector<char>& smem, int nlines, const cplx* tw) {
    std::vector<cplx> full((size_t)C::TWLEN + 1);
    twfull_table<C>(full.data(), tw);
    for (size_t i = 0; i + sizeof(double) <= smem.size(); i += sizeof(double)) { const double nan = NAN; memcpy(&smem[i], &nan, sizeof nan); }
    auto offxi = [&](int line, int& off, int& xi) { if (LL::LSTR == 1) { off = line * LL::line_elems(C::L); xi = 0; } else { off = 0; xi = line; } };
    for (int line = 0; line < nlines; ++line) for (int t = 0; t < C::T; ++t) {
        int off, xi; offxi(line, off, xi);
        cplx (&v)[C::E] = *reinterpret_cast<cplx(*)[C::E]>(regs[line * C::T + t].data());
        if (g_twfull) stage<C, S, DIR, PRUNE, true, false, LL>(v, t, full.data(), smem.data(), off, xi);
        else stage<C, S, DIR, PRUNE, false, false, LL>(v, t, tw, smem.data(), off, xi);
    }
    for (int line = 0; line < nlines; ++line) for (int t = 0; t < C::T; ++t) {
        int off, xi; offxi(line, off, xi);
        cplx (&v)[C::E] = *reinterpret_cast<cplx(*)[C::E]>(regs[line * C::T + t].data());
        lds_read<C, SB, LL, 2>(v, t, smem.data(), off, xi);
    }
}

// Emulated wavefront for the exchange through the lanes (fft_core.hpp: xlane_transpose8).  The three gfx950 primitives are
// modelled from their ISA definitions on arrays of 64 lanes:
//   v_permlane32_swap a, b : lanes 32..63 of a <-> lanes 0..31 of b
//   v_permlane16_swap a, b : odd rows (16 lanes) of a <-> even rows of b
//   v_mov_b32_dpp dst, src row_ror:8 bank_mask:M : lanes of the enabled banks (4 lanes each, per row of 16) take src of lane ^ 8
//   ... row_shr:4 (lane i takes lane i - 4) / row_shl:4 (lane i takes lane i + 4) under bank masks 0xa / 0x5
// and xlane_step32's use of them is checked against its specification (lanes with the bit clear keep a and take the
// partner's a into b; lanes with the bit set keep b and take the partner's b into a).
typedef unsigned long long u64;
struct Wave { double re[64], im[64]; };
static void model_step(int bit, double (&a)[64], double (&b)[64]) {
    double na[64], nb[64];
    for (int l = 0; l < 64; ++l) { na[l] = a[l]; nb[l] = b[l]; }
    if (bit == 5) { for (int l = 0; l < 32; ++l) { na[l + 32] = b[l]; nb[l] = a[l + 32]; } }
    else if (bit == 4) { for (int l = 0; l < 64; ++l) if ((l >> 4) & 1) { na[l] = b[l - 16]; nb[l - 16] = a[l]; } }
    else if (bit == 3) {
        for (int l = 0; l < 64; ++l) { const int bank = (l & 15) >> 2;            // bank_mask 0xc -> banks 2, 3 ; 0x3 -> banks 0, 1
            if (bank >= 2) na[l] = b[(l & ~15) | ((l + 8) & 15)];                 // a = update_dpp(a, b, row_ror:8, bank 0xc)
            else nb[l] = a[(l & ~15) | ((l + 8) & 15)]; }                         // b = update_dpp(b, a, row_ror:8, bank 0x3)
    } else {
        for (int l = 0; l < 64; ++l) { const int bank = (l & 15) >> 2;            // bank_mask 0xa -> banks 1, 3 ; 0x5 -> banks 0, 2
            if (bank & 1) na[l] = b[l - 4];                                       // row_shr:4: lane i takes lane i - 4
            else nb[l] = a[l + 4]; }                                              // row_shl:4: lane i takes lane i + 4
    }
    for (int l = 0; l < 64; ++l) { a[l] = na[l]; b[l] = nb[l]; }
}
static int check_model_steps() {
    int bad = 0;
    for (int bit = 2; bit <= 5; ++bit) {
        double a[64], b[64];
        for (int l = 0; l < 64; ++l) { a[l] = 100 + l; b[l] = 200 + l; }
        model_step(bit, a, b);
        for (int l = 0; l < 64; ++l) {
            const int p = l ^ (1 << bit);
            const double wa = ((l >> bit) & 1) ? 200 + p : 100 + l, wb = ((l >> bit) & 1) ? 200 + l : 100 + p;
            if (a[l] != wa || b[l] != wb) ++bad;
        }
    }
    return bad;
}
// the exchange S -> S + 1 (and back) of two stages of equal radix, run wavefront by wavefront with the schedule the device uses
// (xlane_transpose_with); lane = line + LSTR * t, lane bits from xlane_lowbit
template <class C, int S, int LSTR>
static void emu_xlane(std::vector<std::vector<cplx>>& regs, int nlines) {
    static_assert(64 % LSTR == 0, "interleaved lines");
    constexpr int LB = xlane_lowbit<C, S, LSTR>(), TPW = (64 / LSTR < C::T) ? 64 / LSTR : C::T;   // threads t of a line per wavefront
    constexpr int R = C::template R<S>(), NB = C::E / R, NK = xlane_log2(R);
    static_assert(LB >= 2 && LB + NK <= 6, "lane bits 2..5");
    for (int t0 = 0; t0 < C::T; t0 += TPW) {
        // gather the wavefront: lane l holds thread t0 + l / LSTR of line l % LSTR (lanes beyond the line's threads: scratch)
        std::vector<std::vector<cplx>> scratch(64, std::vector<cplx>(C::E));
        std::vector<cplx*> lane(64);
        for (int l = 0; l < 64; ++l) lane[l] = (l / LSTR < TPW) ? regs[(l % LSTR) * C::T + t0 + l / LSTR].data() : scratch[l].data();
        for (int u = 0; u < NB; ++u) for (int k = 0; k < NK; ++k) for (int q = 0; q < R; ++q) if (!((q >> k) & 1)) {
            const int ea = u + NB * q, eb = u + NB * (q | (1 << k));
            double ar[64], ai[64], br[64], bi[64];
            for (int l = 0; l < 64; ++l) { ar[l] = lane[l][ea].x; ai[l] = lane[l][ea].y; br[l] = lane[l][eb].x; bi[l] = lane[l][eb].y; }
            model_step(LB + k, ar, br); model_step(LB + k, ai, bi);
            for (int l = 0; l < 64; ++l) { lane[l][ea] = make_double2(ar[l], ai[l]); lane[l][eb] = make_double2(br[l], bi[l]); }
        }
    }
    (void)nlines;
}
// ... and the device's own schedule function must visit exactly those (k, slot pair)s in that order
template <class C, int S> static int check_xlane_schedule() {
    constexpr int R = C::template R<S>(), NB = C::E / R, NK = xlane_log2(R);
    std::vector<int> want, got;
    for (int u = 0; u < NB; ++u) for (int k = 0; k < NK; ++k) for (int q = 0; q < R; ++q) if (!((q >> k) & 1)) { want.push_back(k); want.push_back(u + NB * q); want.push_back(u + NB * (q | (1 << k))); }
    cplx v[C::E];
    xlane_transpose_with<C, S>(v, [&](int k, cplx& a, cplx& b) { got.push_back(k); got.push_back((int)(&a - v)); got.push_back((int)(&b - v)); });
    return want == got ? 0 : 1;
}
// the tail of the forward transform from stage S on / of the inverse from stage S down, as fft_core.hpp's fft_*_ws_from
template <class C, class LL, int S> static void emu_forward_from(std::vector<std::vector<cplx>>& regs, std::vector<char>& smem, int nlines, const cplx* tw, bool xl) {
    if constexpr (S == C::NS - 1) emu_stage<C, S, +1, 0>(regs, nlines, tw);
    else {
        bool lanes = false;
        if constexpr (xlane_stage_ok<C, S, LL>()) { if (xl) { lanes = true; emu_stage<C, S, +1, 0>(regs, nlines, tw); emu_xlane<C, S, LL::LSTR>(regs, nlines); } }
        if (!lanes) emu_stage_ws<C, LL, S, S + 1, +1, 0>(regs, smem, nlines, tw);
        emu_forward_from<C, LL, S + 1>(regs, smem, nlines, tw, xl);
    }
}
template <class C, class LL, int S> static void emu_inverse_from(std::vector<std::vector<cplx>>& regs, std::vector<char>& smem, int nlines, const cplx* tw, bool xl) {
    bool lanes = false;
    if constexpr (xlane_stage_ok<C, S - 1, LL>()) { if (xl) { lanes = true; emu_stage<C, S, -1, 0>(regs, nlines, tw); emu_xlane<C, S - 1, LL::LSTR>(regs, nlines); } }
    if (!lanes) emu_stage_ws<C, LL, S, S - 1, -1, 0>(regs, smem, nlines, tw);
    if constexpr (S > 1) emu_inverse_from<C, LL, S - 1>(regs, smem, nlines, tw, xl);
}
static bool g_xlane = false;         // fourth sweep: the radix-8 <-> radix-8 exchange through the emulated lanes where the device can

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
    constexpr bool WSOK = !LL::SPLIT;                                // stage-issued stores: whole-complex layouts
    constexpr bool XLOK = xlane_ok<C, LL>();                         // some exchange can run through the lanes
    const bool ws = g_ws && WSOK, xl = g_xlane && XLOK && nlines == LL::LSTR;
    if (ws) {
        if constexpr (WSOK) {
            if (prune) emu_stage_ws<C, LL, 0, 1, +1, 1>(regs, smem, nlines, tw.data()); else emu_stage_ws<C, LL, 0, 1, +1, 0>(regs, smem, nlines, tw.data());
            emu_forward_from<C, LL, 1>(regs, smem, nlines, tw.data(), xl);
        }
    } else {
    if (prune) emu_stage<C, 0, +1, 1>(regs, nlines, tw.data()); else emu_stage<C, 0, +1, 0>(regs, nlines, tw.data());
    emu_exchange<C, LL, 0, 1>(regs, smem, nlines, 0);
    emu_stage<C, 1, +1, 0>(regs, nlines, tw.data());
    if constexpr (C::NS >= 3) { emu_exchange<C, LL, 1, 2>(regs, smem, nlines, 0); emu_stage<C, 2, +1, 0>(regs, nlines, tw.data()); }
    if constexpr (C::NS >= 4) { emu_exchange<C, LL, 2, 3>(regs, smem, nlines, 0); emu_stage<C, 3, +1, 0>(regs, nlines, tw.data()); }
    }
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
    if (ws) {
        if constexpr (WSOK) {
            emu_inverse_from<C, LL, C::NS - 1>(regs, smem, nlines, tw.data(), xl);
        }
    } else {
    if constexpr (C::NS >= 4) { emu_stage<C, 3, -1, 0>(regs, nlines, tw.data()); emu_exchange<C, LL, 3, 2>(regs, smem, nlines, 0); }
    if constexpr (C::NS >= 3) { emu_stage<C, 2, -1, 0>(regs, nlines, tw.data()); emu_exchange<C, LL, 2, 1>(regs, smem, nlines, 0); }
    emu_stage<C, 1, -1, 0>(regs, nlines, tw.data());
    emu_exchange<C, LL, 1, 0>(regs, smem, nlines, 0);
    }
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
    // the forms of the persistent fused pass: half tiles in the unpadded XOR-swizzled buffer (LdsLayout<4, -1, false>), exchange
    // stores issued from inside the stages + exchange_read, and the radix-8 <-> radix-8 exchange through the lanes
#define RUNZ(CFG) do { \
    worst = fmax(worst, run_cfg<CFG, LdsLayout<4, -1, false>>(#CFG " half-tile swizzled", true)); \
    worst = fmax(worst, run_cfg<CFG, LdsLayout<8, 3, false>>(#CFG " whole tile", true)); \
    worst = fmax(worst, run_cfg<CFG, LdsLayout<4, 3, false>>(#CFG " half tile, padded (one-tile kernels)", true)); \
    worst = fmax(worst, run_cfg<CFG, LdsLayout<8, -1, false>>(#CFG " whole tile swizzled", false)); } while (0)
    for (int pass = 0; pass < 3; ++pass) {
        g_twfull = pass >= 1; g_ws = pass >= 1; g_xlane = pass == 2;
        printf("---- fused-pass forms: %s\n", pass == 0 ? "exchange()" : pass == 1 ? "stores from inside the stages, full table" : "+ lane exchanges between stages of equal radix");
        RUNZ(Cfg512); RUNZ(Cfg1024); RUNZ(Cfg1536); RUNZ(Cfg1280); RUNZ(Cfg2048); RUNZ(Cfg768); RUNZ(Cfg640);
        if (pass == 2) { RUNZ(Cfg128); RUNZ(Cfg192); RUNZ(Cfg320); RUNZ(Cfg384); RUNZ(Cfg1024S); }
    }
    g_twfull = g_ws = g_xlane = false;
    static_assert(xlane_ok<Cfg1024, LdsLayout<8, 3, false>>() && xlane_ok<Cfg1536, LdsLayout<4, -1, false>>() && xlane_ok<Cfg512, LdsLayout<8, 3, false>>(), "lane exchange available");
    static_assert(!xlane_ok<Cfg2048, LdsLayout<8, 3, false>>() && !xlane_ok<Cfg96, LdsLayout<8, 3, false>>() && !xlane_ok<Cfg1024, LdsLayout<8, 3, true>>(), "lane exchange not available");
    static_assert(xlane_stage_ok<Cfg1280, 1, LdsLayout<4, -1, false>>() && xlane_stage_ok<Cfg1280, 2, LdsLayout<4, -1, false>>() && !xlane_stage_ok<Cfg1280, 1, LdsLayout<8, 3, false>>() && xlane_stage_ok<Cfg1280, 2, LdsLayout<8, 3, false>>(), "20.4.4.4: both late exchanges in 4-line workgroups, the last one in 8-line workgroups");
    static_assert(!xlane_stage_ok<Cfg640, 1, LdsLayout<8, 3, false>>() && xlane_stage_ok<Cfg640, 2, LdsLayout<8, 3, false>>() && xlane_stage_ok<Cfg320, 1, LdsLayout<8, 3, false>>() && xlane_stage_ok<Cfg384, 1, LdsLayout<8, 3, false>>(), "radix-4 pairs");
    int bad = check_model_steps() + check_xlane_schedule<Cfg1024, 1>() + check_xlane_schedule<Cfg1536, 1>() + check_xlane_schedule<Cfg512, 1>() + check_xlane_schedule<Cfg1280, 1>() + check_xlane_schedule<Cfg1280, 2>();
    printf("lane-exchange primitive model / schedule mismatches: %d\n", bad);
    // ticket -> (tile, half) of the ticketed fused pass: every (tile, half) exactly once over the eight queues, pairs adjacent
    for (int half = 0; half < 2; ++half) for (unsigned ntiles = 16; ntiles <= 16 * 40; ntiles += 16) {
        const unsigned nwork = half ? ntiles / 4 : ntiles / 8;
        std::vector<int> seen((size_t)ntiles * 2, 0);
        for (unsigned q = 0; q < 8; ++q) for (unsigned c = 0; c < nwork; ++c) {
            unsigned tile, hf;
            if (half) ticket_decode<true>((c << 3) | q, tile, hf); else ticket_decode<false>((c << 3) | q, tile, hf);
            if (tile >= ntiles || hf > (unsigned)half) { ++bad; continue; }
            ++seen[(size_t)tile * 2 + hf];
            // consecutive tickets of a queue: the partner tile of a (row, mirror row) pair is tile ^ 1
            unsigned t2, h2; const unsigned cn = c ^ (half ? 2u : 1u);
            if (half) ticket_decode<true>((cn << 3) | q, t2, h2); else ticket_decode<false>((cn << 3) | q, t2, h2);
            if (t2 != (tile ^ 1u) || h2 != hf) ++bad;
        }
        for (unsigned t = 0; t < ntiles; ++t) for (int hf = 0; hf <= half; ++hf) if (seen[(size_t)t * 2 + hf] != 1) ++bad;
        if (!half) for (unsigned t = 0; t < ntiles; ++t) if (seen[(size_t)t * 2 + 1] != 0) ++bad;
    }
    // row pairs as work items (ticket_decode_pair): tiles 2 p, 2 p + 1 of pair p = q + 8 c; every tile exactly once
    for (unsigned ntiles = 16; ntiles <= 16 * 40; ntiles += 16) {
        std::vector<int> seen(ntiles, 0);
        for (unsigned q = 0; q < 8; ++q) for (unsigned c = 0; c < ntiles / 16; ++c) for (unsigned sub = 0; sub < 2; ++sub) {
            const unsigned tile = ticket_decode_pair((c << 3) | q, sub);
            if (tile >= ntiles || (tile & 1u) != sub) { ++bad; continue; }
            ++seen[tile];
        }
        for (unsigned t = 0; t < ntiles; ++t) if (seen[t] != 1) ++bad;
    }
    printf("ticket decode / lane exchange checks failed: %d\n", bad);
    if (bad) worst = 1.0;
    // the 1024-point line in 8 interleaved lines (the z pass at 512^3) must have its radix-8 <-> radix-8 exchange local
    static_assert(exchange_wave_local<Cfg1024, 1, 2>(8) && exchange_wave_local<Cfg1024, 2, 1>(8) && !exchange_wave_local<Cfg1024, 0, 1>(8), "Cfg1024 locality");
    static_assert(exchange_wave_local<Cfg1024, 0, 1>(64) && exchange_wave_local<Cfg512, 0, 1>(64), "one wave per contiguous line");
    printf("wave-local exchanges emulated group by group: %ld\n", g_local_exchanges);
    printf("worst=%.3e\n", worst);
    return worst < 1e-13 ? 0 : 1;
}
