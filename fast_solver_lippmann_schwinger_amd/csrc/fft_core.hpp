// Hand-written fp64 complex FFT building blocks for gfx950 (wave64, LDS exchange).
//
// One transform line of length L is spread over T threads, E = L/T elements per
// thread held in registers.  The forward transform is a decimation-in-frequency
// Cooley-Tukey factorisation L = R0*R1(*R2) computed IN PLACE: stage s works on
// sub-blocks of length LS_s with radix R_s; between stages the threads swap
// elements through LDS.  Outputs are left in digit-reversed ("storage") order;
// the inverse transform runs the same stages backwards (conjugate twiddle, then
// conjugate butterfly) and therefore consumes exactly that order.  Frequency
// indices are only labels for the convolution: the Green's symbol is permuted
// once at plan creation (see perm_table), so no reordering pass ever runs.
//
// Register slot e of thread t corresponds to
//   time side      : position  t + T*e                      (natural order)
//   frequency side : storage index s = t + T*e, frequency = perm_table[s]
// so zero padding (positions >= L/2 are zero) and cropping (only positions
// < L/2 are wanted) both mean "slots e >= E/2", which lets the first forward /
// last inverse butterfly drop one radix-2 level (PRUNE).
#pragma once
#include <type_traits>
#ifndef LSFC_FFT_HOST_EMULATION      // tests/emu/ compiles this header with g++ to check the index algebra
#include <hip/hip_runtime.h>
#define LSFC_BARRIER() __syncthreads()
#else
#define LSFC_BARRIER() ((void)0)     // (the emulator runs one thread at a time and synchronises in its own driver loop)
#endif
#include <utility>

namespace lsfc { namespace fft {

using cplx = double2;

__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return make_double2(fma(-a.y, b.y, a.x * b.x), fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ cplx cconj(cplx a) { return make_double2(a.x, -a.y); }

// exp(-DIR * 2*pi*i * K / R) * a for compile-time K and R: any divisor of 32 (power-of-two lines), of 24 (lines with
// one factor 3) or of 40 (one factor 5).  Constants are the correctly rounded cosines / sines of the 32-, 24- and
// 40-gon.
template <int B> struct NGon;
template <> struct NGon<24> {
    static constexpr double C[24] = { 1.0, 0.9659258262890683, 0.8660254037844386, 0.7071067811865476, 0.5, 0.25881904510252074, 0.0, -0.25881904510252074, -0.5, -0.7071067811865476, -0.8660254037844386, -0.9659258262890683, -1.0, -0.9659258262890683, -0.8660254037844386, -0.7071067811865476, -0.5, -0.25881904510252074, 0.0, 0.25881904510252074, 0.5, 0.7071067811865476, 0.8660254037844386, 0.9659258262890683 };
    static constexpr double S[24] = { 0.0, 0.25881904510252074, 0.5, 0.7071067811865476, 0.8660254037844386, 0.9659258262890683, 1.0, 0.9659258262890683, 0.8660254037844386, 0.7071067811865476, 0.5, 0.25881904510252074, 0.0, -0.25881904510252074, -0.5, -0.7071067811865476, -0.8660254037844386, -0.9659258262890683, -1.0, -0.9659258262890683, -0.8660254037844386, -0.7071067811865476, -0.5, -0.25881904510252074 };
};
template <> struct NGon<40> {
    static constexpr double C[40] = { 1.0, 0.9876883405951378, 0.9510565162951535, 0.8910065241883679, 0.8090169943749475, 0.7071067811865476, 0.5877852522924731, 0.4539904997395468, 0.30901699437494745, 0.15643446504023087, 0.0, -0.15643446504023087, -0.30901699437494745, -0.4539904997395468, -0.5877852522924731, -0.7071067811865476, -0.8090169943749475, -0.8910065241883679, -0.9510565162951535, -0.9876883405951378, -1.0, -0.9876883405951378, -0.9510565162951535, -0.8910065241883679, -0.8090169943749475, -0.7071067811865476, -0.5877852522924731, -0.4539904997395468, -0.30901699437494745, -0.15643446504023087, 0.0, 0.15643446504023087, 0.30901699437494745, 0.4539904997395468, 0.5877852522924731, 0.7071067811865476, 0.8090169943749475, 0.8910065241883679, 0.9510565162951535, 0.9876883405951378 };
    static constexpr double S[40] = { 0.0, 0.15643446504023087, 0.30901699437494745, 0.4539904997395468, 0.5877852522924731, 0.7071067811865476, 0.8090169943749475, 0.8910065241883679, 0.9510565162951535, 0.9876883405951378, 1.0, 0.9876883405951378, 0.9510565162951535, 0.8910065241883679, 0.8090169943749475, 0.7071067811865476, 0.5877852522924731, 0.4539904997395468, 0.30901699437494745, 0.15643446504023087, 0.0, -0.15643446504023087, -0.30901699437494745, -0.4539904997395468, -0.5877852522924731, -0.7071067811865476, -0.8090169943749475, -0.8910065241883679, -0.9510565162951535, -0.9876883405951378, -1.0, -0.9876883405951378, -0.9510565162951535, -0.8910065241883679, -0.8090169943749475, -0.7071067811865476, -0.5877852522924731, -0.4539904997395468, -0.30901699437494745, -0.15643446504023087 };
};
template <int R, int K, int DIR> struct MulW {
    __device__ __forceinline__ static cplx apply(cplx a) {
        if constexpr (32 % R != 0) {
            constexpr int B = (24 % R == 0) ? 24 : 40;
            static_assert(B % R == 0, "radix must divide 32, 24 or 40");
            constexpr int i = ((K * (B / R)) % B + B) % B;
            if constexpr (i == 0) return a;
            else if constexpr (i == B / 2) return make_double2(-a.x, -a.y);
            else if constexpr (i == B / 4) return DIR > 0 ? make_double2(a.y, -a.x) : make_double2(-a.y, a.x);
            else if constexpr (i == 3 * B / 4) return DIR > 0 ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
            else {
                constexpr double wr = NGon<B>::C[i];
                constexpr double wi = (DIR > 0 ? -1.0 : 1.0) * NGon<B>::S[i];
                return make_double2(fma(-a.y, wi, a.x * wr), fma(a.x, wi, a.y * wr));
            }
        } else {
        constexpr int idx = ((K * (32 / R)) % 32 + 32) % 32;
        constexpr double C32[32] = { 1.00000000000000000000, 0.98078528040323043058, 0.92387953251128673848, 0.83146961230254523567, 0.70710678118654757274, 0.55557023301960228867, 0.38268343236508983729, 0.19509032201612833135, 0.00000000000000006123, -0.19509032201612819257, -0.38268343236508972627, -0.55557023301960195560, -0.70710678118654746172, -0.83146961230254534669, -0.92387953251128673848, -0.98078528040323043058, -1.00000000000000000000, -0.98078528040323043058, -0.92387953251128684951, -0.83146961230254545772, -0.70710678118654768376, -0.55557023301960217765, -0.38268343236509033689, -0.19509032201612866442, -0.00000000000000018370, 0.19509032201612830359, 0.38268343236509000382, 0.55557023301960184458, 0.70710678118654735069, 0.83146961230254523567, 0.92387953251128651644, 0.98078528040323031956 };
        constexpr double S32[32] = { 0.00000000000000000000, 0.19509032201612824808, 0.38268343236508978178, 0.55557023301960217765, 0.70710678118654746172, 0.83146961230254523567, 0.92387953251128673848, 0.98078528040323043058, 1.00000000000000000000, 0.98078528040323043058, 0.92387953251128673848, 0.83146961230254545772, 0.70710678118654757274, 0.55557023301960217765, 0.38268343236508989280, 0.19509032201612860891, 0.00000000000000012246, -0.19509032201612835911, -0.38268343236508967076, -0.55557023301960195560, -0.70710678118654746172, -0.83146961230254523567, -0.92387953251128651644, -0.98078528040323031956, -1.00000000000000000000, -0.98078528040323043058, -0.92387953251128662746, -0.83146961230254545772, -0.70710678118654768376, -0.55557023301960217765, -0.38268343236509039240, -0.19509032201612871993 };
        if constexpr (idx == 0) return a;
        else if constexpr (idx == 16) return make_double2(-a.x, -a.y);
        else if constexpr (idx == 8) return DIR > 0 ? make_double2(a.y, -a.x) : make_double2(-a.y, a.x);
        else if constexpr (idx == 24) return DIR > 0 ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
        else if constexpr (idx == 4) {
            constexpr double c = 0.70710678118654752440;
            return DIR > 0 ? make_double2((a.x + a.y) * c, (a.y - a.x) * c) : make_double2((a.x - a.y) * c, (a.x + a.y) * c);
        } else if constexpr (idx == 12) {
            constexpr double c = 0.70710678118654752440;
            return DIR > 0 ? make_double2((a.y - a.x) * c, -(a.x + a.y) * c) : make_double2(-(a.x + a.y) * c, (a.x - a.y) * c);
        } else {
            constexpr double wr = C32[idx];
            constexpr double wi = (DIR > 0 ? -1.0 : 1.0) * S32[idx];
            return make_double2(fma(-a.y, wi, a.x * wr), fma(a.x, wi, a.y * wr));
        }
        }
    }
};

// In-register DFT of size R (natural order in, natural order out).
// DIR=+1: X[k] = sum_j a[j] exp(-2 pi i jk/R);  DIR=-1: conjugate kernel (unnormalised).
template <int R, int DIR> struct Dft;

template <int DIR> struct Dft<1, DIR> { __device__ __forceinline__ static void run(cplx (&)[1]) {} };
template <int DIR> struct Dft<2, DIR> {
    __device__ __forceinline__ static void run(cplx (&a)[2]) {
        const cplx s = cadd(a[0], a[1]), d = csub(a[0], a[1]); a[0] = s; a[1] = d;
    }
};

// multiply by -i (DIR > 0) or +i (DIR < 0)
template <int DIR> __device__ __forceinline__ cplx mul_mi(cplx a) { return DIR > 0 ? make_double2(a.y, -a.x) : make_double2(-a.y, a.x); }
template <int DIR> struct Dft<3, DIR> {
    __device__ __forceinline__ static void run(cplx (&a)[3]) {
        constexpr double h = 0.8660254037844386;         // sin(2 pi / 3)
        const cplx t1 = cadd(a[1], a[2]);
        const cplx t2 = make_double2(fma(-0.5, t1.x, a[0].x), fma(-0.5, t1.y, a[0].y));
        const cplx d = csub(a[1], a[2]);
        const cplx s = mul_mi<DIR>(make_double2(h * d.x, h * d.y));
        a[0] = cadd(a[0], t1); a[1] = cadd(t2, s); a[2] = csub(t2, s);
    }
};
template <int DIR> struct Dft<5, DIR> {
    __device__ __forceinline__ static void run(cplx (&a)[5]) {
        constexpr double c1 = 0.30901699437494745, c2 = -0.8090169943749475;    // cos(2 pi/5), cos(4 pi/5)
        constexpr double s1 = 0.9510565162951535, s2 = 0.5877852522924731;      // sin(2 pi/5), sin(4 pi/5)
        const cplx t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]), t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
        const cplx m1 = make_double2(fma(c2, t2.x, fma(c1, t1.x, a[0].x)), fma(c2, t2.y, fma(c1, t1.y, a[0].y)));
        const cplx m2 = make_double2(fma(c1, t2.x, fma(c2, t1.x, a[0].x)), fma(c1, t2.y, fma(c2, t1.y, a[0].y)));
        const cplx n1 = mul_mi<DIR>(make_double2(fma(s2, t4.x, s1 * t3.x), fma(s2, t4.y, s1 * t3.y)));
        const cplx n2 = mul_mi<DIR>(make_double2(fma(-s1, t4.x, s2 * t3.x), fma(-s1, t4.y, s2 * t3.y)));
        a[0] = cadd(a[0], cadd(t1, t2));
        a[1] = cadd(m1, n1); a[4] = csub(m1, n1);
        a[2] = cadd(m2, n2); a[3] = csub(m2, n2);
    }
};

template <int R, int DIR, int... K>
__device__ __forceinline__ void dit_combine(cplx (&a)[R], const cplx (&ev)[R / 2], const cplx (&od)[R / 2],
                                            std::integer_sequence<int, K...>) {
    (([&] { const cplx t = MulW<R, K, DIR>::apply(od[K]); a[K] = cadd(ev[K], t); a[K + R / 2] = csub(ev[K], t); }()), ...);
}

template <int R, int DIR> struct Dft {
    __device__ __forceinline__ static void run(cplx (&a)[R]) {
        cplx ev[R / 2], od[R / 2];
#pragma unroll
        for (int j = 0; j < R / 2; ++j) { ev[j] = a[2 * j]; od[j] = a[2 * j + 1]; }
        Dft<R / 2, DIR>::run(ev);
        Dft<R / 2, DIR>::run(od);
        dit_combine<R, DIR>(a, ev, od, std::make_integer_sequence<int, R / 2>{});
    }
};

// Forward DFT whose inputs a[R/2..R) are zero (zero-padded line): one DIF level
// degenerates to a twiddle, X[2k] = DFT_{R/2}(a)[k], X[2k+1] = DFT_{R/2}(a.*W_R^j)[k].
template <int R, int DIR, int... J>
__device__ __forceinline__ void halfzero_twiddle(cplx (&od)[R / 2], const cplx (&a)[R], std::integer_sequence<int, J...>) {
    ((od[J] = MulW<R, J, DIR>::apply(a[J])), ...);
}
template <int R, int DIR>
__device__ __forceinline__ void dft_halfzero(cplx (&a)[R]) {
    if constexpr (R == 2) { a[1] = a[0]; }
    else {
        cplx ev[R / 2], od[R / 2];
#pragma unroll
        for (int j = 0; j < R / 2; ++j) ev[j] = a[j];
        halfzero_twiddle<R, DIR>(od, a, std::make_integer_sequence<int, R / 2>{});
        Dft<R / 2, DIR>::run(ev);
        Dft<R / 2, DIR>::run(od);
#pragma unroll
        for (int k = 0; k < R / 2; ++k) { a[2 * k] = ev[k]; a[2 * k + 1] = od[k]; }
    }
}

// DFT of which only outputs [0, R/2) are wanted (cropped line): the last DIT level
// computes only the "+" halves.  Outputs [R/2, R) are left unspecified.
template <int R, int DIR, int... K>
__device__ __forceinline__ void halfout_combine(cplx (&a)[R], const cplx (&ev)[R / 2], const cplx (&od)[R / 2],
                                                std::integer_sequence<int, K...>) {
    ((a[K] = cadd(ev[K], MulW<R, K, DIR>::apply(od[K]))), ...);
}
template <int R, int DIR>
__device__ __forceinline__ void dft_halfout(cplx (&a)[R]) {
    if constexpr (R == 2) { a[0] = cadd(a[0], a[1]); }
    else {
        cplx ev[R / 2], od[R / 2];
#pragma unroll
        for (int j = 0; j < R / 2; ++j) { ev[j] = a[2 * j]; od[j] = a[2 * j + 1]; }
        Dft<R / 2, DIR>::run(ev);
        Dft<R / 2, DIR>::run(od);
        halfout_combine<R, DIR>(a, ev, od, std::make_integer_sequence<int, R / 2>{});
    }
}

// w[k] = w1^k for k = 1..R-1 by a depth-<=4 product tree (error ~ 4 ulp).
template <int R> __device__ __forceinline__ void twiddle_powers(cplx w1, cplx (&w)[R]) {
    w[0] = make_double2(1.0, 0.0);
    if constexpr ((R & (R - 1)) != 0) {
        // radices with an odd factor (3, 6, 12, 24, 5, 10, 20): the same tree, w[k] = w[p] * w[k - p] with p the
        // largest power of two <= k
        if constexpr (R > 1) w[1] = w1;
#pragma unroll
        for (int k = 2; k < R; ++k) {
            int p = 1; while (2 * p <= k) p *= 2;
            w[k] = (k == p) ? cmul(w[p / 2], w[p / 2]) : cmul(w[p], w[k - p]);
        }
        return;
    }
    if constexpr (R > 1) w[1] = w1;
    if constexpr (R > 2) { w[2] = cmul(w1, w1); w[3] = cmul(w[2], w1); }
    if constexpr (R > 4) { w[4] = cmul(w[2], w[2]); w[5] = cmul(w[4], w1); w[6] = cmul(w[4], w[2]); w[7] = cmul(w[4], w[3]); }
    if constexpr (R > 8) {
        w[8] = cmul(w[4], w[4]);
#pragma unroll
        for (int k = 9; k < 16; ++k) w[k] = cmul(w[8], w[k - 8]);
    }
    if constexpr (R > 16) {
        w[16] = cmul(w[8], w[8]);
#pragma unroll
        for (int k = 17; k < 32; ++k) w[k] = cmul(w[16], w[k - 16]);
    }
}

// Factorisation of one line into 2..4 stages.
template <int L_, int T_, int R0_, int R1_, int R2_ = 1, int R3_ = 1> struct Cfg {
    static constexpr int L = L_, T = T_, E = L_ / T_;
    static constexpr int R0 = R0_, R1 = R1_, R2 = R2_, R3 = R3_;
    static constexpr int NS = 2 + (R2_ > 1 ? 1 : 0) + (R3_ > 1 ? 1 : 0);
    static_assert(R0_ * R1_ * R2_ * R3_ == L_, "radices must multiply to L");
    static_assert(E % R0_ == 0 && E % R1_ == 0 && E % R2_ == 0 && E % R3_ == 0, "radix must divide elements per thread");
    static_assert(R3_ == 1 || R2_ > 1, "a 4th stage needs a 3rd");
    template <int S> static constexpr int LS() { return S == 0 ? L : (S == 1 ? L / R0 : (S == 2 ? L / (R0 * R1) : L / (R0 * R1 * R2))); }
    template <int S> static constexpr int R() { return S == 0 ? R0 : (S == 1 ? R1 : (S == 2 ? R2 : R3)); }
    // full stage-twiddle table (optional, kept in LDS): stage S holds (R_S - 1) * M_S entries
    // tw_full[TWOFF<S>() + (q-1)*M_S + r] = exp(-2 pi i * r*q / LS_S), q = 1..R_S-1, r < M_S = LS_S / R_S
    template <int S> static constexpr int TWCNT() { return (LS<S>() / R<S>() > 1) ? (R<S>() - 1) * (LS<S>() / R<S>()) : 0; }
    template <int S> static constexpr int TWOFF() { return S == 0 ? 0 : TWOFF<(S > 0 ? S - 1 : 0)>() + TWCNT<(S > 0 ? S - 1 : 0)>(); }
    static constexpr int TWLEN = TWCNT<0>() + TWCNT<1>() + (NS >= 3 ? TWCNT<2>() : 0) + (NS >= 4 ? TWCNT<3>() : 0);
};

// In-place position touched by slot e of thread t in stage S.
template <class C, int S> __device__ __forceinline__ int stage_pos(int t, int e) {
    constexpr int LS = C::template LS<S>(), R = C::template R<S>();
    constexpr int M = LS / R, NB = C::E / R;
    const int u = e % NB, q = e / NB;
    const int b = t + C::T * u;
    return (b / M) * LS + (b % M) + M * q;
}

// Which thread (of the line) owns position `pos` in stage S, and the host/compile-time twin of stage_pos.
template <class C, int S> constexpr int stage_owner(int pos) {
    constexpr int LS = C::template LS<S>(), R = C::template R<S>();
    constexpr int M = LS / R;
    const int b = (pos / LS) * M + (pos % LS) % M;
    return b % C::T;
}
template <class C, int S> constexpr int stage_pos_c(int t, int e) {
    constexpr int LS = C::template LS<S>(), R = C::template R<S>();
    constexpr int M = LS / R, NB = C::E / R;
    const int u = e % NB, q = e / NB;
    const int b = t + C::T * u;
    return (b / M) * LS + (b % M) + M * q;
}
// An exchange SA -> SB is WAVE-LOCAL when every element stays among the threads t of one group of G consecutive t
// (G = the threads of a line that share a wavefront: 64 / LINES for interleaved lines, min(T, 64) for contiguous
// ones).  Such an exchange needs no workgroup barrier: LDS executes one wave's instructions in order, and no other
// wave touches the positions involved -- the waves of a workgroup then drift apart and overlap each other's LDS and
// VALU phases instead of marching in lock step.  (Cfg1024 = 16.8.8 with 8 interleaved lines: the exchange between the
// two radix-8 stages moves data inside blocks of 64 positions owned by the 8 threads t = 8w .. 8w+7 of wave w.)
template <class C, int SA, int SB> constexpr bool exchange_wave_local(int G) {
    if (G >= C::T) return true;
    for (int t = 0; t < C::T; ++t)
        for (int e = 0; e < C::E; ++e)
            if (stage_owner<C, SB>(stage_pos_c<C, SA>(t, e)) / G != t / G) return false;
    return true;
}

// How the lines of one workgroup share LDS: address(pos) = off + (pos + pos>>PADSHIFT)*LSTR + xi
// PADSHIFT < 0: no padding; instead the low bits of a position are XOR-ed with the bits above its block of 8, which spreads
// the stride-8 accesses of a last radix-8 stage over the banks just as the padding does (4 interleaved 16-byte lines: the
// four consecutive t of a 16-lane LDS cycle land in four different 64-byte bank groups in every stage of 16.8.8 /
// 24.8.8 / 16.16.8) at no cost in LDS -- what lets two half-tile workgroups plus their twiddle tables share a CU.
template <int LSTR_, int PADSHIFT_, bool SPLIT_> struct LdsLayout {
    static constexpr int LSTR = LSTR_, PADSHIFT = PADSHIFT_;
    static constexpr bool SPLIT = SPLIT_;
    static constexpr bool SWIZZLE = PADSHIFT_ < 0;
    static constexpr int SWZMASK = ((256 / (SPLIT_ ? 8 : 16)) / LSTR_ > 1 ? (256 / (SPLIT_ ? 8 : 16)) / LSTR_ : 1) - 1;
    __device__ __forceinline__ static int addr(int off, int xi, int pos) {
        if constexpr (SWIZZLE) return off + (pos ^ ((pos >> 3) & SWZMASK)) * LSTR + xi;
        else return off + (pos + (pos >> PADSHIFT)) * LSTR + xi;
    }
    // elements (of 8 or 16 bytes) needed per line
    static constexpr int line_elems(int L) { return SWIZZLE ? L : L + (L >> (SWIZZLE ? 0 : PADSHIFT)); }
    static constexpr int elem_bytes() { return SPLIT ? 8 : 16; }
    // threads t of one line that share a wavefront (lane = xi + LSTR * t for interleaved lines, t + T * line otherwise)
    static constexpr int wave_group() { return LSTR == 1 ? 64 : (64 % LSTR == 0 ? 64 / LSTR : 1); }
};

// COMP: 0 = real parts, 1 = imaginary parts (SPLIT layouts), 2 = whole complex numbers
template <class C, int S, class LL, int COMP>
__device__ __forceinline__ void lds_write(const cplx (&v)[C::E], int t, char* smem, int off, int xi) {
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        const int a = LL::addr(off, xi, stage_pos<C, S>(t, e));
        if constexpr (COMP == 2) reinterpret_cast<cplx*>(smem)[a] = v[e];
        else reinterpret_cast<double*>(smem)[a] = (COMP == 0 ? v[e].x : v[e].y);
    }
}
template <class C, int S, class LL, int COMP>
__device__ __forceinline__ void lds_read(cplx (&v)[C::E], int t, const char* smem, int off, int xi) {
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        const int a = LL::addr(off, xi, stage_pos<C, S>(t, e));
        if constexpr (COMP == 2) v[e] = reinterpret_cast<const cplx*>(smem)[a];
        else if constexpr (COMP == 0) v[e].x = reinterpret_cast<const double*>(smem)[a];
        else v[e].y = reinterpret_cast<const double*>(smem)[a];
    }
}

#ifndef LSFC_FFT_HOST_EMULATION
// Compile-time switch of the wave-local form: bit 0 contiguous (x) passes, bit 1 interleaved (y, z) passes.  DEFAULT 0 (every
// exchange synchronises the whole workgroup): measured on MI355X the wave-local form changes nothing at 512^3 (fused
// pass 6.05 -> 6.2 ms, x / y passes equal) and gains ~5 % on the fused pass at 256^3 only, and two of five test runs of
// builds with bit 1 set ended in a GPU memory fault in the first apply of the session that no workgroup-barrier build
// ever showed (profiles/r02_experiment_wave_local_sync.log).  Kept for the record and for A/B builds (make EXTRA=-DLSFC_WAVE_LOCAL_SYNC=1).
#ifndef LSFC_WAVE_LOCAL_SYNC
#define LSFC_WAVE_LOCAL_SYNC 0
#endif
// LOCAL: the exchange stays inside each wavefront -- order the wave's own LDS accesses, nothing else
template <bool LOCAL> __device__ __forceinline__ void lds_barrier() {
    if constexpr (LOCAL) {
        // the wave drains its own LDS queue between the phases (s_waitcnt lgkmcnt(0): LDS only, global loads stay in flight)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else LSFC_BARRIER();
}
template <class C, int SA, int SB, class LL> constexpr bool exchange_is_local() {
    constexpr int bit = LL::LSTR == 1 ? 1 : 2;
    if constexpr ((LSFC_WAVE_LOCAL_SYNC & bit) == 0) return false;
    else return exchange_wave_local<C, SA, SB>(LL::wave_group()) && exchange_wave_local<C, SB, SA>(LL::wave_group());
}
// Move every element from its stage-SA owner to its stage-SB owner through LDS.
template <class C, int SA, int SB, class LL>
__device__ __forceinline__ void exchange(cplx (&v)[C::E], int t, char* smem, int off, int xi) {
    constexpr bool LOC = exchange_is_local<C, SA, SB, LL>();
    if constexpr (LL::SPLIT) {
        lds_write<C, SA, LL, 0>(v, t, smem, off, xi); lds_barrier<LOC>();
        lds_read<C, SB, LL, 0>(v, t, smem, off, xi);  lds_barrier<LOC>();
        lds_write<C, SA, LL, 1>(v, t, smem, off, xi); lds_barrier<LOC>();
        lds_read<C, SB, LL, 1>(v, t, smem, off, xi);  lds_barrier<LOC>();
    } else {
        lds_write<C, SA, LL, 2>(v, t, smem, off, xi); lds_barrier<LOC>();
        lds_read<C, SB, LL, 2>(v, t, smem, off, xi);  lds_barrier<LOC>();
    }
}
// second half of an exchange whose stores were issued by stage<..., LLW>: wait for them, fetch the stage-SB slots
// TRAIL = false: the caller frees the buffer itself before anything is stored into it again (stage<..., BARF>, or a barrier)
template <class C, int SB, class LL, bool TRAIL = true>
__device__ __forceinline__ void exchange_read(cplx (&v)[C::E], int t, char* smem, int off, int xi) {
    static_assert(!LL::SPLIT, "exchange_read: whole-complex layouts only");
    LSFC_BARRIER();
    lds_read<C, SB, LL, 2>(v, t, smem, off, xi);
    if constexpr (TRAIL) LSFC_BARRIER();
}
// whether the forward transform ENDS with a wave-local exchange (callers that re-use the exchange buffer right after
// it, as the fused pass does for the symbol, then need a workgroup barrier of their own)
template <class C, class LL> constexpr bool forward_ends_local() {
    if constexpr (C::NS >= 4) return exchange_is_local<C, 2, 3, LL>();
    else if constexpr (C::NS >= 3) return exchange_is_local<C, 1, 2, LL>();
    else return exchange_is_local<C, 0, 1, LL>();
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// Exchange between the last two stages THROUGH THE LANES of a wavefront instead of LDS (round 3).
// When both stages have radix 8 (16.8.8, 24.8.8, 8.8.8) the exchange between them is, for every u, the 8 x 8 transpose
//     slot (u, q) of thread t   ->   slot (u, t & 7) of thread (t & ~7) | q
// (stage_pos<C, S>(t, u + NB q) = 64 (b / 8) + (b & 7) + 8 q  ==  stage_pos<C, S + 1>(t', u + NB q') = 8 b' + q' with
// b = t + T u: b' = 8 (b / 8) + q, q' = b & 7; checked for every factorisation by tests/emu).  In the interleaved passes
// lane = line + LSTR * t, so the eight threads of a transpose sit in ONE wavefront at lane bits LB .. LB + 2 (LB = 3 for 8
// interleaved lines, 2 for 4) and the transpose is three butterfly steps, step k swapping the slots (q, q | 2^k) between
// the lanes that differ in bit LB + k.  gfx950 swaps across lane bit 5 / 4 with one v_permlane32_swap / v_permlane16_swap
// per dword pair, across bit 3 with two DPP moves (row_ror:8 under complementary bank masks) and across bit 2 with
// row_shr:4 / row_shl:4.  Per 16-byte element that is 2 + 2 + 4 VALU operations against one ds_write_b128 (13 cycles of the
// CU's LDS store path, the slowest LDS operation of the fused pass) + one ds_read_b128, and two workgroup barriers go.
#ifndef LSFC_FFT_HOST_EMULATION
template <int BIT> __device__ __forceinline__ void xlane_step32(unsigned& a, unsigned& b) {
    // lanes with bit BIT clear keep a and take the partner's a into b; lanes with the bit set keep b and take the
    // partner's b into a (partner = lane ^ (1 << BIT))
    if constexpr (BIT == 5) { const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r[0]; b = r[1]; }
    else if constexpr (BIT == 4) { const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = r[0]; b = r[1]; }
    else if constexpr (BIT == 3) {
        // row_ror:8 = lane ^ 8 inside a row of 16; bank_mask 0xc: lanes 8..15 of the row are written, 0x3: lanes 0..7
        const unsigned na = (unsigned)__builtin_amdgcn_update_dpp((int)a, (int)b, 0x128, 0xf, 0xc, false);
        const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x128, 0xf, 0x3, false);
        a = na; b = nb;
    } else {
        static_assert(BIT == 2, "xlane_step32: lane bits 2..5");
        // row_shr:4 (lane i takes lane i - 4) into the lanes with bit 2 set (banks 1, 3); row_shl:4 into the others
        const unsigned na = (unsigned)__builtin_amdgcn_update_dpp((int)a, (int)b, 0x114, 0xf, 0xa, false);
        const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x104, 0xf, 0x5, false);
        a = na; b = nb;
    }
}
template <int BIT> __device__ __forceinline__ void xlane_step(cplx& a, cplx& b) {
    unsigned a0 = (unsigned)__double2loint(a.x), a1 = (unsigned)__double2hiint(a.x), a2 = (unsigned)__double2loint(a.y), a3 = (unsigned)__double2hiint(a.y);
    unsigned b0 = (unsigned)__double2loint(b.x), b1 = (unsigned)__double2hiint(b.x), b2 = (unsigned)__double2loint(b.y), b3 = (unsigned)__double2hiint(b.y);
    xlane_step32<BIT>(a0, b0); xlane_step32<BIT>(a1, b1); xlane_step32<BIT>(a2, b2); xlane_step32<BIT>(a3, b3);
    a = make_double2(__hiloint2double((int)a1, (int)a0), __hiloint2double((int)a3, (int)a2));
    b = make_double2(__hiloint2double((int)b1, (int)b0), __hiloint2double((int)b3, (int)b2));
}
#endif
// Which exchanges can run through the lanes.  For ANY two consecutive stages S, S + 1 of equal radix R the exchange is, for every u,
// an R x R transpose between the slot index q and log2(R) bits of the thread index: with b = t + T u = M_S beta + M' gamma + delta
// (M' = M_S / R the butterfly stride of stage S + 1, gamma < R, delta < M')
//     stage_pos<C, S>(t, u + NB q) = LS_S beta + M' gamma + delta + M_S q  ==  stage_pos<C, S + 1>(t', u + NB q')
//     with b' = M_S beta + M' q + delta and q' = gamma:   slot (u, q) of thread t  <->  slot (u, gamma(t)) of thread t[gamma := q]
// and when M_S divides T the bits of gamma are bits log2(M') .. log2(M_S) - 1 of t itself (checked for every factorisation by
// tests/emu).  Lane = line + LSTR * t, so they sit at lane bits LB = log2(LSTR) + log2(M') and up: usable when they are bits 2..5.
// 16.8.8 / 24.8.8 / 8.8.8: the exchange 1 -> 2 (t bits 0..2); 20.4.4.4 in 4-line workgroups: 1 -> 2 (t bits 2, 3 = lane bits 4, 5:
// two permlane swaps per dword pair) AND 2 -> 3 (t bits 0, 1 = lane bits 2, 3), which leaves ONE exchange through LDS per direction.
constexpr int xlane_log2(int x) { int r = 0; while ((1 << r) < x) ++r; return r; }
template <class C, int S> constexpr int xlane_mnext() { return C::template LS<S>() / C::template R<S>() / C::template R<S>(); }   // M'
template <class C, int S, int LSTR> constexpr int xlane_lowbit() { return xlane_log2(LSTR) + xlane_log2(xlane_mnext<C, S>() > 0 ? xlane_mnext<C, S>() : 1); }
template <class C, int S, class LL> constexpr bool xlane_stage_ok() {
    if constexpr (S < 1 || S + 1 >= C::NS) return false;
    else {
        constexpr int R = C::template R<S>(), M = C::template LS<S>() / R;
        if constexpr (R != C::template R<S + 1>() || (R != 4 && R != 8 && R != 16) || M % R != 0 || C::T % M != 0 || C::E % R != 0) return false;
        else {
            constexpr int MN = M / R, LB = xlane_lowbit<C, S, LL::LSTR>();
            return (MN & (MN - 1)) == 0 && (LL::LSTR == 8 || LL::LSTR == 4) && !LL::SPLIT && LB >= 2 && LB + xlane_log2(R) <= 6;
        }
    }
}
// at least one exchange of C can run through the lanes of layout LL
template <class C, class LL> constexpr bool xlane_ok() { return xlane_stage_ok<C, 1, LL>() || xlane_stage_ok<C, 2, LL>(); }
// the butterfly steps of the transpose after stage S, written over a primitive STEP(k, a, b) (k-th of the log2(R) lane bits) so
// that tests/emu can run the same schedule on an emulated wavefront
template <class C, int S, class STEP> __device__ __forceinline__ void xlane_transpose_with(cplx (&v)[C::E], STEP&& step) {
    constexpr int R = C::template R<S>(), NB = C::E / R, NK = xlane_log2(R);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#pragma unroll
            for (int q = 0; q < R; ++q)
                if (!((q >> k) & 1)) step(k, v[u + NB * q], v[u + NB * (q | (1 << k))]);
        }
    }
}
#ifndef LSFC_FFT_HOST_EMULATION
template <class C, int S, int LSTR> __device__ __forceinline__ void xlane_transpose(cplx (&v)[C::E]) {
    constexpr int LB = xlane_lowbit<C, S, LSTR>(), NK = xlane_log2(C::template R<S>());
    xlane_transpose_with<C, S>(v, [](int k, cplx& a, cplx& b) __attribute__((always_inline)) {
        if (k == 0) xlane_step<LB>(a, b);
        else if (k == 1) xlane_step<LB + 1>(a, b);
        else if (k == 2) { if constexpr (NK > 2) xlane_step<LB + 2>(a, b); }
        else { if constexpr (NK > 3) xlane_step<LB + 3>(a, b); }
    });
}
#endif

// One butterfly stage on the register slots.  PRUNE: 0 full; 1 forward with inputs
// q >= R/2 zero; 2 inverse with only outputs q < R/2 wanted.
// TWFULL: `tw` is the full stage-twiddle table (Cfg::TWOFF layout, normally staged in LDS): every power is read,
// none is computed -- trades the product trees' fp64 work for LDS reads.
// TWCHAIN (computed twiddles only): the powers w1^q are produced one after the other, w1^q = w1^(q-1) * w1, and applied at once
// -- two complex numbers live instead of the R of the product tree, at an error of ~q ulp in the q-th power instead of ~4 ulp
// (3e-15 at radix 16).  For the passes that would otherwise spill (1024-thread workgroups, 128 registers).
// LLW != void (whole-complex layouts): every output also goes to its stage-S position of the exchange buffer as soon as it
// exists (the first half of exchange<C, S, ., LLW>; finish with exchange_read) -- the 16-byte LDS stores, the slowest LDS
// operation of the pass, then queue behind the butterflies and twiddle products still being computed instead of after them.
// BARF: the workgroup barrier that frees the exchange buffer (every wave has finished the loads of the previous exchange) is
// taken here, just before the stage's FIRST store, instead of right after those loads: the first butterfly of every wave runs
// while the slower waves are still loading.  (__syncthreads drains the wave's own LDS queue first, so all of its loads of the
// previous exchange -- also those for its later butterflies -- have completed when it passes.)
template <class C, int S, int DIR, int PRUNE, bool TWFULL = false, bool TWCHAIN = false, class LLW = void, bool BARF = false>
__device__ __forceinline__ void stage(cplx (&v)[C::E], int t, const cplx* __restrict__ tw, char* smem = nullptr, int off = 0, int xi = 0) {
    constexpr int LS = C::template LS<S>(), R = C::template R<S>();
    constexpr int M = LS / R, NB = C::E / R;
    constexpr bool WR = !std::is_void<LLW>::value;
    [[maybe_unused]] bool first = true;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        cplx a[R];
        [[maybe_unused]] auto put = [&](int q) {
            if constexpr (WR) {
                if constexpr (BARF) { if (first) { LSFC_BARRIER(); first = false; } }
                reinterpret_cast<cplx*>(smem)[LLW::addr(off, xi, stage_pos<C, S>(t, u + NB * q))] = a[q];
            }
        };
#pragma unroll
        for (int q = 0; q < R; ++q) a[q] = v[u + NB * q];
        if constexpr (TWFULL && M > 1) {
            // table twiddles: each is read and applied on the spot, four at a time (LSFC_TW_CHUNK), so that at most four
            // of them are live instead of all R - 1 (60 registers at radix 16) -- what keeps the persistent fused pass,
            // which also holds the next tile's data, inside the 256-register budget
            const int r = (t + C::T * u) % M;
            auto tw_at = [&](int q) { const cplx x = tw[C::template TWOFF<S>() + (q - 1) * M + r]; return (DIR < 0) ? cconj(x) : x; };
            if constexpr (DIR < 0) {
#pragma unroll
                for (int q = 1; q < R; ++q) {
                    a[q] = cmul(a[q], tw_at(q));
#ifdef LSFC_TW_CHUNK
                    if (q % LSFC_TW_CHUNK == 0) __builtin_amdgcn_sched_barrier(0);
#endif
                }
            }
            if constexpr (PRUNE == 1) dft_halfzero<R, DIR>(a);
            else if constexpr (PRUNE == 2) dft_halfout<R, DIR>(a);
            else Dft<R, DIR>::run(a);
            if constexpr (DIR > 0) {
                put(0);
#pragma unroll
                for (int q = 1; q < R; ++q) {
                    a[q] = cmul(a[q], tw_at(q));
                    put(q);
#ifdef LSFC_TW_CHUNK
                    if (q % LSFC_TW_CHUNK == 0) __builtin_amdgcn_sched_barrier(0);
#endif
                }
            } else {
#pragma unroll
                for (int q = 0; q < R; ++q) put(q);
            }
        } else if constexpr (TWCHAIN && M > 1) {
            const int r = (t + C::T * u) % M;
            cplx w1 = tw[r * (C::L / LS)];
            if constexpr (DIR < 0) w1 = cconj(w1);
            if constexpr (DIR < 0) {
                cplx wq = w1;
#pragma unroll
                for (int q = 1; q < R; ++q) { a[q] = cmul(a[q], wq); if (q + 1 < R) wq = cmul(wq, w1); }
            }
            if constexpr (PRUNE == 1) dft_halfzero<R, DIR>(a);
            else if constexpr (PRUNE == 2) dft_halfout<R, DIR>(a);
            else Dft<R, DIR>::run(a);
            if constexpr (DIR > 0) {
                cplx wq = w1;
#pragma unroll
                for (int q = 1; q < R; ++q) { a[q] = cmul(a[q], wq); if (q + 1 < R) wq = cmul(wq, w1); }
            }
        } else {
        cplx w[R];
        if constexpr (M > 1) {
            const int r = (t + C::T * u) % M;
            cplx w1 = tw[r * (C::L / LS)];
            if constexpr (DIR < 0) w1 = cconj(w1);
            twiddle_powers<R>(w1, w);
        }
        if constexpr (DIR < 0 && M > 1) {
#pragma unroll
            for (int q = 1; q < R; ++q) a[q] = cmul(a[q], w[q]);
        }
        if constexpr (PRUNE == 1) dft_halfzero<R, DIR>(a);
        else if constexpr (PRUNE == 2) dft_halfout<R, DIR>(a);
        else Dft<R, DIR>::run(a);
        if constexpr (DIR > 0 && M > 1) {
#pragma unroll
            for (int q = 1; q < R; ++q) a[q] = cmul(a[q], w[q]);
        }
        }
        if constexpr (WR && !(TWFULL && M > 1)) {
#pragma unroll
            for (int q = 0; q < R; ++q) put(q);
        }
#pragma unroll
        for (int q = 0; q < R; ++q) v[u + NB * q] = a[q];
    }
}

#ifndef LSFC_FFT_HOST_EMULATION
// natural (time) order in slots -> storage (digit-reversed frequency) order in slots
template <class C, class LL, bool PRUNE_IN, bool TWFULL = false, bool TWCHAIN = false>
__device__ __forceinline__ void fft_forward(cplx (&v)[C::E], int t, const cplx* __restrict__ tw, char* smem, int off, int xi) {
    stage<C, 0, +1, PRUNE_IN ? 1 : 0, TWFULL, TWCHAIN>(v, t, tw);
    exchange<C, 0, 1, LL>(v, t, smem, off, xi);
    stage<C, 1, +1, 0, TWFULL, TWCHAIN>(v, t, tw);
    if constexpr (C::NS >= 3) {
        exchange<C, 1, 2, LL>(v, t, smem, off, xi);
        stage<C, 2, +1, 0, TWFULL, TWCHAIN>(v, t, tw);
    }
    if constexpr (C::NS >= 4) {
        exchange<C, 2, 3, LL>(v, t, smem, off, xi);
        stage<C, 3, +1, 0, TWFULL, TWCHAIN>(v, t, tw);
    }
}

// storage order in slots -> natural (time) order in slots, unnormalised
template <class C, class LL, bool PRUNE_OUT, bool TWFULL = false, bool TWCHAIN = false>
__device__ __forceinline__ void fft_inverse(cplx (&v)[C::E], int t, const cplx* __restrict__ tw, char* smem, int off, int xi) {
    if constexpr (C::NS >= 4) {
        stage<C, 3, -1, 0, TWFULL, TWCHAIN>(v, t, tw);
        exchange<C, 3, 2, LL>(v, t, smem, off, xi);
    }
    if constexpr (C::NS >= 3) {
        stage<C, 2, -1, 0, TWFULL, TWCHAIN>(v, t, tw);
        exchange<C, 2, 1, LL>(v, t, smem, off, xi);
    }
    stage<C, 1, -1, 0, TWFULL, TWCHAIN>(v, t, tw);
    exchange<C, 1, 0, LL>(v, t, smem, off, xi);
    stage<C, 0, -1, PRUNE_OUT ? 2 : 0, TWFULL, TWCHAIN>(v, t, tw);
}
// The same transforms with the exchange stores issued from inside the stages (stage<..., LL>; whole-complex layouts).
// `hook` runs between the first stage and the wait for its stores (the fused pass loads its symbol there).
// DEFER: the barriers that free the exchange buffer are taken inside the following stage, before its first store (BARF), and
// by the CALLER after the last forward stage (before it stores into the buffer) and before the first stage of the next
// forward transform (a stage with BARF, or a barrier).
// XL: every exchange S -> S + 1, S >= 1, that can (xlane_stage_ok) runs through the lanes (xlane_transpose), not LDS.
template <class C, class LL, int S, bool TWFULL, bool DEFER, bool XL>
__device__ __forceinline__ void fft_forward_ws_from(cplx (&v)[C::E], int t, const cplx* __restrict__ tw, char* smem, int off, int xi) {
    if constexpr (S == C::NS - 1) stage<C, S, +1, 0, TWFULL>(v, t, tw);
    else if constexpr (XL && xlane_stage_ok<C, S, LL>()) {
        stage<C, S, +1, 0, TWFULL>(v, t, tw);
        xlane_transpose<C, S, LL::LSTR>(v);
        fft_forward_ws_from<C, LL, S + 1, TWFULL, DEFER, XL>(v, t, tw, smem, off, xi);
    } else {
        stage<C, S, +1, 0, TWFULL, false, LL, DEFER>(v, t, tw, smem, off, xi);
        exchange_read<C, S + 1, LL, !DEFER>(v, t, smem, off, xi);
        fft_forward_ws_from<C, LL, S + 1, TWFULL, DEFER, XL>(v, t, tw, smem, off, xi);
    }
}
template <class C, class LL, bool PRUNE_IN, bool TWFULL, bool DEFER = false, bool XL = false, class F>
__device__ __forceinline__ void fft_forward_ws(cplx (&v)[C::E], int t, const cplx* __restrict__ tw, char* smem, int off, int xi, F&& hook) {
    static_assert(!XL || (xlane_ok<C, LL>() && !DEFER), "fft_forward_ws: lane exchange not available for this line / layout");
    stage<C, 0, +1, PRUNE_IN ? 1 : 0, TWFULL, false, LL, DEFER>(v, t, tw, smem, off, xi);
    hook();
    exchange_read<C, 1, LL, !DEFER>(v, t, smem, off, xi);
    fft_forward_ws_from<C, LL, 1, TWFULL, DEFER, XL>(v, t, tw, smem, off, xi);
}
// stage S (>= 1) backwards and the exchange S -> S - 1 behind it, down to the slots of stage 0
template <class C, class LL, int S, bool TWFULL, bool DEFER, bool XL>
__device__ __forceinline__ void fft_inverse_ws_from(cplx (&v)[C::E], int t, const cplx* __restrict__ tw, char* smem, int off, int xi) {
    if constexpr (XL && xlane_stage_ok<C, S - 1, LL>()) {
        stage<C, S, -1, 0, TWFULL>(v, t, tw);
        xlane_transpose<C, S - 1, LL::LSTR>(v);
    } else {
        // (the caller has synchronised after its own use of the buffer: the first storing stage stores at once)
        stage<C, S, -1, 0, TWFULL, false, LL, (DEFER && S < C::NS - 1)>(v, t, tw, smem, off, xi);
        exchange_read<C, S - 1, LL, !DEFER>(v, t, smem, off, xi);
    }
    if constexpr (S > 1) fft_inverse_ws_from<C, LL, S - 1, TWFULL, DEFER, XL>(v, t, tw, smem, off, xi);
}
template <class C, class LL, bool PRUNE_OUT, bool TWFULL, bool DEFER = false, bool XL = false>
__device__ __forceinline__ void fft_inverse_ws(cplx (&v)[C::E], int t, const cplx* __restrict__ tw, char* smem, int off, int xi) {
    static_assert(!XL || (xlane_ok<C, LL>() && !DEFER), "fft_inverse_ws: lane exchange not available for this line / layout");
    fft_inverse_ws_from<C, LL, C::NS - 1, TWFULL, DEFER, XL>(v, t, tw, smem, off, xi);
    stage<C, 0, -1, PRUNE_OUT ? 2 : 0, TWFULL>(v, t, tw);
}
#endif // !LSFC_FFT_HOST_EMULATION

// Work items of the ticketed fused pass (fft_kernels.hip: k_zfused_persist): w = (ticket c << 3) | queue q, q = 0..7 (one
// queue per XCD), c = 0 .. nwork - 1 with nwork = ntiles / 8 whole tiles or ntiles / 4 half tiles per queue (ntiles a
// multiple of 16).  Whole tiles: ticket c of queue q is tile 2 (q + 8 (c >> 1)) + (c & 1) -- two consecutive tickets are
// two tiles next to each other in block order, i.e. (y-even symbol) a row and its mirror row, which read the same symbol
// rows.  Half tiles: ticket c is half c & 1 of tile 2 (q + 8 (c >> 2)) + ((c >> 1) & 1) -- four consecutive tickets are
// the halves of such a pair.  Every (tile, half) is handed out exactly once over the eight queues (tests/emu).
template <bool HALF> __device__ __forceinline__ void ticket_decode(unsigned w, unsigned& tile, unsigned& half) {
    if constexpr (HALF) { tile = 2u * ((w & 7u) + 8u * (w >> 5)) + ((w >> 4) & 1u); half = (w >> 3) & 1u; }
    else { tile = 2u * ((w & 7u) + 8u * (w >> 4)) + ((w >> 3) & 1u); half = 0u; }
}

// Pairs (round 3): a work item is a PAIR of whole tiles, 2 p and 2 p + 1 with p = q + 8 c -- a row and its mirror row, processed one
// after the other by the SAME workgroup, which keeps the symbol values they share in registers (ntiles / 16 pairs per queue)
__device__ __forceinline__ unsigned ticket_decode_pair(unsigned w, unsigned sub) { return 2u * ((w & 7u) + 8u * (w >> 3)) + sub; }

// Host mirror of the slot bookkeeping: frequency index held at storage index s.
// After the last stage slot e = u + NB*q of thread t sits at position (t + T*u)*RL + q; in-place DIF leaves
// frequency k = k0 + R0*(k1 + R1*(k2 + R2*k3)) at position k0*M0 + k1*M1 + k2*M2 + k3 (M_s = L / (R0..R_s)).
template <class C> inline void perm_table(int* freq_of_storage) {
    const int rad[4] = { C::R0, C::R1, C::R2, C::R3 };
    const int RL = rad[C::NS - 1];
    const int NB = C::E / RL;
    for (int t = 0; t < C::T; ++t)
        for (int e = 0; e < C::E; ++e) {
            const int u = e % NB, q = e / NB;
            int rem = (t + C::T * u) * RL + q;
            int k = 0, weight = 1, M = C::L;
            for (int s = 0; s < C::NS; ++s) {
                M /= rad[s];
                const int ks = rem / M; rem %= M;
                k += ks * weight; weight *= rad[s];
            }
            freq_of_storage[t + C::T * e] = k;
        }
}

// Host mirror: fill the full stage-twiddle table of a factorisation from tw[j] = exp(-2 pi i j / L).
template <class C, int S> inline void twfull_stage(cplx* out, const cplx* tw) {
    constexpr int LS = C::template LS<S>(), R = C::template R<S>();
    constexpr int M = LS / R;
    if (M > 1)
        for (int q = 1; q < R; ++q)
            for (int r = 0; r < M; ++r)
                out[C::template TWOFF<S>() + (q - 1) * M + r] = tw[((long long)r * q * (C::L / LS)) % C::L];
}
template <class C> inline void twfull_table(cplx* out, const cplx* tw) {
    twfull_stage<C, 0>(out, tw);
    twfull_stage<C, 1>(out, tw);
    if (C::NS >= 3) twfull_stage<C, 2>(out, tw);
    if (C::NS >= 4) twfull_stage<C, 3>(out, tw);
}

}} // namespace lsfc::fft
