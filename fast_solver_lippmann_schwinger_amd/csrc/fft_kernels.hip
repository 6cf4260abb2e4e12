// Pruned, fused axis passes of the padded-grid convolution (gfx950).
//
//   y = alpha*x + beta * crop( IFFT( sym .* FFT( pad( nu .* x ) ) ) )
//
// on the reduced grid Lx x Ly x Lz = 2n x 2m x 2l, never touching the zero 7/8 of
// the padded cube (35 complex + 1 real of HBM traffic per point):
//
//   xfwd   x[n][m][l] (*nu)            -> A1[Lx][m][l]            lines contiguous
//   yfwd   A1                          -> A2[8][l][Ly][Lx/8]      128-B chunks in, 128-B chunks out
//   zfused A2 (in place) .* sym        -> A2                      fully contiguous 8-line tiles
//   yinv   A2                          -> A1[Lx][m][l]
//   xinv   A1, x                       -> y[n][m][l]
//
// (2D: xfwd -> zfused along y on the natural A1[Lx][m] -> xinv.)
// The grid sizes n, m, l need not be powers of two: every pass reads only the first n (m, l) entries of a line
// and zero-fills up to L/2, writes only the first n (m, l) on the way back; L = pruned_best_length(n) per axis.
// Frequency-side indices are "storage" indices (fft_core.hpp); A2 is tiled so that
// the eight x'-neighbours of a z-line are interleaved (xi fastest), which makes
// the largest pass (z: 16 of the 35 complex per point) a pure stream.
#include "common.hpp"
#include "fft_configs.hpp"
#include "pruned.hpp"
#include <cstdlib>
#include <mutex>
#include <set>
#include <map>
#include <utility>

// This file is compiled once per family of line lengths (Makefile: -DLSFC_FAMILY=2 | 3 | 5): the power-of-two lines,
// the lines with one factor 3 and the lines with one factor 5.  pruned.hip routes each call to the family of its L.
#ifndef LSFC_FAMILY
#error "compile with -DLSFC_FAMILY=2, 3 or 5"
#endif
#define LSFC_CAT2(a, b) a##b
#define LSFC_CAT(a, b) LSFC_CAT2(a, b)
#define FAM(name) LSFC_CAT(LSFC_CAT(name, _f), LSFC_FAMILY)

namespace lsfc {
using namespace fft;

static constexpr int XB = 8;           // x' lines per strided workgroup: 8 * 16 B = one 128-B line

// EXACT: n == L/2 (the grid fills the line): the end-of-line predicates compile away
template <class C, int LPW, bool SPLIT, bool EXACT>
__global__ __launch_bounds__(C::T * LPW)
void k_xfwd(const VecBatch vb, int64_t obatch, const double* __restrict__ nu, cplx* __restrict__ out,
            const cplx* __restrict__ tw, int64_t nlines, int wmagic, int W, int Wp, int n, int64_t bstride) {
    // blockIdx.y = right-hand side of the batch
    const cplx* __restrict__ x = vb.x[blockIdx.y];
    out += (int64_t)blockIdx.y * obatch;
    using LL = LdsLayout<1, 3, SPLIT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = C::T, E = C::E;
    const int t = threadIdx.x % T, ll = threadIdx.x / T;
    const int64_t line = (int64_t)blockIdx.x * LPW + ll;
    const bool valid = line < nlines;
    const int64_t lc = valid ? line : nlines - 1;
    const cplx* xin = x + lc * n;
    cplx v[E];
#pragma unroll
    for (int e = 0; e < E / 2; ++e) v[e] = (EXACT || t + T * e < n) ? xin[t + T * e] : make_double2(0.0, 0.0);
    if (nu) {
        const double* nin = nu + lc * n;
#pragma unroll
        for (int e = 0; e < E / 2; ++e) { const double s = (EXACT || t + T * e < n) ? nin[t + T * e] : 0.0; v[e].x *= s; v[e].y *= s; }
    }
#pragma unroll
    for (int e = E / 2; e < E; ++e) v[e] = make_double2(0.0, 0.0);
    fft_forward<C, LL, true>(v, t, tw, smem, ll * LL::line_elems(C::L), 0);
    if (valid) {
        // storage index s of this line goes to chunk s / W (one chunk per destination rank and pipeline stage of the
        // slab transpose; a single chunk of width L on one GPU): out[chunk][line][s % W], row pitch Wp >= W.  The
        // division is exact by multiplication: wmagic = floor(2^22 / W) + 1, valid for s < 2048.
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int s = t + T * e;
            const int c = (int)(((unsigned)s * (unsigned)wmagic) >> 22);
            out[(int64_t)c * bstride + line * Wp + (s - c * W)] = v[e];
        }
    }
}

template <class C, int LPW, bool SPLIT, bool EXACT>
__global__ __launch_bounds__(C::T * LPW)
void k_xinv(const cplx* __restrict__ in, const VecBatch vb, int64_t ibatch, double alpha, double beta,
            const cplx* __restrict__ tw, int64_t nlines, int wmagic, int W, int Wp, int n, int64_t bstride) {
    const cplx* xorig = vb.x[blockIdx.y];
    cplx* y = vb.y[blockIdx.y];
    in += (int64_t)blockIdx.y * ibatch;
    using LL = LdsLayout<1, 3, SPLIT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = C::T, E = C::E;
    const int t = threadIdx.x % T, ll = threadIdx.x / T;
    const int64_t line = (int64_t)blockIdx.x * LPW + ll;
    const bool valid = line < nlines;
    const int64_t lc = valid ? line : nlines - 1;
    cplx v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int s = t + T * e;
        const int c = (int)(((unsigned)s * (unsigned)wmagic) >> 22);
        v[e] = in[(int64_t)c * bstride + lc * Wp + (s - c * W)];
    }
    fft_inverse<C, LL, true>(v, t, tw, smem, ll * LL::line_elems(C::L), 0);
    if (valid) {
#pragma unroll
        for (int e = 0; e < E / 2; ++e) {
            if (EXACT || t + T * e < n) {
                const int64_t idx = line * n + t + T * e;
                cplx r = make_double2(beta * v[e].x, beta * v[e].y);
                if (alpha != 0.0) { const cplx xo = xorig[idx]; r.x = fma(alpha, xo.x, r.x); r.y = fma(alpha, xo.y, r.y); }
                y[idx] = r;
            }
        }
    }
}

// (two 512-thread workgroups of the y passes per CU -- 128 registers, split exchanges, chained twiddle powers, no spills -- change
// nothing at 512^3: yfwd 2.53 -> 2.51 ms, profiles/r02_experiment_ticketed_half_tiles.log; the passes are bound by their access pattern)
// A1[Lx][m][l] (natural) -> A2[XB][l][Ly][Lx/XB]
template <class C, int LINES, bool SPLIT, int WPE, bool EXACT>
__global__ __launch_bounds__(C::T * LINES, WPE)
void k_yfwd(const cplx* __restrict__ a1, cplx* __restrict__ a2, const cplx* __restrict__ tw, int Lx, int m, int l, int TG, int TZ, int p1, int p2,
            int64_t batch1, int64_t batch2) {
    using LL = LdsLayout<LINES, 3, SPLIT>;
    a1 += (int64_t)blockIdx.y * batch1; a2 += (int64_t)blockIdx.y * batch2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = C::T, E = C::E, Ly = C::L;
    const int xi = threadIdx.x % LINES, t = threadIdx.x / LINES;
    const int ngrp = Lx / LINES;
    // block order: (TG x TZ) tiles of (x'-group, z), groups fastest inside a tile and across tiles, so that the
    // workgroups in flight together touch short contiguous runs in BOTH the natural and the tiled array
    const int within = blockIdx.x % (TG * TZ), tile = blockIdx.x / (TG * TZ);
    const int ntg = ngrp / TG;
    const int g = (tile % ntg) * TG + within % TG, z = (tile / ntg) * TZ + within / TG;
    const int xp = g * LINES + xi;                       // x' storage index
    const cplx* src = a1 + xp + (int64_t)p1 * m * z;
    cplx v[E];
#pragma unroll
    for (int e = 0; e < E / 2; ++e) v[e] = (EXACT || t + T * e < m) ? src[(int64_t)p1 * (t + T * e)] : make_double2(0.0, 0.0);
#pragma unroll
    for (int e = E / 2; e < E; ++e) v[e] = make_double2(0.0, 0.0);
    fft_forward<C, LL, true, false, (C::T * LINES > 512)>(v, t, tw, smem, 0, xi);     // (1024-thread workgroups, 128 registers: chained twiddle powers, no spills)
    const int xb = xp / XB, xq = xp % XB;
    cplx* dst = a2 + xq + (int64_t)XB * z + (int64_t)p2 * ((int64_t)Ly * xb);
#pragma unroll
    for (int e = 0; e < E; ++e) dst[(int64_t)p2 * (t + T * e)] = v[e];
}

template <class C, int LINES, bool SPLIT, int WPE, bool EXACT>
__global__ __launch_bounds__(C::T * LINES, WPE)
void k_yinv(const cplx* __restrict__ a2, cplx* __restrict__ a1, const cplx* __restrict__ tw, int Lx, int m, int l, int TG, int TZ, int p1, int p2,
            int64_t batch1, int64_t batch2) {
    using LL = LdsLayout<LINES, 3, SPLIT>;
    a1 += (int64_t)blockIdx.y * batch1; a2 += (int64_t)blockIdx.y * batch2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = C::T, E = C::E, Ly = C::L;
    const int xi = threadIdx.x % LINES, t = threadIdx.x / LINES;
    const int ngrp = Lx / LINES;
    // block order: (TG x TZ) tiles of (x'-group, z), groups fastest inside a tile and across tiles, so that the
    // workgroups in flight together touch short contiguous runs in BOTH the natural and the tiled array
    const int within = blockIdx.x % (TG * TZ), tile = blockIdx.x / (TG * TZ);
    const int ntg = ngrp / TG;
    const int g = (tile % ntg) * TG + within % TG, z = (tile / ntg) * TZ + within / TG;
    const int xp = g * LINES + xi;
    const int xb = xp / XB, xq = xp % XB;
    const cplx* src = a2 + xq + (int64_t)XB * z + (int64_t)p2 * ((int64_t)Ly * xb);
    cplx v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = src[(int64_t)p2 * (t + T * e)];
    fft_inverse<C, LL, true, false, (C::T * LINES > 512)>(v, t, tw, smem, 0, xi);
    cplx* dst = a1 + xp + (int64_t)p1 * m * z;
#pragma unroll
    for (int e = 0; e < E / 2; ++e) if (EXACT || t + T * e < m) dst[(int64_t)p1 * (t + T * e)] = v[e];
}

// In-place forward -> .* sym -> inverse along one strided axis.
// Line (g, outer, xi): element j at data[g*dGrp + outer*dOuter + xi + dLine*j],
// symbol entry for storage index s at sym[g*sGrp + outer*sOuter + xi + sLine*s].
// PREFETCH: the symbol loads are issued before the forward transform, so their HBM latency hides behind its
// butterflies (+E complex registers; pays at E = 16 where the kernel runs at 2 waves/SIMD either way).
// Tried and dropped (profiles/r01_experiment_*.log): LDS-only exchange barriers, non-temporal accesses, a persistent
// software-pipelined form of all five kernels, 128-register builds (two 512-thread workgroups per CU), an
// 8-elements-per-thread four-stage factorisation of the 1024-point line, and Infinity-Cache slab blocking of the
// y/z/y passes -- each measured slower or neutral on MI355X.
// HALF: the workgroup transforms only 4 of the 8 interleaved lines of a tile (64 B of every 128-B line); the two
// halves of a tile are blocks b and b+8 of a group of 16, i.e. (by the observed round-robin placement) on the SAME
// XCD, whose L2 merges their reads and writes of the shared lines.  Halving the workgroup to 4 waves lets two (or
// three) independent workgroups share a CU at the same register budget, so one computes while the other waits on HBM.
// ZE (z-even symbol): a symbol line stores only the entries with frequency kz <= L/2 -- L/2 of them in storage order
// (exactly the slots e < E/2 of every thread) plus the kz = L/2 entry at index L/2.  Each thread loads its E/2
// values; the mirror values (kz -> L - kz) of its slots e >= E/2 are held by other threads of the same line and are
// fetched through LDS (zm[s - L/2] = storage index of the partner).  Halves the symbol bytes of the pass again.
// TWL: the full stage-twiddle table (tw points to it) is staged in LDS behind the exchange buffer and read instead of
// computing the power trees (-18 % fp64 instructions, +40 % LDS reads).
// MULTI (several right-hand sides, lsfc_apply_batch): the workgroup runs the tile of every batch member in turn with
// ONE load of its symbol tile -- the symbol's share of the pass (8 of 16 complex per point in the byte model, 4 with
// the half symbol) is paid once per batch instead of once per right-hand side.
template <class C, int LINES, bool SPLIT, bool PREFETCH, int WPE, bool HALF, bool ZE, bool EXACT, bool TWL, bool MULTI = false>
__global__ __launch_bounds__(C::T * LINES, WPE)
void k_zfused(cplx* __restrict__ data, const cplx* __restrict__ sym, const cplx* __restrict__ tw,
              int nouter, int64_t dGrp, int64_t dOuter, int64_t dLine, int64_t sGrp, int64_t sOuter, int64_t sLine,
              const int2* __restrict__ ytab, const int* __restrict__ zm, int nin, int nrhs, int64_t dBatch, int sibling_pairs) {
    using LL = LdsLayout<LINES, 3, SPLIT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = C::T, E = C::E;
    int xi = threadIdx.x % LINES; const int t = threadIdx.x / LINES;
    unsigned tile = blockIdx.x;
    if constexpr (HALF) {
        // groups of 8 tiles x (XB / LINES) parts: the parts of a tile are 8 blocks apart, i.e. on one XCD
        constexpr int PARTS = XB / LINES, SH = PARTS == 4 ? 5 : 4;
        const unsigned b = blockIdx.x;
        tile = (b >> SH) * 8 + (b & 7);
        xi += (int)((b >> 3) & (PARTS - 1)) * LINES;
    } else if (sibling_pairs) {
        // 2D layout, 4-line sub-groups: groups 2p and 2p + 1 are the two halves of the same 128-B lines.  Blocks b and b + 8
        // (same XCD under round-robin placement) take them, so one L2 fetches each line once instead of two L2s once each.
        const unsigned b = blockIdx.x;
        tile = (((b >> 4) * 8 + (b & 7)) << 1) | ((b >> 3) & 1);
    }
    // block order -> (data row, symbol row): with a y-even symbol a row and its mirror share one symbol row and sit
    // next to each other in block order, so the second read of that row is served by the Infinity Cache, not HBM
    const int o = tile % nouter, g = tile / nouter;
    int outer = o, srow = o;
    if (ytab) { const int2 e = ytab[o]; outer = e.x; srow = e.y; }
    cplx* d = data + g * dGrp + outer * dOuter + xi;
    const cplx* s = sym + g * sGrp + srow * sOuter + xi;
    const int li = threadIdx.x % LINES;                 // line slot inside this workgroup's LDS
    if constexpr (TWL) {
        cplx* tl = reinterpret_cast<cplx*>(smem + (size_t)LL::line_elems(C::L) * LINES * LL::elem_bytes());
        for (int i = threadIdx.x; i < C::TWLEN; i += C::T * LINES) tl[i] = tw[i];
        __syncthreads();
        tw = tl;
    }
    // (whole-complex layouts: exchange stores issued from inside the stages, fft_core.hpp)
    // ... and the exchanges between stages of equal radix through the lanes of the wavefront where the line and the layout allow it
    // (fft_core.hpp xlane_stage_ok; whole wavefronts only): one LDS round trip and two workgroup barriers less per direction in a
    // kernel that is a chain of latencies on the small grids and in 2D
#ifdef LSFC_NO_XLANE_ONE_TILE
    constexpr bool XL1 = false;
#else
    constexpr bool XL1 = !SPLIT && xlane_ok<C, LL>() && (C::T * LINES) % 64 == 0;
#endif
    auto fwd = [&](cplx (&w)[E]) {
        if constexpr (!SPLIT) fft_forward_ws<C, LL, true, TWL, false, XL1>(w, t, tw, smem, 0, li, [] {});
        else fft_forward<C, LL, true, TWL>(w, t, tw, smem, 0, li);
    };
    auto inv = [&](cplx (&w)[E]) {
        if constexpr (!SPLIT) fft_inverse_ws<C, LL, true, TWL, false, XL1>(w, t, tw, smem, 0, li);
        else fft_inverse<C, LL, true, TWL>(w, t, tw, smem, 0, li);
    };
    cplx v[E];
    if constexpr (MULTI) {
        // symbol tile once, then every right-hand side of the batch
        constexpr int H = ZE ? E / 2 : E;
        cplx sv[H];
#pragma unroll
        for (int e = 0; e < H; ++e) sv[e] = s[sLine * (t + T * e)];
        int part[ZE ? H : 1];
        cplx smid = make_double2(0.0, 0.0);
        if constexpr (ZE) {
#pragma unroll
            for (int e = 0; e < H; ++e) part[e] = zm[t + T * e];
            if (t == 0) smid = s[sLine * (C::L / 2)];
        }
        for (int r = 0; r < nrhs; ++r) {
            cplx* dr = d + (int64_t)r * dBatch;
#pragma unroll
            for (int e = 0; e < E / 2; ++e) v[e] = (EXACT || t + T * e < nin) ? dr[dLine * (t + T * e)] : make_double2(0.0, 0.0);
#pragma unroll
            for (int e = E / 2; e < E; ++e) v[e] = make_double2(0.0, 0.0);
            fwd(v);
            if constexpr (ZE) {
                cplx* stage = reinterpret_cast<cplx*>(smem);
                if constexpr (forward_ends_local<C, LL>()) LSFC_BARRIER();   // other waves may still read the exchange buffer
#pragma unroll
                for (int e = 0; e < H; ++e) stage[(t + T * e) * LINES + li] = sv[e];
                if (t == 0) stage[(C::L / 2) * LINES + li] = smid;
                LSFC_BARRIER();
#pragma unroll
                for (int e = 0; e < H; ++e) {
                    v[e] = cmul(v[e], sv[e]);
                    v[e + H] = cmul(v[e + H], stage[part[e] * LINES + li]);
                }
                LSFC_BARRIER();
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = cmul(v[e], sv[e]);
            }
            inv(v);
#pragma unroll
            for (int e = 0; e < E / 2; ++e) if (EXACT || t + T * e < nin) dr[dLine * (t + T * e)] = v[e];
            // (every exchange ends with a barrier after its reads: the buffer is free for the next member)
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < E / 2; ++e) v[e] = (EXACT || t + T * e < nin) ? d[dLine * (t + T * e)] : make_double2(0.0, 0.0);
#pragma unroll
    for (int e = E / 2; e < E; ++e) v[e] = make_double2(0.0, 0.0);
    if constexpr (ZE) {
        constexpr int H = E / 2;
        int part[H];
#pragma unroll
        for (int e = 0; e < H; ++e) part[e] = zm[t + T * e];
        cplx sv[H];
        if constexpr (PREFETCH) {
#pragma unroll
            for (int e = 0; e < H; ++e) sv[e] = s[sLine * (t + T * e)];
        }
        fwd(v);
        if constexpr (!PREFETCH) {
#pragma unroll
            for (int e = 0; e < H; ++e) sv[e] = s[sLine * (t + T * e)];
        }
        // stage this line's stored half in LDS (the exchange buffer is free between the two transforms)
        cplx* stage = reinterpret_cast<cplx*>(smem);
        if constexpr (forward_ends_local<C, LL>()) LSFC_BARRIER();   // other waves may still read the exchange buffer
#pragma unroll
        for (int e = 0; e < H; ++e) stage[(t + T * e) * LINES + li] = sv[e];
        if (t == 0) stage[(C::L / 2) * LINES + li] = s[sLine * (C::L / 2)];
        LSFC_BARRIER();
#pragma unroll
        for (int e = 0; e < H; ++e) {
            v[e] = cmul(v[e], sv[e]);
            v[e + H] = cmul(v[e + H], stage[part[e] * LINES + li]);
        }
        LSFC_BARRIER();
    } else if constexpr (PREFETCH) {
        // issue the symbol loads before the forward transform: their HBM latency hides behind its butterflies
        cplx sv[E];
#pragma unroll
        for (int e = 0; e < E; ++e) sv[e] = s[sLine * (t + T * e)];
        fwd(v);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = cmul(v[e], sv[e]);
    } else {
        fwd(v);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = cmul(v[e], s[sLine * (t + T * e)]);
    }
    inv(v);
#pragma unroll
    for (int e = 0; e < E / 2; ++e) if (EXACT || t + T * e < nin) d[dLine * (t + T * e)] = v[e];
}

// the XCD this wave runs on (HW_REG_XCC_ID, bits 3:0)
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v & 7u;
}
// opaque copy of a value: the optimiser may not carry anything derived from the original across this point (used to keep
// loop-invariant address arithmetic from being hoisted into registers that then live across the whole tile loop)
template <class V> __device__ __forceinline__ V launder_v(V x) { asm volatile("" : "+v"(x)); return x; }
template <class P> __device__ __forceinline__ P launder_s(P x) { asm volatile("" : "+s"(x)); return x; }

// Persistent, software-pipelined form of the fused pass (z-even symbol, 8-line tiles of the 3D layout).
// Why: with one workgroup per CU (the exchange buffer of a 1024-point tile fills the LDS) the plain kernel runs its
// memory phases and its compute phases one after the other -- SQ counters at 512^3: waves parked 36 % of their life,
// VALU active 21 %, and a CU cannot pull its 161 KB per tile faster than ~24 GB/s (6.7 us of the 11.2 us per tile).
// Here one workgroup per CU walks over tiles b, b + G, b + 2G, ... and keeps the NEXT tile's data loads in flight under
// the inverse transform of the current tile (they re-use the registers of the symbol values, which are dead after the
// multiply), the symbol loads in flight under the forward transform as before, and the stores of the previous tile
// drain under the next forward transform.  The stage-twiddle table and the mirror-slot table are set up once per
// workgroup instead of once per tile.  Register pressure is that of the plain PREFETCH variant.
// HALF: 4-line workgroups on half tiles (as k_zfused's HALF), TWO per CU: the exchange buffer of a whole 1024-point tile leaves
// room for one workgroup only, whose eight waves then march in lock step through VALU phases and LDS phases that never
// overlap (SQ counters at 512^3: 2.1 ms of VALU issue per SIMD + 2.2 ms of LDS array time inside a 5.3 ms pass).  Two
// independent half-tile workgroups drift apart and fill each other's phases.  The two halves of a tile (64 B each of every
// 128-B line) must pass through the same L2 at about the same time or every line is fetched twice; a static walk does
// not hold that over hundreds of tiles (measured: 29 GB read instead of 17), so the work is handed out per XCD: tickets
// c = 0, 1, 2, ... from tickets[xcd] (one atomic per workgroup and tile, fetched a whole forward transform before it is
// needed) mean the halves of tiles xcd-th pair of each group of 8 pairs (see `locate`), so the halves of a tile -- and the
// tile of the mirror row, which reads the same symbol rows -- go to the next workgroups of that XCD that become free.  A workgroup whose queue is exhausted moves on to the next queue (xcd + 1, ...), so every ticket of
// every queue is drawn whatever the placement of the workgroups (and the tail balances itself); it ends when all eight
// are exhausted.  There is no waiting on other workgroups anywhere.
// the exchange buffer, which between the transforms also stages the symbol's (L / 2 + 1) x LINES whole-complex mirror values
template <class C, int LINES, bool SPLIT, bool HALF> constexpr size_t persist_xbuf_bytes() {
    using LL = LdsLayout<LINES, HALF ? -1 : 3, SPLIT>;
    constexpr size_t ex = (size_t)LL::line_elems(C::L) * LINES * LL::elem_bytes(), stg = (size_t)(C::L / 2 + 1) * LINES * sizeof(cplx);
    return ex > stg ? ex : stg;
}
template <class C, int LINES, bool SPLIT, bool TWL, bool HALF> constexpr size_t persist_lds_bytes() {
    return persist_xbuf_bytes<C, LINES, SPLIT, HALF>() + (TWL ? (size_t)C::TWLEN * sizeof(cplx) : 0);
}
// TICKETS without HALF: whole tiles handed out the same way, in pairs (row, mirror row) per XCD queue.
// XL (round 3) is a set of bits, chosen by the run-time knob "xlane" (0, 1, 3, 5; -1 = 5 where available, else 1):
// XL & 1: the exchange between the two radix-8 stages of the line runs through the lanes of the wavefront (fft_core.hpp:
// xlane_transpose8) instead of LDS -- one LDS exchange and two workgroup barriers less per direction (-0.28 ms at 512^3).
// XL & 4: work items are row PAIRS whose shared symbol values stay in registers (see PAIR below; -0.05 ms).
// XL & 2 (measured slower, kept for A/B runs): in addition the mirror values of the z-even symbol (slots e >= E/2: the entry of frequency L - kz) are loaded by the
// thread that needs them, a second read of lines this workgroup fetches anyway (L2 hits), instead of being staged through
// LDS by the threads that hold them: 8 ds_write_b128 + 8 ds_read_b128 per thread and one of the two barriers around them go.
template <class C, int LINES, bool SPLIT, bool EXACT, bool TWL, bool LATE_SYM = false, bool HALF = false, bool TICKETS = HALF, int XL = 0>
__global__ __launch_bounds__(C::T * LINES, (HALF && 2 * persist_lds_bytes<C, LINES, SPLIT, TWL, HALF>() <= (size_t)160 * 1024) ? 2 : 1)
void k_zfused_persist(cplx* __restrict__ data, const cplx* __restrict__ sym, const cplx* __restrict__ tw,
                      int nouter, int64_t dGrp, int64_t dOuter, int64_t dLine, int64_t sGrp, int64_t sOuter, int64_t sLine,
                      const int2* __restrict__ ytab, const int* __restrict__ zm, int nin, unsigned ntiles, unsigned* __restrict__ tickets) {
    using LL = LdsLayout<LINES, HALF ? -1 : 3, SPLIT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = C::T, E = C::E, H = E / 2;
#ifdef LSFC_NO_STAGE_STORES
    constexpr bool WS = false;
#else
    constexpr bool WS = true;                           // exchange stores from inside the stages (fft_core.hpp: stage<..., LLW>)
#endif
    // ... with the buffer-freeing barriers taken inside the next stage, before its first store (BARF), instead of right after the
    // loads: measured SLOWER (same-box A/B 4.88 -> 5.06 ms: the barrier then separates a wave's butterflies from its own stores);
    // off, kept for A/B builds (make EXTRA=-DLSFC_DEFERRED_BARRIERS)
#ifdef LSFC_DEFERRED_BARRIERS
    constexpr bool DEFER = true;
#else
    constexpr bool DEFER = false;
#endif
    const int li = threadIdx.x % LINES, t = threadIdx.x / LINES;
    // XL & 4 (PAIR; ticketed whole tiles): a work item is a row pair -- the tile of a row and of its mirror row, which multiply by the
    // same symbol values -- and the workgroup runs the two tiles back to back with ONE load of those values: they stay in registers
    // across the first tile's inverse transform and the second tile's forward transform (8 of the 24 loads of every second tile go)
    constexpr bool PAIR = (XL & 4) != 0 && TICKETS && !HALF;
    // work item: a tile (static walk b, b + G, ...) or, HALF, a ticket of this workgroup's XCD
    // (HALF: a work item is (ticket << 3 | queue); DONE = nothing left anywhere)
    constexpr unsigned DONE = 0xFFFFFFFFu;
    unsigned cur = blockIdx.x, nwork = ntiles;
    unsigned* slot = nullptr;                           // HALF: where thread 0 publishes the work item it drew
    unsigned queue = 0, queues_left = 8;                // HALF, thread 0: the queue it draws from, queues not yet seen empty
    [[maybe_unused]] unsigned parity = 1u;              // XL = 2: which of the two ticket slots the next publication uses (slot[0]: the first draw)
    auto draw = [&]() -> unsigned {                     // (thread 0 only)
        while (queues_left > 0) {
            const unsigned c = atomicAdd(tickets + queue, 1u);
            if (c < nwork) return (c << 3) | queue;
            queue = (queue + 1) & 7u; --queues_left;
        }
        return DONE;
    };
    if constexpr (TICKETS) {
        queue = xcc_id();
        nwork = HALF ? ntiles / 4 : (PAIR ? ntiles / 16 : ntiles / 8);   // (ntiles / 8 tiles per queue) [x 2 halves | as pairs]; the host checks ntiles % 16 == 0
        slot = reinterpret_cast<unsigned*>(smem + persist_lds_bytes<C, LINES, SPLIT, TWL, HALF>());
        if (threadIdx.x == 0) *slot = draw();
        __syncthreads();
        cur = __builtin_amdgcn_readfirstlane(*slot);
        __syncthreads();
        if (cur == DONE) return;
    } else if (cur >= nwork) return;                    // (uniform per workgroup; no inter-workgroup synchronisation anywhere)
    if constexpr (TWL) {
        cplx* tl = reinterpret_cast<cplx*>(smem + persist_xbuf_bytes<C, LINES, SPLIT, HALF>());
        for (int i = threadIdx.x; i < C::TWLEN; i += C::T * LINES) tl[i] = tw[i];
        __syncthreads();
        tw = tl;
    }
    // Addressing: a tile's base addresses are uniform (scalar registers), the thread's part is one 32-bit offset --
    // 64-bit per-thread pointers kept across the loop cost the registers that decide between "fits" and "spills", and a
    // spill reload at the loop top would make the wave wait for the previous tile's stores (scratch and global memory
    // share the in-order vmcnt counter).
    auto locate = [&](unsigned w, cplx*& dbase, const cplx*& sbase, unsigned sub = 0) {
        unsigned tl_ = w; int half = 0;
        // (ticket -> tile, half: fft_core.hpp ticket_decode)
        if constexpr (PAIR) tl_ = ticket_decode_pair(w, sub);
        else if constexpr (TICKETS) { unsigned hf; ticket_decode<HALF>(w, tl_, hf); half = (int)hf * LINES; }
        const int o = (int)(tl_ % (unsigned)nouter), g = (int)(tl_ / (unsigned)nouter);
        int outer = o, srow = o;
        if (ytab) { const int2 e2 = ytab[o]; outer = e2.x; srow = e2.y; }
        dbase = data + g * dGrp + outer * dOuter + half;
        sbase = sym + g * sGrp + srow * sOuter + half;
    };
    // (unsigned: scalar base + zero-extended 32-bit lane offset is an addressing mode of the global instructions)
    const unsigned doff = (unsigned)(li + (int)dLine * t), dstep = (unsigned)((int)dLine * T);      // element j = t + T e of this thread's line
    const unsigned soff = (unsigned)(li + (int)sLine * t), sstep = (unsigned)((int)sLine * T);
    // (tried in round 3: the uniform part dstep * e of an element index moved into the scalar base, (base + dstep * e)[doff] -- one
    // lane offset instead of a per-lane add per element: neutral at 1024 points, slower on the 768-point line; not kept)
    constexpr bool MIRG = (XL & 2) != 0;
    [[maybe_unused]] unsigned moff[MIRG ? H : 1];      // MIRG: element index of the mirror entry of slot e + H in this thread's symbol line
    if constexpr (MIRG) {
#pragma unroll
        for (int e = 0; e < H; ++e) moff[e] = (unsigned)(li + (int)sLine * zm[t + T * e]);
    }
    cplx nd[H];                                        // data of the tile about to be transformed
    {
        cplx* d; const cplx* s;
        locate(cur, d, s);
#pragma unroll
        for (int e = 0; e < H; ++e) nd[e] = (EXACT || t + T * e < nin) ? d[doff + dstep * e] : make_double2(0.0, 0.0);
    }
    [[maybe_unused]] unsigned sub = 0;                  // PAIR: which tile of the pair `cur` is being processed
    [[maybe_unused]] const cplx* sheld = nullptr;       // PAIR: the symbol tile whose values sv_pair / smid_pair hold
    [[maybe_unused]] cplx sv_pair[PAIR ? H : 1];        // PAIR: the symbol values live across the two tiles of a pair (loop-carried registers)
    [[maybe_unused]] cplx smid_pair = make_double2(0.0, 0.0);
    for (;;) {
        // HALF: draw the next ticket now; it is published and read around the barriers of the symbol multiply below
        unsigned drawn = 0;
        if constexpr (TICKETS) { if (threadIdx.x == 0 && (!PAIR || sub == 0)) drawn = draw(); }
        cplx v[E];
#pragma unroll
        for (int e = 0; e < H; ++e) { v[e] = nd[e]; v[e + H] = make_double2(0.0, 0.0); }
        cplx sv_tile[PAIR ? 1 : H];
        cplx (&sv)[H] = *reinterpret_cast<cplx (*)[H]>(PAIR ? &sv_pair[0] : &sv_tile[0]);
        [[maybe_unused]] cplx smr[MIRG ? H : 1];
        cplx smid_tile = make_double2(0.0, 0.0);
        cplx& smid = PAIR ? smid_pair : smid_tile;
        auto load_symbol = [&] {
            cplx* d; const cplx* s;
            locate(cur, d, s, sub);
            if constexpr (PAIR) {
                if (s == sheld) return;                 // (uniform) the mirror row's tile: same symbol values, already in registers
                sheld = s;
            }
            const unsigned so = launder_v(soff);
#pragma unroll
            for (int e = 0; e < H; ++e) sv[e] = s[so + sstep * e];
            if constexpr (MIRG) {
#pragma unroll
                for (int e = 0; e < H; ++e) smr[e] = s[moff[e]];
            } else {
                if (t == 0) smid = s[so - (unsigned)((int)sLine * t) + (unsigned)((int)sLine * (C::L / 2))];
            }
        };
        if constexpr (!SPLIT && WS) {
            // exchange stores issued from inside the stages (fft_forward_ws); the symbol is loaded after the first stage
            if constexpr (!LATE_SYM) load_symbol();
            fft_forward_ws<C, LL, true, TWL, DEFER, (XL > 0)>(v, t, tw, smem, 0, li, [&] { if constexpr (LATE_SYM) load_symbol(); });
        } else if constexpr (LATE_SYM) {
            // the first forward stage (the widest butterfly plus its twiddles) runs before the symbol values occupy registers
            stage<C, 0, +1, 1, TWL>(v, t, tw);
            load_symbol();
            exchange<C, 0, 1, LL>(v, t, smem, 0, li);
            stage<C, 1, +1, 0, TWL>(v, t, tw);
            if constexpr (C::NS >= 3) { exchange<C, 1, 2, LL>(v, t, smem, 0, li); stage<C, 2, +1, 0, TWL>(v, t, tw); }
            if constexpr (C::NS >= 4) { exchange<C, 2, 3, LL>(v, t, smem, 0, li); stage<C, 3, +1, 0, TWL>(v, t, tw); }
        } else {
            load_symbol();
            fft_forward<C, LL, true, TWL>(v, t, tw, smem, 0, li);
        }
        cplx* stage = reinterpret_cast<cplx*>(smem);
        unsigned next = cur + gridDim.x;
        if constexpr (MIRG) {
            // no staging: every thread holds its own mirror values.  The ticket travels through a slot of its own per tile parity,
            // so ONE barrier publishes it (the slot of the other parity is rewritten only two barriers later)
#pragma unroll
            for (int e = 0; e < H; ++e) { v[e] = cmul(v[e], sv[e]); v[e + H] = cmul(v[e + H], smr[e]); }
            if constexpr (TICKETS) {
                if (threadIdx.x == 0) slot[parity] = drawn;
                LSFC_BARRIER();
                next = __builtin_amdgcn_readfirstlane(slot[parity]);
                parity ^= 1u;
            }
        } else {
        // other waves may still read the exchange buffer (wave-local last exchange, or the deferred barrier of fft_forward_ws)
        if constexpr (forward_ends_local<C, LL>() || (!SPLIT && WS && DEFER)) LSFC_BARRIER();
#pragma unroll
        for (int e = 0; e < H; ++e) stage[(t + T * e) * LINES + li] = sv[e];
        if (t == 0) stage[(C::L / 2) * LINES + li] = smid;
        if constexpr (TICKETS) { if (threadIdx.x == 0 && (!PAIR || sub == 0)) *slot = drawn; }
        LSFC_BARRIER();
        {
            const int* zmt = launder_s(zm) + t;            // (re-read per tile from L1: eight registers less across the loop)
#pragma unroll
            for (int e = 0; e < H; ++e) {
                v[e] = cmul(v[e], sv[e]);
                v[e + H] = cmul(v[e + H], stage[zmt[T * e] * LINES + li]);
            }
        }
        // next tile: its loads travel while this tile is transformed back
        if constexpr (TICKETS) next = __builtin_amdgcn_readfirstlane(*slot);
        LSFC_BARRIER();
        }
        // PAIR: after the first tile of a pair comes its second; the drawn ticket (in the slot since the first tile) is for afterwards
        const bool more = PAIR ? (sub == 0 || next != DONE) : (TICKETS ? next != DONE : next < nwork);
        if (more) {
            cplx* dn; const cplx* sn;
            if (PAIR && sub == 0) locate(cur, dn, sn, 1u); else locate(next, dn, sn, 0u);
            const unsigned dof = launder_v(doff);
#pragma unroll
            for (int e = 0; e < H; ++e) nd[e] = (EXACT || t + T * e < nin) ? dn[dof + dstep * e] : make_double2(0.0, 0.0);
        }
        if constexpr (!SPLIT && WS) fft_inverse_ws<C, LL, true, TWL, DEFER, (XL > 0)>(v, t, tw, smem, 0, li);
        else fft_inverse<C, LL, true, TWL>(v, t, tw, smem, 0, li);
        {
            cplx* d; const cplx* s;
            locate(cur, d, s, sub);
            const unsigned dof = launder_v(doff);
#pragma unroll
            for (int e = 0; e < H; ++e) if (EXACT || t + T * e < nin) d[dof + dstep * e] = v[e];
        }
        if (!more) break;
        if constexpr (PAIR) { if (sub == 0) sub = 1; else { sub = 0; cur = next; } }
        else cur = next;
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// opt a kernel in to > 48 KiB of dynamic LDS, once per kernel and device (not on every launch)
template <class K> static void allow_lds(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return;
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0; LSFC_HIP(hipGetDevice(&dev));
    const std::pair<const void*, int> key(reinterpret_cast<const void*>(kernel), dev);
    std::lock_guard<std::mutex> lock(mu);
    if (done.count(key)) return;
    LSFC_HIP(hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done.insert(key);
}

// developer switches read once per process (the launch paths run per apply)
static bool env_flag_no_sibling_pairs() { static const bool v = getenv("LSFC_NO_SIBLING_PAIRS") != nullptr; return v; }
static bool env_flag_z_quarter() { static const bool v = [] { const char* e = getenv("LSFC_Z_QUARTER"); return e && e[0] == '1'; }(); return v; }

template <class C> struct Tune {
    // lines per workgroup: contiguous passes use 256-thread workgroups; strided passes
    // interleave XB lines unless that would exceed 512 threads.
    static constexpr int LPW = (256 / C::T) > 0 ? (256 / C::T) : 1;
    static constexpr int LINES = (C::T * XB <= 512 || (C::E <= 8 && C::T * XB <= 1024)) ? XB : 512 / C::T;
    // the y passes (no symbol, no second transform) of the 2048-point line: whole 128-B chunks from 1024-thread workgroups
    // (16 waves, 128 registers each; split exchanges: 147 KB of LDS) instead of 64-B half chunks from 512 threads
#ifdef LSFC_Y2048_HALF
    static constexpr int YLINES = LINES;
#else
    static constexpr int YLINES = (C::L == 2048 && C::T * XB <= 1024) ? XB : LINES;
#endif
};

template <class C, bool SPLIT> static void xfwd_t(const VecBatch& vb, int nrhs, int64_t obatch, const double* nu, cplx* out, const cplx* tw, int64_t nlines, int wmagic, int W, int Wp, int n, int64_t bstride, hipStream_t st) {
    constexpr int LPW = Tune<C>::LPW;
    using LL = LdsLayout<1, 3, SPLIT>;
    const size_t lds = (size_t)LL::line_elems(C::L) * LPW * LL::elem_bytes();
    auto k = (n == C::L / 2) ? k_xfwd<C, LPW, SPLIT, true> : k_xfwd<C, LPW, SPLIT, false>;
    allow_lds(k, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)((nlines + LPW - 1) / LPW), (unsigned)nrhs), dim3(C::T * LPW), lds, st, vb, obatch, nu, out, tw, nlines, wmagic, W, Wp, n, bstride);
}
template <class C, bool SPLIT> static void xinv_t(const cplx* in, const VecBatch& vb, int nrhs, int64_t ibatch, double alpha, double beta, const cplx* tw, int64_t nlines, int wmagic, int W, int Wp, int n, int64_t bstride, hipStream_t st) {
    constexpr int LPW = Tune<C>::LPW;
    using LL = LdsLayout<1, 3, SPLIT>;
    const size_t lds = (size_t)LL::line_elems(C::L) * LPW * LL::elem_bytes();
    auto k = (n == C::L / 2) ? k_xinv<C, LPW, SPLIT, true> : k_xinv<C, LPW, SPLIT, false>;
    allow_lds(k, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)((nlines + LPW - 1) / LPW), (unsigned)nrhs), dim3(C::T * LPW), lds, st, in, vb, ibatch, alpha, beta, tw, nlines, wmagic, W, Wp, n, bstride);
}
static void ytile(const PrunedTuning& tn, int L, int ngrp, int l, int& TG, int& TZ) {
    // auto (0): all groups x 1 plane, except at L >= 1024 where 32 groups x 8 planes keeps the 128-B chunks that the
    // workgroups in flight touch together within a few DRAM pages / TLB entries of both arrays (yinv 3.3 -> 2.8 ms)
    const int ag = (L >= 1024) ? 32 : ngrp, az = (L >= 1024) ? 8 : 1;
    TG = tn.ytile_g > 0 ? tn.ytile_g : ag; if (TG > ngrp) TG = ngrp; while (ngrp % TG) --TG;
    TZ = tn.ytile_z > 0 ? tn.ytile_z : az; if (TZ > l) TZ = l;       while (l % TZ) --TZ;
}
template <class C, bool SPLIT, int WPE> static void yfwd_t(const PrunedTuning& tn, const cplx* a1, cplx* a2, const cplx* tw, int Lx, int m, int l, int p1, int p2, hipStream_t st, int nrhs, int64_t b1, int64_t b2) {
    constexpr int LINES = Tune<C>::YLINES;
    using LL = LdsLayout<LINES, 3, SPLIT>;
    const size_t lds = (size_t)LL::line_elems(C::L) * LINES * LL::elem_bytes();
    auto k = (m == C::L / 2) ? k_yfwd<C, LINES, SPLIT, WPE, true> : k_yfwd<C, LINES, SPLIT, WPE, false>;
    allow_lds(k, lds);
    int TG, TZ; ytile(tn, C::L, Lx / LINES, l, TG, TZ);
    hipLaunchKernelGGL(k, dim3((unsigned)((Lx / LINES) * l), (unsigned)nrhs), dim3(C::T * LINES), lds, st, a1, a2, tw, Lx, m, l, TG, TZ, p1, p2, b1, b2);
}
template <class C, bool SPLIT, int WPE> static void yinv_t(const PrunedTuning& tn, const cplx* a2, cplx* a1, const cplx* tw, int Lx, int m, int l, int p1, int p2, hipStream_t st, int nrhs, int64_t b1, int64_t b2) {
    constexpr int LINES = Tune<C>::YLINES;
    using LL = LdsLayout<LINES, 3, SPLIT>;
    const size_t lds = (size_t)LL::line_elems(C::L) * LINES * LL::elem_bytes();
    auto k = (m == C::L / 2) ? k_yinv<C, LINES, SPLIT, WPE, true> : k_yinv<C, LINES, SPLIT, WPE, false>;
    allow_lds(k, lds);
    int TG, TZ; ytile(tn, C::L, Lx / LINES, l, TG, TZ);
    hipLaunchKernelGGL(k, dim3((unsigned)((Lx / LINES) * l), (unsigned)nrhs), dim3(C::T * LINES), lds, st, a2, a1, tw, Lx, m, l, TG, TZ, p1, p2, b1, b2);
}
template <class C, bool SPLIT, bool PREFETCH, int WPE, bool ZE = false> static void zfused_t(cplx* data, const cplx* sym, const cplx* tw, const cplx* twl, int Lx, int nouter,
                                                    int64_t dTile, int64_t dOuter, int64_t dLine, int64_t sTile, int64_t sOuter, int64_t sLine,
                                                    const int2* ytab, const int* zm, int nin, hipStream_t st, int nrhs, int64_t dBatch) {
    // dTile/sTile are strides per XB-tile of x'; a workgroup covers LINES of the XB lines of a tile.
    // twl != NULL: full stage-twiddle table, staged in LDS by the kernel.
    constexpr int LINES = Tune<C>::LINES;
    static_assert(XB % LINES == 0, "LINES must divide XB");
    using LL = LdsLayout<LINES, 3, SPLIT>;
    size_t lds = (size_t)LL::line_elems(C::L) * LINES * LL::elem_bytes();
    auto k = (nin == C::L / 2) ? k_zfused<C, LINES, SPLIT, PREFETCH, WPE, false, ZE, true, false> : k_zfused<C, LINES, SPLIT, PREFETCH, WPE, false, ZE, false, false>;
    if (twl && lds + (size_t)C::TWLEN * sizeof(cplx) > (size_t)160 * 1024) twl = nullptr;   // table does not fit beside the exchange buffer
    if (twl) {
        k = (nin == C::L / 2) ? k_zfused<C, LINES, SPLIT, PREFETCH, WPE, false, ZE, true, true> : k_zfused<C, LINES, SPLIT, PREFETCH, WPE, false, ZE, false, true>;
        lds += (size_t)C::TWLEN * sizeof(cplx);
        tw = twl;
    }
    if (nrhs > 1) {
        // batch: one symbol load per tile for all right-hand sides (the symbol is always loaded up front there, so the
        // PREFETCH flavours share one instantiation)
        if (twl) k = (nin == C::L / 2) ? k_zfused<C, LINES, SPLIT, false, WPE, false, ZE, true, true, true> : k_zfused<C, LINES, SPLIT, false, WPE, false, ZE, false, true, true>;
        else     k = (nin == C::L / 2) ? k_zfused<C, LINES, SPLIT, false, WPE, false, ZE, true, false, true> : k_zfused<C, LINES, SPLIT, false, WPE, false, ZE, false, false, true>;
    }
    allow_lds(k, lds);
    if (LINES == XB) {
        hipLaunchKernelGGL(k, dim3((unsigned)((Lx / XB) * nouter)), dim3(C::T * LINES), lds, st, data, sym, tw, nouter,
                           dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, nrhs, dBatch,
                           // (pairing a row with its mirror row on one XCD here too -- blocks b and b + 8 -- measured neutral on the
                           // one-tile and the multi-right-hand-side forms at 48^3 ... 256^3: left off)
                           0);
    } else {
        // split each tile into XB/LINES sub-groups: sub-group h starts at xi offset h*LINES
        // (tile, sub-group) collapse to one group index only when tiles are XB-contiguous in xi (2D natural layout)
        LSFC_REQUIRE(dTile == XB && sTile == XB, "sub-tile groups need the natural (2D) layout");
        hipLaunchKernelGGL(k, dim3((unsigned)((Lx / LINES) * nouter)), dim3(C::T * LINES), lds, st, data, sym, tw, nouter,
                           (int64_t)LINES, dOuter, dLine, (int64_t)LINES, sOuter, sLine, ytab, zm, nin, nrhs, dBatch,
                           (XB / LINES == 2 && nouter == 1 && (Lx / LINES) % 16 == 0 && !env_flag_no_sibling_pairs()) ? 1 : 0);
    }
}

// persistent pipelined fused pass: z-even symbol, whole 8-line tiles (3D layout); one workgroup per CU
static int cu_count() {
    static int cus = 0;
    if (!cus) { int dev = 0; LSFC_HIP(hipGetDevice(&dev)); hipDeviceProp_t pr; LSFC_HIP(hipGetDeviceProperties(&pr, dev)); cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    return cus;
}
static unsigned* ticket_set(hipStream_t st);
template <class C, bool SPLIT, bool LATE_SYM = false, bool TICKETS = false> static void zfused_persist_t(cplx* data, const cplx* sym, const cplx* tw, const cplx* twl, int Lx, int nouter,
                                                            int64_t dTile, int64_t dOuter, int64_t dLine, int64_t sTile, int64_t sOuter, int64_t sLine,
                                                            const int2* ytab, const int* zm, int nin, hipStream_t st, int xl = 0) {
    constexpr int LINES = XB;
    using LL = LdsLayout<LINES, 3, SPLIT>;
    size_t lds = persist_xbuf_bytes<C, LINES, SPLIT, false>();
    if (twl && lds + (size_t)C::TWLEN * sizeof(cplx) > (size_t)160 * 1024) twl = nullptr;
    auto k = (nin == C::L / 2) ? k_zfused_persist<C, LINES, SPLIT, true, false, LATE_SYM, false, TICKETS> : k_zfused_persist<C, LINES, SPLIT, false, false, LATE_SYM, false, TICKETS>;
    if (twl) {
        k = (nin == C::L / 2) ? k_zfused_persist<C, LINES, SPLIT, true, true, LATE_SYM, false, TICKETS> : k_zfused_persist<C, LINES, SPLIT, false, true, LATE_SYM, false, TICKETS>;
        lds += (size_t)C::TWLEN * sizeof(cplx);
        tw = twl;
    }
    // lane exchanges between stages of equal radix (whole-complex whole tiles, symbol after the first stage: the 512^3 and 256^3 forms)
    if constexpr (xlane_ok<C, LL>() && !SPLIT && LATE_SYM) {
        if (xl == 1) {
            if (twl) k = (nin == C::L / 2) ? k_zfused_persist<C, LINES, SPLIT, true, true, LATE_SYM, false, TICKETS, 1> : k_zfused_persist<C, LINES, SPLIT, false, true, LATE_SYM, false, TICKETS, 1>;
            else     k = (nin == C::L / 2) ? k_zfused_persist<C, LINES, SPLIT, true, false, LATE_SYM, false, TICKETS, 1> : k_zfused_persist<C, LINES, SPLIT, false, false, LATE_SYM, false, TICKETS, 1>;
        } else if (xl >= 2 && twl) {
            // 3: + mirror symbol values from L2 (XL & 2; measured slower, profiles/r03_experiment_fused_pass_variants.log)
            if (xl == 5 && TICKETS) k = (nin == C::L / 2) ? k_zfused_persist<C, LINES, SPLIT, true, true, LATE_SYM, false, TICKETS, 5> : k_zfused_persist<C, LINES, SPLIT, false, true, LATE_SYM, false, TICKETS, 5>;
            else if (xl == 3) k = (nin == C::L / 2) ? k_zfused_persist<C, LINES, SPLIT, true, true, LATE_SYM, false, TICKETS, 3> : k_zfused_persist<C, LINES, SPLIT, false, true, LATE_SYM, false, TICKETS, 3>;
            else k = (nin == C::L / 2) ? k_zfused_persist<C, LINES, SPLIT, true, true, LATE_SYM, false, TICKETS, 1> : k_zfused_persist<C, LINES, SPLIT, false, true, LATE_SYM, false, TICKETS, 1>;
        }
    }
    if (TICKETS) lds += 16;                             // the ticket slots
    allow_lds(k, lds);
    const int cus = cu_count();
    const unsigned ntiles = (unsigned)((Lx / XB) * nouter);
    // workgroups per CU that fit (LDS-limited); the walk stays interleaved so that co-resident workgroups touch neighbouring tiles
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, ((size_t)160 * 1024) / lds));
    const unsigned grid = std::min<unsigned>(ntiles, (unsigned)(cus * per_cu));
    hipLaunchKernelGGL(k, dim3(grid), dim3(C::T * LINES), lds, st, data, sym, tw, nouter,
                       dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, ntiles, TICKETS ? ticket_set(st) : (unsigned*)nullptr);
}
// Ticket counters of the half-tile pass: 8 (one per XCD) per launch, zeroed on the launch's stream just before it.  A ring of
// 64 sets per device, so that launches in flight on different streams (chunks of a distributed plan) never share a set.
static unsigned* ticket_set(hipStream_t st) {
    static std::mutex mu;
    static std::map<int, std::pair<unsigned*, unsigned>> ring;
    int dev = 0; LSFC_HIP(hipGetDevice(&dev));
    unsigned* base; unsigned idx;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = ring.find(dev);
        if (it == ring.end()) {
            unsigned* p = nullptr;
            LSFC_HIP(hipMalloc(&p, 64 * 8 * sizeof(unsigned)));
            it = ring.emplace(dev, std::make_pair(p, 0u)).first;
        }
        base = it->second.first; idx = it->second.second++ % 64u;
    }
    unsigned* set = base + 8 * idx;
    LSFC_HIP(hipMemsetAsync(set, 0, 8 * sizeof(unsigned), st));
    return set;
}
// the same on half tiles: 4-line workgroups with the twiddle table, as many per CU as the LDS holds (two at L = 1024)
template <class C> static void zfused_persist_half_t(cplx* data, const cplx* sym, const cplx* twl, int Lx, int nouter,
                                                     int64_t dTile, int64_t dOuter, int64_t dLine, int64_t sTile, int64_t sOuter, int64_t sLine,
                                                     const int2* ytab, const int* zm, int nin, hipStream_t st, int xl = 0) {
    if constexpr (C::L >= 1024) {
        constexpr int LINES = XB / 2;
        constexpr size_t lds = persist_lds_bytes<C, LINES, false, true, true>() + 16;   // + the ticket slot
        static_assert(lds <= (size_t)160 * 1024, "half-tile persistent pass: exchange buffer exceeds the LDS");
        auto k = (nin == C::L / 2) ? k_zfused_persist<C, LINES, false, true, true, true, true> : k_zfused_persist<C, LINES, false, false, true, true, true>;
        if constexpr (xlane_ok<C, LdsLayout<LINES, -1, false>>()) {
            if (xl) k = (nin == C::L / 2) ? k_zfused_persist<C, LINES, false, true, true, true, true, true, 1> : k_zfused_persist<C, LINES, false, false, true, true, true, true, 1>;
        }
        // (tried in round 3: the ONE LDS exchange left per direction run split -- real parts, then imaginary parts, through half the
        // buffer -- so that TWO half-tile workgroups fit a CU at 1280 / 1536 points: without the in-stage stores every output of the
        // widest butterfly stays live next to the symbol values in flight, the kernels spill (240 / 472 bytes per lane), 640^3 ties
        // at 30.4 ms and 768^3 loses 49.3 -> 60.7 ms; removed, profiles/r03_experiment_lane_exchange_radix4.log)
        LSFC_REQUIRE(twl != nullptr, "half-tile persistent pass: twiddle table missing");
        allow_lds(k, lds);
        const unsigned ntiles = (unsigned)((Lx / XB) * nouter);
        LSFC_REQUIRE(ntiles % 16 == 0 && (nouter % 2 == 0 || !ytab), "ticketed half-tile z pass needs a multiple of 16 tiles in row pairs");
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, ((size_t)160 * 1024) / lds));
        const unsigned grid = std::min<unsigned>(2 * ntiles, (unsigned)(cu_count() * per_cu));
        unsigned* tickets = ticket_set(st);
        hipLaunchKernelGGL(k, dim3(grid), dim3(C::T * LINES), lds, st, data, sym, twl, nouter,
                           dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, ntiles, tickets);
    } else {
        fail(LSFC_EINVAL, "half-tile persistent pass: lines of %d points run as whole tiles", (int)C::L);
    }
}

// half-tile z pass (L = 1024 and L = 1536 in the 3D tiled layout only): 4-line workgroups, sibling halves 8 blocks apart
// (LINES = 2: quarter tiles -- four sibling workgroups per tile, 32 blocks per group of 8 tiles)
template <class C, bool SPLIT, bool PREFETCH, int WPE, bool ZE = false, int LINES = 4> static void zfused_half_t(cplx* data, const cplx* sym, const cplx* tw, int Lx, int nouter,
                                                         int64_t dTile, int64_t dOuter, int64_t dLine, int64_t sTile, int64_t sOuter, int64_t sLine,
                                                         const int2* ytab, const int* zm, int nin, hipStream_t st) {
    using LL = LdsLayout<LINES, 3, SPLIT>;
    const size_t lds = (size_t)LL::line_elems(C::L) * LINES * LL::elem_bytes();
    auto k = (nin == C::L / 2) ? k_zfused<C, LINES, SPLIT, PREFETCH, WPE, true, ZE, true, false> : k_zfused<C, LINES, SPLIT, PREFETCH, WPE, true, ZE, false, false>;
    allow_lds(k, lds);
    const int64_t ntiles = (int64_t)(Lx / XB) * nouter;
    LSFC_REQUIRE(ntiles % 8 == 0, "half-tile z pass needs a multiple of 8 tiles");
    hipLaunchKernelGGL(k, dim3((unsigned)((XB / LINES) * ntiles)), dim3(C::T * LINES), lds, st, data, sym, tw, nouter,
                       dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, 1, (int64_t)0, 0);
}

#if LSFC_FAMILY == 2
#define LSFC_DISPATCH_L(L, CALL)                                           \
    switch (L) {                                                           \
    case 32:   { using C = Cfg32;   CALL; } break;                         \
    case 64:   { using C = Cfg64;   CALL; } break;                         \
    case 128:  { using C = Cfg128;  CALL; } break;                         \
    case 256:  { using C = Cfg256;  CALL; } break;                         \
    case 512:  { using C = Cfg512;  CALL; } break;                         \
    case 1024: { using C = Cfg1024; CALL; } break;                         \
    case 2048: { using C = Cfg2048; CALL; } break;                         \
    default: fail(LSFC_EINVAL, "pruned pipeline: unsupported padded length %d", (int)(L)); }
#elif LSFC_FAMILY == 3
#define LSFC_DISPATCH_L(L, CALL)                                           \
    switch (L) {                                                           \
    case 48:   { using C = Cfg48;   CALL; } break;                         \
    case 96:   { using C = Cfg96;   CALL; } break;                         \
    case 192:  { using C = Cfg192;  CALL; } break;                         \
    case 384:  { using C = Cfg384;  CALL; } break;                         \
    case 768:  { using C = Cfg768;  CALL; } break;                         \
    case 1536: { using C = Cfg1536; CALL; } break;                         \
    default: fail(LSFC_EINVAL, "pruned pipeline: unsupported padded length %d", (int)(L)); }
#else
#define LSFC_DISPATCH_L(L, CALL)                                           \
    switch (L) {                                                           \
    case 80:   { using C = Cfg80;   CALL; } break;                         \
    case 160:  { using C = Cfg160;  CALL; } break;                         \
    case 320:  { using C = Cfg320;  CALL; } break;                         \
    case 640:  { using C = Cfg640;  CALL; } break;                         \
    case 1280: { using C = Cfg1280; CALL; } break;                         \
    default: fail(LSFC_EINVAL, "pruned pipeline: unsupported padded length %d", (int)(L)); }
#endif

// Loads this family's code object (several MB: hundreds of kernel instantiations) on the current device.  HIP loads code
// objects lazily at the first launch from them; done here, once per device at plan creation and followed by a device
// synchronisation, the load never runs concurrently with a caller's transfers or the first real pass.
__global__ void FAM(k_warmup)(int* p) { if (p) *p = 0; }
void FAM(pruned_warmup)() {
    hipLaunchKernelGGL(FAM(k_warmup), dim3(1), dim3(64), 0, 0, (int*)nullptr);
    LSFC_HIP(hipGetLastError());
}

void FAM(pruned_perm)(int L, int* freq_of_storage) {
    LSFC_DISPATCH_L(L, perm_table<C>(freq_of_storage));
}

// chunk addressing of the x passes: s / W by multiplication with wmagic = floor(2^22 / W) + 1 (exact for s < 2048:
// s * (wmagic * W - 2^22) <= s * W < 2^22; the product s * wmagic stays below 2^32 for W >= 2)
static int chunk_magic(int L, int W) {
    LSFC_REQUIRE(W >= 8 && W <= L && L % W == 0 && L <= 2048, "chunk width %d does not divide the line length %d", W, L);
    return (1 << 22) / W + 1;
}
void FAM(pruned_xfwd)(int L, const PrunedTuning& tn, const VecBatch& vb, int nrhs, int64_t obatch, const double* nu, cplx* out, const cplx* tw, int64_t nlines, int W, int Wp, int n, hipStream_t st, int64_t bstride) {
    const int wmagic = chunk_magic(L, W);
    if (bstride <= 0) bstride = (int64_t)Wp * nlines;       // dense chunks: [chunk][line][Wp]
    if (tn.split_x) { LSFC_DISPATCH_L(L, (xfwd_t<C, true>(vb, nrhs, obatch, nu, out, tw, nlines, wmagic, W, Wp, n, bstride, st))); }
    else            { LSFC_DISPATCH_L(L, (xfwd_t<C, false>(vb, nrhs, obatch, nu, out, tw, nlines, wmagic, W, Wp, n, bstride, st))); }
    LSFC_HIP(hipGetLastError());
}
void FAM(pruned_xinv)(int L, const PrunedTuning& tn, const cplx* in, const VecBatch& vb, int nrhs, int64_t ibatch, double alpha, double beta, const cplx* tw, int64_t nlines, int W, int Wp, int n, hipStream_t st, int64_t bstride) {
    const int wmagic = chunk_magic(L, W);
    if (bstride <= 0) bstride = (int64_t)Wp * nlines;       // dense chunks: [chunk][line][Wp]
    if (tn.split_x) { LSFC_DISPATCH_L(L, (xinv_t<C, true>(in, vb, nrhs, ibatch, alpha, beta, tw, nlines, wmagic, W, Wp, n, bstride, st))); }
    else            { LSFC_DISPATCH_L(L, (xinv_t<C, false>(in, vb, nrhs, ibatch, alpha, beta, tw, nlines, wmagic, W, Wp, n, bstride, st))); }
    LSFC_HIP(hipGetLastError());
}
void FAM(pruned_yfwd)(int L, const PrunedTuning& tn, const cplx* a1, cplx* a2, const cplx* tw, int Lx, int m, int l, int p1, int p2, hipStream_t st, int nrhs, int64_t b1, int64_t b2) {
    size_t full_lds = 0;
    LSFC_DISPATCH_L(L, (full_lds = (size_t)LdsLayout<Tune<C>::YLINES, 3, false>::line_elems(C::L) * Tune<C>::YLINES * 16));
    if (tn.split_s || full_lds > (size_t)160 * 1024) { LSFC_DISPATCH_L(L, (yfwd_t<C, true, 1>(tn, a1, a2, tw, Lx, m, l, p1, p2, st, nrhs, b1, b2))); }
    else            { LSFC_DISPATCH_L(L, (yfwd_t<C, false, 1>(tn, a1, a2, tw, Lx, m, l, p1, p2, st, nrhs, b1, b2))); }
    LSFC_HIP(hipGetLastError());
}
void FAM(pruned_yinv)(int L, const PrunedTuning& tn, const cplx* a2, cplx* a1, const cplx* tw, int Lx, int m, int l, int p1, int p2, hipStream_t st, int nrhs, int64_t b1, int64_t b2) {
    size_t full_lds = 0;
    LSFC_DISPATCH_L(L, (full_lds = (size_t)LdsLayout<Tune<C>::YLINES, 3, false>::line_elems(C::L) * Tune<C>::YLINES * 16));
    if (tn.split_s || full_lds > (size_t)160 * 1024) { LSFC_DISPATCH_L(L, (yinv_t<C, true, 1>(tn, a2, a1, tw, Lx, m, l, p1, p2, st, nrhs, b1, b2))); }
    else            { LSFC_DISPATCH_L(L, (yinv_t<C, false, 1>(tn, a2, a1, tw, Lx, m, l, p1, p2, st, nrhs, b1, b2))); }
    LSFC_HIP(hipGetLastError());
}
int FAM(pruned_twfull_len)(int L) {
    int len = 0;
    LSFC_DISPATCH_L(L, len = C::TWLEN);
    return len;
}
void FAM(pruned_twfull)(int L, const cplx* tw, cplx* out) {
    LSFC_DISPATCH_L(L, twfull_table<C>(out, tw));
}

void FAM(pruned_zfused)(int L, const PrunedTuning& tn, cplx* data, const cplx* sym, const cplx* tw, const cplx* twl, int Lx, int nouter,
                   int64_t dTile, int64_t dOuter, int64_t dLine, int64_t sTile, int64_t sOuter, int64_t sLine, const int2* ytab,
                   const int* zm, int nin, hipStream_t st, int nrhs, int64_t dBatch) {
    // the half-tile forms (below) take one right-hand side per launch: a batch runs through them member by member
    const bool tiled = dLine == 8 && ((int64_t)(Lx / XB) * nouter) % 8 == 0;
    const bool half_form = (LSFC_FAMILY == 2 && ((L == 1024 && (tn.z_half >= 0 ? tn.z_half : (zm ? 0 : 2)) > 0 && tiled) || (L == 2048 && dLine == 8)))
                        || (LSFC_FAMILY == 3 && L == 1536 && (tn.z_half >= 0 ? tn.z_half : 1) > 0 && tiled);
    // persistent pipelined form: z-even symbol, 3D tiled layout, whole 8-line tiles, one right-hand side
    {
        // z_persist: 1 whole-complex exchanges, 2 split exchanges, 3 / 4 the same with the symbol loaded after the first forward
        // stage (its 32 registers stay free during the widest butterfly); auto = 3: 5.24 ms at 512^3 against 5.73 for the
        // one-tile kernel, 0.58 against 0.675 ms at 256^3 (profiles/r02_experiment_persistent_zpass.log)
        // 5: ticketed half tiles.  Auto at L = 1536, whose whole tiles do not fit the register budget (768^3: fused pass 29.5 ->
        // 23.7 ms, apply 56.3 -> 50.4 ms); at L = 1024 it ties with 3 (5.19-5.25 against 5.20-5.25 ms; it moves 24.0 GB instead
        // of 25.8: paired rows meet in L2) and 3 stays (profiles/r02_experiment_ticketed_half_tiles.log)
        // (the 1280-point line, 20 elements per thread, likewise: 640^3 fused pass 18.5 -> 16.5 ms, apply 35.6 -> 33.6 ms)
        // 6 = 3 with the tiles handed out by the same tickets (row pairs per XCD: both reads of a symbol row meet in one L2):
        // 512^3 apply 12.95 -> 12.86 ms over three A/B rounds of bench.py, neutral at 256^3; auto from L = 1024 on
        const int zp = tn.z_persist >= 0 ? tn.z_persist : ((L == 1536 || L == 1280 || L == 2048) ? 5 : (L >= 1024 ? 6 : 3));   // (2048: 1024^3 fused pass 58.8 -> 47.0 ms)
        bool eight_lines = false;
        LSFC_DISPATCH_L(L, (eight_lines = Tune<C>::LINES == XB));
        // (worth it only when a workgroup walks over several tiles: below ~4 tiles per resident workgroup -- grids up to 64^3 --
        // the one-tile kernels finish sooner, 35 against 37.5 us per apply at 48^3)
        const bool enough_tiles = tn.z_persist > 0 || (int64_t)(Lx / XB) * nouter >= (int64_t)4096;
        const bool half5 = zp == 5 && tiled && twl && L >= 1024 && ((int64_t)(Lx / XB) * nouter) % 16 == 0 && (nouter % 2 == 0 || !ytab);
        if (zp > 0 && zm && dLine == 8 && nrhs == 1 && (eight_lines || half5) && (!half_form || half5) && enough_tiles) {
            size_t full_lds = 0;
            LSFC_DISPATCH_L(L, (full_lds = (size_t)LdsLayout<XB, 3, false>::line_elems(C::L) * XB * 16));
            const bool split = zp == 2 || zp == 4 || full_lds > (size_t)160 * 1024;
            // 5: half tiles (4-line workgroups, swizzled unpadded exchange buffer + twiddle table), two workgroups per CU at L = 1024
            if (half5) {
                LSFC_DISPATCH_L(L, (zfused_persist_half_t<C>(data, sym, twl, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st, tn.xlane != 0 ? 1 : 0)));
                LSFC_HIP(hipGetLastError());
                return;
            }
            // 6: whole tiles (as 3) handed out by tickets in row pairs per XCD
            if (zp == 6 && !split && ((int64_t)(Lx / XB) * nouter) % 16 == 0 && nouter % 2 == 0) {
                LSFC_DISPATCH_L(L, (zfused_persist_t<C, false, true, true>(data, sym, tw, tn.tw_lds ? twl : nullptr, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st, tn.xlane < 0 ? 5 : tn.xlane)));
                LSFC_HIP(hipGetLastError());
                return;
            }
            if ((zp == 3 || zp == 6) && !split) { LSFC_DISPATCH_L(L, (zfused_persist_t<C, false, true>(data, sym, tw, tn.tw_lds ? twl : nullptr, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st, tn.xlane < 0 ? 1 : tn.xlane))); }
            else if (zp >= 3) { LSFC_DISPATCH_L(L, (zfused_persist_t<C, true, true>(data, sym, tw, tn.tw_lds ? twl : nullptr, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st))); }
            else if (split) { LSFC_DISPATCH_L(L, (zfused_persist_t<C, true>(data, sym, tw, tn.tw_lds ? twl : nullptr, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st))); }
            else       { LSFC_DISPATCH_L(L, (zfused_persist_t<C, false>(data, sym, tw, tn.tw_lds ? twl : nullptr, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st))); }
            LSFC_HIP(hipGetLastError());
            return;
        }
    }
    if (nrhs > 1 && half_form) {
        for (int r = 0; r < nrhs; ++r)
            FAM(pruned_zfused)(L, tn, data + (int64_t)r * dBatch, sym, tw, twl, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st, 1, 0);
        return;
    }
#define LSFC_ZF(SP, PF) do { if (zm) { LSFC_DISPATCH_L(L, (zfused_t<C, SP, PF, 1, true>(data, sym, tw, tn.tw_lds ? twl : nullptr, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st, nrhs, dBatch))); } \
                             else    { LSFC_DISPATCH_L(L, (zfused_t<C, SP, PF, 1, false>(data, sym, tw, tn.tw_lds ? twl : nullptr, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st, nrhs, dBatch))); } } while (0)
    // auto (-1): half-tile, split exchanges, symbol prefetch -- 6.95 -> 6.6 ms at 512^3 (profiles/r01_experiment_half_tile.log)
    // with the z-even half symbol the full-tile form wins (6.05 ms, profiles/r01_experiment_even_z.log)
#if LSFC_FAMILY == 2
    const int zh = tn.z_half >= 0 ? tn.z_half : (zm ? 0 : 2);
    if (L == 1024 && zh > 0 && dLine == 8 && ((int64_t)(Lx / XB) * nouter) % 8 == 0) {
        using C = Cfg1024;
#define LSFC_ZH(SP, PF, W) do { if (zm) zfused_half_t<C, SP, PF, W, true>(data, sym, tw, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st); \
                                else zfused_half_t<C, SP, PF, W, false>(data, sym, tw, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st); } while (0)
        switch (zh) {
        case 1: LSFC_ZH(false, true, 2); break;
        case 2: LSFC_ZH(true, true, 2); break;
        case 3: LSFC_ZH(true, false, 3); break;
        default: LSFC_ZH(false, false, 2); break;
        }
#undef LSFC_ZH
        LSFC_HIP(hipGetLastError());
        return;
    }
    // the 2048-point line in the 3D tiled layout: a whole 8-line tile does not fit the 160 KiB of LDS, half tiles do
    if (L == 2048 && dLine == 8) {
        LSFC_REQUIRE(((int64_t)(Lx / XB) * nouter) % 8 == 0, "half-tile z pass needs a multiple of 8 tiles");
        using C = Cfg2048;
        // LSFC_Z_QUARTER=1 (experiment, off): quarter tiles, two 256-thread workgroups per CU (74 KB of LDS each).  Measured on
        // the 2D tiled pass at n = 1024 (256 tiles): 35.4 against 34.2 us -- workgroups that start together run their phases
        // together, so two per CU overlap nothing in a one-tile kernel; the ticketed persistent form (2 half tiles per
        // workgroup) takes 54.8 us there (profiles/r02_2d_half_symbol.jsonl)
        const bool quarter = zm && env_flag_z_quarter();
        if (quarter) zfused_half_t<C, false, false, 2, true, 2>(data, sym, tw, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st);
        else if (zm) zfused_half_t<C, false, false, 2, true>(data, sym, tw, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st);
        else zfused_half_t<C, false, false, 2, false>(data, sym, tw, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st);
        LSFC_HIP(hipGetLastError());
        return;
    }
#elif LSFC_FAMILY == 3
    // the 1536-point line holds 24 elements per thread: in 8-line (512-thread) workgroups the fused pass is capped at
    // 256 registers and spills.  Half tiles (4 lines, 256 threads, one wave per SIMD) lift the cap.
    // z_half: 0 off, 1 (auto) whole-complex exchange + prefetch, 2 split + prefetch, 3 split, 4 whole-complex
    // (768^3: fused pass 36.7 -> 29.7 ms, apply 64.7 -> 57.4 ms)
    const int zh = tn.z_half >= 0 ? tn.z_half : 1;
    if (L == 1536 && zh > 0 && dLine == 8 && ((int64_t)(Lx / XB) * nouter) % 8 == 0) {
        using C = Cfg1536;
#define LSFC_ZH(SP, PF) do { if (zm) zfused_half_t<C, SP, PF, 1, true>(data, sym, tw, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st); \
                             else zfused_half_t<C, SP, PF, 1, false>(data, sym, tw, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st); } while (0)
        switch (zh) {
        case 1: LSFC_ZH(false, true); break;
        case 2: LSFC_ZH(true, true); break;
        case 3: LSFC_ZH(true, false); break;
        default: LSFC_ZH(false, false); break;
        }
#undef LSFC_ZH
        LSFC_HIP(hipGetLastError());
        return;
    }
#endif
    // auto (-1): measured on MI355X -- at L >= 1024 (16 elements/thread, 2 waves/SIMD either way) whole-complex
    // exchanges + symbol prefetch win (7.4 -> 6.7 ms at 512^3); below, split exchanges without prefetch (more waves)
    // mixed-radix lines (profiles/r01_experiment_mixed_radix_knobs.log): the symbol prefetch pays from L = 320 on
    // (z pass -15..-30 %), whole-complex exchanges on the factor-3 lines from L = 384; the 1536-point line (24
    // elements per thread in 512-thread workgroups) has no registers left for the prefetch
#if LSFC_FAMILY == 2
    // (the 2048-point line reaches this point only in the 2D layout, 4 lines per workgroup: no prefetch there,
    // fused pass 54 -> 46 us at 2D n = 1024)
    const bool e16 = L == 1024, full = L >= 1024;
#elif LSFC_FAMILY == 3
    const bool e16 = L >= 384 && L < 1536, full = L >= 384;
#else
    const bool e16 = L >= 320, full = false;
#endif
    bool sp = tn.split_z >= 0 ? tn.split_z != 0 : !full;
    const bool pf = tn.sym_prefetch >= 0 ? tn.sym_prefetch != 0 : e16;
    // whole-complex exchange buffers of the longest lines exceed the 160 KiB of LDS: those run split
    // (sized by the workgroup zfused_t launches: Tune<C>::LINES lines -- the y passes alone use YLINES)
    size_t full_lds = 0;
    LSFC_DISPATCH_L(L, (full_lds = (size_t)LdsLayout<Tune<C>::LINES, 3, false>::line_elems(C::L) * Tune<C>::LINES * 16));
    if (full_lds > (size_t)160 * 1024) sp = true;
    if (sp) { if (pf) { LSFC_ZF(true, true); } else { LSFC_ZF(true, false); } }
    else    { if (pf) { LSFC_ZF(false, true); } else { LSFC_ZF(false, false); } }
#undef LSFC_ZF
    LSFC_HIP(hipGetLastError());
}

} // namespace lsfc
