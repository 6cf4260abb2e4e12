// Symbol generators: the builder side of the reference (buildFastConvolution*,
// Gtruncated2D/3D, buildGConv) evaluated on the device (gfx950), setup-time only.
#include "plan.hpp"
#include <future>
#include "pointwise.hpp"
#include <cmath>

namespace lsfc {

static inline unsigned grid_for(int64_t count, int block = 256, int cap = 256 * 16) {
    int64_t g = (count + block - 1) / block; if (g < 1) g = 1; if (g > cap) g = cap; return (unsigned)g;
}

// a*a - b*b without FMA contraction (HIP's __dmul_rn is a plain product and still contracts): at
// |s| == k the reference divides by an exact zero and yields Inf/NaN; keep that behaviour bit for bit.
__device__ __forceinline__ double diff_squares_exact(double a, double b) {
#pragma clang fp contract(off)
    const double aa = a * a;
    const double bb = b * b;
    return aa - bb;
}

// ---- Gtruncated3D (src/Functions.jl:49-51) -----------------------------------
__device__ __forceinline__ cplx gtrunc3d(double L, double k, double s, cplx eiLk, bool patch, cplx limit) {
    if (patch && s == k) return limit;
    const double t = L * s / 3.14159265358979323846;
    const double snc = (t == 0.0) ? 1.0 : sinpi(t) / (3.14159265358979323846 * t);     // Julia sinc
    // cos(L s) - i k L sinc
    const double cr = cos(L * s), ci = -k * L * snc;
    // -1 + e^{iLk} * (cr + i ci)
    const double nr = -1.0 + (eiLk.x * cr - eiLk.y * ci);
    const double ni = eiLk.x * ci + eiLk.y * cr;
    const double den = diff_squares_exact(k, s);
    return make_double2(nr / den, ni / den);
}

// Planes jz in [z0, z0+C) of the literal symbol in FFT order: W[jx + P0*(jy + P1*c)]
__global__ void k_gen_gv3d_planes(cplx* __restrict__ W, int P0, int P1, int P2, int z0, int C, double dk, double L, double k,
                                  cplx eiLk, int patch, cplx limit) {
    const int64_t total = (int64_t)P0 * P1 * C;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int jx = (int)(idx % P0); const int64_t r = idx / P0; const int jy = (int)(r % P1); const int jz = z0 + (int)(r / P1);
        const double kx = dk * (double)(jx < P0 / 2 ? jx : jx - P0);
        const double ky = dk * (double)(jy < P1 / 2 ? jy : jy - P1);
        const double kz = dk * (double)(jz < P2 / 2 ? jz : jz - P2);
        const double s = sqrt(kx * kx + ky * ky + kz * kz);       // src/FastConvolution3D.jl:98
        W[idx] = gtrunc3d(L, k, s, eiLk, patch != 0, limit);
    }
}

// U[dx + Q0*(dy + Q1*jz)] = W[wrap(dx) + P0*(wrap(dy) + P1*c)]  for the planes of one chunk
__global__ void k_crop_xy_planes(const cplx* __restrict__ W, cplx* __restrict__ U, int P0, int P1, int Q0, int Q1, int z0, int C) {
    const int64_t total = (int64_t)Q0 * Q1 * C;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int dx = (int)(idx % Q0); const int64_t r = idx / Q0; const int dy = (int)(r % Q1); const int c = (int)(r / Q1);
        const int sx = dx < Q0 / 2 ? dx : dx + (P0 - Q0);
        const int sy = dy < Q1 / 2 ? dy : dy + (P1 - Q1);
        U[dx + (int64_t)Q0 * (dy + (int64_t)Q1 * (z0 + c))] = W[sx + (int64_t)P0 * (sy + (int64_t)P1 * c)];
    }
}

// T2[dx + Q0*(dy + Q1*dz)] = scale * U[dx + Q0*(dy + Q1*wrap(dz))]
__global__ void k_crop_z(const cplx* __restrict__ U, cplx* __restrict__ T2, int64_t plane, int P2, int Q2, double scale) {
    const int64_t total = plane * Q2;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t xy = idx % plane; const int dz = (int)(idx / plane);
        const int sz = dz < Q2 / 2 ? dz : dz + (P2 - Q2);
        const cplx v = U[xy + plane * sz];
        T2[idx] = make_double2(scale * v.x, scale * v.y);
    }
}

void symbol_gv3d_reduced(lsfc_plan* p, double box, DevBuf<cplx>& G2) {
    const int n = p->dims[0], m = p->dims[1], l = p->dims[2];
    // literal lattice P = 4n (src/FastConvolution3D.jl:79-81); working grid Q = p->pads (2n, or pruned_best_length(n) <= 4n)
    const int P0 = 4 * n, P1 = 4 * m, P2 = 4 * l, Q0 = p->pads[0], Q1 = p->pads[1], Q2 = p->pads[2];
    LSFC_REQUIRE(Q0 <= P0 && Q1 <= P1 && Q2 <= P2 && Q0 >= 2 * n && Q1 >= 2 * m && Q2 >= 2 * l, "internal: working grid outside [2n, 4n]");
    const double Lp = 4.0 * box, L = 1.8 * box, k = p->omega;       // src/FastConvolution3D.jl:72-73
    const double dk = 2.0 * 3.14159265358979323846 / Lp;
    const cplx eiLk = make_double2(cos(L * k), sin(L * k));
    // analytic limit at s == k: (iL - (i/k) sin(Lk) e^{iLk}) / (2k)
    const double sLk = sin(L * k);
    const cplx limit = make_double2((sLk * eiLk.y / k) / (2.0 * k), (L - sLk * eiLk.x / k) / (2.0 * k));
    const int patch = (p->flags & LSFC_FLAG_PATCH_SINGULAR) ? 1 : 0;
    hipStream_t st = p->stream;

    PhaseTimer pt;
    const int64_t plane_lit = (int64_t)P0 * P1, plane_red = (int64_t)Q0 * Q1;
    // chunk of z-frequency planes: ~1 GiB of literal planes at a time
    int C = (int)std::max<int64_t>(1, std::min<int64_t>(P2, ((int64_t)1 << 30) / (plane_lit * (int64_t)sizeof(cplx))));
    while (P2 % C) --C;
    // rocFFT compiles the kernels of a new transform shape at plan creation (0.3-0.8 s each in a fresh process): the
    // three plans are created concurrently on helper threads, and the two needed later keep compiling while the
    // planes are evaluated
    const int dev = p->device;
    RocFft inv2d, invz, fwd;
    const size_t len2[2] = { (size_t)P0, (size_t)P1 };
    const size_t len3[3] = { (size_t)Q0, (size_t)Q1, (size_t)Q2 };
    auto on_device = [dev](auto&& body) { return std::async(std::launch::async, [dev, body] { LSFC_HIP(hipSetDevice(dev)); body(); }); };
    auto f2d = on_device([&inv2d, &len2, C] { inv2d.create(2, len2, false, (size_t)C, true); });
    auto fz = on_device([&invz, P2, plane_red] { invz.create_strided_1d((size_t)P2, (size_t)plane_red, 1, (size_t)plane_red, false, true); });
    auto f3d = on_device([&fwd, &len3] { fwd.create(3, len3, true, 1, true); });
    // (a failure below must not unwind while a helper still references the plans)
    struct Joiner { std::future<void>* f[3]; ~Joiner() { for (auto* x : f) if (x->valid()) x->wait(); } } joiner{{ &f2d, &fz, &f3d }};

    DevBuf<cplx> U; U.alloc((size_t)(plane_red * P2));
    pt.mark("symbol: alloc U");
    {
        DevBuf<cplx> W; W.alloc((size_t)(plane_lit * C));
        f2d.get();
        pt.mark("symbol: alloc W + rocFFT 2D plan");
        for (int z0 = 0; z0 < P2; z0 += C) {
            hipLaunchKernelGGL(k_gen_gv3d_planes, dim3(grid_for(plane_lit * C)), dim3(256), 0, st, W.p, P0, P1, P2, z0, C, dk, L, k, eiLk, patch, limit);
            inv2d.exec(W.p, st);
            hipLaunchKernelGGL(k_crop_xy_planes, dim3(grid_for(plane_red * C)), dim3(256), 0, st, W.p, U.p, P0, P1, Q0, Q1, z0, C);
        }
        LSFC_HIP(hipGetLastError());
        LSFC_HIP(hipStreamSynchronize(st));
        pt.mark("symbol: planes (gen, ifft2, crop)");
        inv2d.release();
    }
    pt.mark("symbol: free W + 2D plan");
    fz.get();
    pt.mark("symbol: rocFFT z plan");
    invz.exec(U.p, st);
    LSFC_HIP(hipStreamSynchronize(st));
    pt.mark("symbol: ifft z");
    invz.release();
    pt.mark("symbol: free z plan");
    G2.alloc((size_t)(plane_red * Q2));
    pt.mark("symbol: alloc G2");
    hipLaunchKernelGGL(k_crop_z, dim3(grid_for(plane_red * Q2)), dim3(256), 0, st, U.p, G2.p, plane_red, P2, Q2,
                       1.0 / ((double)P0 * (double)P1 * (double)P2));
    LSFC_HIP(hipGetLastError());
    LSFC_HIP(hipStreamSynchronize(st));
    pt.mark("symbol: crop z");
    U.release();
    pt.mark("symbol: free U");
    f3d.get();
    pt.mark("symbol: rocFFT 3D plan");
    fwd.exec(G2.p, st);
    LSFC_HIP(hipStreamSynchronize(st));
    pt.mark("symbol: fft 3D");
    fwd.release();
    pt.mark("symbol: free 3D plan");
}

// ---- the same symbol through its symmetry: Gtruncated3D is radial, so the literal symbol, the spatial kernel T and the
// reduced symbol are even in every axis, and the pruned pipeline stores only the frequencies ky <= Q1/2, kz <= Q2/2 (all kx).
// Every stage below works on the non-negative half of the axes it has not transformed yet and mirrors a half-axis to
// full length only for the one strided rocFFT pass that consumes it:
//   planes kz in [0, 2l] of the literal lattice -> ifft2 -> keep offsets dx in [0, Q0/2], dy in [0, Q1/2]   (1/8 of U)
//   mirror in kz, ifft along z (length 4l), keep dz in [0, Q2/2]                                            (octant of T)
//   mirror in dz, fft z (Q2), keep kz <= Q2/2; mirror in dy, fft y (Q1), keep ky <= Q1/2; mirror in dx, fft x (Q0)
// Peak temporary memory at 512^3: 8.6 GB instead of 51 GB (and a quarter of the transform work); 1024^3 fits one GPU.
// Gq[kx + Q0*(ky + (Q1/2+1)*kz)], natural order, unscaled as symbol_gv3d_reduced's G2.
__global__ void k_crop_xy_quarter(const cplx* __restrict__ W, cplx* __restrict__ U, int P0, int P1, int P2, int Hx, int Hy, int z0, int C) {
    const int64_t plane = (int64_t)Hx * Hy, total = plane * C;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int dx = (int)(idx % Hx); const int64_t r = idx / Hx; const int dy = (int)(r % Hy); const int c = (int)(r / Hy);
        const int jz = z0 + c;
        if (jz > P2 / 2) continue;
        const cplx v = W[dx + (int64_t)P0 * (dy + (int64_t)P1 * c)];
        U[dx + (int64_t)Hx * dy + plane * jz] = v;
        if (jz > 0 && jz < P2 / 2) U[dx + (int64_t)Hx * dy + plane * (P2 - jz)] = v;       // the mirror plane
    }
}
// dst[col + ncols*z] = scale * src[col + ncols*min(z, Q - z)], z < Q   (axis = the slowest one)
__global__ void k_mirror_slowest(const cplx* __restrict__ src, cplx* __restrict__ dst, int64_t ncols, int Q, double scale) {
    const int64_t total = ncols * Q;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t col = idx % ncols; const int z = (int)(idx / ncols);
        const cplx v = src[col + ncols * (int64_t)(z <= Q / 2 ? z : Q - z)];
        dst[idx] = make_double2(scale * v.x, scale * v.y);
    }
}
// dst[dx + Hx*(y + Q1*kz)] = src[dx + Hx*(min(y, Q1 - y) + Hy*kz)]   (middle axis)
__global__ void k_mirror_middle(const cplx* __restrict__ src, cplx* __restrict__ dst, int Hx, int Hy, int Q1, int Hz) {
    const int64_t total = (int64_t)Hx * Q1 * Hz;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int dx = (int)(idx % Hx); const int64_t r = idx / Hx; const int y = (int)(r % Q1); const int kz = (int)(r / Q1);
        dst[idx] = src[dx + (int64_t)Hx * ((y <= Q1 / 2 ? y : Q1 - y) + (int64_t)Hy * kz)];
    }
}
// dst[x + Q0*(ky + Hy*kz)] = src[min(x, Q0 - x) + Hx*(ky + Q1*kz)], ky < Hy   (fastest axis; src rows have Q1 entries)
__global__ void k_mirror_fastest(const cplx* __restrict__ src, cplx* __restrict__ dst, int Hx, int Q0, int Hy, int Q1, int Hz) {
    const int64_t total = (int64_t)Q0 * Hy * Hz;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % Q0); const int64_t r = idx / Q0; const int ky = (int)(r % Hy); const int kz = (int)(r / Hy);
        dst[idx] = src[(x <= Q0 / 2 ? x : Q0 - x) + (int64_t)Hx * (ky + (int64_t)Q1 * kz)];
    }
}

void symbol_gv3d_quarter(lsfc_plan* p, double box, DevBuf<cplx>& Gq) {
    const int n = p->dims[0], m = p->dims[1], l = p->dims[2];
    const int P0 = 4 * n, P1 = 4 * m, P2 = 4 * l, Q0 = p->pads[0], Q1 = p->pads[1], Q2 = p->pads[2];
    LSFC_REQUIRE(Q0 <= P0 && Q1 <= P1 && Q2 <= P2 && Q0 >= 2 * n && Q1 >= 2 * m && Q2 >= 2 * l, "internal: working grid outside [2n, 4n]");
    LSFC_REQUIRE(Q0 % 2 == 0 && Q1 % 2 == 0 && Q2 % 2 == 0, "internal: odd working grid");
    const int Hx = Q0 / 2 + 1, Hy = Q1 / 2 + 1, Hz = Q2 / 2 + 1;
    const double Lp = 4.0 * box, L = 1.8 * box, k = p->omega;       // src/FastConvolution3D.jl:72-73
    const double dk = 2.0 * 3.14159265358979323846 / Lp;
    const cplx eiLk = make_double2(cos(L * k), sin(L * k));
    const double sLk = sin(L * k);
    const cplx limit = make_double2((sLk * eiLk.y / k) / (2.0 * k), (L - sLk * eiLk.x / k) / (2.0 * k));
    const int patch = (p->flags & LSFC_FLAG_PATCH_SINGULAR) ? 1 : 0;
    hipStream_t st = p->stream;

    PhaseTimer pt;
    const int64_t plane_lit = (int64_t)P0 * P1, cols = (int64_t)Hx * Hy;
    const int planes = P2 / 2 + 1;                                  // kz = 0 .. 2l
    int C = (int)std::max<int64_t>(1, std::min<int64_t>(planes, ((int64_t)1 << 30) / (plane_lit * (int64_t)sizeof(cplx))));
    const int dev = p->device;
    RocFft inv2d, inv2d_tail, invz, fwdz, fwdy, fwdx;
    const size_t len2[2] = { (size_t)P0, (size_t)P1 };
    const int tail = planes % C;                                    // the last chunk of planes may be shorter
    auto on_device = [dev](auto&& body) { return std::async(std::launch::async, [dev, body] { LSFC_HIP(hipSetDevice(dev)); body(); }); };
    auto f2d = on_device([&inv2d, &inv2d_tail, &len2, C, tail] { inv2d.create(2, len2, false, (size_t)C, true); if (tail) inv2d_tail.create(2, len2, false, (size_t)tail, true); });
    auto fiz = on_device([&invz, P2, cols] { invz.create_strided_1d((size_t)P2, (size_t)cols, 1, (size_t)cols, false, true); });
    auto ffz = on_device([&fwdz, Q2, cols] { fwdz.create_strided_1d((size_t)Q2, (size_t)cols, 1, (size_t)cols, true, true); });
    auto ffy = on_device([&fwdy, Q1, Hx] { fwdy.create_strided_1d((size_t)Q1, (size_t)Hx, 1, (size_t)Hx, true, true); });
    const size_t lenx[1] = { (size_t)Q0 };
    auto ffx = on_device([&fwdx, &lenx, Hy, Hz] { fwdx.create(1, lenx, true, (size_t)Hy * (size_t)Hz, true); });
    struct Joiner { std::future<void>* f[5]; ~Joiner() { for (auto* x : f) if (x->valid()) x->wait(); } } joiner{{ &f2d, &fiz, &ffz, &ffy, &ffx }};

    DevBuf<cplx> U; U.alloc((size_t)(cols * P2));
    {
        DevBuf<cplx> W; W.alloc((size_t)(plane_lit * C));
        f2d.get();
        pt.mark("symbol: alloc + rocFFT 2D plan");
        for (int z0 = 0; z0 < planes; z0 += C) {
            const int c = std::min(C, planes - z0);
            hipLaunchKernelGGL(k_gen_gv3d_planes, dim3(grid_for(plane_lit * c)), dim3(256), 0, st, W.p, P0, P1, P2, z0, c, dk, L, k, eiLk, patch, limit);
            (c == C ? inv2d : inv2d_tail).exec(W.p, st);
            hipLaunchKernelGGL(k_crop_xy_quarter, dim3(grid_for(cols * c)), dim3(256), 0, st, W.p, U.p, P0, P1, P2, Hx, Hy, z0, c);
        }
        LSFC_HIP(hipGetLastError());
        LSFC_HIP(hipStreamSynchronize(st));
        pt.mark("symbol: planes kz >= 0 (gen, ifft2, quarter crop)");
        inv2d.release(); inv2d_tail.release();
    }
    fiz.get();
    invz.exec(U.p, st);
    LSFC_HIP(hipStreamSynchronize(st));
    invz.release();
    pt.mark("symbol: ifft z");
    // octant of the spatial kernel, mirrored to Q2 along z, forward z
    DevBuf<cplx> E1; E1.alloc((size_t)(cols * Q2));
    hipLaunchKernelGGL(k_mirror_slowest, dim3(grid_for(cols * Q2)), dim3(256), 0, st, U.p, E1.p, cols, Q2, 1.0 / ((double)P0 * (double)P1 * (double)P2));
    LSFC_HIP(hipGetLastError());
    LSFC_HIP(hipStreamSynchronize(st));
    U.release();
    ffz.get();
    fwdz.exec(E1.p, st);
    // planes kz <= Q2/2 of E1, mirrored to Q1 along y, forward y plane by plane
    DevBuf<cplx> E2; E2.alloc((size_t)Hx * Q1 * Hz);
    hipLaunchKernelGGL(k_mirror_middle, dim3(grid_for((int64_t)Hx * Q1 * Hz)), dim3(256), 0, st, E1.p, E2.p, Hx, Hy, Q1, Hz);
    LSFC_HIP(hipGetLastError());
    LSFC_HIP(hipStreamSynchronize(st));
    fwdz.release(); E1.release();
    ffy.get();
    for (int kz = 0; kz < Hz; ++kz) fwdy.exec(E2.p + (int64_t)Hx * Q1 * kz, st);
    // rows ky <= Q1/2, mirrored to Q0 along x, forward x
    Gq.alloc((size_t)Q0 * Hy * Hz);
    hipLaunchKernelGGL(k_mirror_fastest, dim3(grid_for((int64_t)Q0 * Hy * Hz)), dim3(256), 0, st, E2.p, Gq.p, Hx, Q0, Hy, Q1, Hz);
    LSFC_HIP(hipGetLastError());
    LSFC_HIP(hipStreamSynchronize(st));
    fwdy.release(); E2.release();
    ffx.get();
    fwdx.exec(Gq.p, st);
    LSFC_HIP(hipStreamSynchronize(st));
    fwdx.release();
    pt.mark("symbol: forward z, y, x on the mirrored halves");
}

// ---- Gtruncated2D (src/Functions.jl:40-42), centred literal (4n x 4m) ---------
__global__ void k_gen_gv2d(cplx* __restrict__ G, int P0, int P1, double dk, double L, double k, cplx a, cplx b, int patch, cplx limit) {
    const int64_t total = (int64_t)P0 * P1;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % P0), j = (int)(idx / P0);
        const double kx = dk * (double)(i - P0 / 2), ky = dk * (double)(j - P1 / 2);       // kx = -(2n):(2n-1)
        const double s = sqrt(kx * kx + ky * ky);
        if (patch && s == k) { G[idx] = limit; continue; }
        const double sj1 = s * j1(L * s), j0v = j0(L * s);
        // 1 + a*(s J1(Ls)) - b*J0(Ls)
        const double nr = 1.0 + a.x * sj1 - b.x * j0v, ni = a.y * sj1 - b.y * j0v;
        const double den = diff_squares_exact(s, k);
        G[idx] = make_double2(nr / den, ni / den);
    }
}

void symbol_gv2d_literal(lsfc_plan* p, double box, DevBuf<cplx>& G, int lit[3]) {
    const int n = p->dims[0], m = p->dims[1];
    lit[0] = 4 * n; lit[1] = 4 * m; lit[2] = 1;
    const double Lp = 4.0 * box, L = 1.5 * box, k = p->omega;       // src/FastConvolution.jl:187-188
    const double dk = 2.0 * 3.14159265358979323846 / Lp;
    const double pi = 3.14159265358979323846;
    // a = i pi/2 L H0(Lk), b = i pi/2 L k H1(Lk); H = J + iY
    const double J0 = ::j0(L * k), Y0 = ::y0(L * k), J1 = ::j1(L * k), Y1 = ::y1(L * k);
    const cplx a = make_double2(-pi / 2 * L * Y0, pi / 2 * L * J0);
    const cplx b = make_double2(-pi / 2 * L * k * Y1, pi / 2 * L * k * J1);
    // limit at s == k: (a L k J0(Lk) + b L J1(Lk)) / (2k)
    const cplx limit = make_double2((a.x * L * k * J0 + b.x * L * J1) / (2 * k), (a.y * L * k * J0 + b.y * L * J1) / (2 * k));
    G.alloc((size_t)lit[0] * lit[1]);
    hipLaunchKernelGGL(k_gen_gv2d, dim3(grid_for((int64_t)lit[0] * lit[1])), dim3(256), 0, p->stream, G.p, lit[0], lit[1], dk, L, k, a, b,
                       (p->flags & LSFC_FLAG_PATCH_SINGULAR) ? 1 : 0, limit);
    LSFC_HIP(hipGetLastError());
    LSFC_HIP(hipStreamSynchronize(p->stream));
}

// ---- buildGConv (src/FastConvolution.jl:425-469) ------------------------------
__global__ void k_gen_trap2d(cplx* __restrict__ Ge, int P0, int P1, int n, int m, double xe0, double ye0, double h, double k, cplx d0) {
    const int64_t total = (int64_t)P0 * P1;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % P0), j = (int)(idx / P0);
        const double h2 = h * h;
        if (i == n - 1 && j == m - 1) { Ge[idx] = make_double2(-0.25 * d0.y * h2, 0.25 * d0.x * h2); continue; }   // i/4 * D0 * h^2
        const double xe = xe0 + h * i, ye = ye0 + h * j;
        const double r = sqrt(xe * xe + ye * ye);
        // i/4 * H0(k r) * h^2 = (-Y0 + i J0)/4 * h^2
        Ge[idx] = make_double2(-0.25 * y0(k * r) * h2, 0.25 * j0(k * r) * h2);
    }
}

void symbol_trap2d_literal(lsfc_plan* p, double x0, double y0, double h, cplx d0, DevBuf<cplx>& G) {
    const int n = p->dims[0], m = p->dims[1];
    const int P0 = 2 * n - 1, P1 = 2 * m - 1;
    G.alloc((size_t)P0 * P1);
    const double xe0 = x0 - (n - 1) / 2.0 * h, ye0 = y0 - (m - 1) / 2.0 * h;      // :433-434
    hipLaunchKernelGGL(k_gen_trap2d, dim3(grid_for((int64_t)P0 * P1)), dim3(256), 0, p->stream, G.p, P0, P1, n, m, xe0, ye0, h, p->omega, d0);
    LSFC_HIP(hipGetLastError());
    const size_t len[2] = { (size_t)P0, (size_t)P1 };
    RocFft fwd; fwd.create(2, len, true);
    fwd.exec(G.p, p->stream);                                                       // GFFT = fft(Ge), :179
    LSFC_HIP(hipStreamSynchronize(p->stream));
}

// loads this translation unit's code object on the current device (pruned.hip: pruned_warmup -- every code object of the library is
// resident before the first transfer or pass of a process exists; DESIGN 3, "The round-2 first-apply GPU fault")
__global__ void k_warmup_symbol(int* p) { if (p) *p = 0; }
void warmup_symbol() {
    hipLaunchKernelGGL(k_warmup_symbol, dim3(1), dim3(64), 0, 0, (int*)nullptr);
    LSFC_HIP(hipGetLastError());
}

} // namespace lsfc
