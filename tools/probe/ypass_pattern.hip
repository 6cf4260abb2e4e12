// Memory-pattern probe for the y passes at 512^3 (no arithmetic): a workgroup reads its 8 x' lines of one z plane from A1
// (128-B chunks, one per row y) and writes 1024 chunks of 128 B to the tiled array A2 -- in today's layout
// [xb][y][z][8] (chunks of one workgroup 66.7 KB apart; the TZ = 8 workgroups of a block-order tile fill 1 KB runs) and in
// a z-blocked layout [xb][z/8][y][z%8][8] (chunks of one workgroup 1 KB apart; the same 8 workgroups fill 1 MB runs).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/ypass_pattern.hip -o gpurun_out/ypass_pattern ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef double2 cplx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, bool INV>
__global__ __launch_bounds__(512) void k_probe(const cplx* __restrict__ a1, cplx* __restrict__ a2, int Lx, int m, int l, int Ly, int p1, int64_t p2) {
    const int TG = 32, TZ = 8;
    const int xi = threadIdx.x % 8, t = threadIdx.x / 8;
    const int ngrp = Lx / 8;
    const int within = blockIdx.x % (TG * TZ), tile = blockIdx.x / (TG * TZ);
    const int ntg = ngrp / TG;
    const int g = (tile % ntg) * TG + within % TG, z = (tile / ntg) * TZ + within / TG;
    const int xp = g * 8 + xi, xb = g;
    // MODE 2: A1 side contiguous per workgroup (64 KB), A2 as today; MODE 3: A1 as today, A2 side contiguous per workgroup
    // (128 KB); MODE 4: both contiguous (a streaming copy with this kernel's shape)
    const bool c1 = MODE == 2 || MODE == 4, c2 = MODE == 3 || MODE == 4;
    cplx* r1 = c1 ? const_cast<cplx*>(a1) + (int64_t)blockIdx.x * 4096 + xi : const_cast<cplx*>(a1) + xp + (int64_t)p1 * m * z;
    const int64_t s1 = c1 ? 8 : p1;
    cplx v[16];
    auto a2at = [&](int y) -> cplx* {
        if (c2) return a2 + xi + (int64_t)blockIdx.x * 8192 + 8 * y;
        if (MODE == 0 || MODE == 2) return a2 + xi + (int64_t)8 * z + p2 * ((int64_t)Ly * xb + y);
        return a2 + xi + 8 * (z % 8) + (int64_t)64 * (y + (int64_t)Ly * ((z / 8) + (int64_t)(l / 8) * xb));
    };
    if (!INV) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = r1[s1 * (t + 64 * e)];
#pragma unroll
        for (int e = 0; e < 16; ++e) *a2at(t + 64 * e) = v[e % 8];
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = *a2at(t + 64 * e);
#pragma unroll
        for (int e = 0; e < 8; ++e) r1[s1 * (t + 64 * e)] = make_double2(v[e].x + v[e + 8].x, v[e].y + v[e + 8].y);
    }
}


// Wider workgroups (round 3): 1024 threads moving either 16 x' lines of one plane (256-B chunks on the A1 side, two A2 tiles) or 8 x'
// lines of TWO consecutive planes (128-B chunks from two planes, 256-B runs on the A2 side) -- what LINES = 16 in the y passes would buy
template <int SHAPE, bool INV>
__global__ __launch_bounds__(1024) void k_probe_wide(const cplx* __restrict__ a1, cplx* __restrict__ a2, int Lx, int m, int l, int Ly, int p1, int64_t p2) {
    const int TG = 32, TZ = 8;
    const int xi = threadIdx.x % 16, t = threadIdx.x / 16;
    // SHAPE 0: xi = 16 x'; the grid covers (Lx / 16) groups x l planes.  SHAPE 1: xi = (z parity, 8 x'); (Lx / 8) groups x (l / 2) plane pairs
    const int ngrp = SHAPE == 0 ? Lx / 16 : Lx / 8, nz = SHAPE == 0 ? l : l / 2;
    const int tg = SHAPE == 0 ? TG / 2 : TG, tz = SHAPE == 0 ? TZ : TZ / 2;
    const int within = blockIdx.x % (tg * tz), tile = blockIdx.x / (tg * tz);
    const int ntg = ngrp / tg;
    const int g = (tile % ntg) * tg + within % tg, zz = (tile / ntg) * tz + within / tg;
    (void)nz;
    const int xp = SHAPE == 0 ? g * 16 + xi : g * 8 + (xi % 8);
    const int z = SHAPE == 0 ? zz : 2 * zz + xi / 8;
    const int xb = xp / 8, xq = xp % 8;
    cplx* r1 = const_cast<cplx*>(a1) + xp + (int64_t)p1 * m * z;
    cplx v[16];
    auto a2at = [&](int y) -> cplx* { return a2 + xq + (int64_t)8 * z + p2 * ((int64_t)Ly * xb + y); };
    if (!INV) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = r1[(int64_t)p1 * (t + 64 * e)];
#pragma unroll
        for (int e = 0; e < 16; ++e) *a2at(t + 64 * e) = v[e % 8];
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = *a2at(t + 64 * e);
#pragma unroll
        for (int e = 0; e < 8; ++e) r1[(int64_t)p1 * (t + 64 * e)] = make_double2(v[e].x + v[e + 8].x, v[e].y + v[e + 8].y);
    }
}

int main() {
    const int n = 512, Lx = 1024, Ly = 1024, m = n, l = n, p1 = Lx + 40;
    const int64_t p2 = 8 * l + 72;
    const size_t e1 = (size_t)p1 * m * l, e2 = (size_t)p2 * Ly * (Lx / 8);
    cplx *a1, *a2;
    CK(hipMalloc(&a1, e1 * sizeof(cplx))); CK(hipMalloc(&a2, e2 * sizeof(cplx)));
    CK(hipMemset(a1, 0, e1 * sizeof(cplx))); CK(hipMemset(a2, 0, e2 * sizeof(cplx)));
    hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    const dim3 grid((Lx / 8) * l), block(512);
    const double gb = 6.0 * 16.0 * (double)n * n * n / 1e9;
    auto time = [&](auto kern, const char* name) {
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, grid, block, 0, 0, a1, a2, Lx, m, l, Ly, p1, p2);
        hipEventRecord(s, 0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, grid, block, 0, 0, a1, a2, Lx, m, l, Ly, p1, p2);
        hipEventRecord(e, 0); hipEventSynchronize(e);
        float ms = 0; hipEventElapsedTime(&ms, s, e); ms /= 10;
        printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, gb / ms);
    };
    time(k_probe<0, false>, "yfwd pattern, layout [xb][y][z][8] (today)");
    time(k_probe<1, false>, "yfwd pattern, layout [xb][z/8][y][z%8][8]");
    time(k_probe<2, false>, "yfwd: A1 side contiguous, A2 as today");
    time(k_probe<3, false>, "yfwd: A1 as today, A2 side contiguous");
    time(k_probe<4, false>, "yfwd shape, both sides contiguous (copy)");
    auto timew = [&](auto kern, const char* name) {
        const dim3 gw((Lx / 16) * l), bw(1024);
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, gw, bw, 0, 0, a1, a2, Lx, m, l, Ly, p1, p2);
        hipEventRecord(s, 0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, gw, bw, 0, 0, a1, a2, Lx, m, l, Ly, p1, p2);
        hipEventRecord(e, 0); hipEventSynchronize(e);
        float ms = 0; hipEventElapsedTime(&ms, s, e); ms /= 10;
        printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, gb / ms);
    };
    timew(k_probe_wide<0, false>, "yfwd pattern, 16 x' per workgroup");
    timew(k_probe_wide<1, false>, "yfwd pattern, 8 x' x 2 planes per workgroup");
    timew(k_probe_wide<0, true>, "yinv pattern, 16 x' per workgroup");
    timew(k_probe_wide<1, true>, "yinv pattern, 8 x' x 2 planes per workgroup");
    time(k_probe<0, true>, "yinv pattern, layout [xb][y][z][8] (today)");
    time(k_probe<1, true>, "yinv pattern, layout [xb][z/8][y][z%8][8]");
    time(k_probe<2, true>, "yinv: A1 side contiguous, A2 as today");
    time(k_probe<3, true>, "yinv: A1 as today, A2 side contiguous");
    time(k_probe<4, true>, "yinv shape, both sides contiguous (copy)");
    return 0;
}
