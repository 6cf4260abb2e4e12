// Memory-pattern probe for the fused z pass at 512^3 (no arithmetic, no LDS): what would a z-blocked A2 layout
// [xb][z/8][y][z%8][8] -- which the y passes like better (ypass_pattern.hip: -0.08 ms for both) -- cost the pass that moves the
// most bytes?  Today a tile (8 x' lines x 512 z) is ONE contiguous run of 64 KB; z-blocked it is 64 chunks of 1 KB, 1 MB apart.
// The kernel mimics k_zfused_persist's traffic: 512-thread workgroups (lane = line + 8 t), one per CU, walking over tiles; per tile
// every thread loads 8 data values (z = t + 64 e, e < 8), 8 symbol values (contiguous quarter-symbol tile, as today) and stores 8.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/zpass_pattern.hip -o gpurun_out/zpass_pattern ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double2 cplx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k_probe(cplx* __restrict__ a2, const cplx* __restrict__ sym, int ntx, int Ly, int l, int64_t p2, int64_t zb, int hz, int64_t out_shift) {
    const int li = threadIdx.x % 8, t = threadIdx.x / 8;
    const unsigned ntiles = (unsigned)ntx * Ly;
    if constexpr (MODE == 2) {
        // software-pipelined: loads of tile i + 1 are in flight while tile i is stored
        auto base = [&](unsigned tile, cplx*& d, const cplx*& s) {
            const int row = tile % Ly, xb = tile / Ly;
            d = a2 + (int64_t)xb * p2 * Ly + (int64_t)row * p2 + li + 8 * t;
            s = sym + ((int64_t)xb * (Ly / 2 + 1) + (row <= Ly / 2 ? row : Ly - row)) * 8 * hz + li + 8 * t;
        };
        cplx v[8], w[8], nv[8], nw[8];
        cplx* d; const cplx* s;
        unsigned tile = blockIdx.x;
        if (tile >= ntiles) return;
        base(tile, d, s);
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] = d[512 * e]; w[e] = s[512 * e]; }
        for (;;) {
            const unsigned next = tile + gridDim.x;
            cplx* dn = d; const cplx* sn = s;
            if (next < ntiles) {
                base(next, dn, sn);
#pragma unroll
                for (int e = 0; e < 8; ++e) { nv[e] = dn[512 * e]; nw[e] = sn[512 * e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) d[512 * e] = make_double2(v[e].x + w[e].x, v[e].y - w[e].y);
            if (next >= ntiles) break;
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] = nv[e]; w[e] = nw[e]; }
            d = dn; s = sn; tile = next;
        }
        return;
    }
    const unsigned per = (ntiles + gridDim.x - 1) / gridDim.x;
    for (unsigned it = 0; it < per; ++it) {
        // MODE 5: every workgroup walks over a CONTIGUOUS range of tiles instead of tiles b, b + G, b + 2G, ...
        const unsigned tile = MODE == 5 ? blockIdx.x * per + it : blockIdx.x + it * gridDim.x;
        if (tile >= ntiles) break;
        const int row = tile % Ly, xb = tile / Ly;
        cplx* d; int64_t step, off;
        if (MODE != 1) { d = a2 + (int64_t)xb * p2 * Ly + (int64_t)row * p2; off = li + 8 * t; step = 8 * 64; }                       // [xb][y][z][8]
        else { d = a2 + (int64_t)xb * zb * (l / 8) + (int64_t)row * 64; off = li + 8 * (t % 8) + (int64_t)(t / 8) * zb; step = 8 * zb; }   // [xb][z/8][y][z%8][8]
        const cplx* s = sym + ((int64_t)xb * (Ly / 2 + 1) + (row <= Ly / 2 ? row : Ly - row)) * 8 * hz + li + 8 * t;
        cplx v[8], w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = d[off + step * e];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = MODE == 3 ? make_double2(1.0, 2.0) : s[8 * 64 * e];
        cplx* o = MODE == 4 ? d + out_shift : d;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[off + step * e] = make_double2(v[e].x + w[e].x, v[e].y - w[e].y);
    }
}

int main() {
    const int n = 512, Lx = 1024, Ly = 1024, l = n, hz = 512 + 8;
    const int64_t p2 = 8 * l + 72, zb = (int64_t)Ly * 64 + 72;            // (both padded off the power-of-two strides)
    const size_t e2 = (size_t)(p2 * Ly > zb * (l / 8) ? p2 * Ly : zb * (l / 8)) * (Lx / 8), es = (size_t)(Lx / 8) * (Ly / 2 + 1) * 8 * hz;
    cplx *a2, *sym;
    CK(hipMalloc(&a2, 2 * e2 * sizeof(cplx))); CK(hipMalloc(&sym, es * sizeof(cplx)));
    CK(hipMemset(a2, 0, 2 * e2 * sizeof(cplx))); CK(hipMemset(sym, 0, es * sizeof(cplx)));
    hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    const double gb = (4.0 + 4.0 + 2.0) * 16.0 * (double)n * n * n / 1e9;      // data in + out (4 N complex each), quarter symbol (mirror rows: cache)
    auto time = [&](auto kern, const char* name, int grid = 256) {
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, a2, sym, Lx / 8, Ly, l, p2, zb, hz, (int64_t)e2);
        hipEventRecord(s, 0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, a2, sym, Lx / 8, Ly, l, p2, zb, hz, (int64_t)e2);
        hipEventRecord(e, 0); hipEventSynchronize(e);
        float ms = 0; hipEventElapsedTime(&ms, s, e); ms /= 10;
        printf("%-58s %.3f ms  %.2f TB/s of 21.5 GB\n", name, ms, gb / ms);
    };
    for (int rep = 0; rep < 2; ++rep) {
        time(k_probe<0>, "z-pass pattern, A2 [xb][y][z][8] (today: 64-KB tiles)");
        time(k_probe<1>, "z-pass pattern, A2 [xb][z/8][y][z%8][8] (1-KB chunks)");
    }
    // how much of the gap to a streaming copy is memory-level parallelism: more workgroups per CU (the real kernel's exchange buffer
    // allows one), and the next tile's loads issued before this tile's stores
    time(k_probe<0>, "today's layout, 2 workgroups per CU", 512);
    time(k_probe<0>, "today's layout, 4 workgroups per CU", 1024);
    time(k_probe<2>, "today's layout, 1 per CU, next tile's loads before the stores");
    time(k_probe<2>, "today's layout, 2 per CU, next tile's loads before the stores", 512);
    time(k_probe<3>, "today's layout, no symbol stream (17.2 GB)");
    time(k_probe<4>, "today's layout, stores to a second array (not in place)");
    time(k_probe<5>, "today's layout, contiguous tile range per workgroup");
    return 0;
}
