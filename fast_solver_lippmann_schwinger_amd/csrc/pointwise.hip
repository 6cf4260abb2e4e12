// Element-wise and reduction kernels (gfx950): padded embed / crop for the rocFFT
// pipelines, symbol preparation at plan creation, and the BLAS-1 pieces of GMRES.
// All are HBM-bound: 16 B per lane accesses, grid-stride loops, wave64 shuffle
// reductions with a fixed-order second stage (bitwise reproducible, no float atomics).
#include "common.hpp"
#include "pointwise.hpp"
#include <algorithm>
#include <vector>

namespace lsfc {

static inline unsigned grid_for(int64_t count, int block = 256, int cap = 256 * 16) {
    int64_t g = (count + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// ---- rocFFT pipelines: embed and crop ---------------------------------------

// W[p0][p1][p2] (x fastest) = (i<n && j<m && k<l) ? (nu?nu:1)*x : 0
__global__ void k_embed(const cplx* __restrict__ x, const double* __restrict__ nu, cplx* __restrict__ W,
                        int n, int m, int l, int p0, int p1, int p2) {
    const int64_t total = (int64_t)p0 * p1 * p2;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % p0); const int64_t r = idx / p0; const int j = (int)(r % p1); const int k = (int)(r / p1);
        cplx v = make_double2(0.0, 0.0);
        if (i < n && j < m && k < l) {
            const int64_t s = i + (int64_t)n * (j + (int64_t)m * k);
            v = x[s];
            if (nu) { const double c = nu[s]; v.x *= c; v.y *= c; }
        }
        W[idx] = v;
    }
}

__global__ void k_mul_inplace(cplx* __restrict__ W, const cplx* __restrict__ S, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const cplx a = W[idx], b = S[idx];
        W[idx] = make_double2(fma(-a.y, b.y, a.x * b.x), fma(a.x, b.y, a.y * b.x));
    }
}

// y = alpha*x + beta*W[o0+i][o1+j][o2+k]
__global__ void k_crop_axpy(const cplx* __restrict__ W, const cplx* x, cplx* y, double alpha, double beta,
                            int n, int m, int l, int p0, int p1, int o0, int o1, int o2) {
    const int64_t total = (int64_t)n * m * l;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % n); const int64_t r = idx / n; const int j = (int)(r % m); const int k = (int)(r / m);
        const cplx w = W[(o0 + i) + (int64_t)p0 * ((o1 + j) + (int64_t)p1 * (o2 + k))];
        cplx v = make_double2(beta * w.x, beta * w.y);
        if (alpha != 0.0) { const cplx xo = x[idx]; v.x = fma(alpha, xo.x, v.x); v.y = fma(alpha, xo.y, v.y); }
        y[idx] = v;
    }
}

void pw_embed(const cplx* x, const double* nu, cplx* W, const int dims[3], const int pads[3], hipStream_t st) {
    const int64_t total = (int64_t)pads[0] * pads[1] * pads[2];
    hipLaunchKernelGGL(k_embed, dim3(grid_for(total)), dim3(256), 0, st, x, nu, W, dims[0], dims[1], dims[2], pads[0], pads[1], pads[2]);
    LSFC_HIP(hipGetLastError());
}
void pw_mul_inplace(cplx* W, const cplx* S, int64_t total, hipStream_t st) {
    hipLaunchKernelGGL(k_mul_inplace, dim3(grid_for(total)), dim3(256), 0, st, W, S, total);
    LSFC_HIP(hipGetLastError());
}
void pw_crop_axpy(const cplx* W, const cplx* x, cplx* y, double alpha, double beta, const int dims[3], const int pads[3],
                  const int off[3], hipStream_t st) {
    const int64_t total = (int64_t)dims[0] * dims[1] * dims[2];
    hipLaunchKernelGGL(k_crop_axpy, dim3(grid_for(total)), dim3(256), 0, st, W, x, y, alpha, beta, dims[0], dims[1], dims[2],
                       pads[0], pads[1], off[0], off[1], off[2]);
    LSFC_HIP(hipGetLastError());
}

// ---- symbol preparation (plan creation only) --------------------------------

// dst[i][j][k] = scale * src[(i+s0)%p0][(j+s1)%p1][(k+s2)%p2]   (ifftshift: s = p/2 for even p, (p+1)/2... caller passes)
__global__ void k_roll_scale(const cplx* __restrict__ src, cplx* __restrict__ dst, int p0, int p1, int p2, int s0, int s1, int s2, double scale) {
    const int64_t total = (int64_t)p0 * p1 * p2;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % p0); const int64_t r = idx / p0; const int j = (int)(r % p1); const int k = (int)(r / p1);
        const int si = (i + s0) % p0, sj = (j + s1) % p1, sk = (k + s2) % p2;
        const cplx v = src[si + (int64_t)p0 * (sj + (int64_t)p1 * sk)];
        dst[idx] = make_double2(scale * v.x, scale * v.y);
    }
}

// Wrap-crop of the spatial kernel: dst on the (q0,q1,q2)=(2n,2m,2l) grid takes offsets
// d in [-q/2, q/2) per axis from the (p0,p1,p2)-periodic src:  dst[d mod q] = scale*src[d mod p].
__global__ void k_wrap_crop(const cplx* __restrict__ src, cplx* __restrict__ dst, int p0, int p1, int p2, int q0, int q1, int q2, double scale) {
    const int64_t total = (int64_t)q0 * q1 * q2;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % q0); const int64_t r = idx / q0; const int j = (int)(r % q1); const int k = (int)(r / q1);
        const int si = (i < q0 / 2 || q0 == 1) ? i : i + (p0 - q0);
        const int sj = (j < q1 / 2 || q1 == 1) ? j : j + (p1 - q1);
        const int sk = (k < q2 / 2 || q2 == 1) ? k : k + (p2 - q2);
        const cplx v = src[si + (int64_t)p0 * (sj + (int64_t)p1 * sk)];
        dst[idx] = make_double2(scale * v.x, scale * v.y);
    }
}

// Re-sample the spatial kernel on the working grid: offset d in [-nmax, nmax] per axis sits at src index
// (d + origin) mod p and goes to dst index d mod q; every other dst entry (offsets the cropped convolution never
// touches) is zero.
__global__ void k_resample_kernel(const cplx* __restrict__ src, cplx* __restrict__ dst, int p0, int p1, int p2, int q0, int q1, int q2,
                                  int o0, int o1, int o2, int n0, int n1, int n2, double scale) {
    const int64_t total = (int64_t)q0 * q1 * q2;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % q0); const int64_t r = idx / q0; const int j = (int)(r % q1); const int k = (int)(r / q1);
        const int d0 = (i <= n0) ? i : i - q0, d1 = (j <= n1) ? j : j - q1, d2 = (k <= n2) ? k : k - q2;
        cplx v = make_double2(0.0, 0.0);
        if (d0 >= -n0 && d1 >= -n1 && d2 >= -n2) {
            const int s0 = ((d0 + o0) % p0 + p0) % p0, s1 = ((d1 + o1) % p1 + p1) % p1, s2 = ((d2 + o2) % p2 + p2) % p2;
            const cplx w = src[s0 + (int64_t)p0 * (s1 + (int64_t)p1 * s2)];
            v = make_double2(scale * w.x, scale * w.y);
        }
        dst[idx] = v;
    }
}
void pw_resample_kernel(const cplx* src, cplx* dst, const int p[3], const int q[3], const int origin[3], const int nmax[3], double scale, hipStream_t st) {
    const int64_t total = (int64_t)q[0] * q[1] * q[2];
    hipLaunchKernelGGL(k_resample_kernel, dim3(grid_for(total)), dim3(256), 0, st, src, dst, p[0], p[1], p[2], q[0], q[1], q[2],
                       origin[0], origin[1], origin[2], nmax[0], nmax[1], nmax[2], scale);
    LSFC_HIP(hipGetLastError());
}

// Natural FFT-order symbol G2[Lx][Ly][Lz] -> storage-order, tile-interleaved layout of the
// pruned pipeline.  3D: out[xi + 8*(sz + Lz*(sy + Ly*xb))];  2D (Lz==1): out[sx + Lx*sy].
// 3D: `rows` symbol rows per tile; pyrow[r] = y frequency of row r (rows == Ly: every storage row; rows == Ly/2+1:
// only the rows with ky <= Ly/2 of a y-even symbol, shared by the mirror rows).
// hz: entries per symbol line.  hz == Lz: all storage slots.  hz == Lz/2 + 8 (z-even symbol): slots j < Lz/2 in
// storage order (frequencies kz < Lz/2), slot Lz/2 = the kz = Lz/2 entry, the rest padding.
__global__ void k_permute_symbol(const cplx* __restrict__ G2, cplx* __restrict__ out, const int* __restrict__ px,
                                 const int* __restrict__ pyrow, const int* __restrict__ pz, int Lx, int Ly, int Lz, int rows, int hz,
                                 int xb0, int ntiles, double scale, int srcLy) {
    const int64_t total = (Lz > 1) ? (int64_t)8 * hz * rows * ntiles : (int64_t)Lx * rows;     // 2D: `rows` symbol rows (pyrow gives their frequency)
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int sx, sy, kz = 0;
        bool pad = false;
        if (Lz > 1) {
            const int xi = (int)(idx % 8); int64_t r = idx / 8;
            const int j = (int)(r % hz); r /= hz; sy = (int)(r % rows); const int xb = (int)(r / rows);
            sx = (xb0 + xb) * 8 + xi;
            if (hz == Lz) kz = pz[j];
            else if (j < Lz / 2) kz = pz[j];
            else if (j == Lz / 2) kz = Lz / 2;
            else pad = true;
        } else { sx = (int)(idx % Lx); sy = (int)(idx / Lx); }
        cplx v = make_double2(0.0, 0.0);
        if (!pad) v = G2[px[sx] + (int64_t)Lx * (pyrow[sy] + (int64_t)srcLy * kz)];       // (srcLy < Ly: the source holds ky, kz >= 0 only)
        out[idx] = make_double2(scale * v.x, scale * v.y);
    }
}

// partial[2b] = max |G[..k..] - G[..(L-k)..]| along `axis`, partial[2b+1] = max |G| over the slice of block b
__global__ void k_mirror_dev(const cplx* __restrict__ G, double* __restrict__ partial, int Lx, int Ly, int Lz, int axis) {
    __shared__ double sh[2][4];
    const int64_t total = (int64_t)Lx * Ly * Lz;
    double dmax = 0.0, amax = 0.0;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % Lx); const int64_t r = idx / Lx; const int j = (int)(r % Ly); const int k = (int)(r / Ly);
        const int mj = axis == 1 ? (Ly - j) % Ly : j, mk = axis == 2 ? (Lz - k) % Lz : k, mi = axis == 0 ? (Lx - i) % Lx : i;
        const cplx a = G[idx], b = G[mi + (int64_t)Lx * (mj + (int64_t)Ly * mk)];
        dmax = fmax(dmax, fmax(fabs(a.x - b.x), fabs(a.y - b.y)));
        amax = fmax(amax, fmax(fabs(a.x), fabs(a.y)));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { dmax = fmax(dmax, __shfl_down(dmax, off, 64)); amax = fmax(amax, __shfl_down(amax, off, 64)); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wave] = dmax; sh[1][wave] = amax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { dmax = fmax(dmax, sh[0][w]); amax = fmax(amax, sh[1][w]); }
        partial[2 * blockIdx.x] = dmax; partial[2 * blockIdx.x + 1] = amax;
    }
}
double pw_mirror_deviation(const cplx* G, const int L[3], int axis, hipStream_t st) {
    const int blocks = 1024;
    DevBuf<double> part; part.alloc(2 * blocks);
    hipLaunchKernelGGL(k_mirror_dev, dim3(blocks), dim3(256), 0, st, G, part.p, L[0], L[1], L[2], axis);
    LSFC_HIP(hipGetLastError());
    std::vector<double> h(2 * blocks);
    LSFC_HIP(hipMemcpyAsync(h.data(), part.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    LSFC_HIP(hipStreamSynchronize(st));
    double d = 0, a = 0;
    for (int b = 0; b < blocks; ++b) { d = std::max(d, h[2 * b]); a = std::max(a, h[2 * b + 1]); }
    if (!(d == d) || !(a == a)) return 1.0;                       // NaN symbol (unpatched singular omega): never treat as even
    return a > 0 ? d / a : 0.0;
}

__global__ void k_scale(cplx* __restrict__ a, double s, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        cplx v = a[idx]; a[idx] = make_double2(s * v.x, s * v.y);
    }
}

void pw_roll_scale(const cplx* src, cplx* dst, const int p[3], const int s[3], double scale, hipStream_t st) {
    const int64_t total = (int64_t)p[0] * p[1] * p[2];
    hipLaunchKernelGGL(k_roll_scale, dim3(grid_for(total)), dim3(256), 0, st, src, dst, p[0], p[1], p[2], s[0], s[1], s[2], scale);
    LSFC_HIP(hipGetLastError());
}
void pw_wrap_crop(const cplx* src, cplx* dst, const int p[3], const int q[3], double scale, hipStream_t st) {
    const int64_t total = (int64_t)q[0] * q[1] * q[2];
    hipLaunchKernelGGL(k_wrap_crop, dim3(grid_for(total)), dim3(256), 0, st, src, dst, p[0], p[1], p[2], q[0], q[1], q[2], scale);
    LSFC_HIP(hipGetLastError());
}
void pw_permute_symbol(const cplx* G2, cplx* out, const int* px, const int* pyrow, const int* pz, const int L[3], int rows, int hz, int xb0, int ntiles, double scale, hipStream_t st, int srcLy) {
    const int64_t total = (L[2] > 1) ? (int64_t)8 * hz * rows * ntiles : (int64_t)L[0] * rows;
    hipLaunchKernelGGL(k_permute_symbol, dim3(grid_for(total)), dim3(256), 0, st, G2, out, px, pyrow, pz, L[0], L[1], L[2], rows, hz, xb0, ntiles, scale, srcLy > 0 ? srcLy : L[1]);
    LSFC_HIP(hipGetLastError());
}
void pw_scale(cplx* a, double s, int64_t total, hipStream_t st) {
    hipLaunchKernelGGL(k_scale, dim3(grid_for(total)), dim3(256), 0, st, a, s, total);
    LSFC_HIP(hipGetLastError());
}

// ---- delta-source sampling (sampleG3D / sampleGConv) ---------------------------------------
// out[s][i] = K[|i0-j0|][|i1-j1|][|i2-j2|] for source s at grid index j: the response to a unit source is the
// spatial kernel shifted to the source, and the kernel is even in every axis, so ONE convolution (source at the
// origin corner) serves every source by a gather.
__global__ void k_gather_sources(const cplx* __restrict__ K, cplx* __restrict__ out, const int64_t* __restrict__ src, int nsrc,
                                 int n, int m, int l) {
    const int64_t N = (int64_t)n * m * l, total = N * nsrc;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(idx / N); const int64_t i = idx % N, j = src[s];
        const int i0 = (int)(i % n), i1 = (int)((i / n) % m), i2 = (int)(i / ((int64_t)n * m));
        const int j0 = (int)(j % n), j1 = (int)((j / n) % m), j2 = (int)(j / ((int64_t)n * m));
        const int d0 = abs(i0 - j0), d1 = abs(i1 - j1), d2 = abs(i2 - j2);
        out[idx] = K[d0 + (int64_t)n * (d1 + (int64_t)m * d2)];
    }
}
void pw_gather_sources(const cplx* K, cplx* out, const int64_t* src, int nsrc, const int dims[3], hipStream_t st) {
    const int64_t total = (int64_t)dims[0] * dims[1] * dims[2] * nsrc;
    hipLaunchKernelGGL(k_gather_sources, dim3(grid_for(total)), dim3(256), 0, st, K, out, src, nsrc, dims[0], dims[1], dims[2]);
    LSFC_HIP(hipGetLastError());
}

// ---- GMRES BLAS-1 ------------------------------------------------------------
// Reductions: each block accumulates a grid-stride slice, reduces across its four
// waves (shuffle, then LDS), and writes one partial; a single-wave finisher sums
// the partials in index order.  RED_BLOCKS is fixed so results do not depend on N
// beyond the slice boundaries -> run-to-run bitwise reproducible.

static constexpr int RED_BLOCKS = 1024;
static constexpr int RED_THREADS = 256;
// blocks actually launched for a vector of n entries: at least ~4 entries per thread (a 48^3 grid gets 108 blocks, not 1024
// blocks of mostly idle threads); a function of n only, so results stay run-to-run reproducible
static inline int red_blocks(int64_t n) {
    const int64_t b = (n + (int64_t)RED_THREADS * 4 - 1) / ((int64_t)RED_THREADS * 4);
    return (int)(b < 1 ? 1 : (b > RED_BLOCKS ? RED_BLOCKS : b));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ cplx block_sum(cplx acc, cplx* sh) {
    acc.x = wave_sum(acc.x); acc.y = wave_sum(acc.y);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = acc;
    __syncthreads();
    cplx r = make_double2(0.0, 0.0);
    if (threadIdx.x == 0) { for (int w = 0; w < RED_THREADS / 64; ++w) { r.x += sh[w].x; r.y += sh[w].y; } }
    return r;   // valid in thread 0
}

// partial[b] = sum_i conj(a[i]) * b[i]   (Julia dot(a, b))
__global__ __launch_bounds__(RED_THREADS) void k_dot_partial(const cplx* __restrict__ a, const cplx* __restrict__ b, cplx* __restrict__ partial, int64_t n) {
    __shared__ cplx sh[RED_THREADS / 64];
    cplx acc = make_double2(0.0, 0.0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const cplx u = a[i], v = b[i];
        acc.x = fma(u.x, v.x, fma(u.y, v.y, acc.x));
        acc.y = fma(u.x, v.y, fma(-u.y, v.x, acc.y));
    }
    const cplx r = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// w -= h[hidx] * v, then partial[b] = sum conj(vnext) * w  (fused MGS step: one pass over w)
__global__ __launch_bounds__(RED_THREADS) void k_axpy_dot_partial(cplx* __restrict__ w, const cplx* __restrict__ v, const cplx* __restrict__ h,
                                                                   const cplx* __restrict__ vnext, cplx* __restrict__ partial, int64_t n) {
    __shared__ cplx sh[RED_THREADS / 64];
    const cplx c = *h;
    cplx acc = make_double2(0.0, 0.0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const cplx q = v[i]; cplx x = w[i];
        x.x -= fma(c.x, q.x, -c.y * q.y);
        x.y -= fma(c.x, q.y, c.y * q.x);
        w[i] = x;
        if (vnext) {
            const cplx u = vnext[i];
            acc.x = fma(u.x, x.x, fma(u.y, x.y, acc.x));
            acc.y = fma(u.x, x.y, fma(-u.y, x.x, acc.y));
        } else {   // last MGS step: accumulate |w|^2 for the norm
            acc.x = fma(x.x, x.x, fma(x.y, x.y, acc.x));
        }
    }
    const cplx r = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// out = sum partial[0..count)   (mode 0) or sqrt(re sum) in out.x (mode 1)
__global__ __launch_bounds__(64) void k_finish(const cplx* __restrict__ partial, int count, cplx* __restrict__ out, int mode) {
    cplx acc = make_double2(0.0, 0.0);
    for (int i = threadIdx.x; i < count; i += 64) { acc.x += partial[i].x; acc.y += partial[i].y; }
    acc.x = wave_sum(acc.x); acc.y = wave_sum(acc.y);
    if (threadIdx.x == 0) { if (mode == 1) acc = make_double2(sqrt(acc.x), 0.0); *out = acc; }
}

// multi-dot for classical Gram-Schmidt: partial[j*RED_BLOCKS + b] = sum conj(V_j) * w for NC columns at once, so w is
// read once per group of NC columns instead of once per column
template <int NC>
__global__ __launch_bounds__(RED_THREADS) void k_multidot_partial(const cplx* __restrict__ V, int64_t ldv, int k, const cplx* __restrict__ w,
                                                                   cplx* __restrict__ partial, int64_t n) {
    __shared__ cplx sh[RED_THREADS / 64];
    cplx acc[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) acc[j] = make_double2(0.0, 0.0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const cplx v = w[i];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            if (j < k) {
                const cplx u = V[(int64_t)j * ldv + i];
                acc[j].x = fma(u.x, v.x, fma(u.y, v.y, acc[j].x));
                acc[j].y = fma(u.x, v.y, fma(-u.y, v.x, acc[j].y));
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        if (j < k) {
            const cplx r = block_sum(acc[j], sh);
            if (threadIdx.x == 0) partial[(int64_t)j * RED_BLOCKS + blockIdx.x] = r;
            __syncthreads();
        }
    }
}
__global__ __launch_bounds__(64) void k_multifinish(const cplx* __restrict__ partial, int count, cplx* __restrict__ out) {
    const cplx* p = partial + (int64_t)blockIdx.x * RED_BLOCKS;
    cplx acc = make_double2(0.0, 0.0);
    for (int i = threadIdx.x; i < count; i += 64) { acc.x += p[i].x; acc.y += p[i].y; }
    acc.x = wave_sum(acc.x); acc.y = wave_sum(acc.y);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

// y += sign * sum_j c[j] * V_j   (k <= 64 columns; coefficients in device memory)
__global__ void k_gemv_acc(cplx* __restrict__ y, const cplx* __restrict__ V, int64_t ldv, int k, const cplx* __restrict__ c, double sign, int64_t n) {
    __shared__ cplx cs[64];
    if (threadIdx.x < k) cs[threadIdx.x] = c[threadIdx.x];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        cplx acc = make_double2(0.0, 0.0);
        for (int j = 0; j < k; ++j) {
            const cplx q = V[(int64_t)j * ldv + i], cj = cs[j];
            acc.x += fma(cj.x, q.x, -cj.y * q.y);
            acc.y += fma(cj.x, q.y, cj.y * q.x);
        }
        cplx v = y[i]; v.x = fma(sign, acc.x, v.x); v.y = fma(sign, acc.y, v.y); y[i] = v;
    }
}

// y = a - b
__global__ void k_sub(cplx* __restrict__ y, const cplx* __restrict__ a, const cplx* __restrict__ b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const cplx u = a[i], v = b[i]; y[i] = make_double2(u.x - v.x, u.y - v.y);
    }
}
// a *= 1 / s[0].x   (s on the device: no host round trip between the norm and the scaling)
__global__ void k_scale_inv_dev(cplx* __restrict__ a, const cplx* __restrict__ s, int64_t n) {
    const double inv = 1.0 / s->x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        cplx v = a[i]; a[i] = make_double2(v.x * inv, v.y * inv);
    }
}

// ---- fused single-device Gram-Schmidt steps -----------------------------------------------------------------------
// On one device no all-reduce sits between a reduction and its consumer, so the finisher launches disappear: every block
// of the CONSUMING kernel sums the producer's block partials itself (<= 1024 values from L2, in exactly the order
// k_finish uses, so every block -- and the host -- sees bit-identical scalars), and block 0 publishes the scalar for the
// host.  A modified Gram-Schmidt sweep over k vectors is then k + 2 launches instead of 2k + 4, a classical one 3.
__device__ __forceinline__ cplx finish_in_wave(const cplx* __restrict__ partial, int count) {      // all 64 lanes of one wave
    cplx acc = make_double2(0.0, 0.0);
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < count; i += 64) { acc.x += partial[i].x; acc.y += partial[i].y; }
    acc.x = wave_sum(acc.x); acc.y = wave_sum(acc.y);
    return make_double2(__shfl(acc.x, 0, 64), __shfl(acc.y, 0, 64));
}

// w -= h * v with h = sum(hpartial[0..nb)); then partial[b] = sum conj(vnext) * w (or |w|^2 when vnext == NULL); block 0: *hout = h
__global__ __launch_bounds__(RED_THREADS) void k_axpy_dot_fused(cplx* __restrict__ w, const cplx* __restrict__ v, const cplx* __restrict__ hpartial, int nb,
                                                                 cplx* __restrict__ hout, const cplx* __restrict__ vnext, cplx* __restrict__ partial, int64_t n) {
    __shared__ cplx sh[RED_THREADS / 64];
    __shared__ cplx hs;
    if (threadIdx.x < 64) { const cplx h = finish_in_wave(hpartial, nb); if (threadIdx.x == 0) { hs = h; if (blockIdx.x == 0) *hout = h; } }
    __syncthreads();
    const cplx c = hs;
    cplx acc = make_double2(0.0, 0.0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const cplx q = v[i]; cplx x = w[i];
        x.x -= fma(c.x, q.x, -c.y * q.y);
        x.y -= fma(c.x, q.y, c.y * q.x);
        w[i] = x;
        if (vnext) {
            const cplx u = vnext[i];
            acc.x = fma(u.x, x.x, fma(u.y, x.y, acc.x));
            acc.y = fma(u.x, x.y, fma(-u.y, x.x, acc.y));
        } else {
            acc.x = fma(x.x, x.x, fma(x.y, x.y, acc.x));
        }
    }
    const cplx r = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// w -= V h with h_j = sum(hpartial[j*RED_BLOCKS + 0..nb)), j < k <= 64; partial[b] = |w|^2 slice; block 0: hout[j] = h_j
__global__ __launch_bounds__(RED_THREADS) void k_cgs_update_fused(cplx* __restrict__ w, const cplx* __restrict__ V, int64_t ldv, int k,
                                                                   const cplx* __restrict__ hpartial, int nb, cplx* __restrict__ hout,
                                                                   cplx* __restrict__ partial, int64_t n) {
    __shared__ cplx sh[RED_THREADS / 64];
    __shared__ cplx cs[64];
    for (int j = threadIdx.x >> 6; j < k; j += RED_THREADS / 64) {
        const cplx h = finish_in_wave(hpartial + (int64_t)j * RED_BLOCKS, nb);
        if ((threadIdx.x & 63) == 0) { cs[j] = h; if (blockIdx.x == 0) hout[j] = h; }
    }
    __syncthreads();
    cplx nrm = make_double2(0.0, 0.0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        cplx acc = make_double2(0.0, 0.0);
        for (int j = 0; j < k; ++j) {
            const cplx q = V[(int64_t)j * ldv + i], cj = cs[j];
            acc.x += fma(cj.x, q.x, -cj.y * q.y);
            acc.y += fma(cj.x, q.y, cj.y * q.x);
        }
        cplx x = w[i]; x.x -= acc.x; x.y -= acc.y; w[i] = x;
        nrm.x = fma(x.x, x.x, fma(x.y, x.y, nrm.x));
    }
    const cplx r = block_sum(nrm, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// a *= 1 / sqrt(sum(npartial[0..nb)).x); block 0: *nout = (norm, 0)
__global__ __launch_bounds__(RED_THREADS) void k_scale_inv_fused(cplx* __restrict__ a, const cplx* __restrict__ npartial, int nb, cplx* __restrict__ nout, int64_t n) {
    __shared__ double ns;
    if (threadIdx.x < 64) { const cplx s2 = finish_in_wave(npartial, nb); if (threadIdx.x == 0) { ns = sqrt(s2.x); if (blockIdx.x == 0) *nout = make_double2(ns, 0.0); } }
    __syncthreads();
    const double inv = 1.0 / ns;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        cplx v = a[i]; a[i] = make_double2(v.x * inv, v.y * inv);
    }
}

// Modified Gram-Schmidt in blocks of MB basis vectors (large vectors: the sweep is pure HBM traffic).  Strict MGS takes the
// inner product with v_{i+1} only after w has been updated with v_i: four passes over N-vectors per basis vector (read w, v_i,
// v_{i+1}, write w).  But  <v_j, w - sum_{l<j} h_l v_l> = <v_j, w> - sum_{l<j} h_l <v_j, v_l>,  so ONE pass over w and the MB
// vectors of a block yields the MB inner products with the not-yet-updated w plus the MB (MB - 1) / 2 inner products of the
// block's vectors with each other (they cost no memory traffic: the vectors are in registers anyway), from which every
// workgroup of the next kernel recovers the MGS coefficients h_0 .. h_{MB-1} by the recurrence above -- the same numbers
// modified Gram-Schmidt produces, up to the order of rounding.  That kernel updates w with the whole block and, in the same
// pass, takes the inner products with the NEXT block (or |w|^2 after the last).  Passes per basis vector: 2 + 2 / MB (MB = 4:
// 2.5 instead of 4).  Partials: slot base + j (j < MB) = <v_j, w>, base + MB + pair(j, l) = <v_j, v_l> for l < j.
template <int MB> __device__ __forceinline__ constexpr int mgs_pair(int j, int l) { return j * (j - 1) / 2 + l; }     // l < j
template <int MB>
__global__ __launch_bounds__(RED_THREADS) void k_mgs_block(cplx* __restrict__ w, const cplx* __restrict__ Vp, int mp, const cplx* __restrict__ ppartial, int nb,
                                                            cplx* __restrict__ hout, const cplx* __restrict__ Vn, int mn, cplx* __restrict__ npartial,
                                                            int64_t ldv, int64_t n) {
    constexpr int NP = MB * (MB - 1) / 2;
    __shared__ cplx sh[RED_THREADS / 64];
    __shared__ cplx fin[MB + NP];
    __shared__ cplx hs[MB];
    if (mp > 0) {
        const int ns = mp + mp * (mp - 1) / 2;             // (slots of a short block are numbered as those of a full one: pair(j, l) does not depend on mp)
        for (int q = threadIdx.x >> 6; q < MB + NP; q += RED_THREADS / 64) {
            const bool used = q < mp || (q >= MB && q - MB < mp * (mp - 1) / 2);
            if (used) { const cplx v = finish_in_wave(ppartial + (int64_t)q * RED_BLOCKS, nb); if ((threadIdx.x & 63) == 0) fin[q] = v; }
        }
        (void)ns;
        __syncthreads();
        if (threadIdx.x == 0) {
            cplx h[MB];
            for (int j = 0; j < mp; ++j) {
                cplx a = fin[j];
                for (int l = 0; l < j; ++l) {               // a -= h_l * <v_j, v_l>
                    const cplx g = fin[MB + mgs_pair<MB>(j, l)];
                    a.x -= fma(h[l].x, g.x, -h[l].y * g.y);
                    a.y -= fma(h[l].x, g.y, h[l].y * g.x);
                }
                h[j] = a; hs[j] = a;
                if (blockIdx.x == 0) hout[j] = a;
            }
        }
        __syncthreads();
    }
    cplx c[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) c[j] = j < mp ? hs[j] : make_double2(0.0, 0.0);
    cplx ad[MB], ag[NP > 0 ? NP : 1];
#pragma unroll
    for (int j = 0; j < MB; ++j) ad[j] = make_double2(0.0, 0.0);
#pragma unroll
    for (int j = 0; j < NP; ++j) ag[j] = make_double2(0.0, 0.0);
    // two elements per thread and trip (i and i + stride): twice the loads in flight per wave
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    auto element = [&](int64_t i) {
        cplx x = w[i];
        if (mp > 0) {
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                if (j < mp) {
                    const cplx q = Vp[(int64_t)j * ldv + i];
                    x.x -= fma(c[j].x, q.x, -c[j].y * q.y);
                    x.y -= fma(c[j].x, q.y, c[j].y * q.x);
                }
            }
            w[i] = x;
        }
        if (mn > 0) {
            cplx u[MB];
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                if (j < mn) {
                    u[j] = Vn[(int64_t)j * ldv + i];
                    ad[j].x = fma(u[j].x, x.x, fma(u[j].y, x.y, ad[j].x));
                    ad[j].y = fma(u[j].x, x.y, fma(-u[j].y, x.x, ad[j].y));
#pragma unroll
                    for (int l = 0; l < j; ++l) {           // <v_j, v_l> = sum conj(v_j) v_l
                        cplx& g = ag[mgs_pair<MB>(j, l)];
                        g.x = fma(u[j].x, u[l].x, fma(u[j].y, u[l].y, g.x));
                        g.y = fma(u[j].x, u[l].y, fma(-u[j].y, u[l].x, g.y));
                    }
                }
            }
        } else {
            ad[0].x = fma(x.x, x.x, fma(x.y, x.y, ad[0].x));            // after the last block: |w|^2
        }
    };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 2 * stride) {
        element(i);
        if (i + stride < n) element(i + stride);
    }
#pragma unroll
    for (int q = 0; q < MB + NP; ++q) {
        const bool used = mn > 0 ? (q < mn || (q >= MB && q - MB < mn * (mn - 1) / 2)) : q == 0;
        if (used) {
            const cplx r = block_sum(q < MB ? ad[q] : ag[q - MB], sh);
            if (threadIdx.x == 0) npartial[(int64_t)q * RED_BLOCKS + blockIdx.x] = r;
            __syncthreads();
        }
    }
}
// one kernel of the blocked sweep: w -= Vp h (h from the partials `ppartial` of the previous kernel, published in hout; mp = 0:
// nothing to subtract, w is only read), then the partials of the inner products with the mn <= MB vectors Vn (mn = 0: of |w|^2,
// slot 0) into `npartial`; both partial areas hold blas_mgs_slots() slots
int blas_mgs_block_size() { return 4; }
int blas_mgs_slots() { return 4 + 6; }
void blas_mgs_block(cplx* w, const cplx* Vp, int mp, const cplx* ppartial, cplx* hout, const cplx* Vn, int mn, cplx* npartial, int64_t ldv, int64_t n, hipStream_t st) {
    LSFC_REQUIRE(mp >= 0 && mp <= 4 && mn >= 0 && mn <= 4, "mgs block: at most 4 vectors per block");
    hipLaunchKernelGGL(k_mgs_block<4>, dim3(red_blocks(n)), dim3(RED_THREADS), 0, st, w, Vp, mp, ppartial, red_blocks(n), hout, Vn, mn, npartial, ldv, n);
    LSFC_HIP(hipGetLastError());
}

int blas_red_blocks(int64_t n) { return red_blocks(n); }
int blas_partial_slot() { return RED_BLOCKS; }
// out.x = sqrt(sum partial[0..nb).x): the norm from its block partials (the DGKS test needs it on the host before scaling)
void blas_finish_norm(const cplx* partial, cplx* out, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(64), 0, st, partial, red_blocks(n), out, 1);
    LSFC_HIP(hipGetLastError());
}
void blas_dot_partial(const cplx* a, const cplx* b, cplx* partial, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_dot_partial, dim3(red_blocks(n)), dim3(RED_THREADS), 0, st, a, b, partial, n);
    LSFC_HIP(hipGetLastError());
}
void blas_axpy_dot_fused(cplx* w, const cplx* v, const cplx* hpartial, cplx* hout, const cplx* vnext, cplx* partial, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_axpy_dot_fused, dim3(red_blocks(n)), dim3(RED_THREADS), 0, st, w, v, hpartial, red_blocks(n), hout, vnext, partial, n);
    LSFC_HIP(hipGetLastError());
}
void blas_multidot_partial(const cplx* V, int64_t ldv, int k, const cplx* w, cplx* partial, int64_t n, hipStream_t st) {
    constexpr int NC = 8;
    const int nb = red_blocks(n);
    for (int j0 = 0; j0 < k; j0 += NC) {
        const int kc = k - j0 < NC ? k - j0 : NC;
        hipLaunchKernelGGL(k_multidot_partial<NC>, dim3(nb), dim3(RED_THREADS), 0, st, V + (int64_t)j0 * ldv, ldv, kc, w, partial + (int64_t)j0 * RED_BLOCKS, n);
    }
    LSFC_HIP(hipGetLastError());
}
void blas_cgs_update_fused(cplx* w, const cplx* V, int64_t ldv, int k, const cplx* hpartial, cplx* hout, cplx* npartial, int64_t n, hipStream_t st) {
    LSFC_REQUIRE(k <= 64, "cgs_update: at most 64 columns");
    hipLaunchKernelGGL(k_cgs_update_fused, dim3(red_blocks(n)), dim3(RED_THREADS), 0, st, w, V, ldv, k, hpartial, red_blocks(n), hout, npartial, n);
    LSFC_HIP(hipGetLastError());
}
void blas_scale_inv_fused(cplx* a, const cplx* npartial, cplx* nout, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_scale_inv_fused, dim3(grid_for(n)), dim3(RED_THREADS), 0, st, a, npartial, red_blocks(n), nout, n);
    LSFC_HIP(hipGetLastError());
}

void blas_dot(const cplx* a, const cplx* b, cplx* partial, cplx* out, int64_t n, hipStream_t st) {
    const int nb = red_blocks(n);
    hipLaunchKernelGGL(k_dot_partial, dim3(nb), dim3(RED_THREADS), 0, st, a, b, partial, n);
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(64), 0, st, partial, nb, out, 0);
    LSFC_HIP(hipGetLastError());
}
__global__ void k_sqrt_dev(cplx* s) { if (threadIdx.x == 0) *s = make_double2(sqrt(s->x), 0.0); }
void blas_sqrt_dev(cplx* s, hipStream_t st) {
    hipLaunchKernelGGL(k_sqrt_dev, dim3(1), dim3(64), 0, st, s);
    LSFC_HIP(hipGetLastError());
}
void blas_nrm2(const cplx* a, cplx* partial, cplx* out, int64_t n, hipStream_t st, bool defer_sqrt) {
    const int nb = red_blocks(n);
    hipLaunchKernelGGL(k_dot_partial, dim3(nb), dim3(RED_THREADS), 0, st, a, a, partial, n);
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(64), 0, st, partial, nb, out, defer_sqrt ? 0 : 1);
    LSFC_HIP(hipGetLastError());
}
void blas_axpy_dot(cplx* w, const cplx* v, const cplx* h, const cplx* vnext, cplx* partial, cplx* out, int64_t n, hipStream_t st, bool defer_sqrt) {
    const int nb = red_blocks(n);
    hipLaunchKernelGGL(k_axpy_dot_partial, dim3(nb), dim3(RED_THREADS), 0, st, w, v, h, vnext, partial, n);
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(64), 0, st, partial, nb, out, (vnext || defer_sqrt) ? 0 : 1);
    LSFC_HIP(hipGetLastError());
}
void blas_multidot(const cplx* V, int64_t ldv, int k, const cplx* w, cplx* partial, cplx* out, int64_t n, hipStream_t st) {
    constexpr int NC = 8;
    const int nb = red_blocks(n);
    for (int j0 = 0; j0 < k; j0 += NC) {
        const int kc = k - j0 < NC ? k - j0 : NC;
        hipLaunchKernelGGL(k_multidot_partial<NC>, dim3(nb), dim3(RED_THREADS), 0, st, V + (int64_t)j0 * ldv, ldv, kc, w,
                           partial + (int64_t)j0 * RED_BLOCKS, n);
    }
    hipLaunchKernelGGL(k_multifinish, dim3(k), dim3(64), 0, st, partial, nb, out);
    LSFC_HIP(hipGetLastError());
}
void blas_gemv_acc(cplx* y, const cplx* V, int64_t ldv, int k, const cplx* c, double sign, int64_t n, hipStream_t st) {
    LSFC_REQUIRE(k <= 64, "gemv_acc: at most 64 columns");
    hipLaunchKernelGGL(k_gemv_acc, dim3(grid_for(n)), dim3(256), 0, st, y, V, ldv, k, c, sign, n);
    LSFC_HIP(hipGetLastError());
}
void blas_sub(cplx* y, const cplx* a, const cplx* b, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_sub, dim3(grid_for(n)), dim3(256), 0, st, y, a, b, n);
    LSFC_HIP(hipGetLastError());
}
void blas_scale_inv_dev(cplx* a, const cplx* s, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_scale_inv_dev, dim3(grid_for(n)), dim3(256), 0, st, a, s, n);
    LSFC_HIP(hipGetLastError());
}
int blas_partial_count() { return RED_BLOCKS * 66; }   // 64 slots of a multidot + two more (norm partials, alternating slots of the fused sweeps)

// loads this translation unit's code object on the current device (pruned.hip: pruned_warmup -- every code object of the library is
// resident before the first transfer or pass of a process exists; DESIGN 3, "The round-2 first-apply GPU fault")
__global__ void k_warmup_pointwise(int* p) { if (p) *p = 0; }
void warmup_pointwise() {
    hipLaunchKernelGGL(k_warmup_pointwise, dim3(1), dim3(64), 0, 0, (int*)nullptr);
    LSFC_HIP(hipGetLastError());
}

} // namespace lsfc
