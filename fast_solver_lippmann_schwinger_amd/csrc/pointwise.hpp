// Host-side interface of pointwise.hip.
#pragma once
#include "common.hpp"

namespace lsfc {

// rocFFT pipelines
void pw_embed(const cplx* x, const double* nu, cplx* W, const int dims[3], const int pads[3], hipStream_t);
void pw_mul_inplace(cplx* W, const cplx* S, int64_t total, hipStream_t);
void pw_crop_axpy(const cplx* W, const cplx* x, cplx* y, double alpha, double beta, const int dims[3], const int pads[3],
                  const int off[3], hipStream_t);
// symbol preparation
void pw_roll_scale(const cplx* src, cplx* dst, const int p[3], const int s[3], double scale, hipStream_t);
void pw_resample_kernel(const cplx* src, cplx* dst, const int p[3], const int q[3], const int origin[3], const int nmax[3], double scale, hipStream_t);
void pw_wrap_crop(const cplx* src, cplx* dst, const int p[3], const int q[3], double scale, hipStream_t);
// 3D: tiles [xb0, xb0+ntiles) of the x' axis only (the symbol slab of one rank); 2D: whole symbol
void pw_permute_symbol(const cplx* G2, cplx* out, const int* px, const int* pyrow, const int* pz, const int L[3], int rows, int hz, int xb0, int ntiles, double scale, hipStream_t,
                       int srcLy = 0 /* rows per z plane of the source when it holds ky, kz >= 0 only (0: L[1]) */);
// max |G(k) - G(L-k)| / max |G| along `axis` of a natural-order symbol (1.0 if it contains NaN)
double pw_mirror_deviation(const cplx* G, const int L[3], int axis, hipStream_t);
void pw_scale(cplx* a, double s, int64_t total, hipStream_t);

void pw_gather_sources(const cplx* K, cplx* out, const int64_t* src, int nsrc, const int dims[3], hipStream_t);

// GMRES BLAS-1 (results land in device scalars; `partial` is scratch of blas_partial_count() entries)
int  blas_partial_count();
void blas_dot(const cplx* a, const cplx* b, cplx* partial, cplx* out, int64_t n, hipStream_t);          // out = a' * b
// defer_sqrt: leave the sum of squares (to be all-reduced across ranks, then blas_sqrt_dev)
void blas_nrm2(const cplx* a, cplx* partial, cplx* out, int64_t n, hipStream_t, bool defer_sqrt = false);  // out.x = ||a||
void blas_sqrt_dev(cplx* s, hipStream_t);
// w -= h[0]*v ; then out = vnext' * w (vnext != NULL) or out.x = ||w|| (vnext == NULL)
void blas_axpy_dot(cplx* w, const cplx* v, const cplx* h, const cplx* vnext, cplx* partial, cplx* out, int64_t n, hipStream_t, bool defer_sqrt = false);
void blas_multidot(const cplx* V, int64_t ldv, int k, const cplx* w, cplx* partial, cplx* out, int64_t n, hipStream_t); // out[j] = V_j' * w
void blas_gemv_acc(cplx* y, const cplx* V, int64_t ldv, int k, const cplx* c, double sign, int64_t n, hipStream_t);      // y += sign * V c
void blas_sub(cplx* y, const cplx* a, const cplx* b, int64_t n, hipStream_t);
// fused single-device Gram-Schmidt steps: the consumer of a reduction sums the producer's block partials itself (no finisher
// launches); `partial` slots are blas_partial_count() / 64 = 1024 entries apart; block 0 publishes the scalars in hout / nout
int  blas_red_blocks(int64_t n);
int  blas_partial_slot();                       // entries between two slots of `partial`
void blas_finish_norm(const cplx* partial, cplx* out, int64_t n, hipStream_t);
void blas_dot_partial(const cplx* a, const cplx* b, cplx* partial, int64_t n, hipStream_t);
void blas_axpy_dot_fused(cplx* w, const cplx* v, const cplx* hpartial, cplx* hout, const cplx* vnext, cplx* partial, int64_t n, hipStream_t);
void blas_multidot_partial(const cplx* V, int64_t ldv, int k, const cplx* w, cplx* partial, int64_t n, hipStream_t);
void blas_cgs_update_fused(cplx* w, const cplx* V, int64_t ldv, int k, const cplx* hpartial, cplx* hout, cplx* npartial, int64_t n, hipStream_t);
void blas_scale_inv_fused(cplx* a, const cplx* npartial, cplx* nout, int64_t n, hipStream_t);
// modified Gram-Schmidt in blocks of blas_mgs_block_size() basis vectors (pointwise.hip: k_mgs_block): 2 + 2 / MB passes over
// N-vectors per basis vector instead of 4
int  blas_mgs_block_size();
int  blas_mgs_slots();                          // partial slots one kernel of the blocked sweep writes / reads
void blas_mgs_block(cplx* w, const cplx* Vp, int mp, const cplx* ppartial, cplx* hout, const cplx* Vn, int mn, cplx* npartial, int64_t ldv, int64_t n, hipStream_t);
void blas_scale_inv_dev(cplx* a, const cplx* s, int64_t n, hipStream_t);                                   // a /= s[0].x

} // namespace lsfc
