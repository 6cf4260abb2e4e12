// Host-side interface of the hand-written pruned axis passes (fft_kernels.hip).
#pragma once
#include "common.hpp"

namespace lsfc {

// Several right-hand sides through one pipeline (lsfc_apply_batch, the batched GMRES): the x passes take the user
// vectors of all of them (arbitrary pointers), the work arrays hold the batch back to back, and the fused pass loads
// every symbol tile ONCE for the whole batch.
static constexpr int LSFC_MAX_BATCH = 8;
struct VecBatch {
    const cplx* x[LSFC_MAX_BATCH];
    cplx* y[LSFC_MAX_BATCH];
};

struct PrunedTuning {
    bool split_x = true;   // contiguous (x) passes: exchange re/im separately (half the LDS, twice the barriers)
    bool split_s = true;   // strided y passes
    int split_z = -1;      // fused z pass: 1 split, 0 whole complex, -1 auto (by line length)
    int sym_prefetch = -1; // z pass: load the symbol before the forward transform; -1 auto
    int pad1 = -1, pad2 = -1;        // row padding (elements, multiples of 8) of the A1 rows / A2 tile rows, read at plan creation
    int tw_lds = 1;                 // fused pass (full-tile form): stage twiddles from an LDS-resident table instead of product
                                    // trees (-2..3 % on the pass, exact twiddles; profiles/r01_experiment_lds_twiddle_table.log)
    int z_half = -1;                // L = 1024 z pass: half-tile 4-wave workgroups (0 off, 1: full+prefetch, 2: split+prefetch, 3: split 3 WG/CU, 4: full; -1 auto)
    int z_persist = -1;             // fused pass with a z-even symbol in the 3D layout: persistent software-pipelined kernel -- 1 whole-complex
                                    // exchanges, 2 split exchanges, 3 (auto) / 4 the same with the symbol loaded after the first forward
                                    // stage; 0 the one-tile-per-workgroup kernels
    int xlane = -1;                 // fused pass, lines with two consecutive stages of equal radix (8.8: 512, 1024, 1536 points; 4.4: 128, 192,
                                    // 320, 384, 640, 1280): the exchange between them through the lanes of the wavefront instead of LDS
                                    // (1 / -1 auto: on; 0: LDS)
    int ytile_g = 0, ytile_z = 0;   // y passes: block-order tile (x'-groups x z planes); 0 = auto
    int batch_fuse = -1;            // several right-hand sides: 1 one fused pass per group, 0 member by member, -1 auto (fused while
                                    // the padded grid has <= 2^24 points, where launches and not bytes bound the apply:
                                    // profiles/r02_batch_apply.jsonl -- 2.3x per right-hand side at 48^3, -5 % at 256^3)
};

bool pruned_length_supported(int64_t L);
// load EVERY code object of the library on `device` (the current device) once, ahead of the first transfer or pass (fft_kernels.hip)
void pruned_warmup(int device);
// smallest supported line length L >= max(2n, 32): L = 2^k, 3 * 2^k or 5 * 2^k (0: none up to 2048)
int pruned_best_length(int64_t n);
// n (x passes), m (y passes), nin (fused pass): the actual grid size along the transformed axis, <= L/2; entries beyond
// it are treated as zero on the way in and not written on the way back.
// W = chunk width of the x'-storage axis in the xfwd output / xinv input: out[s / W][line][s % W] (W = L on one GPU;
// W = L / nranks packs the slab transpose for free).
// freq_of_storage[s] = frequency index held at storage index s after the forward pass of length L
void pruned_perm(int L, int* freq_of_storage);
PrunedTuning pruned_default_tuning();

// nrhs / vb / *batch: right-hand sides per launch (<= LSFC_MAX_BATCH), their vectors, and the distance between the
// batch members in the work arrays
void pruned_xfwd(int L, const PrunedTuning&, const VecBatch& vb, int nrhs, int64_t obatch, const double* nu, cplx* out, const cplx* tw, int64_t nlines, int W, int Wp, int n, hipStream_t,
                 int64_t bstride = 0);     // bstride: distance between chunks of the storage axis (0: dense, Wp * nlines)
void pruned_xinv(int L, const PrunedTuning&, const cplx* in, const VecBatch& vb, int nrhs, int64_t ibatch, double alpha, double beta,
                 const cplx* tw, int64_t nlines, int W, int Wp, int n, hipStream_t, int64_t bstride = 0);
inline void pruned_xfwd(int L, const PrunedTuning& tn, const cplx* x, const double* nu, cplx* out, const cplx* tw, int64_t nlines, int W, int Wp, int n, hipStream_t st,
                        int64_t bstride = 0) {
    VecBatch vb{}; vb.x[0] = x;
    pruned_xfwd(L, tn, vb, 1, 0, nu, out, tw, nlines, W, Wp, n, st, bstride);
}
inline void pruned_xinv(int L, const PrunedTuning& tn, const cplx* in, const cplx* xorig, cplx* y, double alpha, double beta,
                        const cplx* tw, int64_t nlines, int W, int Wp, int n, hipStream_t st, int64_t bstride = 0) {
    VecBatch vb{}; vb.x[0] = xorig; vb.y[0] = y;
    pruned_xinv(L, tn, in, vb, 1, 0, alpha, beta, tw, nlines, W, Wp, n, st, bstride);
}
// p1: row pitch of A1 (>= Lx), p2: pitch of one storage-y row of an A2 tile (>= 8*l); both multiples of 8 elements
void pruned_yfwd(int L, const PrunedTuning&, const cplx* a1, cplx* a2, const cplx* tw, int Lx, int m, int l, int p1, int p2, hipStream_t,
                 int nrhs = 1, int64_t batch1 = 0, int64_t batch2 = 0);
void pruned_yinv(int L, const PrunedTuning&, const cplx* a2, cplx* a1, const cplx* tw, int Lx, int m, int l, int p1, int p2, hipStream_t,
                 int nrhs = 1, int64_t batch1 = 0, int64_t batch2 = 0);
// full stage-twiddle table of the factorisation used for length L (host side; tw[j] = exp(-2 pi i j / L))
int pruned_twfull_len(int L);
void pruned_twfull(int L, const cplx* tw, cplx* out);
void pruned_zfused(int L, const PrunedTuning&, cplx* data, const cplx* sym, const cplx* tw, const cplx* twl /* full table or NULL */, int Lx, int nouter,
                   int64_t dTile, int64_t dOuter, int64_t dLine, int64_t sTile, int64_t sOuter, int64_t sLine,
                   const int2* ytab /* block order -> (data row, symbol row); NULL: identity */,
                   const int* zm /* z-even symbol: partner storage index of every upper-half slot; NULL: full symbol lines */,
                   int nin /* valid entries per line (<= L/2) */, hipStream_t,
                   int nrhs = 1, int64_t dBatch = 0 /* distance between the batch members in `data` */);

} // namespace lsfc
