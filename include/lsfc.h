/*
 * lsfc.h -- C ABI of the MI355X-native Lippmann-Schwinger fast-convolution
 * operator (y = x + omega^2 * G * (nu .* x)) and the GMRES loop that drives it.
 *
 * This is the drop-in boundary for ONE hot path of
 * tanderson92/Fast_solver_Lippmann_Schwinger: the FastM / FastM3D operator
 * apply and the IterativeSolvers.gmres! loop.  Every entry point names the
 * reference interface it replaces (paths relative to the reference root).
 * INTEGRATION.md shows the Julia `ccall` stubs a maintainer would add.
 *
 * Conventions (identical to the reference):
 *   - all grid functions are flat vectors in column-major order, x fastest
 *     (examples/example.jl:39-40, examples/example3D.jl:33-39);
 *   - complex numbers are interleaved (re, im) doubles == Julia Complex{Float64};
 *   - every function returns 0 on success, a negative LSFC_E* code on failure;
 *     the message is available through lsfc_last_error(); no C++ exception
 *     crosses this boundary;
 *   - a plan is not thread-safe; calls are synchronous on return for host
 *     buffers and stream-ordered (plan stream) for device buffers.
 *
 * There is no CPU fallback behind this ABI: every compute entry point runs
 * HIP kernels on a gfx950 device and fails with LSFC_ENODEV when none is present.
 */
#ifndef LSFC_H
#define LSFC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lsfc_plan lsfc_plan;

/* error codes */
#define LSFC_OK        0
#define LSFC_EINVAL   -1   /* bad argument (the reference would throw DimensionMismatch / UndefVarError) */
#define LSFC_ENODEV   -2   /* no HIP device */
#define LSFC_ENOMEM   -3
#define LSFC_EHIP     -4   /* HIP / rocFFT / RCCL runtime failure */
#define LSFC_ENOTCONV -5   /* GMRES hit maxiter without converging (x still updated, like gmres!) */

/* quadRule (src/FastConvolution.jl:20, src/FastConvolution3D.jl:21) */
#define LSFC_QUAD_TRAPEZOIDAL     0   /* "trapezoidal"    */
#define LSFC_QUAD_GREENGARD_VICO  1   /* "Greengard_Vico" */

/* where the caller's vectors live */
#define LSFC_MEM_HOST    0
#define LSFC_MEM_DEVICE  1

/* plan flags */
#define LSFC_FLAG_DEFAULT        0u
#define LSFC_FLAG_LITERAL_PAD    1u   /* keep the reference's literal (ne,me,le) padded grid + shifts
                                         (rocFFT path; for parity tests of the padding identity) */
#define LSFC_FLAG_FORCE_ROCFFT   2u   /* reduced grid, but monolithic rocFFT transforms instead of
                                         the hand-written pruned pipeline */
#define LSFC_FLAG_PATCH_SINGULAR 4u   /* builders only: replace the removable 0/0 of the symbol at
                                         |s| == k by its analytic limit (reference yields Inf/NaN) */

/* orthogonalisation (IterativeSolvers.jl orth_meth) */
#define LSFC_ORTH_MGS   0   /* ModifiedGramSchmidt() -- gmres! default */
#define LSFC_ORTH_CGS   1   /* ClassicalGramSchmidt() */
#define LSFC_ORTH_DGKS  2   /* DGKS() */

/* ---- plan construction -------------------------------------------------- */

/* Replaces the constructor FastM(GFFT,nu,ne,me,n,m,k; quadRule)
 * (src/FastConvolution.jl:11-27).  gfft: ne*me interleaved complex, column-major,
 * in the reference's layout: centred (fftshift) order for Greengard_Vico, plain
 * FFT order for trapezoidal.  nu: n*m doubles.  Both are copied (host pointers). */
int lsfc_plan_create_2d(lsfc_plan** out, int64_t n, int64_t m, int64_t ne, int64_t me,
                        const double* nu, const double* gfft, double omega,
                        int quad_rule, unsigned flags, int device);

/* Replaces FastM3D(GFFT,nu,ne,me,le,n,m,l,k; quadRule) (src/FastConvolution3D.jl:7-26). */
int lsfc_plan_create_3d(lsfc_plan** out, int64_t n, int64_t m, int64_t l,
                        int64_t ne, int64_t me, int64_t le,
                        const double* nu, const double* gfft, double omega,
                        int quad_rule, unsigned flags, int device);

/* Replaces buildFastConvolution(x,y,h,k,nu; quadRule="Greengard_Vico")
 * (src/FastConvolution.jl:170-236 + Gtruncated2D, src/Functions.jl:40-42).
 * box = |x[end]-x[1]| + h; nu already evaluated on the grid (n*m doubles).
 * The symbol is generated on the device. */
int lsfc_plan_create_gv2d(lsfc_plan** out, int64_t n, int64_t m, double box, double omega,
                          const double* nu, unsigned flags, int device);

/* Replaces buildFastConvolution(x,y,h,k,nu; quadRule="trapezoidal")
 * (src/FastConvolution.jl:172-183, buildGConv :425-469).  Odd n,m only, as the
 * reference.  x0,y0 = x[1],y[1]; d0 = D[round(Int,k*h)] (re,im). */
int lsfc_plan_create_trap2d(lsfc_plan** out, int64_t n, int64_t m, double x0, double y0, double h,
                            double omega, double d0_re, double d0_im,
                            const double* nu, unsigned flags, int device);

/* Replaces buildFastConvolution3D(x,y,z,X,Y,Z,h,k,nu) (src/FastConvolution3D.jl:68-101
 * + Gtruncated3D, src/Functions.jl:49-51).  The (4n)^3 symbol cube of the
 * reference (137 GB at n=512) is never materialised: the symbol is evaluated
 * slab-wise on the device and reduced to the equivalent (2n)^3 grid.
 * box = |x[end] - x[1]| + h: as in the reference (:72-81), the truncation radius and the frequency lattice of ALL three
 * axes derive from the x extent, so the symbol is the physical truncated kernel only when n == m == l (the reference's
 * own use); for other shapes the reference's arithmetic is reproduced as it is. */
int lsfc_plan_create_gv3d(lsfc_plan** out, int64_t n, int64_t m, int64_t l, double box, double omega,
                          const double* nu, unsigned flags, int device);

int lsfc_plan_destroy(lsfc_plan* plan);

/* ---- operator traits (src/FastConvolution.jl:31-41) ----------------------- */

/* size(M, dim) == length(nu) */
int64_t lsfc_plan_size(const lsfc_plan* plan);
/* dims[0..2] = n,m,l (l = 1 in 2D); pads[0..2] = working padded grid actually used */
int lsfc_plan_dims(const lsfc_plan* plan, int64_t dims[3], int64_t pads[3]);
/* name of the transform pipeline in use: "pruned-hip", "rocfft-reduced", "rocfft-literal" */
const char* lsfc_plan_pipeline(const lsfc_plan* plan);
/* replace the contrast (nu is a mutable field of the reference struct) */
int lsfc_plan_set_nu(lsfc_plan* plan, const double* nu, int memspace);
/* copy the working symbol out (debug / tests): count complex entries written */
int lsfc_plan_get_symbol(const lsfc_plan* plan, double* out, int64_t capacity_complex, int64_t* count);

/* ---- the apply ------------------------------------------------------------ */

/* y = x + omega^2 * FFTconvolution(nu .* x): replaces `*`, `mul!`, `fastconvolution`
 * (src/FastConvolution.jl:43-107) and `*(M::FastM3D,b)` (src/FastConvolution3D.jl:31-37).
 * x, y: N interleaved complex; y may alias x. */
int lsfc_apply(lsfc_plan* plan, const double* x, double* y, int memspace);

/* y = crop(ifft(GFFT .* fft(pad(apply_nu ? nu.*x : x)))): replaces FFTconvolution
 * (src/FastConvolution.jl:110-154, src/FastConvolution3D.jl:39-63).  The reference
 * multiplies by nu in the 2D trapezoidal branch only; the host wrapper passes
 * apply_nu accordingly. */
int lsfc_convolve(lsfc_plan* plan, const double* x, double* y, int apply_nu, int memspace);

/* nrhs independent applies, vectors stored back to back (x + j*N).  Serves the multi-source callers (sampleG3D applies
 * the operator to up to 27 unit vectors at a time, src/FastConvolution3D.jl:146-159) and several incident fields
 * (tests/plasma_example.jl:160-176).  Groups of up to 8 right-hand sides go through ONE pass of the pipeline: the x and
 * y passes run all of them per launch and the fused pass loads each tile of the Green's symbol once per group, so the
 * symbol's share of the HBM traffic (8 of the 35 complex per point in the byte model) is paid once per group instead of
 * once per vector, and small grids amortise their launch latency. */
int lsfc_apply_batch(lsfc_plan* plan, const double* x, double* y, int64_t nrhs, int mode /*0 apply,1 convolve,2 convolve+nu*/, int memspace);

/* Rows of the discrete Green's matrix for the delta sources at grid indices sources[0..nsrc): out + s*N receives
 * FFTconvolution(M, e_{sources[s]}).  Replaces sampleGConv / sampleG3D(..., fastconv) (src/FastConvolution.jl:278-306,
 * src/FastConvolution3D.jl:136-160), which run one full FFT convolution per source (O(10^3) of them in the
 * preconditioner set-up).  Here the spatial kernel is obtained ONCE (one convolution of a unit source at index 0)
 * and every row is a gather K[|i - j|] -- the kernel is even in every axis -- so the cost per source is one
 * N-vector write.  (2D trapezoidal plans: as in the reference's FFTconvolution, nu is NOT applied to a delta row
 * by this entry; use lsfc_apply_batch for that quirk.) */
int lsfc_sample_sources(lsfc_plan* plan, const int64_t* sources, int64_t nsrc, double* out, int memspace);

/* ---- GMRES ---------------------------------------------------------------- */

/* In-place left preconditioner, mirrors the two-argument ldiv!(Pl, v)
 * (src/preconditioner.jl:147-170): v holds N interleaved complex on the HOST and
 * is overwritten with Pl \ v.  Called synchronously on the calling thread once
 * per Arnoldi step (and once per (re)start).  Return non-zero to abort. */
typedef int (*lsfc_precond_fn)(void* user, double* v, int64_t n);

typedef struct lsfc_gmres_opts {
    int     restart;       /* <=0: min(20, N)                       */
    int64_t maxiter;       /* <=0: N                                 */
    double  reltol;        /* <0: sqrt(eps)                          */
    double  abstol;        /* default 0                              */
    int     orth;          /* LSFC_ORTH_*                            */
    int     initially_zero;/* skip the initial A*x0 (x0 == 0)        */
    lsfc_precond_fn precond; void* precond_user;  /* NULL: Identity() */
    int     precond_on_device; /* 0 (default): v is a HOST pointer (copied over PCIe around the call, like the
                                  reference's host-side ldiv!).  1: v is the DEVICE pointer of the Krylov vector
                                  itself; the callback must enqueue its work on the plan's stream (or synchronise)
                                  -- the hook for a device-resident preconditioner, no PCIe traffic. */
} lsfc_gmres_opts;

typedef struct lsfc_gmres_result {
    int64_t iters;         /* inner iterations performed (history.iters)            */
    int64_t mvps;          /* operator applications (history.mvps)                  */
    int     converged;     /* history.isconverged                                   */
    double  final_resnorm; /* last implicit (preconditioned) residual norm          */
} lsfc_gmres_result;

/* Replaces IterativeSolvers.gmres!(x, fastconv, b; Pl, restart, reltol, abstol,
 * maxiter, log=true) as called at examples/example.jl:85,91 and
 * examples/example3D.jl:78.  x (in/out) and b: N complex.  resnorm (may be NULL)
 * receives up to resnorm_cap entries of history[:resnorm]. */
int lsfc_gmres(lsfc_plan* plan, double* x, const double* b, const lsfc_gmres_opts* opts,
               double* resnorm, int64_t resnorm_cap, lsfc_gmres_result* result, int memspace);

/* nrhs independent solves of the same operator in lock step (x, b: nrhs vectors back to back; resnorm: nrhs rows of
 * resnorm_cap entries; results[nrhs]).  Replaces the back-to-back gmres! calls for several incident directions
 * (tests/plasma_example.jl:160-176): each right-hand side keeps its own Krylov basis, Hessenberg matrix and stopping
 * test -- its iterates are those of lsfc_gmres on that right-hand side alone -- but every Arnoldi step applies the
 * operator to all unconverged right-hand sides in one batched pass (see lsfc_apply_batch).  Host preconditioner
 * callbacks are invoked one at a time.  Returns LSFC_OK even if some right-hand side hit maxiter: check
 * results[j].converged.  Device memory: nrhs * (restart + 2) vectors of N complex for the duration of the call (checked
 * against the free memory up front: LSFC_ENOMEM with the figures; released on return). */
int lsfc_gmres_batch(lsfc_plan* plan, double* x, const double* b, int64_t nrhs, const lsfc_gmres_opts* opts,
                     double* resnorm, int64_t resnorm_cap, lsfc_gmres_result* results, int memspace);

/* ---- device-resident SparsifyingPreconditioner apply ------------------------ */

/* The reference's preconditioner (src/preconditioner.jl:27-58) holds two sparse matrices and a sparse LU:
 *     SparsifyingPreconditioner(Msp, As):  MspInv = lu(Msp)          (:35, UMFPACK)
 *     ldiv!(P, b):  b[:] = MspInv \ (As * b)                          (:132-170)
 * lsfc_precond keeps As and the LU factors on the device and applies v <- Msp^{-1} (As v) there (CSR SpMV, two
 * level-scheduled sparse triangular solves replayed from one hipGraph), so the Krylov vector does not cross PCIe.
 * The factorisation stays with the caller (host): with (Rs .* Msp)[p, q] = L U pass
 *     L, U         CSR, 0-based, diagonals stored (L's unit diagonal included), values interleaved complex
 *     row_gather   k -> row of Msp that becomes row k of L U   (Julia: F.p .- 1;  NULL = identity)
 *     col_scatter  k -> column of Msp behind column k of L U   (Julia: F.q .- 1;  NULL = identity)
 *     row_scale    Rs (Julia: F.Rs), NULL = ones
 * As: CSR, 0-based, N rows.  All index arrays are int64_t.
 * An lsfc_precond owns its work vectors: applies on one object are serialised by the stream they are enqueued on;
 * do not apply the same object from two streams or threads at once. */
typedef struct lsfc_precond lsfc_precond;
int lsfc_precond_create(lsfc_precond** out, int64_t N,
                        const int64_t* As_rowptr, const int64_t* As_col, const double* As_val,
                        const int64_t* L_rowptr, const int64_t* L_col, const double* L_val,
                        const int64_t* U_rowptr, const int64_t* U_col, const double* U_val,
                        const int64_t* row_gather, const int64_t* col_scatter, const double* row_scale, int device);
int lsfc_precond_destroy(lsfc_precond* pc);
/* hipStream_t the apply is enqueued on (NULL = legacy default stream); use the plan's stream under lsfc_gmres */
int lsfc_precond_set_stream(lsfc_precond* pc, void* stream);
/* v <- Msp^{-1} (As v); LSFC_MEM_DEVICE: stream-ordered, returns without synchronising */
int lsfc_precond_apply(lsfc_precond* pc, double* v, int memspace);
/* lsfc_precond_fn for lsfc_gmres_opts: precond = lsfc_precond_callback, precond_user = pc, precond_on_device = 1 */
int lsfc_precond_callback(void* user, double* v, int64_t n);
/* dependency levels of the two triangular solves and kernel launches captured in the graph */
int lsfc_precond_stats(const lsfc_precond* pc, int64_t* levels_L, int64_t* levels_U, int64_t* launches);

/* ---- streams, timing, profiling ------------------------------------------ */

/* Run the plan on a caller-owned hipStream_t (NULL = the legacy default stream). */
int lsfc_plan_set_stream(lsfc_plan* plan, void* hip_stream);
/* Block until all work queued by this plan has finished. */
int lsfc_plan_synchronize(lsfc_plan* plan);
/* Time `reps` back-to-back device-resident applies with HIP events recorded on the
 * plan's stream; *ms_total = elapsed milliseconds for all reps. */
int lsfc_time_apply(lsfc_plan* plan, const double* x_dev, double* y_dev, int reps, double* ms_total);
/* Per-kernel timing of one device-resident apply: HIP events around every stage.
 * names[i] (static strings), ms[i], bytes[i] = algorithmic HBM bytes of stage i. */
int lsfc_profile_apply(lsfc_plan* plan, const double* x_dev, double* y_dev, int reps,
                       int max_stages, const char** names, double* ms, double* bytes, int* nstages);

/* Tuning knobs of the pruned pipeline (benchmarks / autotuning), the run-time form of the LSFC_* environment
 * switches of DESIGN.md section 3: "split_x", "split_s" (re/im-split LDS exchanges of the x / y passes), "split_z",
 * "sym_prefetch", "tw_lds", "z_half" (fused pass: split exchanges, symbol prefetch, LDS-resident stage twiddles,
 * half-tile form), "z_persist" (fused pass: 0 one tile per workgroup, 1-4 persistent pipelined whole tiles (6: handed out by per-XCD tickets), 5 ticketed
 * half tiles, -1 by line length), "xlane" (fused pass, lines with two consecutive stages of equal radix -- 8.8: 512, 1024, 1536 points; 4.4: 128, 192, 320, 384, 640, 1280 --
 * the exchange between them through LDS, 0, or through the lanes of the wavefront, 1; 5 = 1 + row pairs as work items, the default where available; 3 = 1 + mirror symbol values from L2), "ytile_g", "ytile_z" (block-order tile of the y passes), "batch_fuse"
 * (lsfc_apply_batch: 1 one fused pass per group of right-hand sides, 0 member by member, -1 by grid size).  Any other
 * key is LSFC_EINVAL.  Results never depend on them beyond rounding (the forms differ in how twiddles are obtained). */
int lsfc_plan_set_tuning(lsfc_plan* plan, const char* key, int value);

/* ---- device memory helpers for hosts without a HIP binding ---------------- */
int lsfc_device_count(int* count);
int lsfc_malloc(void** dptr, size_t bytes, int device);
int lsfc_free(void* dptr);
int lsfc_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int lsfc_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);
/* Page-lock a long-lived host vector (a Krylov work vector of the caller's solver) so that the host-vector forms of
 * lsfc_apply / lsfc_convolve (mul!(Y, M, b), src/FastConvolution.jl:50-54) move it by DMA instead of through the
 * runtime's staging copies.  Optional: pageable vectors work, at about the same rate on MI355X hosts measured so far.
 * Unregister before the memory is freed. */
int lsfc_host_register(void* ptr, size_t bytes);
int lsfc_host_unregister(void* ptr);
/* Page-locked host memory OWNED BY THE RUNTIME (hipHostMalloc): the preferred way to get DMA-able work vectors -- page-aligned,
 * shares no page with other allocations of the process and is not subject to the kernel's page migration, unlike a registered
 * range of the caller's heap (lsfc_host_register pins user pages in place; ranges that share pages with other heap objects are
 * best avoided, see DESIGN.md "host vectors").  A host that cannot adopt foreign memory for its vectors (Julia can:
 * unsafe_wrap) keeps using pageable vectors or lsfc_host_register. */
int lsfc_host_alloc(void** ptr, size_t bytes);
int lsfc_host_free(void* ptr);

/* ---- slab-distributed 3D operator (one process per GPU, RCCL over xGMI) ---- */

/* 128-byte RCCL unique id: rank 0 calls lsfc_dist_unique_id and ships the bytes to
 * the other ranks with whatever the host already has (torch.distributed, MPI, a file). */
#define LSFC_UNIQUE_ID_BYTES 128
int lsfc_dist_unique_id(unsigned char id[LSFC_UNIQUE_ID_BYTES]);

/* z-slab-partitioned counterpart of lsfc_plan_create_gv3d: this rank owns planes
 * k in [rank*l/nranks, (rank+1)*l/nranks) of nu, x and y.  The padded-grid transform is
 * X-pass -> all-to-all -> Y,Z,symbol,Z^-1,Y^-1 -> all-to-all -> X^-1. */
int lsfc_dist_plan_create_gv3d(lsfc_plan** out, int64_t n, int64_t m, int64_t l, double box, double omega,
                               const double* nu_local, unsigned flags, int device,
                               int rank, int nranks, const unsigned char id[LSFC_UNIQUE_ID_BYTES]);

/* Testing aid for hosts with fewer GPUs than ranks: `nranks` logical ranks as separate plans on ONE device, driven
 * in lock step by lsfc_dist_sim_apply, which performs the two slab exchanges with device-to-device copies instead
 * of RCCL.  Exercises exactly the kernels, layouts and symbol slabs of the multi-GPU path.  x[r], y[r]: host
 * vectors of the local size of rank r.  mode: 0 apply, 1 convolve, 2 convolve with nu. */
int lsfc_dist_sim_plan_create_gv3d(lsfc_plan** out, int64_t n, int64_t m, int64_t l, double box, double omega,
                                   const double* nu_local, unsigned flags, int device, int rank, int nranks);
int lsfc_dist_sim_apply(lsfc_plan** plans, int nranks, const double* const* x, double* const* y, int mode);

/* ---- single-process multi-device 3D operator (ONE host thread drives every GPU) ---- */

/* buildFastConvolution3D for a host that is ONE process -- the reference's own situation (a single Julia process,
 * examples/example3D.jl:54,78) -- on `ndev` devices: rank r = devices[r] owns z planes [r*l/ndev, (r+1)*l/ndev).
 * nu: the full n*m*l contrast (host).  The returned plan behaves like any other with HOST vectors: lsfc_apply,
 * lsfc_convolve, lsfc_apply_batch, lsfc_gmres (host preconditioner callback on the whole vector), lsfc_plan_set_nu,
 * lsfc_profile_apply (x_dev / y_dev ignored), lsfc_plan_synchronize, lsfc_plan_destroy; vectors are scattered over the
 * devices' slabs inside the call and the Krylov basis of lsfc_gmres stays distributed on the devices.  The calling
 * thread enqueues the work of all ranks: x pass -> slab exchange -> y, z, y passes -> slab exchange back -> x pass,
 * in K pipeline chunks with the exchanges on side streams, cross-device ordering by events.
 * Transport of the two exchanges: RCCL grouped ncclSend/ncclRecv over ncclCommInitAll communicators (default when the
 * devices are distinct), or peer-to-peer copies issued by the source rank (environment LSFC_MULTI_TRANSPORT=copy; the
 * only choice when a device is listed more than once, which is how this path is tested on a one-GPU machine).
 * ndev: a power of two dividing l and Lx/8. */
int lsfc_plan_create_gv3d_multi(lsfc_plan** out, int64_t n, int64_t m, int64_t l, double box, double omega,
                                const double* nu, unsigned flags, const int* devices, int ndev);
/* ndev, devices[0..ndev), entries per device slab, name of the exchange transport (any pointer may be NULL) */
int lsfc_multi_info(const lsfc_plan* plan, int* ndev, int* devices, int64_t* local_n, const char** transport);
/* The apply on vectors that already live on the devices: x_dev[r], y_dev[r] = slab of rank r in the memory of
 * devices[r].  Stream-ordered on every device (lsfc_plan_synchronize waits for all of them).  mode: 0 apply,
 * 1 convolve, 2 convolve with nu. */
int lsfc_multi_apply_dev(lsfc_plan* plan, const double* const* x_dev, double* const* y_dev, int mode);

/* Padded line length the hand-written pipeline uses for an axis of n grid points: the smallest of 2^k, 3*2^k, 5*2^k
 * (32 ... 2048) that is >= 2n; 0 if there is none (such axes run through rocFFT on the exact 2n grid).  Pure host
 * arithmetic, no device needed.  (The reference pads every axis to 4n, src/FastConvolution3D.jl:48.) */
int lsfc_padded_length(int64_t n);

/* ---- errors ---------------------------------------------------------------- */
const char* lsfc_last_error(void);
const char* lsfc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LSFC_H */
