// The plan object behind the lsfc C ABI.
#pragma once
#include "common.hpp"
#include "pruned.hpp"
#include <rocfft/rocfft.h>
#include <functional>
#include <memory>
#include <vector>

namespace lsfc {

// Thin RAII wrapper of one in-place complex-double rocFFT transform.
struct RocFft {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    DevBuf<char> work;
    RocFft() = default;
    RocFft(const RocFft&) = delete; RocFft& operator=(const RocFft&) = delete;
    ~RocFft();
    // dense ndim-dimensional transform (lengths fastest first), `batch` contiguous copies.
    // lazy_work: the work buffer is allocated at the first exec() instead of here (plans created ahead of their use,
    // e.g. concurrently on helper threads while rocFFT compiles their kernels, should not hold tens of GB meanwhile)
    void create(int ndim, const size_t* lengths, bool forward, size_t batch = 1, bool lazy_work = false);
    // 1D transforms of `length` with element stride `stride`, `batch` lines `dist` apart
    void create_strided_1d(size_t length, size_t stride, size_t dist, size_t batch, bool forward, bool lazy_work = false);
    void exec(void* buf, hipStream_t stream);
    void release();            // plan, execution info and work buffer
private:
    void finish_create();
};

void rocfft_global_setup();

// GMRES workspace, allocated lazily and kept across solves (gmres.hip)
struct GmresWorkspace {
    int restart = 0;
    DevBuf<cplx> V;          // (restart + 1) basis vectors of length N, column after column
    DevBuf<cplx> hdev;       // Hessenberg column of the current step (restart + 2 entries)
    DevBuf<cplx> ydev;       // least-squares solution at restart
    DevBuf<cplx> partial;    // reduction scratch
    DevBuf<cplx> ax;         // A*x workspace
    cplx* hpin = nullptr;    // pinned host mirror of hdev: two slots of restart + 2 (the step in flight and the one before it)
    cplx* vpin = nullptr;    // pinned host vector for the preconditioner callback
    hipEvent_t fetched[2] = { nullptr, nullptr };   // scalars of a step have landed in their slot
    ~GmresWorkspace();
};

// Host-vector apply as a pipeline (plan.hip: host_pipelined_convolve): copy streams and per-chunk events
struct HostPipe {
    hipStream_t up = nullptr, down = nullptr;      // non-blocking: ordered against the plan's stream by the events only
    std::vector<hipEvent_t> ev_up, ev_down;        // chunk c of x has landed / chunk c of y is ready
    hipEvent_t ev_free = nullptr;                  // everything earlier on the plan's stream has finished with the staging buffers
    hipEvent_t ev_xfree[2] = { nullptr, nullptr }; // staging slot s: its inverse x pass has read xs / its downloads have read ys
    hipEvent_t ev_yfree[2] = { nullptr, nullptr };
    ~HostPipe();
};

// slab-distributed state (dist.hip)
struct DistState {
    int rank = 0, nranks = 1;
    bool force_overlap = false, force_comm = false;
    bool no_overlap = false, split_edges = true;     // LSFC_DIST_OVERLAP=0 / LSFC_DIST_SPLIT_EDGES=0, read at plan creation
    bool sim = false;        // simulated ranks in one process (tests): exchanges done by lsfc_dist_sim_apply
    bool member = false;     // rank of a single-process multi-device plan (MultiState): driven by the parent plan only
    void* comm = nullptr;    // ncclComm_t
    void* comm2 = nullptr;   // second communicator: the way back runs on its own stream, concurrently with the way in
    int lz = 0;              // local z planes (l / nranks)
    int W = 0;               // x' storage indices owned after the transpose (Lx / nranks)
    int K = 1;               // pipeline chunks of the owned x' range (exchange of chunk c+1 overlaps compute on chunk c)
    int Wc = 0;              // W / K
    DevBuf<cplx> S1;         // [nranks][K][Wc][m][lz]: xfwd output packed per (destination rank, chunk) / xinv input
    DevBuf<cplx> R1;         // [K][Wc][m][l]: per chunk the natural layout on its x' range (received blocks concatenated)
    hipStream_t cs1 = nullptr, cs2 = nullptr;      // communication streams (in / back)
    hipStream_t st2 = nullptr;                     // second compute stream (LSFC_DIST_COMPUTE_STREAMS=2): odd chunks run on it, so that the
                                                   // ramp of one chunk's kernels fills the tail of the previous chunk's
    std::vector<hipEvent_t> ev_in, ev_done, ev_back;
    hipEvent_t ev_p1 = nullptr, ev_p1a = nullptr, ev_backa = nullptr;   // x pass done / its first z half done / first z half of the last chunk back
    ~DistState();
};

// Single-process multi-device plan (dist.hip): ONE host thread drives P z-slab ranks, one per device -- the form a
// single-process host (the reference is one Julia process, examples/example3D.jl:54,78) uses to reach the multi-GPU path.
struct MultiState {
    int P = 0;
    std::vector<int> devices;                        // devices[r] of rank r (repeats allowed with the copy transport: tests)
    std::vector<std::unique_ptr<lsfc_plan>> sub;     // sub[r]: the slab plan of rank r (DistState::member)
    bool rccl = false;                               // transport of the two slab exchanges: RCCL grouped send/recv, or peer copies
    std::vector<void*> comm, comm2;                  // ncclComm_t per rank (ncclCommInitAll): way in / way back
    cplx* red_pin = nullptr;                         // pinned host scratch of the scalar reductions (copy transport)
    ~MultiState();
};

} // namespace lsfc

struct lsfc_plan {
    int device = 0;
    hipStream_t stream = nullptr;
    int ndim = 3;
    int dims[3] = {1, 1, 1};        // n, m, l
    int pads[3] = {1, 1, 1};        // padded working grid
    int crop[3] = {0, 0, 0};        // crop window offset in the padded grid
    int64_t N = 0;
    double omega = 0.0;
    int quad_rule = LSFC_QUAD_GREENGARD_VICO;
    unsigned flags = 0;
    enum Pipeline { PRUNED = 0, ROCFFT_REDUCED = 1, ROCFFT_LITERAL = 2 } pipeline = PRUNED;

    lsfc::DevBuf<double> nu;
    lsfc::DevBuf<lsfc::cplx> sym;        // pruned: storage-order tiled layout incl. 1/|pads|; rocFFT: FFT order incl. 1/|pads|

    // pruned pipeline
    lsfc::PrunedTuning tuning;
    lsfc::DevBuf<lsfc::cplx> tw[3];
    lsfc::DevBuf<lsfc::cplx> twl[3];     // full stage-twiddle tables (pruned_twfull)
    lsfc::DevBuf<lsfc::cplx> A1, A2;
    // y-even symbol: only rows with ky <= Ly/2 are stored (sym_rows of them); ytab[o] = (data row, symbol row) in block order
    int sym_rows = 0;
    lsfc::DevBuf<int2> ytab;
    // z-even symbol: lines hold sym_hz = Lz/2 + 8 entries (kz <= Lz/2); zmirror[s - Lz/2] = storage index of the mirror of slot s
    int sym_hz = 0;
    lsfc::DevBuf<int> zmirror;
    int pitch1 = 0, pitch2 = 0;      // row pitch of A1 (>= Lx) and of one storage-y row of an A2 tile (>= 8*l)
    int64_t tile2d = 0;              // 2D with an even symbol: A1 as tiles [Lx/8][m][8] of this pitch (elements); 0: natural rows [m][Lx]
    // several right-hand sides per launch (lsfc_apply_batch, lsfc_gmres_batch): A1 / A2 hold batch_cap members of
    // a1_elems / a2_elems entries back to back (grown on demand)
    int64_t a1_elems = 0, a2_elems = 0;
    int batch_cap = 1;

    // rocFFT pipelines
    std::unique_ptr<lsfc::RocFft> fwd, inv;
    lsfc::DevBuf<lsfc::cplx> W;

    // response to a unit source at grid index 0 (the spatial kernel on [0,n)^d), built on first use by
    // lsfc_sample_sources
    lsfc::DevBuf<lsfc::cplx> kernel0;

    // staging for host-resident vectors
    lsfc::DevBuf<lsfc::cplx> xs, ys;
    std::unique_ptr<lsfc::HostPipe> hostpipe;

    std::unique_ptr<lsfc::GmresWorkspace> gmres;
    std::vector<std::unique_ptr<lsfc::GmresWorkspace>> gmres_batch;    // one workspace per right-hand side of lsfc_gmres_batch
    std::unique_ptr<lsfc::DistState> dist;
    std::unique_ptr<lsfc::MultiState> multi;

    lsfc_plan();
    ~lsfc_plan();
};

namespace lsfc {

// y = alpha*x + beta*conv((use_nu ? nu : 1) .* x); device pointers, stream-ordered.
void plan_convolve_dev(lsfc_plan* p, const cplx* x, cplx* y, bool use_nu, double alpha, double beta);
// nrhs <= LSFC_MAX_BATCH right-hand sides in one pass of the pipeline: y_j = alpha*x_j + beta*conv(...x_j); the symbol is read
// once per batch in the fused pass (pruned pipeline; the other pipelines run the members one after the other)
void plan_convolve_batch_dev(lsfc_plan* p, int nrhs, const VecBatch& vb, bool use_nu, double alpha, double beta);
// the operator M = I + omega^2 G nu
inline void plan_apply_dev(lsfc_plan* p, const cplx* x, cplx* y) { plan_convolve_dev(p, x, y, true, 1.0, p->omega * p->omega); }

// Finish a plan from a natural-layout symbol already on the device.
//   literal: Gd has p->pads entries; centred => ifftshift is folded in.  Takes ownership of nothing.
void plan_finish_literal(lsfc_plan* p, const cplx* Gd, bool centred);
//   reduce a (pe,me,le) symbol (centred or FFT order) to the (2n,2m,2l) grid and pick the pipeline
void plan_finish_reduce(lsfc_plan* p, DevBuf<cplx>& Gd, const int lit[3], bool centred, const int kernel_origin[3]);
//   picks pruned (pruned_best_length(n)) or rocFFT (2n) and sets p->pads / p->pipeline
void plan_choose_reduced_grid(lsfc_plan* p);
//   G2 natural FFT order on the (2n,2m,2l) grid, unscaled
void plan_finish_from_reduced(lsfc_plan* p, DevBuf<cplx>& G2);

void plan_common_init(lsfc_plan* p, int ndim, int64_t n, int64_t m, int64_t l, const double* nu_host, double omega,
                      int quad_rule, unsigned flags, int device);

// symbol generators (symbol.hip)
void symbol_gv3d_reduced(lsfc_plan* p, double box, DevBuf<cplx>& G2);                  // -> (2n,2m,2l) FFT order, unscaled
void symbol_gv3d_quarter(lsfc_plan* p, double box, DevBuf<cplx>& Gq);                  // -> [Q0][Q1/2+1][Q2/2+1] (ky, kz >= 0 only), unscaled
bool plan_quarter_symbol_ok(const lsfc_plan* p);                                       // the stored symbol is the y-even, z-even quarter
void plan_finish_from_quarter(lsfc_plan* p, DevBuf<cplx>& Gq);
void plan_finish_symbol(lsfc_plan* p, DevBuf<cplx>& G2, bool quarter);
void symbol_gv2d_literal(lsfc_plan* p, double box, DevBuf<cplx>& G, int lit[3]);       // -> (4n,4m) centred
void symbol_trap2d_literal(lsfc_plan* p, double x0, double y0, double h, cplx d0, DevBuf<cplx>& G); // -> fft(Ge), (2n-1,2m-1)

// slab-distributed operator (dist.hip)
void dist_convolve_dev(lsfc_plan* p, const cplx* x, cplx* y, bool use_nu, double alpha, double beta);
void dist_allreduce_sum(lsfc_plan* p, cplx* dev, int count);           // no-op unless a real multi-rank plan
void dist_profile_stages(lsfc_plan* p, const cplx* x, cplx* y,
                         std::function<void(const char*, double, std::function<void()>)> add);

// symbol rows / block order of the z pass for a (possibly) y-even symbol: fills p->sym_rows, p->ytab, returns the
// device table of the y frequency of every stored row (plan.hip)
// roots of unity of one axis: the per-line table tw[] and the full stage-twiddle table twl[] (pruned_twfull)
void plan_make_twiddles(lsfc_plan* p, int axis, int L);
void plan_setup_symbol_rows(lsfc_plan* p, const cplx* G2, const std::vector<int>& perm_y, const std::vector<int>& perm_z, DevBuf<int>& pyrow);

// single-process multi-device plan (dist.hip): x[r], y[r] = device pointers of rank r's slab on devices[r]
void multi_convolve_dev(lsfc_plan* root, const cplx* const* x, cplx* const* y, bool use_nu, double alpha, double beta);
void multi_convolve_host(lsfc_plan* root, const cplx* x, cplx* y, bool use_nu, double alpha, double beta);   // scatter, apply, gather
void multi_allreduce_sum(lsfc_plan* root, cplx* const* dev, int count);   // dev[r] on devices[r], stream-ordered on the sub-plan streams
void multi_synchronize(lsfc_plan* root);
void multi_profile(lsfc_plan* root, int reps, int max_stages, const char** names, double* ms, double* bytes, int* nstages);

// GMRES (gmres.hip)
void gmres_run(lsfc_plan* p, cplx* x_dev, const cplx* b_dev, const lsfc_gmres_opts* opts, double* resnorm, int64_t cap,
               lsfc_gmres_result* res);
// nrhs independent solves in lock step (device vectors back to back, nrhs <= 64): every Arnoldi step applies the operator
// to all unconverged right-hand sides in one batch
void gmres_run_batch(lsfc_plan* p, cplx* x_dev, const cplx* b_dev, int nrhs, const lsfc_gmres_opts* opts, double* resnorm, int64_t cap,
                     lsfc_gmres_result* res);
// multi-device plan: x (in/out) and b are HOST vectors of the full size; the Krylov basis is spread over the devices
void gmres_run_multi(lsfc_plan* root, cplx* x_host, const cplx* b_host, const lsfc_gmres_opts* opts, double* resnorm, int64_t cap,
                     lsfc_gmres_result* res);

} // namespace lsfc
