// Slab-distributed 3D operator: one process per GPU, RCCL over xGMI.
//
// Rank p owns z planes [p*lz, (p+1)*lz) of x, y and nu.  Per apply:
//   phase 1 (local)   k_xfwd on the own planes; the output is written already packed per destination
//                     rank: S1[q][W][m][lz], W = Lx/P storage indices of x' for rank q
//   exchange 1        all-to-all (grouped ncclSend/ncclRecv): block q of S1 -> rank q.  The received blocks,
//                     ordered by source rank, ARE the natural array R1[W][m][l] (blocks concatenate along z)
//   phase 2 (local)   k_yfwd, k_zfused (symbol slab of the own x' tiles), k_yinv on R1 / A2
//   exchange 2        block p of R1 (z range of rank p) -> rank p, received into S1[q][W][m][lz]
//   phase 3 (local)   k_xinv reads that packed layout directly, y = alpha*x + beta*(.)
// No pack/unpack/transposition kernel runs: the chunked addressing of the x passes does the packing,
// and the exchange moves 2N complex per transpose, the minimum-volume point of the pipeline.
#include "plan.hpp"
#include "pointwise.hpp"
#include "dist_schedule.hpp"
#include <rccl/rccl.h>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace lsfc {

#define LSFC_NCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) \
    ::lsfc::fail(LSFC_EHIP, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); } while (0)

DistState::~DistState() {
    for (auto& v : { &ev_in, &ev_done, &ev_back }) for (hipEvent_t e : *v) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : { ev_p1, ev_p1a, ev_backa }) if (e) (void)hipEventDestroy(e);
    if (cs1) (void)hipStreamDestroy(cs1);
    if (cs2) (void)hipStreamDestroy(cs2);
    if (st2) (void)hipStreamDestroy(st2);
    if (comm2) (void)ncclCommDestroy((ncclComm_t)comm2);
    if (comm) (void)ncclCommDestroy((ncclComm_t)comm);
}

// block of one (rank, chunk) pair in S1 / R1: [Wc][m][lz]
static int64_t block_elems(const lsfc_plan* p) { return (int64_t)p->dist->Wc * p->dims[1] * p->dist->lz; }

// part = -1: all own planes; part = 0 / 1: the lower / upper half of them (the z halves of every S1 block)
static void phase1(lsfc_plan* p, const cplx* x, bool use_nu, hipStream_t st, int part = -1) {
    const DistState* d = p->dist.get();
    const int64_t all = (int64_t)p->dims[1] * d->lz, nl = part < 0 ? all : all / 2, l0 = part == 1 ? all / 2 : 0;
    // chunk width Wc: storage index s of a line lands in block s / Wc = dest_rank * K + chunk
    pruned_xfwd(p->pads[0], p->tuning, x + l0 * p->dims[0], use_nu ? p->nu.p + l0 * p->dims[0] : nullptr, d->S1.p + l0 * d->Wc, p->tw[0].p,
                nl, d->Wc, d->Wc, p->dims[0], st, all * d->Wc);
}
static void phase2_yfwd(lsfc_plan* p, int c, hipStream_t st) {
    const DistState* d = p->dist.get();
    const int m = p->dims[1], l = p->dims[2];
    pruned_yfwd(p->pads[1], p->tuning, d->R1.p + (int64_t)c * d->Wc * m * l, p->A2.p + (int64_t)c * (d->Wc / 8) * p->pads[1] * p->pitch2,
                p->tw[1].p, d->Wc, m, l, d->Wc, p->pitch2, st);
}
static void phase2_zfused(lsfc_plan* p, int c, hipStream_t st) {
    const DistState* d = p->dist.get();
    const int Ly = p->pads[1], Lz = p->pads[2], l = p->dims[2];
    pruned_zfused(Lz, p->tuning, p->A2.p + (int64_t)c * (d->Wc / 8) * Ly * p->pitch2, p->sym.p + (int64_t)c * d->Wc * p->sym_rows * p->sym_hz, p->tw[2].p, p->twl[2].p, d->Wc, Ly,
                  (int64_t)p->pitch2 * Ly, (int64_t)p->pitch2, 8, (int64_t)8 * p->sym_hz * p->sym_rows, (int64_t)8 * p->sym_hz, 8, p->ytab.p,
                  p->zmirror.p, l, st);
}
static void phase2_yinv(lsfc_plan* p, int c, hipStream_t st) {
    const DistState* d = p->dist.get();
    const int m = p->dims[1], l = p->dims[2];
    pruned_yinv(p->pads[1], p->tuning, p->A2.p + (int64_t)c * (d->Wc / 8) * p->pads[1] * p->pitch2, d->R1.p + (int64_t)c * d->Wc * m * l,
                p->tw[1].p, d->Wc, m, l, d->Wc, p->pitch2, st);
}
static void phase2(lsfc_plan* p, int c, hipStream_t st) { phase2_yfwd(p, c, st); phase2_zfused(p, c, st); phase2_yinv(p, c, st); }
static void phase3(lsfc_plan* p, const cplx* x, cplx* y, double alpha, double beta, hipStream_t st, int part = -1) {
    const DistState* d = p->dist.get();
    const int64_t all = (int64_t)p->dims[1] * d->lz, nl = part < 0 ? all : all / 2, l0 = part == 1 ? all / 2 : 0;
    pruned_xinv(p->pads[0], p->tuning, d->S1.p + l0 * d->Wc, x + l0 * p->dims[0], y + l0 * p->dims[0], alpha, beta, p->tw[0].p,
                nl, d->Wc, d->Wc, p->dims[0], st, all * d->Wc);
}

// Exchange of chunk c.  way in : S1 block (q*K + c)  -> rank q, lands in R1 chunk c at slot <source rank>
//                       way back: R1 chunk c, slot q  -> rank q, lands in S1 block (<source rank>*K + c)
// part = -1: whole blocks; 0 / 1: their lower / upper z halves (blocks are [z][m][Wc], z slowest: halves are contiguous)
static void exchange(lsfc_plan* p, int c, bool back, hipStream_t st, int part = -1) {
    DistState* d = p->dist.get();
    const int P = d->nranks;
    // (the message list -- peers, order, offsets -- is dist_schedule.hpp's, checked on the CPU for P = 2, 4, 8)
    const std::vector<dsched::Msg> msgs = dsched::exchange_messages(d->rank, P, d->K, c, back, part, block_elems(p));
    auto ptr = [&](const dsched::Msg& m) { return (m.in_s1 ? d->S1.p : d->R1.p) + m.off; };
    ncclComm_t comm = (ncclComm_t)(back ? d->comm2 : d->comm);
    if (d->force_comm && P == 1) {
        // single-rank exercise of the RCCL path: the self block travels through ncclSend/ncclRecv
        LSFC_NCCL(ncclGroupStart());
        LSFC_NCCL(ncclSend(ptr(msgs[0]), (size_t)msgs[0].count * 2, ncclDouble, 0, comm, st));
        LSFC_NCCL(ncclRecv(ptr(msgs[1]), (size_t)msgs[1].count * 2, ncclDouble, 0, comm, st));
        LSFC_NCCL(ncclGroupEnd());
        return;
    }
    LSFC_HIP(hipMemcpyAsync(ptr(msgs[1]), ptr(msgs[0]), (size_t)msgs[0].count * sizeof(cplx), hipMemcpyDeviceToDevice, st));
    if (P == 1) return;
    LSFC_NCCL(ncclGroupStart());
    for (size_t i = 2; i < msgs.size(); ++i) {
        const dsched::Msg& m = msgs[i];
        if (m.send) LSFC_NCCL(ncclSend(ptr(m), (size_t)m.count * 2, ncclDouble, m.peer, comm, st));
        else LSFC_NCCL(ncclRecv(ptr(m), (size_t)m.count * 2, ncclDouble, m.peer, comm, st));
    }
    LSFC_NCCL(ncclGroupEnd());
}

void dist_convolve_dev(lsfc_plan* p, const cplx* x, cplx* y, bool use_nu, double alpha, double beta) {
    DistState* d = p->dist.get();
    LSFC_REQUIRE(!d->sim, "simulated ranks are driven through lsfc_dist_sim_apply");
    LSFC_REQUIRE(!d->member, "the ranks of a multi-device plan are driven through their parent plan");
    hipStream_t st = p->stream;
    // LSFC_DIST_OVERLAP=0: no overlap -- every exchange on the compute stream, chunk after chunk (debugging aid)
    if ((d->nranks == 1 && !d->force_overlap) || d->no_overlap) {
        phase1(p, x, use_nu, st);
        for (int c = 0; c < d->K; ++c) { exchange(p, c, false, st); phase2(p, c, st); exchange(p, c, true, st); }
        phase3(p, x, y, alpha, beta, st);
        return;
    }
    // software pipeline over the K chunks of the owned x' range: the all-to-all of chunk c+1 (stream cs1) and the
    // all-to-all back of chunk c-1 (stream cs2, second communicator) run while chunk c is transformed (stream st).
    // The two ends of the pipeline are split once more over the z halves of the own slab, so that the first exchange
    // starts after half of the x pass and the second half of the last exchange back hides half of the inverse x pass.
    const bool edges = d->split_edges && d->lz % 2 == 0;
    const int K = d->K;
    if (edges) {
        phase1(p, x, use_nu, st, 0);
        LSFC_HIP(hipEventRecord(d->ev_p1a, st));
        phase1(p, x, use_nu, st, 1);
    } else {
        phase1(p, x, use_nu, st);
    }
    LSFC_HIP(hipEventRecord(d->ev_p1, st));
    if (edges) {
        LSFC_HIP(hipStreamWaitEvent(d->cs1, d->ev_p1a, 0));
        exchange(p, 0, false, d->cs1, 0);
    }
    LSFC_HIP(hipStreamWaitEvent(d->cs1, d->ev_p1, 0));
    LSFC_HIP(hipStreamWaitEvent(d->cs2, d->ev_p1, 0));      // also orders cs2 after the previous apply's xinv reads of S1
    for (int c = 0; c < K; ++c) {
        if (edges && c == 0) exchange(p, 0, false, d->cs1, 1); else exchange(p, c, false, d->cs1);
        LSFC_HIP(hipEventRecord(d->ev_in[c], d->cs1));
    }
    for (int c = 0; c < K; ++c) {
        // (two compute streams: a chunk always runs on the stream of its parity, so its buffers -- A2 / R1 chunk c -- stay ordered
        // from one apply to the next; everything the second stream does is behind ev_done -> cs2 -> ev_back -> the inverse x pass)
        hipStream_t cst = ((c & 1) && d->st2) ? d->st2 : st;
        LSFC_HIP(hipStreamWaitEvent(cst, d->ev_in[c], 0));
        phase2(p, c, cst);
        LSFC_HIP(hipEventRecord(d->ev_done[c], cst));
        LSFC_HIP(hipStreamWaitEvent(d->cs2, d->ev_done[c], 0));
        // (the S1 blocks this receive overwrites belong to chunk c, whose way-in sends finished before ev_in[c])
        if (edges && c == K - 1) {
            exchange(p, c, true, d->cs2, 0);
            LSFC_HIP(hipEventRecord(d->ev_backa, d->cs2));
            exchange(p, c, true, d->cs2, 1);
        } else {
            exchange(p, c, true, d->cs2);
        }
        LSFC_HIP(hipEventRecord(d->ev_back[c], d->cs2));
    }
    if (edges) {
        // cs2 runs its exchanges in order: ev_backa implies every earlier chunk is back
        LSFC_HIP(hipStreamWaitEvent(st, d->ev_backa, 0));
        phase3(p, x, y, alpha, beta, st, 0);
        LSFC_HIP(hipStreamWaitEvent(st, d->ev_back[K - 1], 0));
        phase3(p, x, y, alpha, beta, st, 1);
    } else {
        for (int c = 0; c < K; ++c) LSFC_HIP(hipStreamWaitEvent(st, d->ev_back[c], 0));
        phase3(p, x, y, alpha, beta, st);
    }
}

void dist_allreduce_sum(lsfc_plan* p, cplx* dev, int count) {
    DistState* d = p->dist.get();
    if (!d || d->sim || d->member || (d->nranks == 1 && !d->force_comm)) return;
    LSFC_NCCL(ncclAllReduce(dev, dev, (size_t)count * 2, ncclDouble, ncclSum, (ncclComm_t)d->comm, p->stream));
}

void dist_profile_stages(lsfc_plan* p, const cplx* x, cplx* y, std::function<void(const char*, double, std::function<void()>)> add) {
    DistState* d = p->dist.get();
    const double N = (double)p->N, C = 16.0, om2 = p->omega * p->omega;     // N = local points
    hipStream_t st = p->stream;
    const int K = d->K;
    // un-overlapped, stage by stage, all on the plan's stream (the production path overlaps the exchanges)
    const bool sim = d->sim || d->member;                       // simulated rank: compute stages only (per-rank kernel times at P ranks)
    if (sim) {
        // no exchange runs here, so the receive buffer would hold the zeros of plan creation and the y / z / y passes would
        // transform zeros -- on which this power-limited chip clocks ~9 % higher.  Prime it once with the x pass's own
        // output (same size: P * K blocks either way), i.e. with data of the statistics the real exchange delivers.
        phase1(p, x, true, st);
        LSFC_HIP(hipMemcpyAsync(d->R1.p, d->S1.p, std::min(d->R1.bytes(), d->S1.bytes()), hipMemcpyDeviceToDevice, st));
    }
    add("xfwd", N * (C + 8) + 2 * N * C, [=] { phase1(p, x, true, st); });
    if (!sim) add("alltoall_in", 2 * N * C, [=] { for (int c = 0; c < K; ++c) exchange(p, c, false, st); });
    add("yfwd", 6 * N * C, [=] { for (int c = 0; c < K; ++c) phase2_yfwd(p, c, st); });
    add("zfused", 16 * N * C, [=] { for (int c = 0; c < K; ++c) phase2_zfused(p, c, st); });
    add("yinv", 6 * N * C, [=] { for (int c = 0; c < K; ++c) phase2_yinv(p, c, st); });
    if (!sim) add("alltoall_back", 2 * N * C, [=] { for (int c = 0; c < K; ++c) exchange(p, c, true, st); });
    add("xinv", 4 * N * C, [=] { phase3(p, x, y, 1.0, om2, st); });
}

// ---------------------------------------------------------------------------
static void create_dist_plan(lsfc_plan** out, int64_t n, int64_t m, int64_t l, double box, double omega, const double* nu_local,
                             unsigned flags, int device, int rank, int nranks, const unsigned char* id, bool sim, bool member = false) {
    LSFC_REQUIRE(out, "NULL argument"); *out = nullptr;
    LSFC_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
    LSFC_REQUIRE(n % 2 == 0 && m % 2 == 0 && l % 2 == 0, "even grid sizes only");
    LSFC_REQUIRE(l % nranks == 0, "l = %lld is not divisible by the number of ranks %d", (long long)l, nranks);
    for (int64_t v : { n, m, l }) LSFC_REQUIRE(pruned_best_length(v) != 0 && pruned_best_length(v) <= 4 * v, "distributed plan: grid sizes from 8 to 1024 per axis");
    LSFC_REQUIRE((pruned_best_length(n) / 8) % nranks == 0 && is_pow2(nranks),
                 "number of ranks must be a power of two dividing Lx/8 (Lx = %d is the padded line length for n = %lld)", pruned_best_length(n), (long long)n);
    std::unique_ptr<lsfc_plan> p(new lsfc_plan());
    const int lz = (int)(l / nranks);
    // local nu: n*m*lz doubles
    plan_common_init(p.get(), 3, n, m, lz, nu_local, omega, LSFC_QUAD_GREENGARD_VICO, flags & ~LSFC_FLAG_LITERAL_PAD, device);
    p->dims[2] = (int)l;                            // dims are global; N stays local
    p->N = n * m * lz;
    for (int d = 0; d < 3; ++d) { p->pads[d] = pruned_best_length(p->dims[d]); p->crop[d] = 0; }
    p->dist.reset(new DistState());
    DistState* d = p->dist.get();
    d->rank = rank; d->nranks = nranks; d->sim = sim; d->member = member; d->lz = lz; d->W = p->pads[0] / nranks;
    // pipeline chunks: up to 4, each at least one 8-wide tile (LSFC_DIST_CHUNKS overrides; 1 disables the overlap)
    int K = 4;
    if (const char* v = getenv("LSFC_DIST_CHUNKS")) K = atoi(v);
    if (nranks == 1 && !sim && !getenv("LSFC_DIST_CHUNKS") && !getenv("LSFC_DIST_FORCE_OVERLAP")) K = 1;
    const dsched::Chunks ch = dsched::plan_chunks(p->pads[0], nranks, K);
    K = ch.K; d->K = ch.K; d->Wc = ch.Wc;
    // LSFC_DIST_FORCE_OVERLAP=1: run the three-stream pipeline even with one rank (tests of the event logic)
    d->force_overlap = getenv("LSFC_DIST_FORCE_OVERLAP") && getenv("LSFC_DIST_FORCE_OVERLAP")[0] == '1';
    // LSFC_DIST_OVERLAP=0: every exchange on the compute stream; LSFC_DIST_SPLIT_EDGES=0: no z-half split of the pipeline ends
    d->no_overlap = getenv("LSFC_DIST_OVERLAP") && getenv("LSFC_DIST_OVERLAP")[0] == '0';
    d->split_edges = !(getenv("LSFC_DIST_SPLIT_EDGES") && getenv("LSFC_DIST_SPLIT_EDGES")[0] == '0');
    // LSFC_DIST_FORCE_COMM=1: build the communicators and route the (self) exchange through RCCL even with one rank
    d->force_comm = !sim && !member && getenv("LSFC_DIST_FORCE_COMM") && getenv("LSFC_DIST_FORCE_COMM")[0] == '1';
    if (member) d->force_overlap = false;               // the parent plan drives the streams of its ranks
    if (!sim && !member && (nranks > 1 || d->force_comm)) {
        LSFC_REQUIRE(id, "NULL unique id");
        // an all-zero id means the caller never received rank 0's id (no broadcast happened): ncclCommInitRank would
        // block forever on every rank instead of failing
        bool nonzero = false;
        for (int i = 0; i < LSFC_UNIQUE_ID_BYTES; ++i) nonzero = nonzero || id[i] != 0;
        LSFC_REQUIRE(nonzero, "the RCCL unique id is all zeros: ship the bytes of lsfc_dist_unique_id() from rank 0 to every rank first");
        ncclUniqueId uid; static_assert(sizeof(uid) == LSFC_UNIQUE_ID_BYTES, "unique id size");
        memcpy(&uid, id, sizeof uid);
        ncclComm_t comm, comm2;
        LSFC_NCCL(ncclCommInitRank(&comm, nranks, uid, rank));
        d->comm = comm;
        LSFC_NCCL(ncclCommSplit(comm, 0, rank, &comm2, nullptr));
        d->comm2 = comm2;
    }
    if (!sim && (nranks > 1 || d->force_overlap || member)) {
        LSFC_HIP(hipStreamCreateWithFlags(&d->cs1, hipStreamNonBlocking));
        LSFC_HIP(hipStreamCreateWithFlags(&d->cs2, hipStreamNonBlocking));
        // LSFC_DIST_COMPUTE_STREAMS=2 (opt-in): odd chunks on a second compute stream.  With chunks of the size a rank of an 8-GPU job
        // transforms (32 x' of 1024: 12 short launches per apply) the ramp of a chunk's kernels fills the tail of the previous chunk's --
        // one rank, 32 such chunks at 512^3: 15.93 -> 15.25-15.5 ms; neutral with 4 or 8 large chunks (profiles/r03_dist_two_compute_streams.log).
        // Not the default: that was measured with device-to-device copies as the exchange; RCCL's transfer kernels need CUs, which
        // they get at the boundaries between compute kernels -- two compute streams close those gaps, and no multi-GPU node has
        // been available to see what that does to the exchange.
        const char* cse = getenv("LSFC_DIST_COMPUTE_STREAMS");
        if (K >= 2 && cse && atoi(cse) == 2) LSFC_HIP(hipStreamCreateWithFlags(&d->st2, hipStreamNonBlocking));
        for (hipEvent_t* e : { &d->ev_p1, &d->ev_p1a, &d->ev_backa }) LSFC_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        d->ev_in.resize((size_t)K); d->ev_done.resize((size_t)K); d->ev_back.resize((size_t)K);
        for (int c = 0; c < K; ++c) {
            LSFC_HIP(hipEventCreateWithFlags(&d->ev_in[c], hipEventDisableTiming));
            LSFC_HIP(hipEventCreateWithFlags(&d->ev_done[c], hipEventDisableTiming));
            LSFC_HIP(hipEventCreateWithFlags(&d->ev_back[c], hipEventDisableTiming));
        }
    }
    // symbol: every rank evaluates the reduced symbol (elementary functions + rocFFT; 0.2 s at 512^3 through the symbol's
    // symmetry, 8.6 GB of temporaries that are freed before the message buffers and A2 are allocated) and keeps
    // only the slab of its own x' tiles in the tiled storage order
    DevBuf<cplx> G2;
    const bool quarter = plan_quarter_symbol_ok(p.get());
    if (quarter) symbol_gv3d_quarter(p.get(), box, G2); else symbol_gv3d_reduced(p.get(), box, G2);
    std::vector<int> perm[3]; DevBuf<int> dperm[3];
    for (int a = 0; a < 3; ++a) {
        perm[a].resize((size_t)p->pads[a]);
        pruned_perm(p->pads[a], perm[a].data());
        dperm[a].alloc(perm[a].size());
        LSFC_HIP(hipMemcpy(dperm[a].p, perm[a].data(), perm[a].size() * sizeof(int), hipMemcpyHostToDevice));
        plan_make_twiddles(p.get(), a, p->pads[a]);
    }
    const int ntiles = d->W / 8;
    const double scale = 1.0 / ((double)p->pads[0] * p->pads[1] * p->pads[2]);
    DevBuf<int> pyrow;
    plan_setup_symbol_rows(p.get(), quarter ? nullptr : G2.p, perm[1], perm[2], pyrow);
    p->sym.alloc((size_t)d->W * p->sym_rows * p->sym_hz);
    pw_permute_symbol(G2.p, p->sym.p, dperm[0].p, pyrow.p, dperm[2].p, p->pads, p->sym_rows, p->sym_hz, rank * ntiles, ntiles, scale, p->stream,
                      quarter ? p->pads[1] / 2 + 1 : 0);
    LSFC_HIP(hipStreamSynchronize(p->stream));
    G2.release();
    d->S1.alloc((size_t)p->pads[0] * m * lz);
    d->R1.alloc((size_t)d->W * m * l);
    // A2 rows are padded like the single-GPU plan's (the message buffers S1 / R1 stay dense: their blocks are the messages)
    const int pad2 = p->tuning.pad2 >= 0 ? p->tuning.pad2 : (p->pads[1] >= 1024 ? 72 : 0);
    p->pitch1 = d->Wc;
    p->pitch2 = 8 * (int)l + pad2 / 8 * 8;
    p->A2.alloc((size_t)(d->W / 8) * p->pads[1] * p->pitch2);
    LSFC_HIP(hipMemset(d->S1.p, 0, d->S1.bytes()));
    LSFC_HIP(hipMemset(d->R1.p, 0, d->R1.bytes()));
    LSFC_HIP(hipMemset(p->A2.p, 0, p->A2.bytes()));
    p->pipeline = lsfc_plan::PRUNED;
    *out = p.release();
}


// ---------------------------------------------------------------------------
// Single-process multi-device plan: ONE host thread enqueues the work of all P ranks (one per device).  The schedule
// is the three-stream pipeline of dist_convolve_dev, made explicit across devices with events (a stream of one device
// may wait on an event of another).  Two transports for the slab exchanges:
//   rccl  grouped ncclSend/ncclRecv over the communicators of ncclCommInitAll (distinct devices; the default)
//   copy  hipMemcpyPeerAsync issued by the SOURCE rank on its communication stream (copy engines instead of CUs; also
//         the only form that runs with a device listed twice, which is how the driver is tested on a one-GPU box)
// ---------------------------------------------------------------------------
MultiState::~MultiState() {
    for (void* c : comm2) if (c) (void)ncclCommDestroy((ncclComm_t)c);
    for (void* c : comm) if (c) (void)ncclCommDestroy((ncclComm_t)c);
    if (red_pin) (void)hipHostFree(red_pin);
}

namespace {
struct Multi {
    lsfc_plan* root; MultiState* ms; int P, K;
    explicit Multi(lsfc_plan* r) : root(r), ms(r->multi.get()), P(ms->P), K(ms->sub[0]->dist->K) {}
    lsfc_plan* sub(int r) const { return ms->sub[(size_t)r].get(); }
    DistState* d(int r) const { return sub(r)->dist.get(); }
    void dev(int r) const { LSFC_HIP(hipSetDevice(ms->devices[(size_t)r])); }
    hipStream_t st(int r) const { return sub(r)->stream; }
    int64_t B() const { return block_elems(sub(0)); }

    // Exchange of chunk c between all ranks.  `on`: 0 = the communication streams (cs1 way in, cs2 way back), 1 = the
    // compute streams (per-stage profile).
    void exchange(int c, bool back, int part, int on) const {
        const int64_t Bfull = B(), Bp = part < 0 ? Bfull : Bfull / 2, off = part == 1 ? Bfull / 2 : 0;
        auto s1 = [&](int r, int q) { return d(r)->S1.p + ((int64_t)q * K + c) * Bfull + off; };     // rank r's block for / from rank q
        auto r1 = [&](int r, int q) { return d(r)->R1.p + ((int64_t)c * P + q) * Bfull + off; };     // rank r's slot of rank q
        auto stream = [&](int r) { return on == 1 ? st(r) : (back ? d(r)->cs2 : d(r)->cs1); };
        const size_t bytes = (size_t)Bp * sizeof(cplx);
        if (!ms->rccl) {
            for (int r = 0; r < P; ++r) {
                dev(r);
                for (int s = 0; s < P; ++s) {
                    const int q = (r + s) % P;                  // every source starts with a different peer
                    const cplx* src = back ? r1(r, q) : s1(r, q);
                    cplx* dst = back ? s1(q, r) : r1(q, r);
                    if (ms->devices[(size_t)q] == ms->devices[(size_t)r]) LSFC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream(r)));
                    else LSFC_HIP(hipMemcpyPeerAsync(dst, ms->devices[(size_t)q], src, ms->devices[(size_t)r], bytes, stream(r)));
                }
            }
            return;
        }
        // every rank's message list is the one its own process would post (dist_schedule.hpp)
        std::vector<std::vector<dsched::Msg>> msgs((size_t)P);
        for (int r = 0; r < P; ++r) msgs[(size_t)r] = dsched::exchange_messages(r, P, K, c, back, part, Bfull);
        auto ptr = [&](int r, const dsched::Msg& m) { return (m.in_s1 ? d(r)->S1.p : d(r)->R1.p) + m.off; };
        for (int r = 0; r < P; ++r) {
            dev(r);
            LSFC_HIP(hipMemcpyAsync(ptr(r, msgs[(size_t)r][1]), ptr(r, msgs[(size_t)r][0]), bytes, hipMemcpyDeviceToDevice, stream(r)));
        }
        if (P == 1) return;
        LSFC_NCCL(ncclGroupStart());
        for (int r = 0; r < P; ++r) {
            ncclComm_t comm = (ncclComm_t)(back ? ms->comm2[(size_t)r] : ms->comm[(size_t)r]);
            for (size_t i = 2; i < msgs[(size_t)r].size(); ++i) {
                const dsched::Msg& m = msgs[(size_t)r][i];
                if (m.send) LSFC_NCCL(ncclSend(ptr(r, m), (size_t)m.count * 2, ncclDouble, m.peer, comm, stream(r)));
                else LSFC_NCCL(ncclRecv(ptr(r, m), (size_t)m.count * 2, ncclDouble, m.peer, comm, stream(r)));
            }
        }
        LSFC_NCCL(ncclGroupEnd());
    }
    void wait_all(hipStream_t s, hipEvent_t DistState::* ev) const { for (int q = 0; q < P; ++q) LSFC_HIP(hipStreamWaitEvent(s, d(q)->*ev, 0)); }
    void wait_all(hipStream_t s, std::vector<hipEvent_t> DistState::* ev, int c) const {
        for (int q = 0; q < P; ++q) LSFC_HIP(hipStreamWaitEvent(s, (d(q)->*ev)[(size_t)c], 0));
    }
};
} // namespace

void multi_synchronize(lsfc_plan* root) {
    Multi M(root);
    for (int r = 0; r < M.P; ++r) { M.dev(r); LSFC_HIP(hipDeviceSynchronize()); }
}

void multi_convolve_dev(lsfc_plan* root, const cplx* const* x, cplx* const* y, bool use_nu, double alpha, double beta) {
    Multi M(root);
    const int P = M.P, K = M.K;
    const bool overlap = !M.d(0)->no_overlap;
    const bool edges = overlap && M.d(0)->split_edges && M.d(0)->lz % 2 == 0;
    // phase 1 on every rank: k_xfwd, output packed per (destination rank, chunk)
    for (int r = 0; r < P; ++r) {
        M.dev(r);
        if (edges) {
            phase1(M.sub(r), x[r], use_nu, M.st(r), 0);
            LSFC_HIP(hipEventRecord(M.d(r)->ev_p1a, M.st(r)));
            phase1(M.sub(r), x[r], use_nu, M.st(r), 1);
        } else phase1(M.sub(r), x[r], use_nu, M.st(r));
        LSFC_HIP(hipEventRecord(M.d(r)->ev_p1, M.st(r)));
    }
    // way in.  A block may only land on rank q once q's previous apply has drained (its R1 / S1 are re-used): every
    // communication stream waits for the x pass of EVERY rank, which on each rank follows the previous apply's last pass.
    if (edges) {
        for (int r = 0; r < P; ++r) { M.dev(r); M.wait_all(M.d(r)->cs1, &DistState::ev_p1a); }
        M.exchange(0, false, 0, 0);
    }
    for (int r = 0; r < P; ++r) { M.dev(r); M.wait_all(M.d(r)->cs1, &DistState::ev_p1); M.wait_all(M.d(r)->cs2, &DistState::ev_p1); }
    for (int c = 0; c < K; ++c) {
        M.exchange(c, false, (edges && c == 0) ? 1 : -1, 0);
        for (int r = 0; r < P; ++r) { M.dev(r); LSFC_HIP(hipEventRecord(M.d(r)->ev_in[(size_t)c], M.d(r)->cs1)); }
    }
    // y / z / y passes chunk by chunk; the way back of chunk c runs under the passes of chunk c + 1
    for (int c = 0; c < K; ++c) {
        for (int r = 0; r < P; ++r) {
            M.dev(r);
            hipStream_t cst = ((c & 1) && M.d(r)->st2) ? M.d(r)->st2 : M.st(r);   // odd chunks: the second compute stream (as dist_convolve_dev)
            M.wait_all(cst, &DistState::ev_in, c);              // copy transport: every source signals its own blocks
            phase2(M.sub(r), c, cst);
            LSFC_HIP(hipEventRecord(M.d(r)->ev_done[(size_t)c], cst));
            LSFC_HIP(hipStreamWaitEvent(M.d(r)->cs2, M.d(r)->ev_done[(size_t)c], 0));
        }
        if (!overlap) for (int r = 0; r < P; ++r) { M.dev(r); LSFC_HIP(hipStreamSynchronize(M.st(r))); if (M.d(r)->st2) LSFC_HIP(hipStreamSynchronize(M.d(r)->st2)); }
        if (edges && c == K - 1) {
            M.exchange(c, true, 0, 0);
            for (int r = 0; r < P; ++r) { M.dev(r); LSFC_HIP(hipEventRecord(M.d(r)->ev_backa, M.d(r)->cs2)); }
            M.exchange(c, true, 1, 0);
        } else M.exchange(c, true, -1, 0);
        for (int r = 0; r < P; ++r) { M.dev(r); LSFC_HIP(hipEventRecord(M.d(r)->ev_back[(size_t)c], M.d(r)->cs2)); }
    }
    // phase 3: k_xinv reads the packed blocks; rank r needs the blocks of every source (their cs2 streams run in order,
    // so the event of the last chunk covers the earlier ones)
    for (int r = 0; r < P; ++r) {
        M.dev(r);
        if (edges) {
            M.wait_all(M.st(r), &DistState::ev_backa);
            phase3(M.sub(r), x[r], y[r], alpha, beta, M.st(r), 0);
            M.wait_all(M.st(r), &DistState::ev_back, K - 1);
            phase3(M.sub(r), x[r], y[r], alpha, beta, M.st(r), 1);
        } else {
            M.wait_all(M.st(r), &DistState::ev_back, K - 1);
            phase3(M.sub(r), x[r], y[r], alpha, beta, M.st(r));
        }
    }
}

void multi_convolve_host(lsfc_plan* root, const cplx* x, cplx* y, bool use_nu, double alpha, double beta) {
    Multi M(root);
    std::vector<const cplx*> xd((size_t)M.P); std::vector<cplx*> yd((size_t)M.P);
    int64_t off = 0;
    for (int r = 0; r < M.P; ++r) {
        lsfc_plan* p = M.sub(r);
        M.dev(r);
        if (p->xs.n < (size_t)p->N) { p->xs.alloc((size_t)p->N); p->ys.alloc((size_t)p->N); }
        LSFC_HIP(hipMemcpyAsync(p->xs.p, x + off, (size_t)p->N * sizeof(cplx), hipMemcpyHostToDevice, p->stream));
        xd[(size_t)r] = p->xs.p; yd[(size_t)r] = p->ys.p; off += p->N;
    }
    multi_convolve_dev(root, xd.data(), yd.data(), use_nu, alpha, beta);
    off = 0;
    for (int r = 0; r < M.P; ++r) {
        lsfc_plan* p = M.sub(r);
        M.dev(r);
        LSFC_HIP(hipMemcpyAsync(y + off, p->ys.p, (size_t)p->N * sizeof(cplx), hipMemcpyDeviceToHost, p->stream));
        off += p->N;
    }
    for (int r = 0; r < M.P; ++r) { M.dev(r); LSFC_HIP(hipStreamSynchronize(M.st(r))); }
}

// sum over the ranks of `count` complex scalars held at dev[r] on devices[r]; the result replaces every copy
void multi_allreduce_sum(lsfc_plan* root, cplx* const* dev, int count) {
    Multi M(root);
    if (M.P == 1) return;
    if (M.ms->rccl) {
        LSFC_NCCL(ncclGroupStart());
        for (int r = 0; r < M.P; ++r)
            LSFC_NCCL(ncclAllReduce(dev[r], dev[r], (size_t)count * 2, ncclDouble, ncclSum, (ncclComm_t)M.ms->comm[(size_t)r], M.st(r)));
        LSFC_NCCL(ncclGroupEnd());
        return;
    }
    // copy transport: O(restart) scalars through pinned host memory, summed in rank order (reproducible); the pinned scratch
    // holds 256 scalars per rank, longer reductions (classical Gram-Schmidt with restart > 256) go in pieces
    if (count > 256) {
        for (int c0 = 0; c0 < count; c0 += 256) {
            std::vector<cplx*> piece((size_t)M.P);
            for (int r = 0; r < M.P; ++r) piece[(size_t)r] = dev[r] + c0;
            multi_allreduce_sum(root, piece.data(), std::min(256, count - c0));
        }
        return;
    }
    cplx* pin = M.ms->red_pin;
    for (int r = 0; r < M.P; ++r) { M.dev(r); LSFC_HIP(hipMemcpyAsync(pin + (size_t)(r + 1) * 256, dev[r], (size_t)count * sizeof(cplx), hipMemcpyDeviceToHost, M.st(r))); }
    for (int r = 0; r < M.P; ++r) { M.dev(r); LSFC_HIP(hipStreamSynchronize(M.st(r))); }
    for (int i = 0; i < count; ++i) {
        cplx acc = make_double2(0.0, 0.0);
        for (int r = 0; r < M.P; ++r) { acc.x += pin[(size_t)(r + 1) * 256 + i].x; acc.y += pin[(size_t)(r + 1) * 256 + i].y; }
        pin[i] = acc;
    }
    for (int r = 0; r < M.P; ++r) { M.dev(r); LSFC_HIP(hipMemcpyAsync(dev[r], pin, (size_t)count * sizeof(cplx), hipMemcpyHostToDevice, M.st(r))); }
    // (the next reduction rewrites the pinned scratch: it starts with copies INTO it that are stream-ordered behind these, and the
    // host only touches it after synchronising every rank's stream above)
}

// per-stage timing of one apply on the ranks' staging vectors: every stage runs on all ranks, un-overlapped, with a
// device-wide synchronisation in between; the time of a stage is the slowest rank's
void multi_profile(lsfc_plan* root, int reps, int max_stages, const char** names, double* ms, double* bytes, int* nstages) {
    Multi M(root);
    const int P = M.P, K = M.K;
    const double N = (double)M.sub(0)->N, C = 16.0, om2 = root->omega * root->omega;
    for (int r = 0; r < P; ++r) {
        lsfc_plan* p = M.sub(r);
        M.dev(r);
        if (p->xs.n < (size_t)p->N) { p->xs.alloc((size_t)p->N); p->ys.alloc((size_t)p->N); LSFC_HIP(hipMemset(p->xs.p, 0, p->xs.bytes())); }
    }
    struct Stage { const char* name; double bytes; std::function<void()> run; };
    std::vector<Stage> stages;
    auto each = [&M, P](std::function<void(int)> f) { return [&M, P, f] { for (int r = 0; r < P; ++r) { M.dev(r); f(r); } }; };
    stages.push_back({"xfwd", N * (C + 8) + 2 * N * C, each([&M](int r) { phase1(M.sub(r), M.sub(r)->xs.p, true, M.st(r)); })});
    stages.push_back({"alltoall_in", 2 * N * C, [&M, K] { for (int c = 0; c < K; ++c) M.exchange(c, false, -1, 1); }});
    stages.push_back({"yfwd", 6 * N * C, each([&M, K](int r) { for (int c = 0; c < K; ++c) phase2_yfwd(M.sub(r), c, M.st(r)); })});
    stages.push_back({"zfused", 16 * N * C, each([&M, K](int r) { for (int c = 0; c < K; ++c) phase2_zfused(M.sub(r), c, M.st(r)); })});
    stages.push_back({"yinv", 6 * N * C, each([&M, K](int r) { for (int c = 0; c < K; ++c) phase2_yinv(M.sub(r), c, M.st(r)); })});
    stages.push_back({"alltoall_back", 2 * N * C, [&M, K] { for (int c = 0; c < K; ++c) M.exchange(c, true, -1, 1); }});
    stages.push_back({"xinv", 4 * N * C, each([&M, om2](int r) { phase3(M.sub(r), M.sub(r)->xs.p, M.sub(r)->ys.p, 1.0, om2, M.st(r)); })});
    LSFC_REQUIRE((int)stages.size() <= max_stages, "max_stages too small (need %d)", (int)stages.size());
    *nstages = (int)stages.size();
    std::vector<hipEvent_t> e0((size_t)P), e1((size_t)P);
    for (int r = 0; r < P; ++r) { M.dev(r); LSFC_HIP(hipEventCreate(&e0[(size_t)r])); LSFC_HIP(hipEventCreate(&e1[(size_t)r])); }
    for (size_t i = 0; i < stages.size(); ++i) { names[i] = stages[i].name; bytes[i] = stages[i].bytes; ms[i] = 0.0; }
    for (int rep = 0; rep < reps; ++rep)
        for (size_t i = 0; i < stages.size(); ++i) {
            multi_synchronize(root);
            for (int r = 0; r < P; ++r) { M.dev(r); LSFC_HIP(hipEventRecord(e0[(size_t)r], M.st(r))); }
            stages[i].run();
            for (int r = 0; r < P; ++r) { M.dev(r); LSFC_HIP(hipEventRecord(e1[(size_t)r], M.st(r))); }
            multi_synchronize(root);
            float worst = 0;
            for (int r = 0; r < P; ++r) { float t = 0; LSFC_HIP(hipEventElapsedTime(&t, e0[(size_t)r], e1[(size_t)r])); worst = std::max(worst, t); }
            ms[i] += worst;
        }
    for (size_t i = 0; i < stages.size(); ++i) ms[i] /= reps;
    for (int r = 0; r < P; ++r) { (void)hipEventDestroy(e0[(size_t)r]); (void)hipEventDestroy(e1[(size_t)r]); }
}

static void create_multi_plan(lsfc_plan** out, int64_t n, int64_t m, int64_t l, double box, double omega, const double* nu,
                              unsigned flags, const int* devices, int ndev) {
    LSFC_REQUIRE(out, "NULL argument"); *out = nullptr;
    LSFC_REQUIRE(devices && ndev >= 1 && ndev <= 64, "devices[] / ndev: need 1..64 devices");
    LSFC_REQUIRE(nu, "nu is NULL");
    LSFC_REQUIRE(l % ndev == 0, "l = %lld is not divisible by the number of devices %d", (long long)l, ndev);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) fail(LSFC_ENODEV, "no HIP device available: the lsfc operator has no CPU fallback");
    bool distinct = true;
    for (int r = 0; r < ndev; ++r) {
        LSFC_REQUIRE(devices[r] >= 0 && devices[r] < count, "device %d out of range (have %d)", devices[r], count);
        for (int q = 0; q < r; ++q) if (devices[q] == devices[r]) distinct = false;
    }
    std::unique_ptr<lsfc_plan> root(new lsfc_plan());
    root->device = devices[0]; root->ndim = 3;
    root->dims[0] = (int)n; root->dims[1] = (int)m; root->dims[2] = (int)l;
    root->N = n * m * l; root->omega = omega; root->quad_rule = LSFC_QUAD_GREENGARD_VICO; root->flags = flags;
    root->pipeline = lsfc_plan::PRUNED;
    root->multi.reset(new MultiState());
    MultiState* ms = root->multi.get();
    ms->P = ndev; ms->devices.assign(devices, devices + ndev);
    // transport: RCCL whenever the devices are distinct (LSFC_MULTI_TRANSPORT=copy selects the copy engines instead);
    // a device listed more than once (tests on a one-GPU box) can only use copies
    const char* tr = getenv("LSFC_MULTI_TRANSPORT");
    ms->rccl = distinct && !(tr && std::string(tr) == "copy");
    LSFC_REQUIRE(distinct || !(tr && std::string(tr) == "rccl"), "LSFC_MULTI_TRANSPORT=rccl needs distinct devices");
    // the ranks are built concurrently, one helper thread per device (symbol evaluation + rocFFT plan compilation are
    // the long part); ranks that share a device are built one after the other
    const int64_t nloc = n * m * (l / ndev);
    ms->sub.resize((size_t)ndev);
    std::vector<int> rc((size_t)ndev, LSFC_OK); std::vector<std::string> msg((size_t)ndev);
    auto build = [&](int r) {
        lsfc_plan* sp = nullptr;
        rc[(size_t)r] = guarded([&] { create_dist_plan(&sp, n, m, l, box, omega, nu + (int64_t)r * nloc, flags, devices[r], r, ndev, nullptr, false, true); });
        if (rc[(size_t)r] != LSFC_OK) msg[(size_t)r] = lsfc_last_error();
        ms->sub[(size_t)r].reset(sp);
    };
    if (distinct && ndev > 1) {
        std::vector<std::thread> th;
        for (int r = 0; r < ndev; ++r) th.emplace_back(build, r);
        for (auto& t : th) t.join();
    } else for (int r = 0; r < ndev; ++r) build(r);
    for (int r = 0; r < ndev; ++r) if (rc[(size_t)r] != LSFC_OK) fail(rc[(size_t)r], "rank %d on device %d: %s", r, devices[r], msg[(size_t)r].c_str());
    for (int d = 0; d < 3; ++d) root->pads[d] = ms->sub[0]->pads[d];
    if (distinct && ndev > 1) {
        // peer access for the copy transport (and for RCCL's own P2P paths)
        for (int r = 0; r < ndev; ++r) {
            LSFC_HIP(hipSetDevice(devices[r]));
            for (int q = 0; q < ndev; ++q) if (q != r) {
                int can = 0; LSFC_HIP(hipDeviceCanAccessPeer(&can, devices[r], devices[q]));
                if (can) { hipError_t pe = hipDeviceEnablePeerAccess(devices[q], 0); if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) LSFC_HIP(pe); (void)hipGetLastError(); }
                else LSFC_REQUIRE(ms->rccl, "device %d cannot access device %d: the copy transport needs peer access", devices[r], devices[q]);
            }
        }
    }
    if (ms->rccl && ndev > 1) {
        std::vector<ncclComm_t> c1((size_t)ndev), c2((size_t)ndev);
        LSFC_NCCL(ncclCommInitAll(c1.data(), ndev, devices));
        ms->comm.assign(c1.begin(), c1.end());
        LSFC_NCCL(ncclCommInitAll(c2.data(), ndev, devices));
        ms->comm2.assign(c2.begin(), c2.end());
    } else ms->rccl = ms->rccl && ndev > 1;
    LSFC_HIP(hipSetDevice(devices[0]));
    LSFC_HIP(hipHostMalloc((void**)&ms->red_pin, (size_t)(ndev + 1) * 256 * sizeof(cplx)));
    *out = root.release();
}

} // namespace lsfc

using namespace lsfc;

extern "C" {

int lsfc_plan_create_gv3d_multi(lsfc_plan** out, int64_t n, int64_t m, int64_t l, double box, double omega, const double* nu,
                                unsigned flags, const int* devices, int ndev) {
    return guarded([&] { create_multi_plan(out, n, m, l, box, omega, nu, flags, devices, ndev); });
}

int lsfc_multi_info(const lsfc_plan* plan, int* ndev, int* devices, int64_t* local_n, const char** transport) {
    return guarded([&] {
        LSFC_REQUIRE(plan && plan->multi, "not a multi-device plan");
        const MultiState* ms = plan->multi.get();
        if (ndev) *ndev = ms->P;
        if (devices) for (int r = 0; r < ms->P; ++r) devices[r] = ms->devices[(size_t)r];
        if (local_n) *local_n = ms->sub[0]->N;
        if (transport) *transport = ms->P == 1 ? "none (one device)" : (ms->rccl ? "rccl send/recv (ncclCommInitAll), pairwise schedule" : "peer copies (hipMemcpyPeerAsync), source-issued");
    });
}

int lsfc_multi_apply_dev(lsfc_plan* plan, const double* const* x_dev, double* const* y_dev, int mode) {
    return guarded([&] {
        LSFC_REQUIRE(plan && plan->multi && x_dev && y_dev, "not a multi-device plan / NULL argument");
        LSFC_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (apply), 1 (convolve) or 2 (convolve with nu)");
        const double om2 = plan->omega * plan->omega;
        multi_convolve_dev(plan, (const cplx* const*)x_dev, (cplx* const*)y_dev, mode != 1, mode == 0 ? 1.0 : 0.0, mode == 0 ? om2 : 1.0);
    });
}

int lsfc_dist_unique_id(unsigned char id[LSFC_UNIQUE_ID_BYTES]) {
    return guarded([&] {
        LSFC_REQUIRE(id, "NULL argument");
        ncclUniqueId uid;
        LSFC_NCCL(ncclGetUniqueId(&uid));
        memcpy(id, &uid, LSFC_UNIQUE_ID_BYTES);
    });
}

int lsfc_dist_plan_create_gv3d(lsfc_plan** out, int64_t n, int64_t m, int64_t l, double box, double omega, const double* nu_local,
                               unsigned flags, int device, int rank, int nranks, const unsigned char id[LSFC_UNIQUE_ID_BYTES]) {
    return guarded([&] { create_dist_plan(out, n, m, l, box, omega, nu_local, flags, device, rank, nranks, id, false); });
}

int lsfc_dist_sim_plan_create_gv3d(lsfc_plan** out, int64_t n, int64_t m, int64_t l, double box, double omega, const double* nu_local,
                                   unsigned flags, int device, int rank, int nranks) {
    return guarded([&] { create_dist_plan(out, n, m, l, box, omega, nu_local, flags, device, rank, nranks, nullptr, true); });
}

int lsfc_dist_sim_apply(lsfc_plan** plans, int nranks, const double* const* x, double* const* y, int mode) {
    return guarded([&] {
        LSFC_REQUIRE(plans && x && y && nranks >= 1, "bad argument");
        for (int r = 0; r < nranks; ++r)
            LSFC_REQUIRE(plans[r] && plans[r]->dist && plans[r]->dist->sim && plans[r]->dist->nranks == nranks && plans[r]->dist->rank == r,
                         "plan %d is not simulated rank %d of %d", r, r, nranks);
        LSFC_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (apply), 1 (convolve) or 2 (convolve with nu)");
        LSFC_HIP(hipSetDevice(plans[0]->device));
        const int64_t B = block_elems(plans[0]);
        auto sync_all = [&] { for (int r = 0; r < nranks; ++r) LSFC_HIP(hipStreamSynchronize(plans[r]->stream)); };
        // host vectors: stage through each plan's staging buffers
        std::vector<const cplx*> xd((size_t)nranks); std::vector<cplx*> yd((size_t)nranks);
        for (int r = 0; r < nranks; ++r) {
            lsfc_plan* p = plans[r];
            if (p->xs.n < (size_t)p->N) { p->xs.alloc((size_t)p->N); p->ys.alloc((size_t)p->N); }
            LSFC_HIP(hipMemcpy(p->xs.p, x[r], (size_t)p->N * sizeof(cplx), hipMemcpyHostToDevice));
            xd[r] = p->xs.p; yd[r] = p->ys.p;
        }
        const int K = plans[0]->dist->K;
        for (int r = 0; r < nranks; ++r) phase1(plans[r], xd[r], mode != 1, plans[r]->stream);
        sync_all();
        for (int c = 0; c < K; ++c) {
            // way in: block (q*K + c) of rank r's S1 -> slot r of chunk c of rank q's R1
            for (int r = 0; r < nranks; ++r) for (int q = 0; q < nranks; ++q)
                LSFC_HIP(hipMemcpy(plans[q]->dist->R1.p + ((int64_t)c * nranks + r) * B, plans[r]->dist->S1.p + ((int64_t)q * K + c) * B,
                                   (size_t)B * sizeof(cplx), hipMemcpyDeviceToDevice));
            for (int r = 0; r < nranks; ++r) phase2(plans[r], c, plans[r]->stream);
            sync_all();
            // way back: slot q of chunk c of rank r's R1 -> block (r*K + c) of rank q's S1
            for (int r = 0; r < nranks; ++r) for (int q = 0; q < nranks; ++q)
                LSFC_HIP(hipMemcpy(plans[q]->dist->S1.p + ((int64_t)r * K + c) * B, plans[r]->dist->R1.p + ((int64_t)c * nranks + q) * B,
                                   (size_t)B * sizeof(cplx), hipMemcpyDeviceToDevice));
        }
        for (int r = 0; r < nranks; ++r) {
            lsfc_plan* p = plans[r];
            if (mode == 0) phase3(p, xd[r], yd[r], 1.0, p->omega * p->omega, p->stream); else phase3(p, xd[r], yd[r], 0.0, 1.0, p->stream);
        }
        sync_all();
        for (int r = 0; r < nranks; ++r) LSFC_HIP(hipMemcpy(y[r], yd[r], (size_t)plans[r]->N * sizeof(cplx), hipMemcpyDeviceToHost));
    });
}

} // extern "C"
