// Restarted, left-preconditioned GMRES on the device, arithmetic of
// IterativeSolvers.jl gmres! (call sites: examples/example.jl:85,91,
// examples/example3D.jl:78): Arnoldi with modified / classical / DGKS
// Gram-Schmidt, residual estimate from the null vector of the Hessenberg matrix,
// Givens least squares at restart or convergence.  The Krylov basis stays in HBM;
// only O(restart) scalars per step cross to the host.
#include "plan.hpp"
#include "pointwise.hpp"
#include <cmath>
#include <cstdlib>
#include <complex>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace lsfc {

using zc = std::complex<double>;

GmresWorkspace::~GmresWorkspace() {
    if (hpin) (void)hipHostFree(hpin);
    if (vpin) (void)hipHostFree(vpin);
    for (auto& e : fetched) if (e) (void)hipEventDestroy(e);
}

// vectors = false: only the pinned host vector (the root of a multi-device plan, whose Krylov basis lives in its ranks)
static GmresWorkspace* workspace(lsfc_plan* p, int restart, bool need_vpin, bool vectors = true) {
    if (!p->gmres || (vectors && p->gmres->restart < restart)) {
        p->gmres.reset(new GmresWorkspace());
        GmresWorkspace* w = p->gmres.get();
        if (vectors) {
            w->restart = restart;
            w->V.alloc((size_t)(restart + 1) * (size_t)p->N);
            w->hdev.alloc((size_t)restart + 2);
            w->ydev.alloc((size_t)restart + 2);
            w->partial.alloc((size_t)blas_partial_count());
            w->ax.alloc((size_t)p->N);
            LSFC_HIP(hipHostMalloc((void**)&w->hpin, 2 * ((size_t)restart + 2) * sizeof(cplx)));
            for (auto& e : w->fetched) LSFC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
    }
    if (need_vpin && !p->gmres->vpin) LSFC_HIP(hipHostMalloc((void**)&p->gmres->vpin, (size_t)p->N * sizeof(cplx)));
    return p->gmres.get();
}

// LinearAlgebra.givensAlgorithm for complex f, g: [c s; -conj(s) c] [f; g] = [r; 0], c real
static void givens(zc f, zc g, double& c, zc& s) {
    if (g == zc(0)) { c = 1.0; s = 0.0; return; }
    if (f == zc(0)) { c = 0.0; s = std::conj(g) / std::abs(g); return; }
    const double d = std::hypot(std::abs(f), std::abs(g));
    c = std::abs(f) / d;
    s = (f / std::abs(f)) * std::conj(g) / d;
}

// hessenberg.jl: least squares of the k x (k-1) Hessenberg block against beta*e1
static void solve_least_squares(const std::vector<zc>& H, int ldh, double beta, int k, std::vector<zc>& y) {
    const int width = k - 1;
    std::vector<zc> Hh((size_t)k * width), rhs((size_t)k, zc(0));
    for (int j = 0; j < width; ++j) for (int i = 0; i < k; ++i) Hh[i + (size_t)k * j] = H[i + (size_t)ldh * j];
    rhs[0] = beta;
    auto A = [&](int i, int j) -> zc& { return Hh[i + (size_t)k * j]; };
    for (int i = 0; i < width; ++i) {
        double c; zc s; givens(A(i, i), A(i + 1, i), c, s);
        A(i, i) = c * A(i, i) + s * A(i + 1, i);
        for (int j = i + 1; j < width; ++j) {
            const zc tmp = -std::conj(s) * A(i, j) + c * A(i + 1, j);
            A(i, j) = c * A(i, j) + s * A(i + 1, j);
            A(i + 1, j) = tmp;
        }
        const zc tmp = -std::conj(s) * rhs[i] + c * rhs[i + 1];
        rhs[i] = c * rhs[i] + s * rhs[i + 1];
        rhs[i + 1] = tmp;
    }
    y.assign((size_t)width, zc(0));
    for (int i = width - 1; i >= 0; --i) {
        zc acc = rhs[i];
        for (int j = i + 1; j < width; ++j) acc -= A(i, j) * y[j];
        y[i] = acc / A(i, i);
    }
}

// The solver is written once over a TEAM of members, each holding one slab of every vector on its own device:
//   one member   ordinary plans, and one rank of a multi-PROCESS slab plan (inner products completed by ncclAllReduce)
//   P members    the ranks of a single-process multi-DEVICE plan, all driven from the calling thread
// Every vector operation is enqueued member by member; the O(restart) scalars of a step are identical on all members
// after the reduction and are read back from member 0.
namespace {
struct Member { lsfc_plan* p; GmresWorkspace* w; cplx* x; const cplx* b; };

// lsfc_gmres_batch: every right-hand side runs the ordinary solver on its own host thread; the threads meet here
// whenever they need the operator, and the last one to arrive applies it to all of them in ONE batched pass of the
// pipeline (plan_convolve_batch_dev).  A solve that has converged leaves the group.
struct ApplyBatcher {
    lsfc_plan* p; int active;
    std::mutex mu, cb_mu;                // cb_mu serialises the user's preconditioner callback
    std::condition_variable cv;
    int waiting = 0; uint64_t gen = 0;
    std::vector<const cplx*> in; std::vector<cplx*> out;
    std::exception_ptr err;
    void run_locked() {
        try {
            LSFC_HIP(hipSetDevice(p->device));
            for (size_t j0 = 0; j0 < in.size(); j0 += LSFC_MAX_BATCH) {
                const int cnt = (int)std::min<size_t>(LSFC_MAX_BATCH, in.size() - j0);
                VecBatch vb{};
                for (int j = 0; j < cnt; ++j) { vb.x[j] = in[j0 + j]; vb.y[j] = out[j0 + j]; }
                plan_convolve_batch_dev(p, cnt, vb, true, 1.0, p->omega * p->omega);
            }
        } catch (...) { err = std::current_exception(); }
        in.clear(); out.clear(); waiting = 0; ++gen;
        cv.notify_all();
    }
    void apply(const cplx* x, cplx* y) {
        std::unique_lock<std::mutex> lk(mu);
        in.push_back(x); out.push_back(y); ++waiting;
        if (waiting == active) run_locked();
        else { const uint64_t g = gen; cv.wait(lk, [&] { return gen != g; }); }
        if (err) std::rethrow_exception(err);
    }
    void leave() {
        std::unique_lock<std::mutex> lk(mu);
        --active;
        if (active > 0 && waiting == active) run_locked();
    }
};

struct Team {
    lsfc_plan* root;
    std::vector<Member> mem;
    bool reduce = false;                // local inner products are partial sums
    GmresWorkspace* rootw = nullptr;    // pinned host vector of the preconditioner callback
    ApplyBatcher* batcher = nullptr;    // lsfc_gmres_batch: operator applications go through the rendezvous

    static void dev(const Member& m) { LSFC_HIP(hipSetDevice(m.p->device)); }
    static cplx* V(const Member& m, int j) { return m.w->V.p + (size_t)j * (size_t)m.p->N; }
    template <class F> void each(F&& f) { for (auto& m : mem) { dev(m); f(m); } }

    void apply(const std::function<const cplx*(const Member&)>& in, const std::function<cplx*(const Member&)>& out) {
        if (batcher) { batcher->apply(in(mem[0]), out(mem[0])); return; }
        if (!root->multi) { dev(mem[0]); plan_apply_dev(root, in(mem[0]), out(mem[0])); return; }
        std::vector<const cplx*> xi; std::vector<cplx*> yo;
        for (auto& m : mem) { xi.push_back(in(m)); yo.push_back(out(m)); }
        multi_convolve_dev(root, xi.data(), yo.data(), true, 1.0, root->omega * root->omega);
    }
    void allreduce(int off, int count) {
        if (!reduce) return;
        if (!root->multi) { dev(mem[0]); dist_allreduce_sum(root, mem[0].w->hdev.p + off, count); return; }
        std::vector<cplx*> ptr;
        for (auto& m : mem) ptr.push_back(m.w->hdev.p + off);
        multi_allreduce_sum(root, ptr.data(), count);
    }
    void finish_nrm(int off) {
        if (!reduce) return;
        allreduce(off, 1);
        each([&](Member& m) { blas_sqrt_dev(m.w->hdev.p + off, m.p->stream); });
    }
    // scalars of the step -> pinned host memory of member 0, slot 0 or 1: posted behind the step's kernels, awaited when
    // the host needs them (a pipelined solve posts the next step's kernels in between)
    void fetch_post(int count, int slot) {
        Member& m = mem[0];
        dev(m);
        LSFC_HIP(hipMemcpyAsync(hpin(slot), m.w->hdev.p, (size_t)count * sizeof(cplx), hipMemcpyDeviceToHost, m.p->stream));
        LSFC_HIP(hipEventRecord(m.w->fetched[slot], m.p->stream));
    }
    void fetch_wait(int slot) { dev(mem[0]); LSFC_HIP(hipEventSynchronize(mem[0].w->fetched[slot])); }
    void fetch_h(int count, int slot = 0) { fetch_post(count, slot); fetch_wait(slot); }
    cplx* hpin(int slot = 0) { return mem[0].w->hpin + (size_t)slot * ((size_t)mem[0].w->restart + 2); }
    void sync() { each([](Member& m) { LSFC_HIP(hipStreamSynchronize(m.p->stream)); }); }
};

void solve(Team& T, const lsfc_gmres_opts* opts_in, double* resnorm, int64_t cap, lsfc_gmres_result* res, int restart,
           int64_t maxiter, double reltol, double abstol, const lsfc_gmres_opts& o) {
    const int64_t Ntot = T.root->N;

    auto precondition = [&](int col) {
        if (!o.precond) return;
        std::unique_lock<std::mutex> cb_lock;
        if (T.batcher) cb_lock = std::unique_lock<std::mutex>(T.batcher->cb_mu);
        if (o.precond_on_device) {
            LSFC_REQUIRE(T.mem.size() == 1, "a device-resident preconditioner callback needs a single-device plan");
            const int rc = o.precond(o.precond_user, (double*)Team::V(T.mem[0], col), T.mem[0].p->N);
            if (rc != 0) fail(LSFC_EINVAL, "preconditioner callback returned %d", rc);
            return;
        }
        // host callback, in place on the whole vector (the two-argument ldiv!): gather the slabs, call, scatter
        cplx* vpin = T.rootw->vpin;
        int64_t off = 0;
        T.each([&](Member& m) { LSFC_HIP(hipMemcpyAsync(vpin + off, Team::V(m, col), (size_t)m.p->N * sizeof(cplx), hipMemcpyDeviceToHost, m.p->stream)); off += m.p->N; });
        T.sync();
        const int rc = o.precond(o.precond_user, (double*)vpin, T.mem.size() == 1 ? T.mem[0].p->N : Ntot);
        if (rc != 0) fail(LSFC_EINVAL, "preconditioner callback returned %d", rc);
        off = 0;
        T.each([&](Member& m) { LSFC_HIP(hipMemcpyAsync(Team::V(m, col), vpin + off, (size_t)m.p->N * sizeof(cplx), hipMemcpyHostToDevice, m.p->stream)); off += m.p->N; });
    };

    // init!: V1 = Pl \ (b - A x) normalised, returns beta
    auto init = [&](bool skip_mv) -> double {
        if (skip_mv) T.each([&](Member& m) { LSFC_HIP(hipMemcpyAsync(Team::V(m, 0), m.b, (size_t)m.p->N * sizeof(cplx), hipMemcpyDeviceToDevice, m.p->stream)); });
        else {
            T.apply([](const Member& m) { return (const cplx*)m.x; }, [](const Member& m) { return m.w->ax.p; });
            T.each([&](Member& m) { blas_sub(Team::V(m, 0), m.b, m.w->ax.p, m.p->N, m.p->stream); });
        }
        precondition(0);
        T.each([&](Member& m) { blas_nrm2(Team::V(m, 0), m.w->partial.p, m.w->hdev.p, m.p->N, m.p->stream, T.reduce); });
        T.finish_nrm(0);
        T.each([&](Member& m) { blas_scale_inv_dev(Team::V(m, 0), m.w->hdev.p, m.p->N, m.p->stream); });
        T.fetch_h(1);
        return T.hpin()[0].x;
    };
    // h[j0..j0+k) = V[:, 0..k)' w  and  w -= V h   (classical Gram-Schmidt sweep over the first k columns), then ||w||
    // one device, no all-reduce between a reduction and its consumer: the fused kernels (pointwise.hip) sum block partials
    // inside the consuming kernel -- same summation order, bit-identical scalars, roughly half the launches
    const bool fused = !T.reduce && T.mem.size() == 1;
    // modified Gram-Schmidt in blocks (pointwise.hip: k_mgs_block) where the sweep is bound by HBM traffic: vectors of >= 2^22
    // entries (LSFC_MGS_BLOCK=0 / 1 forces the strict one-vector-at-a-time sweep / the blocked one)
    const char* mb_env = getenv("LSFC_MGS_BLOCK");
    const bool mgs_blocked = mb_env ? atoi(mb_env) != 0 : Ntot >= ((int64_t)1 << 22);
    auto slot = [&](int s2) { return T.mem[0].w->partial.p + (size_t)s2 * (size_t)blas_partial_slot(); };
    auto cgs_sweep = [&](int k, bool scale_now, int hslot = 0, bool wait = true) {
        if (fused && k <= 64) {
            Member& m = T.mem[0];
            Team::dev(m);
            blas_multidot_partial(Team::V(m, 0), m.p->N, k, Team::V(m, k), slot(0), m.p->N, m.p->stream);
            blas_cgs_update_fused(Team::V(m, k), Team::V(m, 0), m.p->N, k, slot(0), m.w->hdev.p, slot(64), m.p->N, m.p->stream);
            if (scale_now) blas_scale_inv_fused(Team::V(m, k), slot(64), m.w->hdev.p + k, m.p->N, m.p->stream);
            else blas_finish_norm(slot(64), m.w->hdev.p + k, m.p->N, m.p->stream);
            T.fetch_post(k + 1, hslot);
            if (wait) T.fetch_wait(hslot);
            return;
        }
        T.each([&](Member& m) {
            for (int j0 = 0; j0 < k; j0 += 64) {
                const int kc = std::min(64, k - j0);
                blas_multidot(Team::V(m, j0), m.p->N, kc, Team::V(m, k), m.w->partial.p, m.w->hdev.p + j0, m.p->N, m.p->stream);
            }
        });
        T.allreduce(0, k);
        T.each([&](Member& m) {
            for (int j0 = 0; j0 < k; j0 += 64) {
                const int kc = std::min(64, k - j0);
                blas_gemv_acc(Team::V(m, k), Team::V(m, j0), m.p->N, kc, m.w->hdev.p + j0, -1.0, m.p->N, m.p->stream);
            }
            blas_nrm2(Team::V(m, k), m.w->partial.p, m.w->hdev.p + k, m.p->N, m.p->stream, T.reduce);
        });
        T.finish_nrm(k);
        if (scale_now) T.each([&](Member& m) { blas_scale_inv_dev(Team::V(m, k), m.w->hdev.p + k, m.p->N, m.p->stream); });
        T.fetch_post(k + 1, hslot);
        if (wait) T.fetch_wait(hslot);
    };

    std::vector<zc> H((size_t)(restart + 1) * restart, zc(0));
    const int ldh = restart + 1;
    std::vector<zc> nullvec((size_t)restart + 1, zc(1));
    int64_t mvps = o.initially_zero ? 1 : 0;
    double beta = init(o.initially_zero != 0);
    double current = beta, accumulator = 1.0;
    nullvec[0] = 1.0;
    const double tol = std::max(reltol * current, abstol);
    int k = 1; int64_t iteration = 0;
    std::vector<zc> y;

    // One step ahead: on one device nothing the host computes feeds back into the kernels of a step (the fused kernels read
    // their scalars from device memory), so the host may post step k + 1 before it has seen the scalars of step k and the
    // device never idles through the read-back, the Hessenberg update and the next launches.  The step posted past
    // convergence is surplus (its column is not used); none is posted past maxiter or past the restart length.  Worth it
    // where a step is short (small grids); off for DGKS (the host decides about the second sweep) and for callbacks of
    // the caller (they would see one call more than the reference makes) -- the library's own device preconditioner
    // (lsfc_precond_callback: stream-ordered, no side effects) is fine.
    const char* lag_env = getenv("LSFC_GMRES_LOOKAHEAD");
    const bool own_precond = o.precond_on_device && o.precond == &lsfc_precond_callback;
    const bool lookahead = fused && !T.batcher && o.orth != LSFC_ORTH_DGKS && (!o.precond || own_precond) &&
                           (lag_env ? atoi(lag_env) != 0 : Ntot <= ((int64_t)1 << 22));
    // expand!: V[:,kk+1] = A V[:,kk], ldiv!(Pl, V[:,kk+1]), orthogonalise; the scalars travel to pinned slot kk & 1
    auto post_step = [&](int kk) {
        const int hs = kk & 1;
        T.apply([kk](const Member& m) { return (const cplx*)Team::V(m, kk - 1); }, [kk](const Member& m) { return Team::V(m, kk); });
        precondition(kk);
        if (o.orth == LSFC_ORTH_MGS && fused && mgs_blocked) {
            // blocks of MB basis vectors: one pass takes the inner products with a whole block (and of its vectors with each
            // other), the next one recovers the MGS coefficients, updates w with the block and takes the next block's products
            Member& m = T.mem[0];
            Team::dev(m);
            const int MB = blas_mgs_block_size(), S = blas_mgs_slots();
            const int nblk = (kk + MB - 1) / MB;
            blas_mgs_block(Team::V(m, kk), nullptr, 0, nullptr, nullptr, Team::V(m, 0), std::min(MB, kk), slot(0), m.p->N, m.p->N, m.p->stream);
            for (int b = 0; b < nblk; ++b) {
                const int i0 = b * MB, mp = std::min(MB, kk - i0), mn = std::max(0, std::min(MB, kk - i0 - MB));
                blas_mgs_block(Team::V(m, kk), Team::V(m, i0), mp, slot((b & 1) * S), m.w->hdev.p + i0, mn ? Team::V(m, i0 + MB) : nullptr, mn,
                               slot(((b + 1) & 1) * S), m.p->N, m.p->N, m.p->stream);
            }
            blas_scale_inv_fused(Team::V(m, kk), slot((nblk & 1) * S), m.w->hdev.p + kk, m.p->N, m.p->stream);
            T.fetch_post(kk + 1, hs);
        } else if (o.orth == LSFC_ORTH_MGS && fused) {
            Member& m = T.mem[0];
            Team::dev(m);
            blas_dot_partial(Team::V(m, 0), Team::V(m, kk), slot(0), m.p->N, m.p->stream);
            for (int i = 0; i < kk; ++i)
                blas_axpy_dot_fused(Team::V(m, kk), Team::V(m, i), slot(i & 1), m.w->hdev.p + i, (i + 1 < kk) ? Team::V(m, i + 1) : nullptr,
                                    slot((i + 1) & 1), m.p->N, m.p->stream);
            blas_scale_inv_fused(Team::V(m, kk), slot(kk & 1), m.w->hdev.p + kk, m.p->N, m.p->stream);
            T.fetch_post(kk + 1, hs);
        } else if (o.orth == LSFC_ORTH_MGS) {
            // h_i = <V_i, w>; w -= h_i V_i, each sweep fused with the next inner product (norm after the last)
            T.each([&](Member& m) { blas_dot(Team::V(m, 0), Team::V(m, kk), m.w->partial.p, m.w->hdev.p, m.p->N, m.p->stream); });
            T.allreduce(0, 1);
            for (int i = 0; i < kk; ++i) {
                T.each([&](Member& m) {
                    blas_axpy_dot(Team::V(m, kk), Team::V(m, i), m.w->hdev.p + i, (i + 1 < kk) ? Team::V(m, i + 1) : nullptr, m.w->partial.p,
                                  m.w->hdev.p + i + 1, m.p->N, m.p->stream, T.reduce);
                });
                if (i + 1 < kk) T.allreduce(i + 1, 1); else T.finish_nrm(i + 1);
            }
            T.each([&](Member& m) { blas_scale_inv_dev(Team::V(m, kk), m.w->hdev.p + kk, m.p->N, m.p->stream); });
            T.fetch_post(kk + 1, hs);
        } else {
            cgs_sweep(kk, o.orth == LSFC_ORTH_CGS, hs, false);
        }
    };

    int posted = 0;                                         // columns of this cycle whose kernels are in the stream (>= k - 1)
    while (!(iteration >= maxiter || current <= tol)) {
        // steps in flight: the one the host is about to read and, with lookahead, the one after it
        const int want = k + (lookahead ? 1 : 0);
        while (posted < want && posted < restart && iteration + (posted - (k - 1)) < maxiter) { ++posted; post_step(posted); }
        ++mvps;
        double nrm;
        cplx* hp = T.hpin(k & 1);
        T.fetch_wait(k & 1);
        nrm = hp[k].x;
        if (o.orth == LSFC_ORTH_DGKS) {
            // IterativeSolvers orthogonalize.jl: `while nrm < projection_size / sqrt(2)`, projection_size being
            // the norm of the latest correction; the corrections accumulate into the Hessenberg column
            double proj = 0.0;
            for (int i = 0; i < k; ++i) proj += hp[i].x * hp[i].x + hp[i].y * hp[i].y;
            proj = std::sqrt(proj);
            std::vector<cplx> hsum(hp, hp + k);
            bool again = false;
            for (int pass = 0; nrm < proj / std::sqrt(2.0) && pass < 8; ++pass) {     // (8: guard against a stagnating loop)
                again = true;
                cgs_sweep(k, false, k & 1, true);
                proj = 0.0;
                for (int i = 0; i < k; ++i) {
                    proj += hp[i].x * hp[i].x + hp[i].y * hp[i].y;
                    hsum[(size_t)i].x += hp[i].x; hsum[(size_t)i].y += hp[i].y;
                }
                proj = std::sqrt(proj);
                nrm = hp[k].x;
            }
            if (again) for (int i = 0; i < k; ++i) hp[i] = hsum[(size_t)i];
            T.each([&](Member& m) { blas_scale_inv_dev(Team::V(m, k), m.w->hdev.p + k, m.p->N, m.p->stream); });
        }
        for (int i = 0; i < k; ++i) H[i + (size_t)ldh * (k - 1)] = zc(hp[i].x, hp[i].y);
        H[k + (size_t)ldh * (k - 1)] = nrm;

        // update_residual!: nullvec[k+1] = -conj(dot(nullvec[1:k], H[1:k,k]) / H[k+1,k])
        zc acc(0);
        for (int i = 0; i < k; ++i) acc += std::conj(nullvec[i]) * H[i + (size_t)ldh * (k - 1)];
        nullvec[k] = -std::conj(acc / nrm);
        accumulator += std::norm(nullvec[k]);
        current = beta / std::sqrt(accumulator);
        ++k;

        if (k == restart + 1 || current <= tol) {
            solve_least_squares(H, ldh, beta, k, y);
            T.sync();                                       // (members other than 0 may still read their pinned scalars)
            T.each([&](Member& m) {
                for (int i = 0; i < k - 1; ++i) m.w->hpin[i] = make_double2(y[i].real(), y[i].imag());
                LSFC_HIP(hipMemcpyAsync(m.w->ydev.p, m.w->hpin, (size_t)(k - 1) * sizeof(cplx), hipMemcpyHostToDevice, m.p->stream));
                for (int j0 = 0; j0 < k - 1; j0 += 64) {
                    const int kc = std::min(64, k - 1 - j0);
                    blas_gemv_acc(m.x, Team::V(m, j0), m.p->N, kc, m.w->ydev.p + j0, +1.0, m.p->N, m.p->stream);      // x += V y
                }
            });
            T.sync();
            k = 1; posted = 0;
            if (!(current <= tol)) {
                beta = init(false);
                accumulator = 1.0;
                nullvec[0] = 1.0;
                ++mvps;
            }
        }
        if (resnorm && iteration < cap) resnorm[iteration] = current;
        ++iteration;
    }
    T.sync();
    res->iters = iteration; res->mvps = mvps; res->converged = current <= tol ? 1 : 0; res->final_resnorm = current;
    (void)opts_in;
}

struct Resolved { lsfc_gmres_opts o; int restart; int64_t maxiter; double reltol, abstol; };
Resolved resolve(const lsfc_gmres_opts* opts_in, int64_t N) {
    Resolved r;
    if (opts_in) r.o = *opts_in;
    else { r.o.restart = 0; r.o.maxiter = 0; r.o.reltol = -1; r.o.abstol = 0; r.o.orth = LSFC_ORTH_MGS; r.o.initially_zero = 0; r.o.precond = nullptr; r.o.precond_user = nullptr; r.o.precond_on_device = 0; }
    r.restart = (int)(r.o.restart > 0 ? r.o.restart : std::min<int64_t>(20, N));
    r.maxiter = r.o.maxiter > 0 ? r.o.maxiter : N;
    r.reltol = r.o.reltol >= 0 ? r.o.reltol : std::sqrt(2.220446049250313e-16);
    r.abstol = r.o.abstol > 0 ? r.o.abstol : 0.0;
    LSFC_REQUIRE(r.o.orth == LSFC_ORTH_MGS || r.o.orth == LSFC_ORTH_CGS || r.o.orth == LSFC_ORTH_DGKS, "unknown orthogonalisation %d", r.o.orth);
    LSFC_REQUIRE(r.restart >= 1, "restart must be >= 1");
    return r;
}
} // namespace

void gmres_run(lsfc_plan* p, cplx* x, const cplx* b, const lsfc_gmres_opts* opts_in, double* resnorm, int64_t cap, lsfc_gmres_result* res) {
    const Resolved r = resolve(opts_in, p->N);
    Team T; T.root = p;
    GmresWorkspace* w = workspace(p, r.restart, r.o.precond != nullptr && !r.o.precond_on_device);
    T.mem.push_back({p, w, x, b});
    T.rootw = w;
    // slab-distributed plan (one process per GPU): every inner product / squared norm is completed by an all-reduce over the ranks
    T.reduce = p->dist && !p->dist->sim && !p->dist->member && (p->dist->nranks > 1 || p->dist->force_comm);
    solve(T, opts_in, resnorm, cap, res, r.restart, r.maxiter, r.reltol, r.abstol, r.o);
}

void gmres_run_batch(lsfc_plan* p, cplx* x, const cplx* b, int nrhs, const lsfc_gmres_opts* opts_in, double* resnorm, int64_t cap,
                     lsfc_gmres_result* res) {
    const Resolved r = resolve(opts_in, p->N);
    const bool need_vpin = r.o.precond != nullptr && !r.o.precond_on_device;
    // one workspace (Krylov basis, scalars, pinned buffers) per right-hand side: nrhs * (restart + 2) vectors of N complex.
    // Checked against the free device memory up front (a failed hipMalloc half way through leaves a half-built batch), and
    // released again when the call returns -- the single-solve workspace of the plan is the one that is kept across calls.
    {
        size_t free_b = 0, total_b = 0;
        LSFC_HIP(hipMemGetInfo(&free_b, &total_b));
        size_t have = 0;
        for (auto& w : p->gmres_batch) if (w && w->restart >= r.restart) have += w->V.bytes() + w->ax.bytes();
        const double need = (double)nrhs * ((double)r.restart + 2.0) * (double)p->N * sizeof(cplx);
        if (need > (double)free_b + (double)have)
            fail(LSFC_ENOMEM, "lsfc_gmres_batch: %d right-hand sides x (restart %d + 2) vectors of %lld complex need %.1f GB of device memory, %.1f GB are free "
                 "-- solve fewer right-hand sides per call or lower the restart length", nrhs, r.restart, (long long)p->N, need / 1e9, ((double)free_b + (double)have) / 1e9);
    }
    struct Release { lsfc_plan* p; ~Release() { p->gmres_batch.clear(); } } release_on_return{p};
    if ((int)p->gmres_batch.size() < nrhs) p->gmres_batch.resize((size_t)nrhs);
    for (int j = 0; j < nrhs; ++j) {
        std::unique_ptr<GmresWorkspace> keep = std::move(p->gmres);
        p->gmres = std::move(p->gmres_batch[(size_t)j]);
        workspace(p, r.restart, need_vpin);
        p->gmres_batch[(size_t)j] = std::move(p->gmres);
        p->gmres = std::move(keep);
    }
    LSFC_HIP(hipStreamSynchronize(p->stream));
    ApplyBatcher batcher; batcher.p = p; batcher.active = nrhs;
    std::vector<std::exception_ptr> errs((size_t)nrhs);
    std::vector<std::thread> th;
    for (int j = 0; j < nrhs; ++j)
        th.emplace_back([&, j] {
            try {
                LSFC_HIP(hipSetDevice(p->device));
                Team T; T.root = p; T.batcher = &batcher;
                GmresWorkspace* w = p->gmres_batch[(size_t)j].get();
                T.mem.push_back({p, w, x + (int64_t)j * p->N, b + (int64_t)j * p->N});
                T.rootw = w;
                solve(T, opts_in, resnorm ? resnorm + (int64_t)j * cap : nullptr, cap, res + j, r.restart, r.maxiter, r.reltol, r.abstol, r.o);
            } catch (...) { errs[(size_t)j] = std::current_exception(); }
            batcher.leave();
        });
    for (auto& t : th) t.join();
    LSFC_HIP(hipSetDevice(p->device));
    LSFC_HIP(hipStreamSynchronize(p->stream));
    for (auto& e : errs) if (e) std::rethrow_exception(e);
}

void gmres_run_multi(lsfc_plan* root, cplx* x_host, const cplx* b_host, const lsfc_gmres_opts* opts_in, double* resnorm, int64_t cap,
                     lsfc_gmres_result* res) {
    LSFC_REQUIRE(root->multi, "not a multi-device plan");
    const Resolved r = resolve(opts_in, root->N);
    LSFC_REQUIRE(!(r.o.precond && r.o.precond_on_device), "a multi-device plan takes a host preconditioner callback");
    Team T; T.root = root;
    T.rootw = workspace(root, r.restart, r.o.precond != nullptr, false);
    T.reduce = root->multi->P > 1;
    int64_t off = 0;
    for (auto& sp : root->multi->sub) {
        lsfc_plan* p = sp.get();
        LSFC_HIP(hipSetDevice(p->device));
        GmresWorkspace* w = workspace(p, r.restart, false);
        if (p->xs.n < (size_t)p->N) { p->xs.alloc((size_t)p->N); p->ys.alloc((size_t)p->N); }
        LSFC_HIP(hipMemcpyAsync(p->xs.p, x_host + off, (size_t)p->N * sizeof(cplx), hipMemcpyHostToDevice, p->stream));
        LSFC_HIP(hipMemcpyAsync(p->ys.p, b_host + off, (size_t)p->N * sizeof(cplx), hipMemcpyHostToDevice, p->stream));
        T.mem.push_back({p, w, p->xs.p, p->ys.p});
        off += p->N;
    }
    solve(T, opts_in, resnorm, cap, res, r.restart, r.maxiter, r.reltol, r.abstol, r.o);
    off = 0;
    for (auto& m : T.mem) {
        Team::dev(m);
        LSFC_HIP(hipMemcpyAsync(x_host + off, m.x, (size_t)m.p->N * sizeof(cplx), hipMemcpyDeviceToHost, m.p->stream));
        off += m.p->N;
    }
    T.sync();
    LSFC_HIP(hipSetDevice(root->device));
}

} // namespace lsfc
