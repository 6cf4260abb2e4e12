// Restarted, left-preconditioned GMRES on the device, arithmetic of
// IterativeSolvers.jl gmres! (call sites: examples/example.jl:85,91,
// examples/example3D.jl:78): Arnoldi with modified / classical / DGKS
// Gram-Schmidt, residual estimate from the null vector of the Hessenberg matrix,
// Givens least squares at restart or convergence.  The Krylov basis stays in HBM;
// only O(restart) scalars per step cross to the host.
#include "plan.hpp"
#include "pointwise.hpp"
#include <cmath>
#include <complex>
#include <vector>

namespace lsfc {

using zc = std::complex<double>;

GmresWorkspace::~GmresWorkspace() {
    if (hpin) (void)hipHostFree(hpin);
    if (vpin) (void)hipHostFree(vpin);
}

static GmresWorkspace* workspace(lsfc_plan* p, int restart, bool need_vpin) {
    if (!p->gmres || p->gmres->restart < restart) {
        p->gmres.reset(new GmresWorkspace());
        GmresWorkspace* w = p->gmres.get();
        w->restart = restart;
        w->V.alloc((size_t)(restart + 1) * (size_t)p->N);
        w->hdev.alloc((size_t)restart + 2);
        w->ydev.alloc((size_t)restart + 2);
        w->partial.alloc((size_t)blas_partial_count());
        w->ax.alloc((size_t)p->N);
        LSFC_HIP(hipHostMalloc((void**)&w->hpin, ((size_t)restart + 2) * sizeof(cplx)));
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

void gmres_run(lsfc_plan* p, cplx* x, const cplx* b, const lsfc_gmres_opts* opts_in, double* resnorm, int64_t cap, lsfc_gmres_result* res) {
    lsfc_gmres_opts o;
    if (opts_in) o = *opts_in; else { o.restart = 0; o.maxiter = 0; o.reltol = -1; o.abstol = 0; o.orth = LSFC_ORTH_MGS; o.initially_zero = 0; o.precond = nullptr; o.precond_user = nullptr; o.precond_on_device = 0; }
    const int64_t N = p->N;
    const int restart = (int)(o.restart > 0 ? o.restart : std::min<int64_t>(20, N));
    const int64_t maxiter = o.maxiter > 0 ? o.maxiter : N;
    const double reltol = o.reltol >= 0 ? o.reltol : std::sqrt(2.220446049250313e-16);
    const double abstol = o.abstol > 0 ? o.abstol : 0.0;
    LSFC_REQUIRE(o.orth == LSFC_ORTH_MGS || o.orth == LSFC_ORTH_CGS || o.orth == LSFC_ORTH_DGKS, "unknown orthogonalisation %d", o.orth);
    LSFC_REQUIRE(restart >= 1, "restart must be >= 1");

    GmresWorkspace* w = workspace(p, restart, o.precond != nullptr && !o.precond_on_device);
    hipStream_t st = p->stream;
    auto Vcol = [&](int j) { return w->V.p + (size_t)j * (size_t)N; };
    // slab-distributed plan: every inner product / squared norm is completed by an all-reduce over the ranks
    const bool multi = p->dist && !p->dist->sim && (p->dist->nranks > 1 || p->dist->force_comm);
    auto finish_dot = [&](cplx* s, int count) { if (multi) dist_allreduce_sum(p, s, count); };
    auto finish_nrm = [&](cplx* s) { if (multi) { dist_allreduce_sum(p, s, 1); blas_sqrt_dev(s, st); } };

    auto precondition = [&](cplx* v) {
        if (!o.precond) return;
        if (o.precond_on_device) {
            const int rc = o.precond(o.precond_user, (double*)v, N);
            if (rc != 0) fail(LSFC_EINVAL, "preconditioner callback returned %d", rc);
            return;
        }
        LSFC_HIP(hipMemcpyAsync(w->vpin, v, (size_t)N * sizeof(cplx), hipMemcpyDeviceToHost, st));
        LSFC_HIP(hipStreamSynchronize(st));
        const int rc = o.precond(o.precond_user, (double*)w->vpin, N);
        if (rc != 0) fail(LSFC_EINVAL, "preconditioner callback returned %d", rc);
        LSFC_HIP(hipMemcpyAsync(v, w->vpin, (size_t)N * sizeof(cplx), hipMemcpyHostToDevice, st));
    };
    auto fetch_h = [&](int count) {
        LSFC_HIP(hipMemcpyAsync(w->hpin, w->hdev.p, (size_t)count * sizeof(cplx), hipMemcpyDeviceToHost, st));
        LSFC_HIP(hipStreamSynchronize(st));
    };

    // init!: V1 = Pl \ (b - A x) normalised, returns beta
    auto init = [&](bool skip_mv) -> double {
        cplx* v0 = Vcol(0);
        if (skip_mv) LSFC_HIP(hipMemcpyAsync(v0, b, (size_t)N * sizeof(cplx), hipMemcpyDeviceToDevice, st));
        else { plan_apply_dev(p, x, w->ax.p); blas_sub(v0, b, w->ax.p, N, st); }
        precondition(v0);
        blas_nrm2(v0, w->partial.p, w->hdev.p, N, st, multi);
        finish_nrm(w->hdev.p);
        blas_scale_inv_dev(v0, w->hdev.p, N, st);
        fetch_h(1);
        return w->hpin[0].x;
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

    while (!(iteration >= maxiter || current <= tol)) {
        cplx* wv = Vcol(k);
        plan_apply_dev(p, Vcol(k - 1), wv);            // expand!: V[:,k+1] = A V[:,k]
        precondition(wv);                              //          ldiv!(Pl, V[:,k+1])
        ++mvps;
        double nrm;
        if (o.orth == LSFC_ORTH_MGS) {
            // h_i = <V_i, w>; w -= h_i V_i, each sweep fused with the next inner product (norm after the last)
            blas_dot(Vcol(0), wv, w->partial.p, w->hdev.p, N, st);
            finish_dot(w->hdev.p, 1);
            for (int i = 0; i < k; ++i) {
                blas_axpy_dot(wv, Vcol(i), w->hdev.p + i, (i + 1 < k) ? Vcol(i + 1) : nullptr, w->partial.p, w->hdev.p + i + 1, N, st, multi);
                if (i + 1 < k) finish_dot(w->hdev.p + i + 1, 1); else finish_nrm(w->hdev.p + i + 1);
            }
            blas_scale_inv_dev(wv, w->hdev.p + k, N, st);
            fetch_h(k + 1);
            nrm = w->hpin[k].x;
        } else {
            for (int j0 = 0; j0 < k; j0 += 64) {
                const int kc = std::min(64, k - j0);
                blas_multidot(Vcol(j0), N, kc, wv, w->partial.p, w->hdev.p + j0, N, st);
            }
            finish_dot(w->hdev.p, k);
            for (int j0 = 0; j0 < k; j0 += 64) {
                const int kc = std::min(64, k - j0);
                blas_gemv_acc(wv, Vcol(j0), N, kc, w->hdev.p + j0, -1.0, N, st);
            }
            blas_nrm2(wv, w->partial.p, w->hdev.p + k, N, st, multi);
            finish_nrm(w->hdev.p + k);
            fetch_h(k + 1);
            nrm = w->hpin[k].x;
            if (o.orth == LSFC_ORTH_DGKS) {
                // IterativeSolvers orthogonalize.jl: `while nrm < projection_size / sqrt(2)`, projection_size being
                // the norm of the latest correction; the corrections accumulate into the Hessenberg column
                double proj = 0.0;
                for (int i = 0; i < k; ++i) proj += w->hpin[i].x * w->hpin[i].x + w->hpin[i].y * w->hpin[i].y;
                proj = std::sqrt(proj);
                std::vector<cplx> hsum(w->hpin, w->hpin + k);
                bool again = false;
                for (int pass = 0; nrm < proj / std::sqrt(2.0) && pass < 8; ++pass) {     // (8: guard against a NaN-free but stagnating loop)
                    again = true;
                    for (int j0 = 0; j0 < k; j0 += 64) {
                        const int kc = std::min(64, k - j0);
                        blas_multidot(Vcol(j0), N, kc, wv, w->partial.p, w->hdev.p + j0, N, st);
                    }
                    finish_dot(w->hdev.p, k);
                    for (int j0 = 0; j0 < k; j0 += 64) {
                        const int kc = std::min(64, k - j0);
                        blas_gemv_acc(wv, Vcol(j0), N, kc, w->hdev.p + j0, -1.0, N, st);
                    }
                    blas_nrm2(wv, w->partial.p, w->hdev.p + k, N, st, multi);
                    finish_nrm(w->hdev.p + k);
                    fetch_h(k + 1);
                    proj = 0.0;
                    for (int i = 0; i < k; ++i) {
                        proj += w->hpin[i].x * w->hpin[i].x + w->hpin[i].y * w->hpin[i].y;
                        hsum[i].x += w->hpin[i].x; hsum[i].y += w->hpin[i].y;
                    }
                    proj = std::sqrt(proj);
                    nrm = w->hpin[k].x;
                }
                if (again) for (int i = 0; i < k; ++i) w->hpin[i] = hsum[i];
            }
            blas_scale_inv_dev(wv, w->hdev.p + k, N, st);
        }
        for (int i = 0; i < k; ++i) H[i + (size_t)ldh * (k - 1)] = zc(w->hpin[i].x, w->hpin[i].y);
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
            for (int i = 0; i < k - 1; ++i) w->hpin[i] = make_double2(y[i].real(), y[i].imag());
            LSFC_HIP(hipMemcpyAsync(w->ydev.p, w->hpin, (size_t)(k - 1) * sizeof(cplx), hipMemcpyHostToDevice, st));
            for (int j0 = 0; j0 < k - 1; j0 += 64) {
                const int kc = std::min(64, k - 1 - j0);
                blas_gemv_acc(x, Vcol(j0), N, kc, w->ydev.p + j0, +1.0, N, st);      // x += V y
            }
            LSFC_HIP(hipStreamSynchronize(st));
            k = 1;
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
    LSFC_HIP(hipStreamSynchronize(st));
    res->iters = iteration; res->mvps = mvps; res->converged = current <= tol ? 1 : 0; res->final_resnorm = current;
}

} // namespace lsfc
