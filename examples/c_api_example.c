/* Plain-C use of the lsfc ABI (include/lsfc.h): build the 3D operator with the device-side symbol generator,
 * apply it, and solve (I + w^2 G nu) u = rhs with GMRES -- the sequence of examples/example3D.jl:54-79.
 *   gcc -std=c99 -Iinclude examples/c_api_example.c -Lfast_solver_lippmann_schwinger_amd -llsfc -lm \
 *       -Wl,-rpath,$PWD/fast_solver_lippmann_schwinger_amd -o c_api_example && ./c_api_example 32
 * This process loads only the system ROCm stack (HIP runtime, rocFFT, RCCL under /opt/rocm) -- no Python, no PyTorch: the
 * situation of a Julia host that `ccall`s the library.  Besides the single-GPU solve it runs the two multi-GPU entry points
 * on whatever devices are visible: the single-process multi-device plan, and the one-process-per-GPU plan as rank 0 of 1
 * with the slab exchange routed through RCCL send/recv (LSFC_DIST_FORCE_COMM=1).                                            */
#define _POSIX_C_SOURCE 200112L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "lsfc.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != LSFC_OK && rc_ != LSFC_ENOTCONV) { \
    fprintf(stderr, "%s -> %d: %s\n", #call, rc_, lsfc_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 32;
    const long N = (long)n * n * n;
    const double h = 1.0 / n, k = 10.0;
    double* nu = malloc(N * sizeof(double));
    double* uinc = malloc(2 * N * sizeof(double));
    double* rhs = malloc(2 * N * sizeof(double));
    double* u = calloc(2 * N, sizeof(double));
    double* t = malloc(2 * N * sizeof(double));
    for (int p = 0; p < n; ++p) for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) {
        const double x = -0.5 + h * i, y = -0.5 + h * j, z = -0.5 + h * p;
        const long idx = i + (long)n * (j + (long)n * p);                 /* column-major, x fastest */
        nu[idx] = 0.3 * exp(-40 * (x * x + y * y + z * z));
        uinc[2 * idx] = cos(k * x); uinc[2 * idx + 1] = sin(k * x);       /* plane wave exp(i k x) */
    }
    lsfc_plan* plan = NULL;
    CHECK(lsfc_plan_create_gv3d(&plan, n, n, n, /*box = |x_end - x_1| + h*/ 1.0, k, nu, LSFC_FLAG_DEFAULT, 0));
    printf("%s  N=%lld  pipeline=%s\n", lsfc_version(), (long long)lsfc_plan_size(plan), lsfc_plan_pipeline(plan));
    CHECK(lsfc_apply(plan, uinc, t, LSFC_MEM_HOST));                       /* t = fastconv * u_inc */
    for (long i = 0; i < 2 * N; ++i) rhs[i] = -(t[i] - uinc[i]);           /* rhs = -(fastconv*u_inc - u_inc) */
    lsfc_gmres_opts o = {0};
    o.restart = 10; o.reltol = 1e-8; o.orth = LSFC_ORTH_MGS;
    lsfc_gmres_result r;
    double resnorm[256];
    CHECK(lsfc_gmres(plan, u, rhs, &o, resnorm, 256, &r, LSFC_MEM_HOST));
    CHECK(lsfc_apply(plan, u, t, LSFC_MEM_HOST));
    double num = 0, den = 0;
    for (long i = 0; i < 2 * N; ++i) { num += (t[i] - rhs[i]) * (t[i] - rhs[i]); den += rhs[i] * rhs[i]; }
    printf("gmres: %lld iterations, %lld operator applies, converged=%d, true relative residual %.3e\n",
           (long long)r.iters, (long long)r.mvps, r.converged, sqrt(num / den));
    /* ---- the multi-GPU forms of the same operator --------------------------------------------------------------- */
    double* t2 = malloc(2 * N * sizeof(double));
    int ndev = 0, devs[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    CHECK(lsfc_device_count(&ndev));
    int use = 1; while (use * 2 <= ndev && use * 2 <= 8 && n % (use * 2) == 0) use *= 2;      /* power of two dividing n */
    lsfc_plan* multi = NULL;
    CHECK(lsfc_plan_create_gv3d_multi(&multi, n, n, n, 1.0, k, nu, LSFC_FLAG_DEFAULT, devs, use));
    const char* transport = "";
    CHECK(lsfc_multi_info(multi, NULL, NULL, NULL, &transport));
    CHECK(lsfc_apply(multi, u, t2, LSFC_MEM_HOST));
    double dm = 0, nm = 0;
    for (long i = 0; i < 2 * N; ++i) { dm += (t2[i] - t[i]) * (t2[i] - t[i]); nm += t[i] * t[i]; }
    printf("multi-device plan on %d device(s), exchange transport: %s; |y_multi - y_single| / |y_single| = %.2e\n", use, transport, sqrt(dm / nm));
    CHECK(lsfc_plan_destroy(multi));
    setenv("LSFC_DIST_FORCE_COMM", "1", 1);                  /* one rank, but the self exchange goes through ncclSend / ncclRecv */
    setenv("LSFC_DIST_FORCE_OVERLAP", "1", 1);
    setenv("LSFC_DIST_CHUNKS", "2", 1);
    unsigned char id[LSFC_UNIQUE_ID_BYTES];
    CHECK(lsfc_dist_unique_id(id));
    lsfc_plan* dist = NULL;
    CHECK(lsfc_dist_plan_create_gv3d(&dist, n, n, n, 1.0, k, nu, LSFC_FLAG_DEFAULT, 0, /*rank*/ 0, /*nranks*/ 1, id));
    CHECK(lsfc_apply(dist, u, t2, LSFC_MEM_HOST));
    double dd = 0;
    for (long i = 0; i < 2 * N; ++i) dd += (t2[i] - t[i]) * (t2[i] - t[i]);
    printf("slab plan (rank 0 of 1, RCCL send/recv self exchange, 3-stream pipeline): |y_dist - y_single| / |y_single| = %.2e\n", sqrt(dd / nm));
    CHECK(lsfc_plan_destroy(dist));
    CHECK(lsfc_plan_destroy(plan));
    free(nu); free(uinc); free(rhs); free(u); free(t); free(t2);
    return (r.converged && sqrt(num / den) < 1e-7 && sqrt(dm / nm) < 1e-12 && sqrt(dd / nm) < 1e-12) ? 0 : 2;
}
