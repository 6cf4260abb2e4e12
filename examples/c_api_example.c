/* Plain-C use of the lsfc ABI (include/lsfc.h): build the 3D operator with the device-side symbol generator,
 * apply it, and solve (I + w^2 G nu) u = rhs with GMRES -- the sequence of examples/example3D.jl:54-79.
 *   gcc -std=c99 -Iinclude examples/c_api_example.c -Lfast_solver_lippmann_schwinger_amd -llsfc -lm \
 *       -Wl,-rpath,$PWD/fast_solver_lippmann_schwinger_amd -o c_api_example && ./c_api_example 32            */
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
    CHECK(lsfc_plan_destroy(plan));
    free(nu); free(uinc); free(rhs); free(u); free(t);
    return (r.converged && sqrt(num / den) < 1e-7) ? 0 : 2;
}
