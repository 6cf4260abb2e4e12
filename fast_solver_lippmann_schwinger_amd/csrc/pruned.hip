// Front end of the hand-written axis passes: routes every call to the family of its line length (fft_kernels.hip is
// compiled once per family: power-of-two lines, lines with one factor 3, lines with one factor 5).
#include "pruned.hpp"
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>

namespace lsfc {

#define LSFC_FAMILY_DECLS(F)                                                                                                   \
    void pruned_xfwd_f##F(int, const PrunedTuning&, const VecBatch&, int, int64_t, const double*, cplx*, const cplx*, int64_t, int, int, int, hipStream_t, int64_t); \
    void pruned_xinv_f##F(int, const PrunedTuning&, const cplx*, const VecBatch&, int, int64_t, double, double, const cplx*, int64_t, int, int, int, hipStream_t, int64_t); \
    void pruned_yfwd_f##F(int, const PrunedTuning&, const cplx*, cplx*, const cplx*, int, int, int, int, int, hipStream_t, int, int64_t, int64_t);     \
    void pruned_yinv_f##F(int, const PrunedTuning&, const cplx*, cplx*, const cplx*, int, int, int, int, int, hipStream_t, int, int64_t, int64_t);     \
    void pruned_zfused_f##F(int, const PrunedTuning&, cplx*, const cplx*, const cplx*, const cplx*, int, int, int64_t, int64_t, int64_t, \
                            int64_t, int64_t, int64_t, const int2*, const int*, int, hipStream_t, int, int64_t);                \
    void pruned_perm_f##F(int, int*);                                                                                     \
    void pruned_warmup_f##F();                                                                                         \
    int pruned_twfull_len_f##F(int);                                                                                            \
    void pruned_twfull_f##F(int, const cplx*, cplx*);
void warmup_pointwise(); void warmup_symbol(); void warmup_precond();      // pointwise.hip, symbol.hip, precond.hip
LSFC_FAMILY_DECLS(2)
LSFC_FAMILY_DECLS(3)
LSFC_FAMILY_DECLS(5)

// 2: L = 2^k (32..2048);  3: L = 3 * 2^k (48..1536);  5: L = 5 * 2^k (80..1280);  0: not supported
static int family(int64_t L) {
    if (L < 32 || L > 2048) return 0;
    int64_t v = L; while (v % 2 == 0) v /= 2;
    if (v == 1) return 2;
    if (v == 3) return L >= 48 && L <= 1536 ? 3 : 0;
    if (v == 5) return L >= 80 && L <= 1280 ? 5 : 0;
    return 0;
}
bool pruned_length_supported(int64_t L) { return family(L) != 0; }

void pruned_warmup(int device) {
    static std::mutex mu;
    static std::set<int> done;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count(device)) return;
    // LSFC_EAGER_LOAD=0 (developer switch, diagnostics only): leave the code objects to HIP's lazy loading at first launch
    if (const char* e = getenv("LSFC_EAGER_LOAD")) if (e[0] == '0') return;
    pruned_warmup_f2(); pruned_warmup_f3(); pruned_warmup_f5();
    warmup_pointwise(); warmup_symbol(); warmup_precond();
    LSFC_HIP(hipDeviceSynchronize());
    done.insert(device);
}

int pruned_best_length(int64_t n) {
    // LSFC_POW2_ONLY=1 (developer switch): power-of-two lines only, for A/B timing of the mixed-radix lines
    const char* e = getenv("LSFC_POW2_ONLY");
    const bool pow2_only = e && e[0] == '1';
    for (int64_t L = (2 * n > 32 ? 2 * n : 32); L <= 2048; ++L) if (family(L) && (!pow2_only || family(L) == 2)) return (int)L;
    return 0;
}

// LSFC_DEBUG_SYNC=1 (developer switch): synchronise the device after every pass and name the pass that failed -- turns an
// asynchronous GPU fault or launch error into an error message at the kernel that caused it
static bool debug_sync() { static const bool on = getenv("LSFC_DEBUG_SYNC") && getenv("LSFC_DEBUG_SYNC")[0] == '1'; return on; }
static void debug_check(const char* pass, int L) {
    if (!debug_sync()) return;
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipGetLastError(); fail(LSFC_EHIP, "%s (line length %d) failed: %s", pass, L, hipGetErrorString(e)); }
    fprintf(stderr, "[lsfc debug] %s L=%d ok\n", pass, L);
}
#define LSFC_ROUTE(L, NAME, ...)                                            \
    switch (family(L)) {                                                    \
    case 2: NAME##_f2(__VA_ARGS__); break;                                  \
    case 3: NAME##_f3(__VA_ARGS__); break;                                  \
    case 5: NAME##_f5(__VA_ARGS__); break;                                  \
    default: fail(LSFC_EINVAL, "pruned pipeline: unsupported padded length %d", (int)(L)); } \
    debug_check(#NAME, (int)(L));

void pruned_xfwd(int L, const PrunedTuning& tn, const VecBatch& vb, int nrhs, int64_t obatch, const double* nu, cplx* out, const cplx* tw, int64_t nlines, int W, int Wp, int n, hipStream_t st, int64_t bstride) {
    LSFC_REQUIRE(nrhs >= 1 && nrhs <= LSFC_MAX_BATCH, "batch of %d right-hand sides (1..%d per launch)", nrhs, LSFC_MAX_BATCH);
    LSFC_ROUTE(L, pruned_xfwd, L, tn, vb, nrhs, obatch, nu, out, tw, nlines, W, Wp, n, st, bstride);
}
void pruned_xinv(int L, const PrunedTuning& tn, const cplx* in, const VecBatch& vb, int nrhs, int64_t ibatch, double alpha, double beta, const cplx* tw, int64_t nlines, int W, int Wp, int n, hipStream_t st, int64_t bstride) {
    LSFC_REQUIRE(nrhs >= 1 && nrhs <= LSFC_MAX_BATCH, "batch of %d right-hand sides (1..%d per launch)", nrhs, LSFC_MAX_BATCH);
    LSFC_ROUTE(L, pruned_xinv, L, tn, in, vb, nrhs, ibatch, alpha, beta, tw, nlines, W, Wp, n, st, bstride);
}
void pruned_yfwd(int L, const PrunedTuning& tn, const cplx* a1, cplx* a2, const cplx* tw, int Lx, int m, int l, int p1, int p2, hipStream_t st, int nrhs, int64_t b1, int64_t b2) {
    LSFC_ROUTE(L, pruned_yfwd, L, tn, a1, a2, tw, Lx, m, l, p1, p2, st, nrhs, b1, b2);
}
void pruned_yinv(int L, const PrunedTuning& tn, const cplx* a2, cplx* a1, const cplx* tw, int Lx, int m, int l, int p1, int p2, hipStream_t st, int nrhs, int64_t b1, int64_t b2) {
    LSFC_ROUTE(L, pruned_yinv, L, tn, a2, a1, tw, Lx, m, l, p1, p2, st, nrhs, b1, b2);
}
void pruned_zfused(int L, const PrunedTuning& tn, cplx* data, const cplx* sym, const cplx* tw, const cplx* twl, int Lx, int nouter,
                   int64_t dTile, int64_t dOuter, int64_t dLine, int64_t sTile, int64_t sOuter, int64_t sLine, const int2* ytab,
                   const int* zm, int nin, hipStream_t st, int nrhs, int64_t dBatch) {
    LSFC_ROUTE(L, pruned_zfused, L, tn, data, sym, tw, twl, Lx, nouter, dTile, dOuter, dLine, sTile, sOuter, sLine, ytab, zm, nin, st, nrhs, dBatch);
}
void pruned_perm(int L, int* freq_of_storage) {
    switch (family(L)) {
    case 2: pruned_perm_f2(L, freq_of_storage); break;
    case 3: pruned_perm_f3(L, freq_of_storage); break;
    case 5: pruned_perm_f5(L, freq_of_storage); break;
    default: fail(LSFC_EINVAL, "pruned pipeline: unsupported padded length %d", L); }
}
int pruned_twfull_len(int L) {
    switch (family(L)) {
    case 2: return pruned_twfull_len_f2(L);
    case 3: return pruned_twfull_len_f3(L);
    case 5: return pruned_twfull_len_f5(L);
    default: fail(LSFC_EINVAL, "pruned pipeline: unsupported padded length %d", L);
    }
}
void pruned_twfull(int L, const cplx* tw, cplx* out) {
    switch (family(L)) {
    case 2: pruned_twfull_f2(L, tw, out); break;
    case 3: pruned_twfull_f3(L, tw, out); break;
    case 5: pruned_twfull_f5(L, tw, out); break;
    default: fail(LSFC_EINVAL, "pruned pipeline: unsupported padded length %d", L); }
}

static bool env_flag(const char* name, bool dflt) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    return v[0] == '1' || v[0] == 'y' || v[0] == 't';
}

PrunedTuning pruned_default_tuning() {
    PrunedTuning t;
    t.split_x = env_flag("LSFC_SPLIT_X", true);
    t.split_s = env_flag("LSFC_SPLIT_S", true);
    if (const char* v = getenv("LSFC_SPLIT_Z")) t.split_z = atoi(v);
    if (const char* v = getenv("LSFC_PAD1")) t.pad1 = atoi(v);
    if (const char* v = getenv("LSFC_PAD2")) t.pad2 = atoi(v);
    if (const char* v = getenv("LSFC_Z_HALF")) t.z_half = atoi(v);
    if (const char* v = getenv("LSFC_TW_LDS")) t.tw_lds = atoi(v);
    if (const char* v = getenv("LSFC_SYM_PREFETCH")) t.sym_prefetch = atoi(v);
    if (const char* v = getenv("LSFC_YTILE_G")) t.ytile_g = atoi(v);
    if (const char* v = getenv("LSFC_YTILE_Z")) t.ytile_z = atoi(v);
    if (const char* v = getenv("LSFC_BATCH_FUSE")) t.batch_fuse = atoi(v);
    if (const char* v = getenv("LSFC_Z_PERSIST")) t.z_persist = atoi(v);
    if (const char* v = getenv("LSFC_XLANE")) t.xlane = atoi(v);
    return t;
}

} // namespace lsfc
