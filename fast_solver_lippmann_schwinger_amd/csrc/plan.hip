// lsfc C ABI: plan construction, the operator apply, timing helpers (gfx950).
#include "plan.hpp"
#include "pointwise.hpp"
#include <cmath>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <thread>
#include <functional>
#include <mutex>

namespace lsfc {

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_last_error[1024] = "";
void set_last_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_last_error, sizeof g_last_error, fmt, ap); va_end(ap);
}

// ---------------------------------------------------------------------------
// rocFFT wrapper
// ---------------------------------------------------------------------------
#define LSFC_ROCFFT(expr) do { rocfft_status s_ = (expr); if (s_ != rocfft_status_success) \
    ::lsfc::fail(LSFC_EHIP, "%s failed with rocfft_status %d (%s:%d)", #expr, (int)s_, __FILE__, __LINE__); } while (0)

void rocfft_global_setup() {
    static std::once_flag once;
    std::call_once(once, [] { rocfft_setup(); });
}

RocFft::~RocFft() { release(); }
void RocFft::release() {
    if (info) { rocfft_execution_info_destroy(info); info = nullptr; }
    if (plan) { rocfft_plan_destroy(plan); plan = nullptr; }
    work.release();
}
void RocFft::finish_create() {
    size_t wsz = 0;
    LSFC_ROCFFT(rocfft_plan_get_work_buffer_size(plan, &wsz));
    LSFC_ROCFFT(rocfft_execution_info_create(&info));
    if (wsz) { work.alloc(wsz); LSFC_ROCFFT(rocfft_execution_info_set_work_buffer(info, work.p, wsz)); }
}
void RocFft::create(int ndim, const size_t* lengths, bool forward, size_t batch, bool lazy_work) {
    rocfft_global_setup();
    // drop unit dimensions
    size_t len[3]; int nd = 0;
    for (int d = 0; d < ndim; ++d) if (lengths[d] > 1) len[nd++] = lengths[d];
    if (nd == 0) { len[0] = 1; nd = 1; }
    LSFC_ROCFFT(rocfft_plan_create(&plan, rocfft_placement_inplace,
                                   forward ? rocfft_transform_type_complex_forward : rocfft_transform_type_complex_inverse,
                                   rocfft_precision_double, nd, len, batch, nullptr));
    if (!lazy_work) finish_create();
}
void RocFft::create_strided_1d(size_t length, size_t stride, size_t dist, size_t batch, bool forward, bool lazy_work) {
    rocfft_global_setup();
    rocfft_plan_description desc = nullptr;
    LSFC_ROCFFT(rocfft_plan_description_create(&desc));
    const size_t strides[1] = { stride };
    rocfft_status s = rocfft_plan_description_set_data_layout(desc, rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved,
                                                              nullptr, nullptr, 1, strides, dist, 1, strides, dist);
    if (s == rocfft_status_success)
        s = rocfft_plan_create(&plan, rocfft_placement_inplace,
                               forward ? rocfft_transform_type_complex_forward : rocfft_transform_type_complex_inverse,
                               rocfft_precision_double, 1, &length, batch, desc);
    rocfft_plan_description_destroy(desc);
    LSFC_ROCFFT(s);
    if (!lazy_work) finish_create();
}
void RocFft::exec(void* buf, hipStream_t stream) {
    if (!info) finish_create();
    LSFC_ROCFFT(rocfft_execution_info_set_stream(info, stream));
    void* in[1] = { buf };
    LSFC_ROCFFT(rocfft_execute(plan, in, nullptr, info));
}

// ---------------------------------------------------------------------------
// plan construction helpers
// ---------------------------------------------------------------------------
static void select_device(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) fail(LSFC_ENODEV, "no HIP device available (%s): the lsfc operator has no CPU fallback",
                                            e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    LSFC_REQUIRE(device >= 0 && device < count, "device %d out of range (have %d)", device, count);
    LSFC_HIP(hipSetDevice(device));
}

void plan_common_init(lsfc_plan* p, int ndim, int64_t n, int64_t m, int64_t l, const double* nu_host, double omega,
                      int quad_rule, unsigned flags, int device) {
    LSFC_REQUIRE(n >= 1 && m >= 1 && l >= 1 && n <= 8192 && m <= 8192 && l <= 8192, "grid size out of range");
    LSFC_REQUIRE(quad_rule == LSFC_QUAD_TRAPEZOIDAL || quad_rule == LSFC_QUAD_GREENGARD_VICO,
                 "unknown quadRule %d (the reference leaves B undefined, src/FastConvolution.jl:106)", quad_rule);
    LSFC_REQUIRE(nu_host != nullptr, "nu is NULL");
    select_device(device);
    pruned_warmup(device);
    p->device = device; p->ndim = ndim;
    p->dims[0] = (int)n; p->dims[1] = (int)m; p->dims[2] = (int)l;
    p->N = n * m * l; p->omega = omega; p->quad_rule = quad_rule; p->flags = flags;
    p->nu.alloc((size_t)p->N);
    LSFC_HIP(hipMemcpy(p->nu.p, nu_host, p->N * sizeof(double), hipMemcpyHostToDevice));
    p->tuning = pruned_default_tuning();
}

void plan_make_twiddles(lsfc_plan* p, int axis, int L) {
    std::vector<cplx> tw((size_t)L);
    for (int j = 0; j < L; ++j) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)j / (long double)L;
        tw[j] = make_double2((double)cosl(a), (double)sinl(a));
    }
    p->tw[axis].alloc((size_t)L);
    LSFC_HIP(hipMemcpy(p->tw[axis].p, tw.data(), (size_t)L * sizeof(cplx), hipMemcpyHostToDevice));
    std::vector<cplx> full((size_t)pruned_twfull_len(L) + 1);
    pruned_twfull(L, tw.data(), full.data());
    p->twl[axis].alloc(full.size());
    LSFC_HIP(hipMemcpy(p->twl[axis].p, full.data(), full.size() * sizeof(cplx), hipMemcpyHostToDevice));
}

// The hand-written pipeline works on L = the smallest of 2^k, 3*2^k, 5*2^k that is >= 2n, per axis (the grid itself
// may have any size: lines are zero-extended in registers).  It is chosen unless that embedding would more than
// quadruple the number of grid points (tiny or very elongated grids), where the monolithic rocFFT transform on the
// exact 2n grid is the better deal.
static bool pruned_eligible(const lsfc_plan* p) {
    if (p->flags & (LSFC_FLAG_FORCE_ROCFFT | LSFC_FLAG_LITERAL_PAD)) return false;
    double ratio = 1.0;
    for (int d = 0; d < p->ndim; ++d) {
        const int L = pruned_best_length(p->dims[d]);
        if (L == 0) return false;
        // the slab-wise Greengard-Vico generator samples the literal 4n lattice: a working line longer than 4n (an axis of
        // fewer than 8 points next to long ones) has no such samples -> exact 2n grid through rocFFT
        if (p->ndim == 3 && L > 4 * p->dims[d]) return false;
        ratio *= (double)L / (2.0 * (double)p->dims[d]);
    }
    return ratio <= 4.0;
}

// working padded grid of the reduced pipelines: pruned_best_length(n) (pruned) or 2n (rocFFT)
void plan_choose_reduced_grid(lsfc_plan* p) {
    const bool pr = pruned_eligible(p);
    for (int d = 0; d < 3; ++d) {
        p->crop[d] = 0;
        if (d >= p->ndim) p->pads[d] = 1;
        else p->pads[d] = pr ? pruned_best_length(p->dims[d]) : 2 * p->dims[d];
    }
    p->pipeline = pr ? lsfc_plan::PRUNED : lsfc_plan::ROCFFT_REDUCED;
}

static void setup_rocfft_pipeline(lsfc_plan* p) {
    const size_t len[3] = { (size_t)p->pads[0], (size_t)p->pads[1], (size_t)p->pads[2] };
    p->fwd.reset(new RocFft()); p->fwd->create(3, len, true);
    p->inv.reset(new RocFft()); p->inv->create(3, len, false);
    p->W.alloc(len[0] * len[1] * len[2]);
}

void plan_finish_literal(lsfc_plan* p, const cplx* Gd, bool centred) {
    // working grid == the caller's padded grid; symbol pre-shifted and pre-scaled once
    const int64_t total = (int64_t)p->pads[0] * p->pads[1] * p->pads[2];
    p->sym.alloc((size_t)total);
    int shift[3];
    for (int d = 0; d < 3; ++d) shift[d] = centred ? p->pads[d] / 2 : 0;
    pw_roll_scale(Gd, p->sym.p, p->pads, shift, 1.0 / (double)total, p->stream);
    p->pipeline = lsfc_plan::ROCFFT_LITERAL;
    setup_rocfft_pipeline(p);
    LSFC_HIP(hipStreamSynchronize(p->stream));
}

void plan_setup_symbol_rows(lsfc_plan* p, const cplx* G2, const std::vector<int>& perm_y, const std::vector<int>& perm_z, DevBuf<int>& pyrow) {
    const int Ly = p->pads[1], Lz = p->pads[2];
    // The Green's symbols of the reference are even in every axis.  When the reduced symbol handed to us is even in
    // y (checked numerically: the literal-symbol constructors accept arbitrary data) only the rows with ky <= Ly/2
    // are stored, and each row is scheduled right next to its mirror: the second read of the shared symbol row is
    // served on-die by the Infinity Cache, which removes ~1/8 of the apply's HBM traffic.
    bool even = false;
    const char* env = getenv("LSFC_SYM_EVEN_Y");
    // (G2 == NULL: the caller built the symbol through its symmetry, symbol_gv3d_quarter, and vouches for it)
    if (!(env && env[0] == '0') && p->ndim == 3 && Ly >= 4) even = !G2 || pw_mirror_deviation(G2, p->pads, 1, p->stream) < 1e-13;
    std::vector<int> inv((size_t)Ly), rowky;
    for (int s = 0; s < Ly; ++s) inv[perm_y[s]] = s;
    std::vector<int2> tab;
    if (even) {
        std::vector<int> rowof((size_t)Ly, -1);
        for (int s = 0; s < Ly; ++s) if (perm_y[s] <= Ly / 2) { rowof[s] = (int)rowky.size(); rowky.push_back(perm_y[s]); }
        // block order: the self-mirrored rows (ky = 0, Ly/2) first, then every row followed by its mirror -- with an even
        // number of the former every pair sits at (2j, 2j + 1), which the ticketed fused pass relies on to send both rows
        // of a pair through the same L2
        for (int s = 0; s < Ly; ++s)
            if (rowof[s] >= 0 && inv[(Ly - perm_y[s]) % Ly] == s) tab.push_back(make_int2(s, rowof[s]));
        for (int s = 0; s < Ly; ++s) {
            if (rowof[s] < 0) continue;
            const int mirror = inv[(Ly - perm_y[s]) % Ly];
            if (mirror == s) continue;
            tab.push_back(make_int2(s, rowof[s]));
            tab.push_back(make_int2(mirror, rowof[s]));
        }
    } else {
        for (int s = 0; s < Ly; ++s) { rowky.push_back(perm_y[s]); tab.push_back(make_int2(s, s)); }
    }
    LSFC_REQUIRE((int)tab.size() == Ly, "internal: row table has %d entries for %d rows", (int)tab.size(), Ly);
    p->sym_rows = (int)rowky.size();
    p->ytab.alloc(tab.size());
    LSFC_HIP(hipMemcpy(p->ytab.p, tab.data(), tab.size() * sizeof(int2), hipMemcpyHostToDevice));
    pyrow.alloc(rowky.size());
    LSFC_HIP(hipMemcpy(pyrow.p, rowky.data(), rowky.size() * sizeof(int), hipMemcpyHostToDevice));

    // z-even symbol: store only kz <= Lz/2 per line; the kernel fetches the mirror values of its upper slots from
    // the threads that loaded them (through LDS).  Needs: storage slots s < Lz/2 <-> kz < Lz/2 (true for every
    // factorisation in fft_configs.hpp, re-checked here).
    bool zeven = false;
    const char* envz = getenv("LSFC_SYM_EVEN_Z");
    if (!(envz && envz[0] == '0') && p->ndim == 3 && Lz >= 32) {
        bool ok = true;
        for (int s2 = 0; s2 < Lz; ++s2) if ((perm_z[s2] < Lz / 2) != (s2 < Lz / 2)) ok = false;
        if (ok) zeven = !G2 || pw_mirror_deviation(G2, p->pads, 2, p->stream) < 1e-13;
    }
    if (zeven) {
        std::vector<int> invz((size_t)Lz), zm((size_t)Lz / 2);
        for (int s2 = 0; s2 < Lz; ++s2) invz[perm_z[s2]] = s2;
        for (int s2 = Lz / 2; s2 < Lz; ++s2) {
            const int km = (Lz - perm_z[s2]) % Lz;                  // in [1, Lz/2]
            zm[s2 - Lz / 2] = (km == Lz / 2) ? Lz / 2 : invz[km];
        }
        p->sym_hz = Lz / 2 + 8;
        p->zmirror.alloc(zm.size());
        LSFC_HIP(hipMemcpy(p->zmirror.p, zm.data(), zm.size() * sizeof(int), hipMemcpyHostToDevice));
    } else {
        p->sym_hz = Lz;
        p->zmirror.release();
    }
}

// whether the pruned pipeline will store the y-even, z-even quarter of a 3D symbol that is even in every axis
bool plan_quarter_symbol_ok(const lsfc_plan* p) {
    if (p->ndim != 3) return false;                    // (the caller knows that the plan runs the pruned pipeline)
    const char* ey = getenv("LSFC_SYM_EVEN_Y"); const char* ez = getenv("LSFC_SYM_EVEN_Z"); const char* full = getenv("LSFC_SYMBOL_FULL");
    if ((ey && ey[0] == '0') || (ez && ez[0] == '0') || (full && full[0] == '1')) return false;
    const int Ly = p->pads[1], Lz = p->pads[2];
    if (Ly < 4 || Lz < 32 || p->pads[0] % 2 || Ly % 2 || Lz % 2) return false;
    std::vector<int> pz((size_t)Lz);
    pruned_perm(Lz, pz.data());
    for (int s2 = 0; s2 < Lz; ++s2) if ((pz[(size_t)s2] < Lz / 2) != (s2 < Lz / 2)) return false;
    return true;
}

void plan_finish_from_reduced(lsfc_plan* p, DevBuf<cplx>& G2) { plan_finish_symbol(p, G2, false); }
void plan_finish_from_quarter(lsfc_plan* p, DevBuf<cplx>& Gq) { plan_finish_symbol(p, Gq, true); }

void plan_finish_symbol(lsfc_plan* p, DevBuf<cplx>& G2, bool quarter) {
    // G2: natural FFT-order symbol on the grid chosen by plan_choose_reduced_grid; quarter: only its rows ky <= Ly/2 and
    // planes kz <= Lz/2 (symbol_gv3d_quarter)
    const int64_t total = (int64_t)p->pads[0] * p->pads[1] * p->pads[2];
    const int64_t have = quarter ? (int64_t)p->pads[0] * (p->pads[1] / 2 + 1) * (p->pads[2] / 2 + 1) : total;
    LSFC_REQUIRE((int64_t)G2.n == have, "internal: reduced symbol has %lld entries, expected %lld", (long long)G2.n, (long long)have);
    LSFC_REQUIRE(!quarter || (p->pipeline == lsfc_plan::PRUNED && p->ndim == 3), "internal: quarter symbol outside the 3D pruned pipeline");
    const double scale = 1.0 / (double)total;
    if (p->pipeline == lsfc_plan::PRUNED) {
        std::vector<int> perm[3];
        DevBuf<int> dperm[3];
        for (int d = 0; d < p->ndim; ++d) {
            perm[d].resize((size_t)p->pads[d]);
            pruned_perm(p->pads[d], perm[d].data());
            dperm[d].alloc(perm[d].size());
            LSFC_HIP(hipMemcpy(dperm[d].p, perm[d].data(), perm[d].size() * sizeof(int), hipMemcpyHostToDevice));
            plan_make_twiddles(p, d, p->pads[d]);
        }
        if (p->ndim == 3) {
            DevBuf<int> pyrow;
            plan_setup_symbol_rows(p, quarter ? nullptr : G2.p, perm[1], perm[2], pyrow);
            LSFC_REQUIRE(!quarter || (p->sym_rows == p->pads[1] / 2 + 1 && p->sym_hz == p->pads[2] / 2 + 8), "internal: quarter symbol but full storage");
            p->sym.alloc((size_t)p->pads[0] * p->sym_rows * p->sym_hz);
            pw_permute_symbol(G2.p, p->sym.p, dperm[0].p, pyrow.p, dperm[2].p, p->pads, p->sym_rows, p->sym_hz, 0, p->pads[0] / 8, scale, p->stream,
                              quarter ? p->pads[1] / 2 + 1 : 0);
        } else {
            // 2D: the fused pass runs along y.  An even symbol (every Green's symbol of the reference) is stored for the
            // frequencies ky <= Ly/2 only -- rows s < Ly/2 in storage order plus the ky = Ly/2 row -- and the fused pass
            // fetches the mirror values through LDS exactly as the 3D pass does along z (half the symbol bytes of the pass).
            const int Ly = p->pads[1];
            bool even = false;
            const char* env = getenv("LSFC_SYM_EVEN_Z");
            if (!(env && env[0] == '0') && Ly >= 32) {
                bool ok = true;
                for (int s2 = 0; s2 < Ly; ++s2) if ((perm[1][(size_t)s2] < Ly / 2) != (s2 < Ly / 2)) ok = false;
                if (ok) even = pw_mirror_deviation(G2.p, p->pads, 1, p->stream) < 1e-13;
            }
            if (even) {
                std::vector<int> inv((size_t)Ly), zm((size_t)Ly / 2), rowk((size_t)Ly / 2 + 1);
                for (int s2 = 0; s2 < Ly; ++s2) inv[(size_t)perm[1][(size_t)s2]] = s2;
                for (int s2 = Ly / 2; s2 < Ly; ++s2) {
                    const int km = (Ly - perm[1][(size_t)s2]) % Ly;              // in [1, Ly/2]
                    zm[(size_t)(s2 - Ly / 2)] = (km == Ly / 2) ? Ly / 2 : inv[(size_t)km];
                }
                for (int s2 = 0; s2 < Ly / 2; ++s2) rowk[(size_t)s2] = perm[1][(size_t)s2];
                rowk[(size_t)Ly / 2] = Ly / 2;
                p->zmirror.alloc(zm.size());
                LSFC_HIP(hipMemcpy(p->zmirror.p, zm.data(), zm.size() * sizeof(int), hipMemcpyHostToDevice));
                // Tiled form (default from Ly = 2048 on; LSFC_2D_TILED=0|1): the x passes write / read the x'-expanded array as
                // tiles [Lx/8][m][8] -- the chunked output they already have for the slab transposes, chunk width 8 -- so the
                // fused pass along y finds its eight interleaved lines in ONE contiguous run of 128-B lines, exactly the 3D fused
                // pass's situation (same kernels, same z-even symbol layout), instead of 64-B pieces a whole row apart.
                const char* t2 = getenv("LSFC_2D_TILED");
                const bool tiled = t2 ? t2[0] == '1' : Ly >= 2048;      // (at 1024 points the 8-line tiles are too few for the chip: 128 workgroups)
                // (the tiled fused pass is the 3D half-tile kernel: whole groups of 8 tiles -- Lx / 8 tiles here; a short x axis,
                // e.g. n = 80 next to m = 1024, keeps the natural rows)
                if (tiled && p->pads[0] % 8 == 0 && (p->pads[0] / 8) % 8 == 0 && p->pads[0] <= 2048) {
                    p->sym_hz = Ly / 2 + 8; p->sym_rows = 1;
                    DevBuf<int> zero; zero.alloc(1);
                    LSFC_HIP(hipMemset(zero.p, 0, sizeof(int)));
                    const int L3[3] = { p->pads[0], 1, Ly };        // the line axis plays z; one symbol row per tile
                    p->sym.alloc((size_t)p->pads[0] * p->sym_hz);
                    pw_permute_symbol(G2.p, p->sym.p, dperm[0].p, zero.p, dperm[1].p, L3, 1, p->sym_hz, 0, p->pads[0] / 8, scale, p->stream);
                    LSFC_HIP(hipStreamSynchronize(p->stream));
                    p->tile2d = (int64_t)8 * p->dims[1] + (Ly >= 1024 ? 72 : 0);
                } else {
                DevBuf<int> drow; drow.alloc(rowk.size());
                LSFC_HIP(hipMemcpy(drow.p, rowk.data(), rowk.size() * sizeof(int), hipMemcpyHostToDevice));
                p->sym.alloc((size_t)p->pads[0] * rowk.size());
                pw_permute_symbol(G2.p, p->sym.p, dperm[0].p, drow.p, dperm[2].p, p->pads, (int)rowk.size(), 1, 0, p->pads[0] / 8, scale, p->stream);
                LSFC_HIP(hipStreamSynchronize(p->stream));
                }
            } else {
                p->zmirror.release();
                p->sym.alloc((size_t)total);
                pw_permute_symbol(G2.p, p->sym.p, dperm[0].p, dperm[1].p, dperm[2].p, p->pads, p->pads[1], 1, 0, p->pads[0] / 8, scale, p->stream);
            }
        }
        LSFC_HIP(hipStreamSynchronize(p->stream));
        G2.release();
        // row padding of the work arrays (auto: at 2m >= 1024 the 16-KB / 64-KB power-of-two row strides of the y passes
        // are broken up by +40 / +72 elements: yfwd 2.86 -> 2.59 ms, yinv 2.92 -> 2.57 ms at 512^3; neutral or worse below)
        const bool big = p->ndim == 3 && p->pads[1] >= 1024;
        const int pad1 = p->tuning.pad1 >= 0 ? p->tuning.pad1 : (big ? 40 : 0);
        const int pad2 = p->tuning.pad2 >= 0 ? p->tuning.pad2 : (big ? 72 : 0);
        p->pitch1 = p->pads[0] + ((p->ndim == 3) ? pad1 / 8 * 8 : 0);
        p->pitch2 = 8 * p->dims[2] + pad2 / 8 * 8;
        p->a1_elems = (int64_t)p->pitch1 * p->dims[1] * p->dims[2];
        if (p->tile2d) p->a1_elems = std::max<int64_t>(p->a1_elems, p->tile2d * (p->pads[0] / 8));
        p->a2_elems = (p->ndim == 3) ? (int64_t)p->pitch2 * p->pads[1] * (p->pads[0] / 8) : 0;
        p->A1.alloc((size_t)p->a1_elems);
        if (p->ndim == 3) p->A2.alloc((size_t)p->a2_elems);
        p->batch_cap = 1;
    } else {
        pw_scale(G2.p, scale, total, p->stream);
        LSFC_HIP(hipStreamSynchronize(p->stream));
        p->sym = std::move(G2);
        setup_rocfft_pipeline(p);
    }
}

void plan_finish_reduce(lsfc_plan* p, DevBuf<cplx>& Gd, const int lit[3], bool centred, const int kernel_origin[3]) {
    // T = ifft(ifftshift(G)) on the literal grid is the spatial kernel; the cropped convolution only needs its
    // offsets -(n-1)..n-1 per axis, so it is re-sampled on the working grid q and transformed back: G2 = fft(T2).
    //   Greengard-Vico: offset d sits at index d mod lit (kernel_origin = 0), crop window [0, n).
    //   trapezoidal:    offset d sits at index d + (n-1) (kernel_origin = n-1; crop window [n-1, 2n-2] of the reference).
    plan_choose_reduced_grid(p);
    const int* q = p->pads;
    for (int d = 0; d < p->ndim; ++d)
        LSFC_REQUIRE(lit[d] >= 2 * p->dims[d] - 1, "padded size %d along axis %d is smaller than 2n-1=%d", lit[d], d, 2 * p->dims[d] - 1);
    const int64_t ltotal = (int64_t)lit[0] * lit[1] * lit[2];
    const int64_t qtotal = (int64_t)q[0] * q[1] * q[2];
    {
        DevBuf<cplx> tmp; tmp.alloc((size_t)ltotal);
        int shift[3]; for (int d = 0; d < 3; ++d) shift[d] = centred ? lit[d] / 2 : 0;
        pw_roll_scale(Gd.p, tmp.p, lit, shift, 1.0, p->stream);
        LSFC_HIP(hipStreamSynchronize(p->stream));
        Gd.release();
        const size_t len[3] = { (size_t)lit[0], (size_t)lit[1], (size_t)lit[2] };
        RocFft inv; inv.create(3, len, false);
        inv.exec(tmp.p, p->stream);
        Gd.alloc((size_t)qtotal);
        int nmax[3]; for (int d = 0; d < 3; ++d) nmax[d] = (d < p->ndim) ? p->dims[d] - 1 : 0;
        pw_resample_kernel(tmp.p, Gd.p, lit, q, kernel_origin, nmax, 1.0 / (double)ltotal, p->stream);
        LSFC_HIP(hipStreamSynchronize(p->stream));
    }
    {
        const size_t len[3] = { (size_t)q[0], (size_t)q[1], (size_t)q[2] };
        RocFft fwd; fwd.create(3, len, true);
        fwd.exec(Gd.p, p->stream);
        LSFC_HIP(hipStreamSynchronize(p->stream));
    }
    plan_finish_from_reduced(p, Gd);
}

// ---------------------------------------------------------------------------
// the apply
// ---------------------------------------------------------------------------
// the three passes of the 2D pipeline (natural rows, or tiles: lsfc_plan::tile2d)
static void pass2d_xfwd(lsfc_plan* p, const VecBatch& vb, int nrhs, const double* nu, hipStream_t st) {
    const int Lx = p->pads[0], m = p->dims[1];
    if (p->tile2d) pruned_xfwd(Lx, p->tuning, vb, nrhs, p->a1_elems, nu, p->A1.p, p->tw[0].p, m, 8, 8, p->dims[0], st, p->tile2d);
    else pruned_xfwd(Lx, p->tuning, vb, nrhs, p->a1_elems, nu, p->A1.p, p->tw[0].p, m, Lx, p->pitch1, p->dims[0], st);
}
static void pass2d_yfused(lsfc_plan* p, int nrhs, hipStream_t st) {
    const int Lx = p->pads[0], Ly = p->pads[1], m = p->dims[1];
    if (p->tile2d) pruned_zfused(Ly, p->tuning, p->A1.p, p->sym.p, p->tw[1].p, p->twl[1].p, Lx, 1, p->tile2d, 0, 8, (int64_t)8 * p->sym_hz, 0, 8,
                                 nullptr, p->zmirror.p, m, st, nrhs, p->a1_elems);
    else pruned_zfused(Ly, p->tuning, p->A1.p, p->sym.p, p->tw[1].p, p->twl[1].p, Lx, 1, 8, 0, p->pitch1, 8, 0, Lx, nullptr, p->zmirror.p, m, st, nrhs, p->a1_elems);
}
static void pass2d_xinv(lsfc_plan* p, const VecBatch& vb, int nrhs, double alpha, double beta, hipStream_t st) {
    const int Lx = p->pads[0], m = p->dims[1];
    if (p->tile2d) pruned_xinv(Lx, p->tuning, p->A1.p, vb, nrhs, p->a1_elems, alpha, beta, p->tw[0].p, m, 8, 8, p->dims[0], st, p->tile2d);
    else pruned_xinv(Lx, p->tuning, p->A1.p, vb, nrhs, p->a1_elems, alpha, beta, p->tw[0].p, m, Lx, p->pitch1, p->dims[0], st);
}

void plan_convolve_dev(lsfc_plan* p, const cplx* x, cplx* y, bool use_nu, double alpha, double beta) {
    const double* nu = use_nu ? p->nu.p : nullptr;
    hipStream_t st = p->stream;
    if (p->dist) { dist_convolve_dev(p, x, y, use_nu, alpha, beta); return; }
    if (p->pipeline == lsfc_plan::PRUNED) {
        const int Lx = p->pads[0], Ly = p->pads[1], Lz = p->pads[2];
        const int m = p->dims[1], l = p->dims[2];
        const int64_t nlines = (int64_t)m * l;
        if (p->ndim == 2) {
            VecBatch vb{}; vb.x[0] = x; vb.y[0] = y;
            pass2d_xfwd(p, vb, 1, nu, st);
            pass2d_yfused(p, 1, st);
            pass2d_xinv(p, vb, 1, alpha, beta, st);
            return;
        }
        // (round 3, measured slower and removed again -- profiles/r03_experiment_a1_window.log: the x and y passes chunk by chunk over
        // groups of z planes through one chunk-sized window of A1, hoping the Infinity Cache would absorb the 2 x 4.3 GB of A1 traffic)
        pruned_xfwd(Lx, p->tuning, x, nu, p->A1.p, p->tw[0].p, nlines, Lx, p->pitch1, p->dims[0], st);
        if (p->ndim == 3) {
            const int p1 = p->pitch1, p2 = p->pitch2;
            pruned_yfwd(Ly, p->tuning, p->A1.p, p->A2.p, p->tw[1].p, Lx, m, l, p1, p2, st);
            pruned_zfused(Lz, p->tuning, p->A2.p, p->sym.p, p->tw[2].p, p->twl[2].p, Lx, Ly,
                          (int64_t)p2 * Ly, (int64_t)p2, 8, (int64_t)8 * p->sym_hz * p->sym_rows, (int64_t)8 * p->sym_hz, 8, p->ytab.p,
                          p->zmirror.p, l, st);
            pruned_yinv(Ly, p->tuning, p->A2.p, p->A1.p, p->tw[1].p, Lx, m, l, p1, p2, st);
        } else {
            pruned_zfused(Ly, p->tuning, p->A1.p, p->sym.p, p->tw[1].p, p->twl[1].p, Lx, 1, 8, 0, p->pitch1, 8, 0, Lx, nullptr, p->zmirror.p, m, st);
        }
        pruned_xinv(Lx, p->tuning, p->A1.p, x, y, alpha, beta, p->tw[0].p, nlines, Lx, p->pitch1, p->dims[0], st);
    } else {
        const int64_t total = (int64_t)p->pads[0] * p->pads[1] * p->pads[2];
        pw_embed(x, nu, p->W.p, p->dims, p->pads, st);
        p->fwd->exec(p->W.p, st);
        pw_mul_inplace(p->W.p, p->sym.p, total, st);
        p->inv->exec(p->W.p, st);
        pw_crop_axpy(p->W.p, x, y, alpha, beta, p->dims, p->pads, p->crop, st);
    }
}

void plan_convolve_batch_dev(lsfc_plan* p, int nrhs, const VecBatch& vb, bool use_nu, double alpha, double beta) {
    LSFC_REQUIRE(nrhs >= 1 && nrhs <= LSFC_MAX_BATCH, "batch of %d right-hand sides (1..%d per pass)", nrhs, LSFC_MAX_BATCH);
    const bool fuse = p->tuning.batch_fuse > 0 ||
                      (p->tuning.batch_fuse < 0 && (int64_t)p->pads[0] * p->pads[1] * p->pads[2] <= ((int64_t)1 << 24));
    if (nrhs == 1 || p->dist || p->multi || p->pipeline != lsfc_plan::PRUNED || !fuse) {
        for (int j = 0; j < nrhs; ++j) plan_convolve_dev(p, vb.x[j], vb.y[j], use_nu, alpha, beta);
        return;
    }
    hipStream_t st = p->stream;
    if (nrhs > p->batch_cap) {
        // the work arrays hold the batch back to back: grow them (the plan is idle once its stream has drained)
        LSFC_HIP(hipStreamSynchronize(st));
        p->A1.alloc((size_t)(p->a1_elems * nrhs));
        if (p->ndim == 3) p->A2.alloc((size_t)(p->a2_elems * nrhs));
        p->batch_cap = nrhs;
    }
    const double* nu = use_nu ? p->nu.p : nullptr;
    const int Lx = p->pads[0], Ly = p->pads[1], Lz = p->pads[2];
    const int m = p->dims[1], l = p->dims[2];
    const int64_t nlines = (int64_t)m * l;
    if (p->ndim == 2) {
        pass2d_xfwd(p, vb, nrhs, nu, st);
        pass2d_yfused(p, nrhs, st);
        pass2d_xinv(p, vb, nrhs, alpha, beta, st);
        return;
    }
    pruned_xfwd(Lx, p->tuning, vb, nrhs, p->a1_elems, nu, p->A1.p, p->tw[0].p, nlines, Lx, p->pitch1, p->dims[0], st);
    if (p->ndim == 3) {
        const int p1 = p->pitch1, p2 = p->pitch2;
        pruned_yfwd(Ly, p->tuning, p->A1.p, p->A2.p, p->tw[1].p, Lx, m, l, p1, p2, st, nrhs, p->a1_elems, p->a2_elems);
        pruned_zfused(Lz, p->tuning, p->A2.p, p->sym.p, p->tw[2].p, p->twl[2].p, Lx, Ly,
                      (int64_t)p2 * Ly, (int64_t)p2, 8, (int64_t)8 * p->sym_hz * p->sym_rows, (int64_t)8 * p->sym_hz, 8, p->ytab.p,
                      p->zmirror.p, l, st, nrhs, p->a2_elems);
        pruned_yinv(Ly, p->tuning, p->A2.p, p->A1.p, p->tw[1].p, Lx, m, l, p1, p2, st, nrhs, p->a1_elems, p->a2_elems);
    } else {
        pruned_zfused(Ly, p->tuning, p->A1.p, p->sym.p, p->tw[1].p, p->twl[1].p, Lx, 1, 8, 0, p->pitch1, 8, 0, Lx, nullptr, p->zmirror.p, m, st, nrhs, p->a1_elems);
    }
    pruned_xinv(Lx, p->tuning, p->A1.p, vb, nrhs, p->a1_elems, alpha, beta, p->tw[0].p, nlines, Lx, p->pitch1, p->dims[0], st);
}

static void ensure_staging(lsfc_plan* p, int64_t count) {
    if (p->xs.n < (size_t)count) { p->xs.alloc((size_t)count); p->ys.alloc((size_t)count); }
}

HostPipe::~HostPipe() {
    for (hipEvent_t e : ev_up) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_down) (void)hipEventDestroy(e);
    if (ev_free) (void)hipEventDestroy(ev_free);
    for (hipEvent_t e : { ev_xfree[0], ev_xfree[1], ev_yfree[0], ev_yfree[1] }) if (e) (void)hipEventDestroy(e);
    if (up) (void)hipStreamDestroy(up);
    if (down) (void)hipStreamDestroy(down);
}

// The drop-in path the reference exercises, mul!(Y, M, b) with HOST vectors (src/FastConvolution.jl:50-54), as a pipeline over
// K chunks of z planes: chunk c of x travels host -> device on its own stream while the x and y passes of the chunks already
// landed run (those passes work plane by plane: yfwd of plane z needs xfwd of plane z only); then the fused z pass, which needs
// every plane; then the inverse y and x passes chunk by chunk, each chunk of y leaving for the host as soon as it exists.
// What cannot overlap: the first byte of y depends on the last byte of x (the Green's kernel is dense), so the upload and
// the download of ONE apply are strictly one after the other -- the floor is 2 N 16 B / PCIe rate + the fused pass
// (512^3: 2 x 37.5 ms at the 57 GB/s this box moves per direction + 4.7 ms), not the full-duplex rate.
// Caller memory: pageable vectors go through the runtime's staged copies (measured 56 / 50 GB/s up / down, no slower than
// pinned on this box); vectors registered with lsfc_host_register (or allocated pinned) are read and written by DMA directly.
static int host_pipeline_chunks(const lsfc_plan* p) {
    const char* e = getenv("LSFC_HOST_PIPELINE");                  // developer switch: 0 / 1 off, K >= 2 chunks whatever the size
    const int forced = e ? atoi(e) : -1;
    if (forced == 0 || forced == 1) return 1;
    if (p->pipeline != lsfc_plan::PRUNED || p->ndim != 3 || p->dist || p->multi) return 1;
    if (forced < 0 && p->N * (int64_t)sizeof(cplx) < ((int64_t)64 << 20)) return 1;       // small vectors: one copy each way
    const int l = p->dims[2];
    for (int K = forced > 1 ? forced : 8; K >= 2; --K) if (l % K == 0) return K;
    return 1;
}

// nrhs right-hand sides through the pipeline with TWO staging slots: while right-hand side j is transformed back and its chunks leave
// for the host, the chunks of right-hand side j + 1 arrive -- across right-hand sides the two PCIe directions do overlap (512^3: 53-55
// ms per right-hand side against 83-87 for one alone).
static void host_pipelined_convolve(lsfc_plan* p, const cplx* x, cplx* y, int64_t nrhs, bool use_nu, double alpha, double beta, int K) {
    if (!p->hostpipe) {
        std::unique_ptr<HostPipe> hp(new HostPipe());
        LSFC_HIP(hipStreamCreateWithFlags(&hp->up, hipStreamNonBlocking));
        LSFC_HIP(hipStreamCreateWithFlags(&hp->down, hipStreamNonBlocking));
        LSFC_HIP(hipEventCreateWithFlags(&hp->ev_free, hipEventDisableTiming));
        for (hipEvent_t* e : { &hp->ev_xfree[0], &hp->ev_xfree[1], &hp->ev_yfree[0], &hp->ev_yfree[1] }) LSFC_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        p->hostpipe = std::move(hp);
    }
    HostPipe* hp = p->hostpipe.get();
    while ((int)hp->ev_up.size() < 2 * K) {
        hipEvent_t a, b;
        LSFC_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming)); hp->ev_up.push_back(a);
        LSFC_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming)); hp->ev_down.push_back(b);
    }
    const int slots = nrhs > 1 ? 2 : 1;
    if (p->xs.n < (size_t)(slots * p->N)) { LSFC_HIP(hipStreamSynchronize(p->stream)); p->xs.alloc((size_t)(slots * p->N)); p->ys.alloc((size_t)(slots * p->N)); }
    const double* nu = use_nu ? p->nu.p : nullptr;
    hipStream_t st = p->stream;
    const int Lx = p->pads[0], Ly = p->pads[1], Lz = p->pads[2];
    const int n = p->dims[0], m = p->dims[1], l = p->dims[2], lz = l / K;
    const int p1 = p->pitch1, p2 = p->pitch2;
    const int64_t chunk = (int64_t)n * m * lz, lines = (int64_t)m * lz;
    // the staging buffers may still be read by work queued earlier on the plan's stream (device-memory calls are asynchronous)
    LSFC_HIP(hipEventRecord(hp->ev_free, st));
    LSFC_HIP(hipStreamWaitEvent(hp->up, hp->ev_free, 0));
    // Several right-hand sides: the downloads are issued from a helper thread.  With pageable caller memory the runtime stages every
    // copy through pinned buffers ON THE CALLING THREAD and returns when it is done, so downloads issued here would keep the
    // uploads of the next right-hand side from even starting; from a second thread the two directions run side by side.
    struct Down { hipEvent_t ev; cplx* dst; const cplx* src; int slot; bool last; int64_t rhs; };
    std::mutex dmu; std::condition_variable dcv; std::deque<Down> dq; bool ddone = false; std::exception_ptr derr;
    int64_t yfree_recorded[2] = { 0, 0 };              // slot s: 1 + the last right-hand side whose final download has been issued (and ev_yfree recorded)
    std::thread dthread;
    const bool helper = nrhs > 1;
    if (helper) dthread = std::thread([&] {
        try {
            LSFC_HIP(hipSetDevice(p->device));
            for (;;) {
                Down d;
                { std::unique_lock<std::mutex> lk(dmu); dcv.wait(lk, [&] { return !dq.empty() || ddone; }); if (dq.empty()) return; d = dq.front(); dq.pop_front(); }
                LSFC_HIP(hipStreamWaitEvent(hp->down, d.ev, 0));
                LSFC_HIP(hipMemcpyAsync(d.dst, d.src, (size_t)chunk * sizeof(cplx), hipMemcpyDeviceToHost, hp->down));
                if (d.last) {
                    LSFC_HIP(hipEventRecord(hp->ev_yfree[d.slot], hp->down));
                    { std::lock_guard<std::mutex> lk(dmu); yfree_recorded[d.slot] = d.rhs + 1; }
                    dcv.notify_all();
                }
            }
        } catch (...) { { std::lock_guard<std::mutex> lk(dmu); derr = std::current_exception(); } dcv.notify_all(); }
    });
    struct Join { std::thread& t; std::mutex& mu; std::condition_variable& cv; bool& done;
                  ~Join() { if (t.joinable()) { { std::lock_guard<std::mutex> lk(mu); done = true; } cv.notify_all(); t.join(); } } } join{dthread, dmu, dcv, ddone};
    for (int64_t j = 0; j < nrhs; ++j) {
        const int sl = (int)(j % slots);
        const cplx* xj = x + j * p->N; cplx* yj = y + j * p->N;
        cplx* xs = p->xs.p + (int64_t)sl * p->N; cplx* ys = p->ys.p + (int64_t)sl * p->N;
        // slot sl was last used by right-hand side j - 2: its inverse x pass (which reads xs) and its downloads (which read ys) must be over
        if (j >= slots) {
            LSFC_HIP(hipStreamWaitEvent(hp->up, hp->ev_xfree[sl], 0));
            if (helper) {
                // (the event of the slot's last download is recorded by the helper: wait until it has been, then order the stream behind it)
                std::unique_lock<std::mutex> lk(dmu);
                dcv.wait(lk, [&] { return yfree_recorded[sl] >= j - slots + 1 || derr; });
                if (derr) std::rethrow_exception(derr);
            }
            LSFC_HIP(hipStreamWaitEvent(st, hp->ev_yfree[sl], 0));
        }
        for (int c = 0; c < K; ++c) {
            const int64_t off = c * chunk;
            hipEvent_t ev = hp->ev_up[(size_t)(sl * K + c)];
            LSFC_HIP(hipMemcpyAsync(xs + off, xj + off, (size_t)chunk * sizeof(cplx), hipMemcpyHostToDevice, hp->up));
            LSFC_HIP(hipEventRecord(ev, hp->up));
            LSFC_HIP(hipStreamWaitEvent(st, ev, 0));
            cplx* a1 = p->A1.p + (int64_t)c * lines * p1;
            pruned_xfwd(Lx, p->tuning, xs + off, nu ? nu + off : nullptr, a1, p->tw[0].p, lines, Lx, p1, n, st);
            pruned_yfwd(Ly, p->tuning, a1, p->A2.p + (int64_t)8 * c * lz, p->tw[1].p, Lx, m, lz, p1, p2, st);
        }
        pruned_zfused(Lz, p->tuning, p->A2.p, p->sym.p, p->tw[2].p, p->twl[2].p, Lx, Ly,
                      (int64_t)p2 * Ly, (int64_t)p2, 8, (int64_t)8 * p->sym_hz * p->sym_rows, (int64_t)8 * p->sym_hz, 8, p->ytab.p, p->zmirror.p, l, st);
        for (int c = 0; c < K; ++c) {
            const int64_t off = c * chunk;
            hipEvent_t ev = hp->ev_down[(size_t)(sl * K + c)];
            cplx* a1 = p->A1.p + (int64_t)c * lines * p1;
            pruned_yinv(Ly, p->tuning, p->A2.p + (int64_t)8 * c * lz, a1, p->tw[1].p, Lx, m, lz, p1, p2, st);
            pruned_xinv(Lx, p->tuning, a1, xs + off, ys + off, alpha, beta, p->tw[0].p, lines, Lx, p1, n, st);
            LSFC_HIP(hipEventRecord(ev, st));
            if (helper) {
                { std::lock_guard<std::mutex> lk(dmu); dq.push_back(Down{ev, yj + off, ys + off, sl, c == K - 1, j}); }
                dcv.notify_all();
            } else {
                LSFC_HIP(hipStreamWaitEvent(hp->down, ev, 0));
                LSFC_HIP(hipMemcpyAsync(yj + off, ys + off, (size_t)chunk * sizeof(cplx), hipMemcpyDeviceToHost, hp->down));
            }
        }
        LSFC_HIP(hipEventRecord(hp->ev_xfree[sl], st));
        if (!helper) LSFC_HIP(hipEventRecord(hp->ev_yfree[sl], hp->down));
    }
    if (helper) {
        { std::lock_guard<std::mutex> lk(dmu); ddone = true; }
        dcv.notify_all();
        dthread.join();
        if (derr) std::rethrow_exception(derr);
    }
    LSFC_HIP(hipStreamSynchronize(hp->down));
    LSFC_HIP(hipStreamSynchronize(st));
}

static void convolve_any(lsfc_plan* p, const double* x, double* y, int64_t nrhs, bool use_nu, double alpha, double beta, int memspace) {
    LSFC_REQUIRE(p && x && y, "NULL argument");
    LSFC_REQUIRE(nrhs >= 1, "nrhs must be >= 1");
    if (p->multi) {
        // one host process, several devices: the vector is scattered over the ranks' z-slabs, applied, gathered
        LSFC_REQUIRE(memspace == LSFC_MEM_HOST, "a multi-device plan takes host vectors here; per-device slabs go through lsfc_multi_apply_dev");
        for (int64_t j = 0; j < nrhs; ++j) multi_convolve_host(p, (const cplx*)x + j * p->N, (cplx*)y + j * p->N, use_nu, alpha, beta);
        return;
    }
    LSFC_HIP(hipSetDevice(p->device));
    // groups of up to LSFC_MAX_BATCH right-hand sides share one pass of the pipeline (one symbol read per group)
    if (memspace == LSFC_MEM_DEVICE) {
        for (int64_t j0 = 0; j0 < nrhs; j0 += LSFC_MAX_BATCH) {
            const int cnt = (int)std::min<int64_t>(LSFC_MAX_BATCH, nrhs - j0);
            VecBatch vb{};
            for (int j = 0; j < cnt; ++j) { vb.x[j] = (const cplx*)x + (j0 + j) * p->N; vb.y[j] = (cplx*)y + (j0 + j) * p->N; }
            plan_convolve_batch_dev(p, cnt, vb, use_nu, alpha, beta);
        }
        return;
    }
    LSFC_REQUIRE(memspace == LSFC_MEM_HOST, "unknown memspace %d", memspace);
    const int group = (int)std::min<int64_t>(LSFC_MAX_BATCH, nrhs);
    // large 3D vectors: the chunked pipeline above (two staging slots, allocated there)
    const int K = host_pipeline_chunks(p);
    if (K <= 1) ensure_staging(p, p->N * group);
    if (K > 1) {
        // (y aliasing x: right-hand side j + 1 is uploaded while j is downloaded into the same memory -- distinct vectors, no hazard)
        host_pipelined_convolve(p, (const cplx*)x, (cplx*)y, nrhs, use_nu, alpha, beta, K);
        return;
    }
    // LSFC_HOST_COPY=sync (developer switch, diagnostics of the round-2 first-apply fault): the round-2 workaround, a
    // synchronous copy behind a drained stream, instead of the stream-ordered copy
    static const bool sync_copy = [] { const char* e = getenv("LSFC_HOST_COPY"); return e && e[0] == 's'; }();
    for (int64_t j0 = 0; j0 < nrhs; j0 += group) {
        const int cnt = (int)std::min<int64_t>(group, nrhs - j0);
        const size_t bytes = (size_t)cnt * p->N * sizeof(cplx);
        if (sync_copy) {
            LSFC_HIP(hipStreamSynchronize(p->stream));
            LSFC_HIP(hipMemcpy(p->xs.p, (const cplx*)x + j0 * p->N, bytes, hipMemcpyHostToDevice));
        } else LSFC_HIP(hipMemcpyAsync(p->xs.p, (const cplx*)x + j0 * p->N, bytes, hipMemcpyHostToDevice, p->stream));
        VecBatch vb{};
        for (int j = 0; j < cnt; ++j) { vb.x[j] = p->xs.p + (int64_t)j * p->N; vb.y[j] = p->ys.p + (int64_t)j * p->N; }
        plan_convolve_batch_dev(p, cnt, vb, use_nu, alpha, beta);
        LSFC_HIP(hipMemcpyAsync((cplx*)y + j0 * p->N, p->ys.p, bytes, hipMemcpyDeviceToHost, p->stream));
        LSFC_HIP(hipStreamSynchronize(p->stream));
    }
}

} // namespace lsfc

using namespace lsfc;

lsfc_plan::lsfc_plan() {}
lsfc_plan::~lsfc_plan() {}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

const char* lsfc_last_error(void) { return g_last_error; }
int lsfc_padded_length(int64_t n) { return (n >= 1 && n <= 1024) ? lsfc::pruned_best_length(n) : 0; }

const char* lsfc_version(void) { return "lsfc 0.1 (gfx950, rocFFT + hand-written pruned FFT passes)"; }

static int create_from_literal(lsfc_plan** out, int ndim, int64_t n, int64_t m, int64_t l, int64_t ne, int64_t me, int64_t le,
                               const double* nu, const double* gfft, double omega, int quad_rule, unsigned flags, int device) {
    return guarded([&] {
        LSFC_REQUIRE(out && gfft, "NULL argument");
        *out = nullptr;
        std::unique_ptr<lsfc_plan> p(new lsfc_plan());
        plan_common_init(p.get(), ndim, n, m, l, nu, omega, quad_rule, flags, device);
        LSFC_REQUIRE(ne >= 1 && me >= 1 && le >= 1 && ne <= 16384 && me <= 16384 && le <= 16384, "padded size out of range");
        const int lit[3] = { (int)ne, (int)me, (int)le };
        const int64_t ltotal = ne * me * le;
        DevBuf<cplx> Gd; Gd.alloc((size_t)ltotal);
        LSFC_HIP(hipMemcpy(Gd.p, gfft, (size_t)ltotal * sizeof(cplx), hipMemcpyHostToDevice));
        const bool trap = quad_rule == LSFC_QUAD_TRAPEZOIDAL;
        // trapezoidal: plain FFT-order symbol on the (2n-1) grid, crop window [n-1, 2n-2] (src/FastConvolution.jl:70-82)
        for (int d = 0; d < ndim; ++d)
            LSFC_REQUIRE(lit[d] >= (trap ? 2 * p->dims[d] - 1 : p->dims[d]), "padded size %d too small along axis %d", lit[d], d);
        int origin[3]; for (int d = 0; d < 3; ++d) origin[d] = (trap && d < ndim) ? p->dims[d] - 1 : 0;
        if (flags & LSFC_FLAG_LITERAL_PAD) {
            for (int d = 0; d < 3; ++d) { p->pads[d] = lit[d]; p->crop[d] = origin[d]; }
            plan_finish_literal(p.get(), Gd.p, !trap);
        } else {
            plan_finish_reduce(p.get(), Gd, lit, !trap, origin);
        }
        *out = p.release();
    });
}

int lsfc_plan_create_2d(lsfc_plan** out, int64_t n, int64_t m, int64_t ne, int64_t me, const double* nu, const double* gfft,
                        double omega, int quad_rule, unsigned flags, int device) {
    return create_from_literal(out, 2, n, m, 1, ne, me, 1, nu, gfft, omega, quad_rule, flags, device);
}
int lsfc_plan_create_3d(lsfc_plan** out, int64_t n, int64_t m, int64_t l, int64_t ne, int64_t me, int64_t le, const double* nu,
                        const double* gfft, double omega, int quad_rule, unsigned flags, int device) {
    return create_from_literal(out, 3, n, m, l, ne, me, le, nu, gfft, omega, quad_rule, flags, device);
}

int lsfc_plan_create_gv3d(lsfc_plan** out, int64_t n, int64_t m, int64_t l, double box, double omega, const double* nu,
                          unsigned flags, int device) {
    return guarded([&] {
        LSFC_REQUIRE(out, "NULL argument"); *out = nullptr;
        LSFC_REQUIRE(n % 2 == 0 && m % 2 == 0 && l % 2 == 0, "buildFastConvolution3D: even n only (the reference's odd branch is broken)");
        LSFC_REQUIRE(box > 0, "box must be positive");
        PhaseTimer pt;
        std::unique_ptr<lsfc_plan> p(new lsfc_plan());
        plan_common_init(p.get(), 3, n, m, l, nu, omega, LSFC_QUAD_GREENGARD_VICO, flags & ~LSFC_FLAG_LITERAL_PAD, device);
        pt.mark("init + nu upload");
        DevBuf<cplx> G2;
        plan_choose_reduced_grid(p.get());
        if (p->pipeline == lsfc_plan::PRUNED && plan_quarter_symbol_ok(p.get())) {
            symbol_gv3d_quarter(p.get(), box, G2);
            pt.mark("symbol (total; through its symmetry)");
            plan_finish_from_quarter(p.get(), G2);
        } else {
            symbol_gv3d_reduced(p.get(), box, G2);
            pt.mark("symbol (total)");
            plan_finish_from_reduced(p.get(), G2);
        }
        pt.mark("tables, symbol permutation, work arrays");
        *out = p.release();
    });
}

int lsfc_plan_create_gv2d(lsfc_plan** out, int64_t n, int64_t m, double box, double omega, const double* nu, unsigned flags, int device) {
    return guarded([&] {
        LSFC_REQUIRE(out, "NULL argument"); *out = nullptr;
        LSFC_REQUIRE(box > 0, "box must be positive");
        std::unique_ptr<lsfc_plan> p(new lsfc_plan());
        plan_common_init(p.get(), 2, n, m, 1, nu, omega, LSFC_QUAD_GREENGARD_VICO, flags, device);
        DevBuf<cplx> G; int lit[3];
        symbol_gv2d_literal(p.get(), box, G, lit);
        const int origin[3] = { 0, 0, 0 };
        if (flags & LSFC_FLAG_LITERAL_PAD) {
            for (int d = 0; d < 3; ++d) { p->pads[d] = lit[d]; p->crop[d] = 0; }
            plan_finish_literal(p.get(), G.p, true);
        } else plan_finish_reduce(p.get(), G, lit, true, origin);
        *out = p.release();
    });
}

int lsfc_plan_create_trap2d(lsfc_plan** out, int64_t n, int64_t m, double x0, double y0, double h, double omega,
                            double d0_re, double d0_im, const double* nu, unsigned flags, int device) {
    return guarded([&] {
        LSFC_REQUIRE(out, "NULL argument"); *out = nullptr;
        LSFC_REQUIRE(n % 2 == 1 && m % 2 == 1, "so far only works for n odd (src/FastConvolution.jl:455)");
        std::unique_ptr<lsfc_plan> p(new lsfc_plan());
        plan_common_init(p.get(), 2, n, m, 1, nu, omega, LSFC_QUAD_TRAPEZOIDAL, flags, device);
        DevBuf<cplx> G;
        symbol_trap2d_literal(p.get(), x0, y0, h, make_double2(d0_re, d0_im), G);
        const int lit[3] = { 2 * (int)n - 1, 2 * (int)m - 1, 1 };
        const int origin[3] = { (int)n - 1, (int)m - 1, 0 };
        if (flags & LSFC_FLAG_LITERAL_PAD) {
            for (int d = 0; d < 3; ++d) { p->pads[d] = lit[d]; p->crop[d] = origin[d]; }
            plan_finish_literal(p.get(), G.p, false);
        } else plan_finish_reduce(p.get(), G, lit, false, origin);
        *out = p.release();
    });
}

int lsfc_plan_destroy(lsfc_plan* plan) {
    return guarded([&] {
        if (!plan) return;
        if (plan->multi) multi_synchronize(plan);
        else { (void)hipSetDevice(plan->device); (void)hipStreamSynchronize(plan->stream); }
        delete plan;
    });
}

int64_t lsfc_plan_size(const lsfc_plan* plan) { return plan ? plan->N : -1; }

int lsfc_plan_dims(const lsfc_plan* plan, int64_t dims[3], int64_t pads[3]) {
    return guarded([&] {
        LSFC_REQUIRE(plan, "NULL plan");
        for (int d = 0; d < 3; ++d) { if (dims) dims[d] = plan->dims[d]; if (pads) pads[d] = plan->pads[d]; }
    });
}

const char* lsfc_plan_pipeline(const lsfc_plan* plan) {
    if (!plan) return "";
    switch (plan->pipeline) {
    case lsfc_plan::PRUNED: return "pruned-hip";
    case lsfc_plan::ROCFFT_REDUCED: return "rocfft-reduced";
    default: return "rocfft-literal";
    }
}

// (kernel0 does not depend on nu: lsfc_plan_set_nu keeps it)
int lsfc_plan_set_nu(lsfc_plan* plan, const double* nu, int memspace) {
    return guarded([&] {
        LSFC_REQUIRE(plan && nu, "NULL argument");
        if (plan->multi) {
            LSFC_REQUIRE(memspace == LSFC_MEM_HOST, "a multi-device plan takes nu from the host");
            int64_t off = 0;
            for (auto& sp : plan->multi->sub) {
                LSFC_HIP(hipSetDevice(sp->device));
                LSFC_HIP(hipMemcpy(sp->nu.p, nu + off, sp->N * sizeof(double), hipMemcpyHostToDevice));
                off += sp->N;
            }
            return;
        }
        LSFC_HIP(hipSetDevice(plan->device));
        LSFC_HIP(hipMemcpyAsync(plan->nu.p, nu, plan->N * sizeof(double),
                                memspace == LSFC_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, plan->stream));
        LSFC_HIP(hipStreamSynchronize(plan->stream));
    });
}

int lsfc_plan_get_symbol(const lsfc_plan* plan, double* out, int64_t capacity_complex, int64_t* count) {
    return guarded([&] {
        LSFC_REQUIRE(plan && count, "NULL argument");
        LSFC_REQUIRE(!plan->multi, "not available on a multi-device plan");
        *count = (int64_t)plan->sym.n;
        if (out) {
            LSFC_REQUIRE(capacity_complex >= (int64_t)plan->sym.n, "buffer too small");
            LSFC_HIP(hipMemcpy(out, plan->sym.p, plan->sym.bytes(), hipMemcpyDeviceToHost));
        }
    });
}

int lsfc_apply(lsfc_plan* plan, const double* x, double* y, int memspace) {
    return guarded([&] { LSFC_REQUIRE(plan, "NULL plan"); convolve_any(plan, x, y, 1, true, 1.0, plan->omega * plan->omega, memspace); });
}
int lsfc_convolve(lsfc_plan* plan, const double* x, double* y, int apply_nu, int memspace) {
    return guarded([&] { convolve_any(plan, x, y, 1, apply_nu != 0, 0.0, 1.0, memspace); });
}
int lsfc_apply_batch(lsfc_plan* plan, const double* x, double* y, int64_t nrhs, int mode, int memspace) {
    return guarded([&] {
        LSFC_REQUIRE(plan, "NULL plan");
        LSFC_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (apply), 1 (convolve) or 2 (convolve with nu)");
        if (mode == 0) convolve_any(plan, x, y, nrhs, true, 1.0, plan->omega * plan->omega, memspace);
        else convolve_any(plan, x, y, nrhs, mode == 2, 0.0, 1.0, memspace);
    });
}

int lsfc_sample_sources(lsfc_plan* plan, const int64_t* sources, int64_t nsrc, double* out, int memspace) {
    return guarded([&] {
        LSFC_REQUIRE(plan && sources && out && nsrc >= 1, "bad argument");
        LSFC_REQUIRE(!plan->dist && !plan->multi, "not available on a distributed plan");
        LSFC_HIP(hipSetDevice(plan->device));
        lsfc_plan* p = plan;
        const int64_t N = p->N;
        for (int64_t s = 0; s < nsrc; ++s) LSFC_REQUIRE(sources[s] >= 0 && sources[s] < N, "source index %lld out of range", (long long)sources[s]);
        // kernel samples: the response to a unit source at grid index 0 (one FFT convolution, no nu)
        if (p->kernel0.n != (size_t)N) {
            p->kernel0.alloc((size_t)N);
            DevBuf<cplx> e; e.alloc((size_t)N);
            LSFC_HIP(hipMemsetAsync(e.p, 0, (size_t)N * sizeof(cplx), p->stream));
            const cplx one = make_double2(1.0, 0.0);
            LSFC_HIP(hipMemcpyAsync(e.p, &one, sizeof one, hipMemcpyHostToDevice, p->stream));
            plan_convolve_dev(p, e.p, p->kernel0.p, false, 0.0, 1.0);
            LSFC_HIP(hipStreamSynchronize(p->stream));
        }
        DevBuf<int64_t> dsrc; dsrc.alloc((size_t)nsrc);
        LSFC_HIP(hipMemcpyAsync(dsrc.p, sources, (size_t)nsrc * sizeof(int64_t), hipMemcpyHostToDevice, p->stream));
        if (memspace == LSFC_MEM_DEVICE) {
            pw_gather_sources(p->kernel0.p, (cplx*)out, dsrc.p, (int)nsrc, p->dims, p->stream);
            LSFC_HIP(hipStreamSynchronize(p->stream));
        } else {
            // stage in batches of at most ~256 MiB
            const int64_t per = std::max<int64_t>(1, ((int64_t)1 << 24) / N);
            DevBuf<cplx> buf; buf.alloc((size_t)(std::min(per, nsrc) * N));
            for (int64_t s0 = 0; s0 < nsrc; s0 += per) {
                const int64_t cnt = std::min(per, nsrc - s0);
                pw_gather_sources(p->kernel0.p, buf.p, dsrc.p + s0, (int)cnt, p->dims, p->stream);
                LSFC_HIP(hipMemcpyAsync((cplx*)out + s0 * N, buf.p, (size_t)(cnt * N) * sizeof(cplx), hipMemcpyDeviceToHost, p->stream));
                LSFC_HIP(hipStreamSynchronize(p->stream));
            }
        }
    });
}

int lsfc_gmres(lsfc_plan* plan, double* x, const double* b, const lsfc_gmres_opts* opts, double* resnorm, int64_t resnorm_cap,
               lsfc_gmres_result* result, int memspace) {
    int conv_code = LSFC_OK;
    int rc = guarded([&] {
        LSFC_REQUIRE(plan && x && b, "NULL argument");
        LSFC_HIP(hipSetDevice(plan->device));
        lsfc_gmres_result local; lsfc_gmres_result* res = result ? result : &local;
        if (plan->multi) {
            LSFC_REQUIRE(memspace == LSFC_MEM_HOST, "a multi-device plan takes host vectors");
            gmres_run_multi(plan, (cplx*)x, (const cplx*)b, opts, resnorm, resnorm_cap, res);
        } else if (memspace == LSFC_MEM_DEVICE) {
            gmres_run(plan, (cplx*)x, (const cplx*)b, opts, resnorm, resnorm_cap, res);
        } else {
            LSFC_REQUIRE(memspace == LSFC_MEM_HOST, "unknown memspace %d", memspace);
            ensure_staging(plan, plan->N);
            LSFC_HIP(hipMemcpy(plan->xs.p, x, plan->N * sizeof(cplx), hipMemcpyHostToDevice));
            LSFC_HIP(hipMemcpy(plan->ys.p, b, plan->N * sizeof(cplx), hipMemcpyHostToDevice));
            gmres_run(plan, plan->xs.p, plan->ys.p, opts, resnorm, resnorm_cap, res);
            LSFC_HIP(hipMemcpy(x, plan->xs.p, plan->N * sizeof(cplx), hipMemcpyDeviceToHost));
        }
        if (!res->converged) conv_code = LSFC_ENOTCONV;
    });
    if (rc == LSFC_OK && conv_code != LSFC_OK) { set_last_error("gmres: maxiter reached without convergence"); return conv_code; }
    return rc;
}

int lsfc_gmres_batch(lsfc_plan* plan, double* x, const double* b, int64_t nrhs, const lsfc_gmres_opts* opts, double* resnorm,
                     int64_t resnorm_cap, lsfc_gmres_result* results, int memspace) {
    return guarded([&] {
        LSFC_REQUIRE(plan && x && b && results, "NULL argument");
        LSFC_REQUIRE(nrhs >= 1 && nrhs <= 64, "nrhs must be in 1..64");
        LSFC_REQUIRE(!plan->multi && !plan->dist, "batched GMRES runs on a single-device plan");
        LSFC_HIP(hipSetDevice(plan->device));
        if (memspace == LSFC_MEM_DEVICE) {
            gmres_run_batch(plan, (cplx*)x, (const cplx*)b, (int)nrhs, opts, resnorm, resnorm_cap, results);
        } else {
            LSFC_REQUIRE(memspace == LSFC_MEM_HOST, "unknown memspace %d", memspace);
            const size_t bytes = (size_t)nrhs * plan->N * sizeof(cplx);
            DevBuf<cplx> xd, bd; xd.alloc((size_t)nrhs * plan->N); bd.alloc((size_t)nrhs * plan->N);
            LSFC_HIP(hipMemcpy(xd.p, x, bytes, hipMemcpyHostToDevice));
            LSFC_HIP(hipMemcpy(bd.p, b, bytes, hipMemcpyHostToDevice));
            gmres_run_batch(plan, xd.p, bd.p, (int)nrhs, opts, resnorm, resnorm_cap, results);
            LSFC_HIP(hipMemcpy(x, xd.p, bytes, hipMemcpyDeviceToHost));
        }
    });
}

int lsfc_plan_set_stream(lsfc_plan* plan, void* hip_stream) {
    return guarded([&] {
        LSFC_REQUIRE(plan, "NULL plan");
        LSFC_REQUIRE(!plan->multi, "a multi-device plan runs on the streams of its ranks");
        plan->stream = (hipStream_t)hip_stream;
    });
}
int lsfc_plan_synchronize(lsfc_plan* plan) {
    return guarded([&] {
        LSFC_REQUIRE(plan, "NULL plan");
        if (plan->multi) { multi_synchronize(plan); return; }
        LSFC_HIP(hipSetDevice(plan->device)); LSFC_HIP(hipStreamSynchronize(plan->stream));
    });
}

int lsfc_plan_set_tuning(lsfc_plan* plan, const char* key, int value) {
    return guarded([&] {
        LSFC_REQUIRE(plan && key, "NULL argument");
        if (plan->multi) { for (auto& sp : plan->multi->sub) { const int rc = lsfc_plan_set_tuning(sp.get(), key, value); if (rc != LSFC_OK) fail(rc, "%s", lsfc_last_error()); } return; }
        const std::string k(key);
        if (k == "split_x") plan->tuning.split_x = value != 0;
        else if (k == "split_s") plan->tuning.split_s = value != 0;
        else if (k == "split_z") plan->tuning.split_z = value;
        else if (k == "z_half") plan->tuning.z_half = value;
        else if (k == "tw_lds") plan->tuning.tw_lds = value;
        else if (k == "sym_prefetch") plan->tuning.sym_prefetch = value;
        else if (k == "ytile_g") plan->tuning.ytile_g = value;
        else if (k == "ytile_z") plan->tuning.ytile_z = value;
        else if (k == "batch_fuse") plan->tuning.batch_fuse = value;
        else if (k == "z_persist") plan->tuning.z_persist = value;
        else if (k == "xlane") plan->tuning.xlane = value;
        else fail(LSFC_EINVAL, "unknown tuning key '%s'", key);
    });
}

int lsfc_time_apply(lsfc_plan* plan, const double* x_dev, double* y_dev, int reps, double* ms_total) {
    return guarded([&] {
        LSFC_REQUIRE(plan && x_dev && y_dev && ms_total && reps >= 1, "bad argument");
        LSFC_REQUIRE(!plan->multi, "time a multi-device plan from the host around lsfc_multi_apply_dev + lsfc_plan_synchronize");
        LSFC_HIP(hipSetDevice(plan->device));
        hipEvent_t e0, e1;
        LSFC_HIP(hipEventCreate(&e0)); LSFC_HIP(hipEventCreate(&e1));
        LSFC_HIP(hipEventRecord(e0, plan->stream));
        for (int r = 0; r < reps; ++r) plan_apply_dev(plan, (const cplx*)x_dev, (cplx*)y_dev);
        LSFC_HIP(hipEventRecord(e1, plan->stream));
        LSFC_HIP(hipEventSynchronize(e1));
        float ms = 0; LSFC_HIP(hipEventElapsedTime(&ms, e0, e1));
        *ms_total = ms;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    });
}

int lsfc_profile_apply(lsfc_plan* plan, const double* x_dev, double* y_dev, int reps, int max_stages, const char** names,
                       double* ms, double* bytes, int* nstages) {
    return guarded([&] {
        LSFC_REQUIRE(plan && names && ms && bytes && nstages && reps >= 1, "bad argument");
        if (plan->multi) { multi_profile(plan, reps, max_stages, names, ms, bytes, nstages); return; }    // (on the ranks' staging vectors)
        LSFC_REQUIRE(x_dev && y_dev, "bad argument");
        LSFC_HIP(hipSetDevice(plan->device));
        lsfc_plan* p = plan;
        const cplx* x = (const cplx*)x_dev; cplx* y = (cplx*)y_dev;
        hipStream_t st = p->stream;
        const double N = (double)p->N, C = 16.0;
        struct Stage { const char* name; double bytes; std::function<void()> run; };
        std::vector<Stage> stages;
        auto stages_add_fn = [](std::vector<Stage>& v) {
            return [&v](const char* name, double bytes, std::function<void()> run) { v.push_back({name, bytes, std::move(run)}); };
        };
        const double om2 = p->omega * p->omega;
        if (p->dist) {
            dist_profile_stages(p, x, y, stages_add_fn(stages));
        } else if (p->pipeline == lsfc_plan::PRUNED) {
            const int Lx = p->pads[0], Ly = p->pads[1], Lz = p->pads[2];
            const int m = p->dims[1], l = p->dims[2];
            const int64_t nlines = (int64_t)m * l;
            if (p->ndim == 2) {
                VecBatch vb{}; vb.x[0] = x; vb.y[0] = y;
                stages.push_back({"xfwd", N * (C + 8) + 2 * N * C, [=] { pass2d_xfwd(p, vb, 1, p->nu.p, st); }});
                stages.push_back({"yfused", (2 + 4 + 2) * N * C, [=] { pass2d_yfused(p, 1, st); }});
                stages.push_back({"xinv", (2 + 1 + 1) * N * C, [=] { pass2d_xinv(p, vb, 1, 1.0, om2, st); }});
            } else {
            stages.push_back({"xfwd", N * (C + 8) + 2 * N * C, [=] { pruned_xfwd(Lx, p->tuning, x, p->nu.p, p->A1.p, p->tw[0].p, nlines, Lx, p->pitch1, p->dims[0], st); }});
            if (p->ndim == 3) {
                stages.push_back({"yfwd", (2 + 4) * N * C, [=] { pruned_yfwd(Ly, p->tuning, p->A1.p, p->A2.p, p->tw[1].p, Lx, m, l, p->pitch1, p->pitch2, st); }});
                stages.push_back({"zfused", (4 + 8 + 4) * N * C, [=] { pruned_zfused(Lz, p->tuning, p->A2.p, p->sym.p, p->tw[2].p, p->twl[2].p, Lx, Ly,
                                  (int64_t)p->pitch2 * Ly, (int64_t)p->pitch2, 8, (int64_t)8 * p->sym_hz * p->sym_rows, (int64_t)8 * p->sym_hz, 8,
                                  p->ytab.p, p->zmirror.p, l, st); }});
                stages.push_back({"yinv", (4 + 2) * N * C, [=] { pruned_yinv(Ly, p->tuning, p->A2.p, p->A1.p, p->tw[1].p, Lx, m, l, p->pitch1, p->pitch2, st); }});
            } else {
                stages.push_back({"yfused", (2 + 4 + 2) * N * C, [=] { pruned_zfused(Ly, p->tuning, p->A1.p, p->sym.p, p->tw[1].p, p->twl[1].p, Lx, 1, 8, 0, p->pitch1, 8, 0, Lx, nullptr, p->zmirror.p, m, st); }});
            }
            stages.push_back({"xinv", (2 + 1 + 1) * N * C, [=] { pruned_xinv(Lx, p->tuning, p->A1.p, x, y, 1.0, om2, p->tw[0].p, nlines, Lx, p->pitch1, p->dims[0], st); }});
            }
        } else {
            const int64_t total = (int64_t)p->pads[0] * p->pads[1] * p->pads[2];
            const double P = (double)total;
            stages.push_back({"embed", N * (C + 8) + P * C, [=] { pw_embed(x, p->nu.p, p->W.p, p->dims, p->pads, st); }});
            stages.push_back({"rocfft_fwd", 2 * P * C, [=] { p->fwd->exec(p->W.p, st); }});
            stages.push_back({"symbol_mul", 3 * P * C, [=] { pw_mul_inplace(p->W.p, p->sym.p, total, st); }});
            stages.push_back({"rocfft_inv", 2 * P * C, [=] { p->inv->exec(p->W.p, st); }});
            stages.push_back({"crop_axpy", 3 * N * C, [=] { pw_crop_axpy(p->W.p, x, y, 1.0, om2, p->dims, p->pads, p->crop, st); }});
        }
        LSFC_REQUIRE((int)stages.size() <= max_stages, "max_stages too small (need %d)", (int)stages.size());
        *nstages = (int)stages.size();
        std::vector<hipEvent_t> ev(stages.size() + 1);
        for (auto& e : ev) LSFC_HIP(hipEventCreate(&e));
        for (size_t i = 0; i < stages.size(); ++i) { names[i] = stages[i].name; bytes[i] = stages[i].bytes; ms[i] = 0.0; }
        for (int r = 0; r < reps; ++r) {
            LSFC_HIP(hipEventRecord(ev[0], st));
            for (size_t i = 0; i < stages.size(); ++i) { stages[i].run(); LSFC_HIP(hipEventRecord(ev[i + 1], st)); }
            LSFC_HIP(hipEventSynchronize(ev[stages.size()]));
            for (size_t i = 0; i < stages.size(); ++i) { float t = 0; LSFC_HIP(hipEventElapsedTime(&t, ev[i], ev[i + 1])); ms[i] += t; }
        }
        for (size_t i = 0; i < stages.size(); ++i) ms[i] /= reps;
        for (auto& e : ev) (void)hipEventDestroy(e);
    });
}

int lsfc_device_count(int* count) {
    return guarded([&] { LSFC_REQUIRE(count, "NULL argument"); int c = 0; hipError_t e = hipGetDeviceCount(&c); *count = (e == hipSuccess) ? c : 0; });
}
int lsfc_malloc(void** dptr, size_t bytes, int device) {
    return guarded([&] { LSFC_REQUIRE(dptr, "NULL argument"); select_device(device); LSFC_HIP(hipMalloc(dptr, bytes)); });
}
int lsfc_free(void* dptr) { return guarded([&] { if (dptr) LSFC_HIP(hipFree(dptr)); }); }
int lsfc_memcpy_h2d(void* dst, const void* src, size_t bytes) { return guarded([&] { LSFC_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); }); }
int lsfc_memcpy_d2h(void* dst, const void* src, size_t bytes) { return guarded([&] { LSFC_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); }); }
int lsfc_host_register(void* ptr, size_t bytes) {
    return guarded([&] { LSFC_REQUIRE(ptr && bytes, "NULL argument"); LSFC_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault)); });
}
int lsfc_host_unregister(void* ptr) { return guarded([&] { LSFC_REQUIRE(ptr, "NULL argument"); LSFC_HIP(hipHostUnregister(ptr)); }); }
int lsfc_host_alloc(void** ptr, size_t bytes) {
    return guarded([&] { LSFC_REQUIRE(ptr && bytes, "NULL argument"); *ptr = nullptr; LSFC_HIP(hipHostMalloc(ptr, bytes, hipHostMallocDefault)); });
}
int lsfc_host_free(void* ptr) { return guarded([&] { if (ptr) LSFC_HIP(hipHostFree(ptr)); }); }

} // extern "C"
