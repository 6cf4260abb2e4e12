// Device-resident apply of the reference's SparsifyingPreconditioner (src/preconditioner.jl:27-58, 132-170):
//
//     v  <-  Msp^{-1} (As v)
//
// As is a sparse matrix (SpMV), Msp^{-1} is applied through the LU factors the caller computed on the host (UMFPACK
// `lu(Msp)` in the reference, src/preconditioner.jl:35): (Rs .* Msp)[p, q] = L U.  The factorisation itself, and the
// assembly of As / Msp, stay on the host and out of scope; what moves to the device is the part that runs once per
// Arnoldi step, so the Krylov vector no longer crosses PCIe twice per step.
//
// Sparse triangular solves by level scheduling: rows are grouped by dependency depth.  The factors of a 2D/3D stencil
// matrix are deep and thin, so four kinds of steps are scheduled at set-up:
//   * a wide or heavy level              one launch over all its rows, 8..64 lanes per row by row length
//   * a run of light, narrow levels      ONE single-workgroup launch that walks them with barriers, 1024 / rows lanes
//                                        per row; consecutive thin levels are fused into groups of <= 16 rows whose
//                                        mutual coupling is a dense 16 x 16 block resolved by one wave in registers
//   * a heavy group                      the same in two launches: one workgroup per row, then the dense block
//   * a long sequence of thin levels     a dense run: coupling stored as a dense R x R block, blocked right-looking
//                                        forward substitution (the trailing separator blocks of the factor)
// The whole sequence SpMV -> L solve -> U solve -> scatter is captured once in a hipGraph on fixed internal buffers
// and replayed per apply.
#include "common.hpp"
#include "pruned.hpp"
#include <algorithm>
#include <complex>
#include <cstring>
#include <memory>
#include <vector>

namespace lsfc {

// ---- kernels -------------------------------------------------------------------------------------------------------

// y[k] = scale[row] * sum_j A[row, j] x[j],  row = gather[k]   (CSR, LPR lanes per row)
template <int LPR>
__global__ void k_spmv_gather(const int64_t* __restrict__ rowptr, const int* __restrict__ col, const cplx* __restrict__ val,
                              const int* __restrict__ gather, const double* __restrict__ scale, const cplx* __restrict__ x,
                              cplx* __restrict__ y, int nrows) {
    const int g = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / LPR), lane = threadIdx.x % LPR;
    if (g >= nrows) return;                       // (whole LPR-groups leave together: blockDim is a multiple of LPR)
    const int row = gather ? gather[g] : g;
    double sx = 0.0, sy = 0.0;
    for (int64_t e = rowptr[row] + lane; e < rowptr[row + 1]; e += LPR) {
        const cplx a = val[e], b = x[col[e]];
        sx += a.x * b.x - a.y * b.y; sy += a.x * b.y + a.y * b.x;
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, LPR); sy += __shfl_xor(sy, o, LPR); }
    if (lane == 0) { const double s = scale ? scale[row] : 1.0; y[g] = make_double2(s * sx, s * sy); }
}

// one row of a triangular solve, LANES lanes: x[row] = (b[row] - sum_{j != row} a_j x[col_j]) * inv_diag
// (rows are stored in level order: entry k of the sorted arrays is original row rowid[k]; off-diagonal entries only)
template <int LANES>
__device__ __forceinline__ void trsv_row(int k, int lane, const int64_t* __restrict__ rowptr, const int* __restrict__ col,
                                         const cplx* __restrict__ val, const int* __restrict__ rowid, const cplx* __restrict__ invd,
                                         const cplx* __restrict__ b, cplx* x) {
    double sx = 0.0, sy = 0.0;
    for (int64_t e = rowptr[k] + lane; e < rowptr[k + 1]; e += LANES) {
        const cplx a = val[e], v = x[col[e]];
        sx += a.x * v.x - a.y * v.y; sy += a.x * v.y + a.y * v.x;
    }
#pragma unroll
    for (int o = LANES / 2; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, LANES); sy += __shfl_xor(sy, o, LANES); }
    if (lane == 0) {
        const int row = rowid[k];
        const cplx r = make_double2(b[row].x - sx, b[row].y - sy), d = invd[k];
        x[row] = make_double2(r.x * d.x - r.y * d.y, r.x * d.y + r.y * d.x);
    }
}

// one level as its own launch over many workgroups: rows [first, first + nrows) of the sorted order, LPR lanes per row
// (8 for the short rows of the early levels, a whole wave for the long rows of the fill)
template <int LPR>
__global__ void k_trsv_level(int first, int nrows, const int64_t* __restrict__ rowptr, const int* __restrict__ col, const cplx* __restrict__ val,
                             const int* __restrict__ rowid, const cplx* __restrict__ invd, const cplx* __restrict__ b, cplx* x) {
    const int g = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / LPR), lane = threadIdx.x % LPR;
    if (g >= nrows) return;
    trsv_row<LPR>(first + g, lane, rowptr, col, val, rowid, invd, b, x);
}

// a run of narrow levels in ONE workgroup of 1024 threads, a barrier between steps.  The tail of a sparse LU factor
// is deep and thin -- at 2D 257 x 257 a third of the fill sits in 1391 levels of ONE row with ~740 entries each (the
// dense trailing separator blocks) -- so
//  * the lanes are dealt out per step: 1024 / (rows rounded up to a power of two) lanes per row, all 1024 on a single
//    row; sums are folded by wave shuffles and, beyond one wave per row, through LDS;
//  * consecutive thin levels are fused into GROUPS of at most 16 rows: the entries that couple rows of the same group
//    are taken out of the CSR rows into a dense 16 x 16 block at set-up, every row of the group sums its remaining
//    (external) entries at the same time, and one wave then resolves the little dense triangle in registers
//    (16 shuffle steps, no memory latency).  One barrier-bound step thus retires up to 16 dependency levels.
// A group with dense offset -1 is a single level (independent rows).
static constexpr int GROUP_ROWS = 16;

// shared by the walking kernel and the heavy-group kernel: lanes i < nrows of ONE wave hold acc_i = b_i - (external sum);
// forward substitution through the dense block, x_i written by lane i
__device__ __forceinline__ void dense_resolve(int i, int nrows, int first, int dofs, double ax, double ay, const cplx* __restrict__ dense,
                                              const int* __restrict__ rowid, const cplx* __restrict__ invd, cplx* x) {
    const bool mine = i < nrows;
    const cplx di = mine ? invd[first + i] : make_double2(0.0, 0.0);
    cplx drow[GROUP_ROWS];
#pragma unroll
    for (int j = 0; j < GROUP_ROWS; ++j) drow[j] = (mine && j < i) ? dense[(int64_t)dofs + i * GROUP_ROWS + j] : make_double2(0.0, 0.0);
    double myx = 0.0, myy = 0.0;
#pragma unroll
    for (int j = 0; j < GROUP_ROWS; ++j) {
        if (j < nrows) {                                       // uniform
            const double tx = ax * di.x - ay * di.y, ty = ax * di.y + ay * di.x;      // x_j on lane j
            const double xjx = __shfl(tx, j, 64), xjy = __shfl(ty, j, 64);
            if (i == j) { myx = xjx; myy = xjy; }
            ax -= drow[j].x * xjx - drow[j].y * xjy; ay -= drow[j].x * xjy + drow[j].y * xjx;
        }
    }
    if (mine) x[rowid[first + i]] = make_double2(myx, myy);
}


__global__ __launch_bounds__(1024)
void k_trsv_chain(int g0, int g1, const int* __restrict__ grpptr, const int* __restrict__ grpcnt, const int* __restrict__ grpdense,
                  const cplx* __restrict__ dense,
                  const int64_t* __restrict__ rowptr, const int* __restrict__ col, const cplx* __restrict__ val,
                  const int* __restrict__ rowid, const cplx* __restrict__ invd, const cplx* __restrict__ b, cplx* x) {
    __shared__ double psx[16], psy[16], rx[GROUP_ROWS], ry[GROUP_ROWS];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int gi = g0; gi < g1; ++gi) {
        const int first = grpptr[gi], nrows = grpcnt[gi], dofs = grpdense[gi];     // uniform over the workgroup
        int p2 = 1; while (p2 < nrows) p2 <<= 1;
        const int lanes = 1024 / p2;                                       // 8 ... 1024, a power of two
        const int g = tid / lanes, lane = tid % lanes;
        const bool active = g < nrows;
        const int k = first + (active ? g : 0);
        double sx = 0.0, sy = 0.0;
        if (active) {
            const int64_t e1 = rowptr[k + 1];
#pragma unroll 4
            for (int64_t e = rowptr[k] + lane; e < e1; e += lanes) {
                const cplx a = val[e], v = x[col[e]];
                sx += a.x * v.x - a.y * v.y; sy += a.x * v.y + a.y * v.x;
            }
        }
        const int wl = lanes < 64 ? lanes : 64;                            // lanes of this row inside one wave
        for (int o = wl >> 1; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); }
        if (lanes > 64) {                                                  // several waves per row: fold through LDS
            if ((tid & 63) == 0) { psx[wave] = sx; psy[wave] = sy; }
            __syncthreads();
            if (lane == 0) {
                const int nw = lanes >> 6;
                sx = 0.0; sy = 0.0;
                for (int w = 0; w < nw; ++w) { sx += psx[wave + w]; sy += psy[wave + w]; }
            }
        }
        if (dofs < 0) {                                                    // independent rows
            if (active && lane == 0) {
                const int row = rowid[k];
                const cplx r = make_double2(b[row].x - sx, b[row].y - sy), d = invd[k];
                x[row] = make_double2(r.x * d.x - r.y * d.y, r.x * d.y + r.y * d.x);
            }
        } else {                                                           // rows coupled through the dense block
            if (active && lane == 0) { const int row = rowid[k]; rx[g] = b[row].x - sx; ry[g] = b[row].y - sy; }
            __syncthreads();
            if (wave == 0) dense_resolve(tid, nrows, first, dofs, tid < nrows ? rx[tid] : 0.0, tid < nrows ? ry[tid] : 0.0, dense, rowid, invd, x);
        }
        __syncthreads();                          // this step's rows are visible to the next step (same workgroup)
    }
}
// (tried: loading the descriptor and first matrix entry of the next step while the current one is reduced -- slower:
// the walk is bound by the barrier / reduction chain, not by those loads)

// A HEAVY group (<= 16 coupled rows with more entries than one CU should stream) as two launches:
//   k_trsv_group_ext   one 512-thread workgroup per row sums the row's external entries (many CUs, whole rows in flight)
//   k_trsv_group_dense one wave resolves the dense coupling block from those sums and writes x
__global__ __launch_bounds__(512)
void k_trsv_group_ext(int first, const int64_t* __restrict__ rowptr, const int* __restrict__ col, const cplx* __restrict__ val,
                      const int* __restrict__ rowid, const cplx* __restrict__ b, const cplx* __restrict__ x, cplx* __restrict__ gacc) {
    __shared__ double psx[8], psy[8];
    const int k = first + blockIdx.x, tid = threadIdx.x;
    double sx = 0.0, sy = 0.0;
    const int64_t e1 = rowptr[k + 1];
#pragma unroll 4
    for (int64_t e = rowptr[k] + tid; e < e1; e += 512) {
        const cplx a = val[e], v = x[col[e]];
        sx += a.x * v.x - a.y * v.y; sy += a.x * v.y + a.y * v.x;
    }
    for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); }
    if ((tid & 63) == 0) { psx[tid >> 6] = sx; psy[tid >> 6] = sy; }
    __syncthreads();
    if (tid == 0) {
        sx = 0.0; sy = 0.0;
        for (int w = 0; w < 8; ++w) { sx += psx[w]; sy += psy[w]; }
        const int row = rowid[k];
        gacc[blockIdx.x] = make_double2(b[row].x - sx, b[row].y - sy);
    }
}

__global__ __launch_bounds__(64)
void k_trsv_group_dense(int first, int nrows, int dofs, const cplx* __restrict__ dense, const int* __restrict__ rowid,
                        const cplx* __restrict__ invd, const cplx* __restrict__ gacc, cplx* x) {
    const int i = threadIdx.x;
    const cplx a = i < nrows ? gacc[i] : make_double2(0.0, 0.0);
    dense_resolve(i, nrows, first, dofs, a.x, a.y, dense, rowid, invd, x);
}

// A dense RUN: a long sequence of consecutive thin levels (the trailing separator block of a sparse LU factor is a
// chain of one-row levels, each row coupled to almost all rows before it).  Its R rows are solved as a dense
// lower-triangular system in blocks of 64, right-looking:
//   k_run_ext     acc_i = b_i - (entries outside the run), one workgroup per row                       1 launch
//   k_run_diag    one wave solves the 64 x 64 diagonal block J from acc (block staged in LDS)          R / 64 launches
//   k_run_update  acc_i -= D[i, block J] x_J for the rows below, one wave per row, many CUs            R / 64 - 1 launches
// The coupling D is stored dense (R x R, row-major), taken out of the CSR rows at set-up.
static constexpr int RUN_BLOCK = 64;

__global__ __launch_bounds__(256)
void k_run_ext(int first, const int64_t* __restrict__ rowptr, const int* __restrict__ col, const cplx* __restrict__ val,
               const int* __restrict__ rowid, const cplx* __restrict__ b, const cplx* __restrict__ x, cplx* __restrict__ racc) {
    __shared__ double psx[4], psy[4];
    const int k = first + blockIdx.x, tid = threadIdx.x;
    double sx = 0.0, sy = 0.0;
    const int64_t e1 = rowptr[k + 1];
#pragma unroll 4
    for (int64_t e = rowptr[k] + tid; e < e1; e += 256) {
        const cplx a = val[e], v = x[col[e]];
        sx += a.x * v.x - a.y * v.y; sy += a.x * v.y + a.y * v.x;
    }
    for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); }
    if ((tid & 63) == 0) { psx[tid >> 6] = sx; psy[tid >> 6] = sy; }
    __syncthreads();
    if (tid == 0) {
        const int row = rowid[k];
        racc[blockIdx.x] = make_double2(b[row].x - (psx[0] + psx[1] + psx[2] + psx[3]), b[row].y - (psy[0] + psy[1] + psy[2] + psy[3]));
    }
}

__global__ __launch_bounds__(64)
void k_run_diag(int first, int R, int J, const cplx* __restrict__ D, const int* __restrict__ rowid, const cplx* __restrict__ invd,
                const cplx* __restrict__ racc, cplx* __restrict__ rx, cplx* x) {
    __shared__ cplx blk[RUN_BLOCK][RUN_BLOCK + 1];
    const int lane = threadIdx.x, r0 = J * RUN_BLOCK;
    const int nb = R - r0 < RUN_BLOCK ? R - r0 : RUN_BLOCK;
    for (int r = 0; r < nb; ++r) if (lane < nb) blk[r][lane] = D[(int64_t)(r0 + r) * R + r0 + lane];   // row r, coalesced
    __syncthreads();
    const bool mine = lane < nb;
    cplx acc = mine ? racc[r0 + lane] : make_double2(0.0, 0.0);
    const cplx di = mine ? invd[first + r0 + lane] : make_double2(0.0, 0.0);
    double myx = 0.0, myy = 0.0;
    for (int j = 0; j < nb; ++j) {
        const double tx = acc.x * di.x - acc.y * di.y, ty = acc.x * di.y + acc.y * di.x;                  // x_j on lane j
        const double xjx = __shfl(tx, j, 64), xjy = __shfl(ty, j, 64);
        if (lane == j) { myx = xjx; myy = xjy; }
        if (mine && lane > j) { const cplx d = blk[lane][j]; acc.x -= d.x * xjx - d.y * xjy; acc.y -= d.x * xjy + d.y * xjx; }
    }
    if (mine) { const cplx v = make_double2(myx, myy); rx[r0 + lane] = v; x[rowid[first + r0 + lane]] = v; }
}

__global__ __launch_bounds__(256)
void k_run_update(int R, int J, const cplx* __restrict__ D, const cplx* __restrict__ rx, cplx* __restrict__ racc) {
    const int lane = threadIdx.x & 63, r0 = J * RUN_BLOCK;
    const int i = r0 + RUN_BLOCK + (int)blockIdx.x * 4 + (threadIdx.x >> 6);          // a wave per row below block J
    if (i >= R) return;                                                               // (whole waves leave together)
    const cplx d = D[(int64_t)i * R + r0 + lane], v = rx[r0 + lane];                  // block J is a full block here
    double sx = d.x * v.x - d.y * v.y, sy = d.x * v.y + d.y * v.x;
    for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); }
    if (lane == 0) { cplx a = racc[i]; a.x -= sx; a.y -= sy; racc[i] = a; }
}

__global__ void k_scatter(const cplx* __restrict__ w, const int* __restrict__ dst, cplx* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[dst[i]] = w[i];
}

// ---- host side -------------------------------------------------------------------------------------------------------

struct TriFactor {                 // one triangular factor, rows in level order
    DevBuf<int64_t> rowptr; DevBuf<int> col; DevBuf<cplx> val; DevBuf<int> rowid; DevBuf<cplx> invd; DevBuf<int> lvlptr;
    std::vector<int> h_lvlptr;     // host copy of the level boundaries
    int nlevels = 0;
    // launch schedule: kind 0 one level [l0, l0 + 1) as its own launch (lpr lanes per row); 1 a run of groups [l0, l1) walked by
    // one workgroup; 2 the heavy group l0 as two launches; 3 the dense run l0
    struct Seg { int kind; int l0, l1; int lpr; };
    std::vector<Seg> segs;
    DevBuf<int> grpptr, grpcnt, grpdense; DevBuf<cplx> dense;            // groups of the chain segments
    std::vector<int> h_grpptr, h_grpcnt, h_grpdense;
    DevBuf<cplx> gacc;                                                   // b - external sums of a heavy group
    struct Run { int first, R; int64_t dofs; };                         // dense runs: first sorted row, rows, offset of the R x R block
    std::vector<Run> runs;
    DevBuf<cplx> rdense, racc, rx;
    int ngroups = 0;
};

static int narrow_rows() {               // levels with at most this many rows (default 32: 42 -> 37 ms at N = 263 169 against
                                         // 128; LSFC_PRECOND_NARROW, 8..128) ...
    static const int v = [] { const char* e = getenv("LSFC_PRECOND_NARROW"); int n = e ? atoi(e) : 32; return n < 8 ? 8 : (n > 128 ? 128 : n); }();
    return v;
}
#define NARROW narrow_rows()
static constexpr int64_t CHAIN_NNZ = 8192; // ... and at most this many entries are walked inside a single workgroup
static constexpr int RUN_THIN = 4;        // dense runs: consecutive levels of at most this many rows ...
static constexpr int RUN_MIN = 96;        // ... totalling at least this many rows, cut into pieces of at most
static constexpr int RUN_MAX = 1024;      // this many rows (R x R dense block: 16 MB)

static void build_factor(TriFactor& F, int64_t N, const int64_t* rowptr, const int64_t* col, const double* val, bool lower) {
    // levels: depth of each row in the dependency graph of the triangular solve
    std::vector<int> level((size_t)N, 0);
    std::vector<std::complex<double>> diag((size_t)N, std::complex<double>(0, 0));
    int nlev = 0;
    auto scan_row = [&](int64_t r) {
        int lv = 0; bool have_diag = false;
        for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            const int64_t c = col[e];
            LSFC_REQUIRE(c >= 0 && c < N, "preconditioner factor: column index out of range");
            if (c == r) { diag[(size_t)r] = std::complex<double>(val[2 * e], val[2 * e + 1]); have_diag = true; continue; }
            LSFC_REQUIRE(lower ? c < r : c > r, "preconditioner factor: %s factor has an entry on the wrong side of the diagonal (row %lld, col %lld)",
                         lower ? "L" : "U", (long long)r, (long long)c);
            lv = std::max(lv, level[(size_t)c] + 1);
        }
        LSFC_REQUIRE(have_diag && diag[(size_t)r] != std::complex<double>(0, 0), "preconditioner factor: zero or missing diagonal in row %lld", (long long)r);
        level[(size_t)r] = lv; nlev = std::max(nlev, lv + 1);
    };
    if (lower) for (int64_t r = 0; r < N; ++r) scan_row(r); else for (int64_t r = N - 1; r >= 0; --r) scan_row(r);
    // counting sort of the rows by level
    std::vector<int> lvlptr((size_t)nlev + 1, 0);
    for (int64_t r = 0; r < N; ++r) ++lvlptr[(size_t)level[(size_t)r] + 1];
    for (int l = 0; l < nlev; ++l) lvlptr[(size_t)l + 1] += lvlptr[(size_t)l];
    std::vector<int> order((size_t)N), fill(lvlptr.begin(), lvlptr.end() - 1);
    for (int64_t r = 0; r < N; ++r) order[(size_t)fill[(size_t)level[(size_t)r]]++] = (int)r;
    // schedule: a wide level is one launch; a run of consecutive narrow levels is one single-workgroup launch that walks
    // GROUPS: consecutive thin levels with at most GROUP_ROWS rows together, coupled through a dense block
    std::vector<int> grpptr, grpcnt, grpdense;      // group -> first sorted row, rows, dense offset (-1: a single level)
    std::vector<int> group_of((size_t)N, -1);       // sorted row -> group, for the rows of multi-level groups only
    int64_t ndense = 0;
    // a level is walked inside the single-workgroup chain only if it is narrow AND light (one CU streams it); a level
    // with many rows or many entries gets its own launch over many CUs, a wave per row when the rows are long
    std::vector<int64_t> lvlnnz((size_t)nlev, 0);
    for (int64_t r = 0; r < N; ++r) lvlnnz[(size_t)level[(size_t)r]] += rowptr[r + 1] - rowptr[r] - 1;
    // dense runs: long sequences of consecutive thin levels (trailing separator blocks), cut into pieces of <= RUN_MAX rows
    std::vector<int> run_of_level((size_t)nlev, -1), run_of((size_t)N, -1);
    int64_t nrdense = 0;
    for (int l = 0; l < nlev;) {
        int e = l, rows = 0;
        while (e < nlev && lvlptr[(size_t)e + 1] - lvlptr[(size_t)e] <= RUN_THIN) { rows += lvlptr[(size_t)e + 1] - lvlptr[(size_t)e]; ++e; }
        if (rows < RUN_MIN) { l = e > l ? e : l + 1; continue; }
        while (l < e) {                             // pieces end at level boundaries
            int pe = l, prow = 0;
            while (pe < e && prow + (lvlptr[(size_t)pe + 1] - lvlptr[(size_t)pe]) <= RUN_MAX) { prow += lvlptr[(size_t)pe + 1] - lvlptr[(size_t)pe]; ++pe; }
            const int ri = (int)F.runs.size();
            F.runs.push_back({lvlptr[(size_t)l], prow, nrdense});
            nrdense += (int64_t)prow * prow;
            for (int q = l; q < pe; ++q) run_of_level[(size_t)q] = ri;
            for (int k = lvlptr[(size_t)l]; k < lvlptr[(size_t)pe]; ++k) run_of[(size_t)k] = ri;
            l = pe;
        }
    }
    auto chainable = [&](int l) { return run_of_level[(size_t)l] < 0 && lvlptr[(size_t)l + 1] - lvlptr[(size_t)l] <= NARROW && lvlnnz[(size_t)l] <= CHAIN_NNZ; };
    for (int l = 0; l < nlev;) {
        const int rows = lvlptr[(size_t)l + 1] - lvlptr[(size_t)l];
        if (run_of_level[(size_t)l] >= 0) {         // a dense run: its own sequence of launches
            const int ri = run_of_level[(size_t)l];
            F.segs.push_back({3, ri, ri + 1, 0});
            while (l < nlev && run_of_level[(size_t)l] == ri) ++l;
            continue;
        }
        if (!chainable(l)) {                        // lanes per row ~ entries per row
            const int64_t avg = lvlnnz[(size_t)l] / (rows > 0 ? rows : 1);
            F.segs.push_back({0, l, l + 1, avg >= 48 ? 64 : (avg >= 24 ? 32 : (avg >= 12 ? 16 : 8))}); ++l; continue;
        }
        int gfirst = (int)grpptr.size();
        while (l < nlev && chainable(l)) {
            int e = l + 1, tot = lvlptr[(size_t)l + 1] - lvlptr[(size_t)l];
            int64_t gnnz = lvlnnz[(size_t)l];
            while (e < nlev && chainable(e) && tot + (lvlptr[(size_t)e + 1] - lvlptr[(size_t)e]) <= GROUP_ROWS) {
                tot += lvlptr[(size_t)e + 1] - lvlptr[(size_t)e]; gnnz += lvlnnz[(size_t)e]; ++e;
            }
            grpptr.push_back(lvlptr[(size_t)l]); grpcnt.push_back(tot);
            if (e > l + 1) {                        // several levels: rows coupled through a dense block
                for (int k = lvlptr[(size_t)l]; k < lvlptr[(size_t)e]; ++k) group_of[(size_t)k] = (int)grpdense.size();
                grpdense.push_back((int)ndense); ndense += GROUP_ROWS * GROUP_ROWS;
                if (gnnz > CHAIN_NNZ) {             // too much for one CU: its own pair of launches, the walk resumes after it
                    const int gi = (int)grpptr.size() - 1;
                    if (gi > gfirst) F.segs.push_back({1, gfirst, gi, 0});
                    F.segs.push_back({2, gi, gi + 1, 0});
                    gfirst = gi + 1;
                }
            } else grpdense.push_back(-1);
            l = e;
        }
        if ((int)grpptr.size() > gfirst) F.segs.push_back({1, gfirst, (int)grpptr.size(), 0});
    }
    LSFC_REQUIRE(ndense < ((int64_t)1 << 31), "preconditioner factor: too many dense blocks");
    // sorted CSR without the diagonal; entries coupling two rows of the same group go to that group's dense block
    std::vector<int> sorted_of((size_t)N);
    for (int64_t k = 0; k < N; ++k) sorted_of[(size_t)order[(size_t)k]] = (int)k;
    std::vector<cplx> dense((size_t)ndense, make_double2(0.0, 0.0)), rdense((size_t)nrdense, make_double2(0.0, 0.0));
    std::vector<int64_t> rp((size_t)N + 1, 0);
    std::vector<int> cc; std::vector<cplx> vv; std::vector<cplx> invd((size_t)N);
    cc.reserve((size_t)(rowptr[N])); vv.reserve((size_t)(rowptr[N]));
    for (int64_t k = 0; k < N; ++k) {
        const int64_t r = order[(size_t)k];
        const int grp = group_of[(size_t)k];
        for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            if (col[e] == r) continue;
            const int kc = sorted_of[(size_t)col[e]];
            if (run_of[(size_t)k] >= 0 && run_of[(size_t)kc] == run_of[(size_t)k]) {
                const TriFactor::Run& rn = F.runs[(size_t)run_of[(size_t)k]];
                const int i = (int)k - rn.first, j = kc - rn.first;                             // local indices, j < i
                LSFC_REQUIRE(j >= 0 && j < i && i < rn.R, "internal: run block index (%d, %d)", i, j);
                rdense[(size_t)(rn.dofs + (int64_t)i * rn.R + j)] = make_double2(val[2 * e], val[2 * e + 1]);
                continue;
            }
            if (grp >= 0 && group_of[(size_t)kc] == grp) {
                const int i = (int)k - grpptr[(size_t)grp], j = kc - grpptr[(size_t)grp];       // local indices, j < i
                LSFC_REQUIRE(j >= 0 && j < i && i < GROUP_ROWS, "internal: dense block index (%d, %d)", i, j);
                dense[(size_t)grpdense[(size_t)grp] + (size_t)i * GROUP_ROWS + (size_t)j] = make_double2(val[2 * e], val[2 * e + 1]);
                continue;
            }
            cc.push_back((int)col[e]); vv.push_back(make_double2(val[2 * e], val[2 * e + 1]));
        }
        rp[(size_t)k + 1] = (int64_t)cc.size();
        const std::complex<double> id = 1.0 / diag[(size_t)r];
        invd[(size_t)k] = make_double2(id.real(), id.imag());
    }
    auto up = [](auto& dev, const auto& host) {
        dev.alloc(host.size());
        if (!host.empty()) LSFC_HIP(hipMemcpy(dev.p, host.data(), host.size() * sizeof(host[0]), hipMemcpyHostToDevice));
    };
    up(F.rowptr, rp); up(F.col, cc); up(F.val, vv); up(F.rowid, order); up(F.invd, invd); up(F.lvlptr, lvlptr);
    up(F.grpptr, grpptr); up(F.grpcnt, grpcnt); up(F.grpdense, grpdense); up(F.dense, dense);
    F.h_grpptr = grpptr; F.h_grpcnt = grpcnt; F.h_grpdense = grpdense;
    F.gacc.alloc(GROUP_ROWS);
    up(F.rdense, rdense); F.racc.alloc(RUN_MAX); F.rx.alloc(RUN_MAX);
    F.h_lvlptr = lvlptr; F.nlevels = nlev; F.ngroups = (int)grpdense.size();
}

static void launch_factor(const TriFactor& F, const cplx* b, cplx* x, hipStream_t st) {
    for (const auto& s : F.segs) {
        if (s.kind == 1) {
            hipLaunchKernelGGL(k_trsv_chain, dim3(1), dim3(1024), 0, st, s.l0, s.l1, F.grpptr.p, F.grpcnt.p, F.grpdense.p, F.dense.p,
                               F.rowptr.p, F.col.p, F.val.p, F.rowid.p, F.invd.p, b, x);
        } else if (s.kind == 3) {
            const TriFactor::Run& r = F.runs[(size_t)s.l0];
            const cplx* D = F.rdense.p + r.dofs;
            hipLaunchKernelGGL(k_run_ext, dim3((unsigned)r.R), dim3(256), 0, st, r.first, F.rowptr.p, F.col.p, F.val.p, F.rowid.p, b, x, F.racc.p);
            for (int J = 0; J * RUN_BLOCK < r.R; ++J) {
                hipLaunchKernelGGL(k_run_diag, dim3(1), dim3(64), 0, st, r.first, r.R, J, D, F.rowid.p, F.invd.p, F.racc.p, F.rx.p, x);
                const int rem = r.R - (J + 1) * RUN_BLOCK;
                if (rem > 0) hipLaunchKernelGGL(k_run_update, dim3((unsigned)((rem + 3) / 4)), dim3(256), 0, st, r.R, J, D, F.rx.p, F.racc.p);
            }
        } else if (s.kind == 2) {
            const int first = F.h_grpptr[(size_t)s.l0], nrows = F.h_grpcnt[(size_t)s.l0], dofs = F.h_grpdense[(size_t)s.l0];
            hipLaunchKernelGGL(k_trsv_group_ext, dim3((unsigned)nrows), dim3(512), 0, st, first, F.rowptr.p, F.col.p, F.val.p, F.rowid.p, b, x, F.gacc.p);
            hipLaunchKernelGGL(k_trsv_group_dense, dim3(1), dim3(64), 0, st, first, nrows, dofs, F.dense.p, F.rowid.p, F.invd.p, F.gacc.p, x);
        } else {
            const int first = F.h_lvlptr[(size_t)s.l0], nrows = F.h_lvlptr[(size_t)s.l0 + 1] - first;
#define LSFC_LEVEL(LPR) hipLaunchKernelGGL(k_trsv_level<LPR>, dim3((unsigned)(((int64_t)nrows * LPR + 255) / 256)), dim3(256), 0, st, first, nrows, \
                                          F.rowptr.p, F.col.p, F.val.p, F.rowid.p, F.invd.p, b, x)
            switch (s.lpr) { case 64: LSFC_LEVEL(64); break; case 32: LSFC_LEVEL(32); break; case 16: LSFC_LEVEL(16); break; default: LSFC_LEVEL(8); }
#undef LSFC_LEVEL
        }
    }
}

} // namespace lsfc

struct lsfc_precond {
    int device = 0;
    int64_t N = 0;
    hipStream_t stream = nullptr;
    lsfc::DevBuf<int64_t> a_rowptr; lsfc::DevBuf<int> a_col; lsfc::DevBuf<lsfc::cplx> a_val;     // As, CSR
    lsfc::DevBuf<int> rgather, cscatter; lsfc::DevBuf<double> rscale;
    lsfc::TriFactor L, U;
    lsfc::DevBuf<lsfc::cplx> vin, y0, z, w, vout, hstage;
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; hipStream_t captured_on = nullptr;
    int launches = 0;
    ~lsfc_precond() { if (exec) (void)hipGraphExecDestroy(exec); if (graph) (void)hipGraphDestroy(graph); }
};

namespace lsfc {

static void enqueue_all(lsfc_precond* pc, hipStream_t st) {
    const int N = (int)pc->N;
    hipLaunchKernelGGL(k_spmv_gather<8>, dim3((unsigned)(((int64_t)N * 8 + 255) / 256)), dim3(256), 0, st, pc->a_rowptr.p, pc->a_col.p, pc->a_val.p,
                       pc->rgather.p, pc->rscale.p, pc->vin.p, pc->y0.p, N);
    launch_factor(pc->L, pc->y0.p, pc->z.p, st);
    launch_factor(pc->U, pc->z.p, pc->w.p, st);
    hipLaunchKernelGGL(k_scatter, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, pc->w.p, pc->cscatter.p, pc->vout.p, N);
}

// v (device, N complex) <- Msp^{-1} (As v), stream-ordered on `st`
static void precond_apply_dev(lsfc_precond* pc, cplx* v, hipStream_t st) {
    const size_t bytes = (size_t)pc->N * sizeof(cplx);
    LSFC_HIP(hipMemcpyAsync(pc->vin.p, v, bytes, hipMemcpyDeviceToDevice, st));
    // LSFC_PRECOND_GRAPH=0 (developer switch): plain stream launches instead of the captured graph
    static const bool no_graph = getenv("LSFC_PRECOND_GRAPH") && getenv("LSFC_PRECOND_GRAPH")[0] == '0';
    if (no_graph) {
        enqueue_all(pc, st);
        LSFC_HIP(hipGetLastError());
        LSFC_HIP(hipMemcpyAsync(v, pc->vout.p, bytes, hipMemcpyDeviceToDevice, st));
        return;
    }
    if (!pc->exec) {
        // capture the fixed sequence once (internal buffers only), then replay it
        hipStream_t cs; LSFC_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            enqueue_all(pc, cs);
            e = hipStreamEndCapture(cs, &pc->graph);
        }
        if (e == hipSuccess) e = hipGraphInstantiate(&pc->exec, pc->graph, nullptr, nullptr, 0);
        (void)hipStreamDestroy(cs);
        if (e != hipSuccess) { (void)hipGetLastError(); fail(LSFC_EHIP, "preconditioner: graph capture failed: %s", hipGetErrorString(e)); }
    }
    LSFC_HIP(hipGraphLaunch(pc->exec, st));
    LSFC_HIP(hipMemcpyAsync(v, pc->vout.p, bytes, hipMemcpyDeviceToDevice, st));
}

// loads this translation unit's code object on the current device (pruned.hip: pruned_warmup -- every code object of the library is
// resident before the first transfer or pass of a process exists; DESIGN 3, "The round-2 first-apply GPU fault")
__global__ void k_warmup_precond(int* p) { if (p) *p = 0; }
void warmup_precond() {
    hipLaunchKernelGGL(k_warmup_precond, dim3(1), dim3(64), 0, 0, (int*)nullptr);
    LSFC_HIP(hipGetLastError());
}

} // namespace lsfc

using namespace lsfc;

extern "C" {

int lsfc_precond_create(lsfc_precond** out, int64_t N,
                        const int64_t* As_rowptr, const int64_t* As_col, const double* As_val,
                        const int64_t* L_rowptr, const int64_t* L_col, const double* L_val,
                        const int64_t* U_rowptr, const int64_t* U_col, const double* U_val,
                        const int64_t* row_gather, const int64_t* col_scatter, const double* row_scale, int device) {
    return guarded([&] {
        LSFC_REQUIRE(out, "NULL argument"); *out = nullptr;
        LSFC_REQUIRE(N >= 1 && N < ((int64_t)1 << 31), "preconditioner: N out of range");
        LSFC_REQUIRE(As_rowptr && As_col && As_val && L_rowptr && L_col && L_val && U_rowptr && U_col && U_val, "NULL argument");
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) fail(LSFC_ENODEV, "no HIP device available: the preconditioner apply has no CPU fallback");
        LSFC_REQUIRE(device >= 0 && device < count, "device %d out of range (have %d)", device, count);
        LSFC_HIP(hipSetDevice(device));
        pruned_warmup(device);                             // every code object of the library resident before any work of this process is queued
        std::unique_ptr<lsfc_precond> pc(new lsfc_precond());
        pc->device = device; pc->N = N;
        // As: CSR, 32-bit columns on the device
        {
            const int64_t nnz = As_rowptr[N];
            std::vector<int> c((size_t)nnz);
            for (int64_t e = 0; e < nnz; ++e) { LSFC_REQUIRE(As_col[e] >= 0 && As_col[e] < N, "As: column index out of range"); c[(size_t)e] = (int)As_col[e]; }
            pc->a_rowptr.alloc((size_t)N + 1); pc->a_col.alloc((size_t)nnz); pc->a_val.alloc((size_t)nnz);
            LSFC_HIP(hipMemcpy(pc->a_rowptr.p, As_rowptr, ((size_t)N + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
            if (nnz) {
                LSFC_HIP(hipMemcpy(pc->a_col.p, c.data(), (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
                LSFC_HIP(hipMemcpy(pc->a_val.p, As_val, (size_t)nnz * sizeof(cplx), hipMemcpyHostToDevice));
            }
        }
        auto perm_up = [&](DevBuf<int>& dev, const int64_t* p, const char* what) {
            std::vector<int> h((size_t)N); std::vector<char> seen((size_t)N, 0);
            for (int64_t i = 0; i < N; ++i) {
                const int64_t v = p ? p[i] : i;
                LSFC_REQUIRE(v >= 0 && v < N && !seen[(size_t)v], "preconditioner: %s is not a permutation", what);
                seen[(size_t)v] = 1; h[(size_t)i] = (int)v;
            }
            dev.alloc((size_t)N);
            LSFC_HIP(hipMemcpy(dev.p, h.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice));
        };
        perm_up(pc->rgather, row_gather, "row_gather");
        perm_up(pc->cscatter, col_scatter, "col_scatter");
        if (row_scale) { pc->rscale.alloc((size_t)N); LSFC_HIP(hipMemcpy(pc->rscale.p, row_scale, (size_t)N * sizeof(double), hipMemcpyHostToDevice)); }
        build_factor(pc->L, N, L_rowptr, L_col, L_val, true);
        build_factor(pc->U, N, U_rowptr, U_col, U_val, false);
        for (DevBuf<cplx>* b : { &pc->vin, &pc->y0, &pc->z, &pc->w, &pc->vout }) { b->alloc((size_t)N); LSFC_HIP(hipMemset(b->p, 0, b->bytes())); }
        pc->launches = 2;
        for (const TriFactor* F : { &pc->L, &pc->U }) for (const auto& sg : F->segs) {
            if (sg.kind == 3) { const int nb = (F->runs[(size_t)sg.l0].R + RUN_BLOCK - 1) / RUN_BLOCK; pc->launches += 2 * nb; }
            else pc->launches += sg.kind == 2 ? 2 : 1;
        }
        *out = pc.release();
    });
}

int lsfc_precond_destroy(lsfc_precond* pc) {
    return guarded([&] { if (pc) { (void)hipSetDevice(pc->device); (void)hipDeviceSynchronize(); delete pc; } });
}

int lsfc_precond_set_stream(lsfc_precond* pc, void* stream) {
    return guarded([&] { LSFC_REQUIRE(pc, "NULL preconditioner"); pc->stream = (hipStream_t)stream; });
}

int lsfc_precond_apply(lsfc_precond* pc, double* v, int memspace) {
    return guarded([&] {
        LSFC_REQUIRE(pc && v, "NULL argument");
        LSFC_REQUIRE(memspace == LSFC_MEM_HOST || memspace == LSFC_MEM_DEVICE, "bad memspace %d", memspace);
        LSFC_HIP(hipSetDevice(pc->device));
        if (memspace == LSFC_MEM_DEVICE) { precond_apply_dev(pc, (cplx*)v, pc->stream); return; }
        if (pc->hstage.n < (size_t)pc->N) pc->hstage.alloc((size_t)pc->N);
        const size_t bytes = (size_t)pc->N * sizeof(cplx);
        LSFC_HIP(hipMemcpyAsync(pc->hstage.p, v, bytes, hipMemcpyHostToDevice, pc->stream));
        precond_apply_dev(pc, pc->hstage.p, pc->stream);
        LSFC_HIP(hipMemcpyAsync(v, pc->hstage.p, bytes, hipMemcpyDeviceToHost, pc->stream));
        LSFC_HIP(hipStreamSynchronize(pc->stream));
    });
}

/* lsfc_precond_fn-compatible entry: opts.precond = lsfc_precond_callback, opts.precond_user = pc,
 * opts.precond_on_device = 1 (v is the device-resident Krylov vector; work is enqueued on pc's stream). */
int lsfc_precond_callback(void* user, double* v, int64_t n) {
    lsfc_precond* pc = (lsfc_precond*)user;
    if (!pc || n != pc->N) { set_last_error("preconditioner callback: size mismatch (%lld vs %lld)", (long long)n, pc ? (long long)pc->N : -1LL); return 1; }
    return lsfc_precond_apply(pc, v, LSFC_MEM_DEVICE) == LSFC_OK ? 0 : 1;
}

int lsfc_precond_stats(const lsfc_precond* pc, int64_t* levels_L, int64_t* levels_U, int64_t* launches) {
    return guarded([&] {
        LSFC_REQUIRE(pc, "NULL preconditioner");
        if (levels_L) *levels_L = pc->L.nlevels;
        if (levels_U) *levels_U = pc->U.nlevels;
        if (launches) *launches = pc->launches;
    });
}

} // extern "C"
