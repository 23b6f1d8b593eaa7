// fit.hip -- create_anomaly_detector (CAE_improved_modeltrain.py:394-446) on gfx950: the FIT of what
// detector.hip evaluates.
//
//   RobustScaler().fit          :408-409  sklearn/preprocessing/_data.py:1656-1677 (nanmedian, nanpercentile 25/75)
//   PCA(n_components).fit       :412-414  sklearn/decomposition/_pca.py (mean_, principal axes of the centred data)
//   OneClassSVM(rbf, nu).fit    :420-427  sklearn/svm/src/libsvm/svm.cpp:663-941 (Solver::Solve), :943-1036
//                                         (select_working_set), :1123-1160 (calculate_rho), :1710-1750 (solve_one_class)
//
// What runs where.  Everything proportional to the number of training cells runs here: the order statistics
// of every feature column (radix select, exact), the scaled/centred moments XᵀX (fp64 MFMA), the projection,
// and the whole SMO iteration (two launches per iteration, the chosen pair and the stopping test stay on the
// device; the host only polls a flag every few hundred iterations).  The host does the O(1)-per-feature
// arithmetic on the selected order statistics (numpy's lerp, restated in C below) and libsvm's sequential
// rho / objective sums over the final gradient, so those are bit-for-bit what numpy / libsvm produce.
//
// The SMO solver is libsvm's algorithm without shrinking and without the kernel cache: rows of Q are
// recomputed (N x D fp64 FMAs, the training set stays in the Infinity Cache) and rounded to float exactly
// where libsvm rounds them (Qfloat), the gradient is kept in double and updated with libsvm's expression,
// and both selections break ties the way libsvm's sequential `>=` / `<=` scans do (last index wins).
// Shrinking only removes variables that cannot be selected, so the iterates agree with libsvm's until
// rounding in the kernel values (BLAS ddot vs a sequential FMA chain, libm vs ocml exp) flips a float or a
// near-tie; the solution then still agrees to the solver's own stopping tolerance.
#include "api_internal.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <limits>

namespace cs {

typedef double f64x4 __attribute__((ext_vector_type(4)));

namespace {

// ================================================================== RobustScaler.fit
// keys: order-preserving map float -> uint32 (-0.0 folded onto +0.0, as numpy's sort treats them equal)
__device__ __forceinline__ unsigned f2key(float v)
{
    unsigned u = __float_as_uint(v);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// features [n][F] fp32 -> keysT [F][ld] (32x32 tiles through LDS); counts NaNs
__global__ __launch_bounds__(256) void keys_transpose_kernel(const float* __restrict__ x, long n, int F, long ld,
                                                             unsigned* __restrict__ keysT, int* __restrict__ nan_count)
{
    __shared__ unsigned tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    const long r0 = (long)blockIdx.x * 32;                             // rows on grid.x (no 65,535 limit), feature tiles on grid.y
    const int c0 = blockIdx.y * 32;
    int nans = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long r = r0 + ty + 8 * k;
        const int c = c0 + tx;
        unsigned key = 0u;
        if (r < n && c < F) {
            const float v = x[r * F + c];
            nans += (v != v);
            key = f2key(v);
        }
        tile[ty + 8 * k][tx] = key;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k;
        const long r = r0 + tx;
        if (r < n && c < F) keysT[(size_t)c * ld + r] = tile[tx][ty + 8 * k];
    }
    if (nans) atomicAdd(nan_count, nans);
}

constexpr int SEL_THREADS = 1024, SEL_T = 6;
struct SelRanks { unsigned r[SEL_T]; };

// One workgroup per feature column: MSB-first radix select (4 passes of 8 bits) of SEL_T ranks at once.
// Pass 0 shares one histogram; later passes keep one per target (targets may have left for different bins).
__global__ __launch_bounds__(SEL_THREADS) void column_select_kernel(const unsigned* __restrict__ keysT, long n, long ld,
                                                                    SelRanks ranks, float* __restrict__ out /* [F][SEL_T] */)
{
    __shared__ unsigned hist[SEL_T][256];
    __shared__ unsigned s_prefix[SEL_T], s_rank[SEL_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned* col = keysT + (size_t)blockIdx.x * ld;
    if (tid < SEL_T) { s_prefix[tid] = 0u; s_rank[tid] = ranks.r[tid]; }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int e = tid; e < SEL_T * 256; e += SEL_THREADS) (&hist[0][0])[e] = 0u;
        __syncthreads();
        unsigned pf[SEL_T];
#pragma unroll
        for (int t = 0; t < SEL_T; ++t) pf[t] = s_prefix[t];
        for (long e = tid; e < n; e += SEL_THREADS) {
            const unsigned key = col[e];
            const unsigned bin = (key >> shift) & 255u;
            if (pass == 0) {
                atomicAdd(&hist[0][bin], 1u);
            } else {
                const unsigned hi = key >> (shift + 8);
#pragma unroll
                for (int t = 0; t < SEL_T; ++t)
                    if (hi == pf[t]) atomicAdd(&hist[t][bin], 1u);
            }
        }
        __syncthreads();
        if (wave < SEL_T) {                                            // wave t walks target t's histogram
            const unsigned* h = hist[pass == 0 ? 0 : wave];
            const unsigned c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
            unsigned incl = c0 + c1 + c2 + c3;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            unsigned cum = incl - (c0 + c1 + c2 + c3);
            const unsigned rk = s_rank[wave];
            const unsigned cs_[4] = {c0, c1, c2, c3};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (rk >= cum && rk < cum + cs_[b]) {
                    s_prefix[wave] = (pf[wave] << 8) | (unsigned)(4 * lane + b);
                    s_rank[wave] = rk - cum;
                }
                cum += cs_[b];
            }
        }
        __syncthreads();
    }
    if (tid < SEL_T) out[(size_t)blockIdx.x * SEL_T + tid] = key2f(s_prefix[tid]);
}

// ================================================================== PCA moments
// xs = float32(double(float32(x - center)) / scale)   (RobustScaler.transform, _data.py:1715-1718)
__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ x, const float* __restrict__ center,
                                                    const double* __restrict__ scale, long total, int F, float* __restrict__ xs)
{
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int f = (int)(e % F);
        const float t = __fsub_rn(x[e], center[f]);
        xs[e] = (float)__ddiv_rn((double)t, scale[f]);
    }
}

// mean_[f]: float32 sum over the rows in order, then / n -- what np.mean(X, axis=0) does for a C-ordered
// float32 matrix (the reduction over the slow axis adds row after row)
__global__ __launch_bounds__(64) void column_mean_kernel(const float* __restrict__ xs, long n, int F, float* __restrict__ mean)
{
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= F) return;
    float s = 0.0f;
    long r = 0;
    for (; r + 8 <= n; r += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = xs[(r + k) * F + f];
#pragma unroll
        for (int k = 0; k < 8; ++k) s = __fadd_rn(s, v[k]);
    }
    for (; r < n; ++r) s = __fadd_rn(s, xs[r * F + f]);
    mean[f] = __fdiv_rn(s, (float)n);
}

__global__ __launch_bounds__(256) void center_kernel(float* __restrict__ xs, const float* __restrict__ mean, long total, int F)
{
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256)
        xs[e] = __fsub_rn(xs[e], mean[(int)(e % F)]);
}

// scatter = Xc^T Xc in fp64: v_mfma_f64_16x16x4_f64, A[row = l & 15][k = l >> 4] = xc[n + k][a0 + row],
// B[k][col = l & 15] = xc[n + k][b0 + col], D[row = (l >> 4) + 4 reg][col = l & 15].
// A workgroup (4 waves) owns a 128 x 128 tile of the upper triangle for one slice of the rows; a wave a
// 64 x 64 quarter (16 accumulator tiles = 128 VGPRs).  Operands come straight from global memory (a row's
// 64-float run per quarter): 8 loads per 16 MFMAs.
constexpr int COV_T = 128;
__global__ __launch_bounds__(256) void scatter_kernel(const float* __restrict__ xc, long n, int F, int ksplit,
                                                      double* __restrict__ part /* [ksplit][F][F] */)
{
    const int T = F / COV_T;
    // upper-triangular tile index -> (ta, tb), tb >= ta
    int t = blockIdx.x, ta = 0;
    while (t >= T - ta) { t -= T - ta; ++ta; }
    const int tb = ta + t;
    const int ks = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int a0 = ta * COV_T + (wave >> 1) * 64, b0 = tb * COV_T + (wave & 1) * 64;
    const long rows = (n + ksplit - 1) / ksplit;
    const long r0 = (long)ks * rows, r1 = n < r0 + rows ? n : r0 + rows;

    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

    auto load = [&](long r, float a[4], float b[4]) {
        const long row = r + kq;
        const bool ok = row < r1;
        const float* p = xc + (ok ? row : r0) * F;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float va = p[a0 + 16 * i + li], vb = p[b0 + 16 * i + li];
            a[i] = ok ? va : 0.0f;
            b[i] = ok ? vb : 0.0f;
        }
    };
    if (r0 < r1) {
        float a[4], b[4], na[4], nb[4];
        load(r0, a, b);
        for (long r = r0; r < r1; r += 4) {
            const long rn = r + 4 < r1 ? r + 4 : r0;
            load(rn, na, nb);                                   // prefetch; the wrap-around load of the last step is unused
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[i], (double)b[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = na[i]; b[i] = nb[i]; }
        }
    }
    double* o = part + (size_t)ks * F * F;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                o[(size_t)(a0 + 16 * i + kq + 4 * rg) * F + b0 + 16 * j + li] = acc[i][j][rg];
}

// scatter[a][b] = sum over slices (fixed order) for b's tile >= a's tile, mirrored below
__global__ __launch_bounds__(256) void scatter_merge_kernel(const double* __restrict__ part, int F, int ksplit, double* __restrict__ out)
{
    const long total = (long)F * F;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        int a = (int)(e / F), b = (int)(e % F);
        if (b / COV_T < a / COV_T) { const int t = a; a = b; b = t; }
        double s = 0.0;
        for (int k = 0; k < ksplit; ++k) s += part[(size_t)k * total + (size_t)a * F + b];
        out[e] = s;
    }
}

// ================================================================== one-class SVM: SMO
constexpr int SMO_DP = 128;                    // component capacity (the scoring path's limit too)
constexpr int SMO_B = 256;                     // training points per workgroup
constexpr double SMO_TAU = 1e-12;              // svm.cpp: #define TAU 1e-12
constexpr double SMO_INF = std::numeric_limits<double>::infinity();

struct SmoPickI { double v; int i; int pad; };
struct SmoPickJ { double obj; double g2; int j; int pad; };
struct SmoState {
    int done;            // 0 running, 1 converged (select_working_set returned 1), 2 max_iter reached
    int final_parity;    // which alpha/G buffer holds the solution
    int i;               // working-set member chosen by the row-i launch
    int pad;
    double gmax;
    long long n_iter;
};
struct SmoArgs {
    const void* xT;      // [D][ld], double -- or float when every training value is a float (PCA output is): same numbers, half the bytes
    const double* xsq;   // [n]
    long ld;
    int n, D, nb, pad;
    double gamma, eps;
    long long max_iter;
    double* alpha[2];
    double* G[2];
    float* Qi;
    SmoPickI* part_i;
    SmoPickJ* part_j;
    SmoState* state;
};

// x [n][D] row-major -> xT [D][ld]; xsq[j] = x_j . x_j
template <class XT>
__global__ __launch_bounds__(256) void smo_prepare_kernel(const double* __restrict__ x, int n, int D, long ld,
                                                          XT* __restrict__ xT, double* __restrict__ xsq)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double s = 0.0;
    for (int d = 0; d < D; ++d) {
        const double v = x[(size_t)j * D + d];
        xT[(size_t)d * ld + j] = (XT)v;
        s = fma(v, v, s);
    }
    xsq[j] = s;
}

// libsvm's Qfloat kernel value: (float) exp(-gamma * (x_square[i] + x_square[j] - 2 * dot))   svm.cpp:346-349
__device__ __forceinline__ float rbf_q(double gamma, double xsqi, double xsqj, double dot)
{
    return (float)exp(__dmul_rn(-gamma, __dsub_rn(__dadd_rn(xsqi, xsqj), __dmul_rn(2.0, dot))));
}

__device__ __forceinline__ bool later_max(double v, int i, double bv, int bi) { return v > bv || (v == bv && i > bi); }
__device__ __forceinline__ bool later_min(double v, int i, double bv, int bi) { return v < bv || (v == bv && i > bi); }

__device__ __forceinline__ SmoPickI reduce_pick_i(SmoPickI p, SmoPickI* sh /* [4] */)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double v = __shfl_xor(p.v, o);
        const int i = __shfl_xor(p.i, o);
        if (later_max(v, i, p.v, p.i)) { p.v = v; p.i = i; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = p;
    __syncthreads();
    p = sh[0];
#pragma unroll
    for (int w = 1; w < SMO_B / 64; ++w)
        if (later_max(sh[w].v, sh[w].i, p.v, p.i)) p = sh[w];
    return p;
}

__device__ __forceinline__ SmoPickJ reduce_pick_j(SmoPickJ p, SmoPickJ* sh /* [4] */)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double v = __shfl_xor(p.obj, o), g = __shfl_xor(p.g2, o);
        const int j = __shfl_xor(p.j, o);
        if (later_min(v, j, p.obj, p.j)) { p.obj = v; p.j = j; }
        p.g2 = fmax(p.g2, g);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = p;
    __syncthreads();
    p = sh[0];
#pragma unroll
    for (int w = 1; w < SMO_B / 64; ++w) {
        if (later_min(sh[w].obj, sh[w].j, p.obj, p.j)) { p.obj = sh[w].obj; p.j = sh[w].j; }
        p.g2 = fmax(p.g2, sh[w].g2);
    }
    return p;
}

// Initial gradient (svm.cpp:700-716): G[j] = sum over i with alpha_i > 0, in index order, of
// alpha_i * (double) Q_i[j].  solve_one_class puts the non-zero alphas on the first n0 points (:1725-1733).
// A thread owns point j with x_j in registers (SMO_DP doubles); x_i are staged through LDS 16 at a time and
// read as broadcasts.  Also emits the first row-i candidates.
constexpr int INIT_ROWS = 16;
template <class XT>
__global__ __launch_bounds__(SMO_B, 1) void smo_init_kernel(SmoArgs a, int n0)
{
    const XT* __restrict__ xT = (const XT*)a.xT;
    __shared__ double xi[INIT_ROWS][SMO_DP];
    __shared__ double xsqi[INIT_ROWS], ali[INIT_ROWS];
    __shared__ SmoPickI shp[SMO_B / 64];
    const int tid = threadIdx.x;
    const int j = blockIdx.x * SMO_B + tid;
    const int jj = j < a.n ? j : a.n - 1;
    double xj[SMO_DP];
#pragma unroll
    for (int d = 0; d < SMO_DP; ++d) xj[d] = d < a.D ? (double)xT[(size_t)d * a.ld + jj] : 0.0;
    const double xsqj = a.xsq[jj];
    const double* alpha = a.alpha[0];
    double g = 0.0;
    for (int i0 = 0; i0 < n0; i0 += INIT_ROWS) {
        __syncthreads();
        for (int e = tid; e < INIT_ROWS * SMO_DP; e += SMO_B) {
            const int r = e / SMO_DP, d = e % SMO_DP, i = i0 + r;
            xi[r][d] = (i < n0 && d < a.D) ? (double)xT[(size_t)d * a.ld + i] : 0.0;
        }
        if (tid < INIT_ROWS) {
            const int i = i0 + tid;
            xsqi[tid] = i < n0 ? a.xsq[i] : 0.0;
            ali[tid] = i < n0 ? alpha[i] : 0.0;
        }
        __syncthreads();
        const int nr = min(INIT_ROWS, n0 - i0);
        for (int r = 0; r < nr; ++r) {
            double dot = 0.0;
#pragma unroll
            for (int d = 0; d < SMO_DP; ++d) dot = fma(xj[d], xi[r][d], dot);
            const float q = rbf_q(a.gamma, xsqi[r], xsqj, dot);
            g = __dadd_rn(g, __dmul_rn(ali[r], (double)q));
        }
    }
    SmoPickI p{-SMO_INF, -1, 0};
    if (j < a.n) {
        a.G[0][j] = g;
        if (alpha[j] < 1.0) { p.v = -g; p.i = j; }
    }
    p = reduce_pick_i(p, shp);
    if (tid == 0) a.part_i[blockIdx.x] = p;
}

// Launch 1 of an iteration: i = argmax over {alpha_t < C} of -G_t (svm.cpp:957-974), row Q_i, and the
// second-order choice of j over {alpha_j > 0} (:981-1008) as per-workgroup partials.
template <class XT>
__global__ __launch_bounds__(SMO_B) void smo_row_i_kernel(SmoArgs a, int parity)
{
    const XT* __restrict__ xT = (const XT*)a.xT;
    __shared__ double xs[SMO_DP];
    __shared__ SmoPickI shi[SMO_B / 64];
    __shared__ SmoPickJ shj[SMO_B / 64];
    const int tid = threadIdx.x;
    if (a.state->done) return;
    if (a.max_iter >= 0 && a.state->n_iter >= a.max_iter) {           // svm.cpp:725-729
        if (blockIdx.x == 0 && tid == 0) { a.state->done = 2; a.state->final_parity = parity; }
        return;
    }
    const double* alpha = a.alpha[parity];
    const double* G = a.G[parity];
    SmoPickI pi{-SMO_INF, -1, 0};
    for (int b = tid; b < a.nb; b += SMO_B) {
        const SmoPickI q = a.part_i[b];
        if (later_max(q.v, q.i, pi.v, pi.i)) pi = q;
    }
    pi = reduce_pick_i(pi, shi);
    const int i = pi.i;
    const double gmax = pi.v;
    if (i < 0) {                                                       // every alpha at its upper bound: nothing to select
        if (blockIdx.x == 0 && tid == 0) { a.state->i = -1; a.state->gmax = gmax; }
        if (tid == 0) a.part_j[blockIdx.x] = SmoPickJ{SMO_INF, -SMO_INF, -1, 0};
        return;
    }
    if (tid < a.D) xs[tid] = (double)xT[(size_t)tid * a.ld + i];
    __syncthreads();
    const double xsqi = a.xsq[i];
    const int j = blockIdx.x * SMO_B + tid;
    SmoPickJ pj{SMO_INF, -SMO_INF, -1, 0};
    if (j < a.n) {
        double dot = 0.0;
        for (int d = 0; d < a.D; ++d) dot = fma((double)xT[(size_t)d * a.ld + j], xs[d], dot);
        const float q = rbf_q(a.gamma, xsqi, a.xsq[j], dot);
        a.Qi[j] = q;
        if (alpha[j] > 0.0) {                                          // !is_lower_bound(j)
            const double gj = G[j];
            pj.g2 = gj;
            const double gd = __dadd_rn(gmax, gj);
            if (gd > 0.0) {
                const double quad = __dsub_rn(2.0, __dmul_rn(2.0, (double)q));   // QD[i] + QD[j] - 2 y_i Q_ij, QD = 1
                const double num = __dmul_rn(gd, gd);
                pj.obj = -__ddiv_rn(num, quad > 0.0 ? quad : SMO_TAU);
                pj.j = j;
            }
        }
    }
    pj = reduce_pick_j(pj, shj);
    if (tid == 0) a.part_j[blockIdx.x] = pj;
    if (blockIdx.x == 0 && tid == 0) { a.state->i = i; a.state->gmax = gmax; }
}

// Launch 2: j from the partials, the stopping test (svm.cpp:1029-1030), the two-variable update (:806-845),
// row Q_j, G += Q_i d_alpha_i + Q_j d_alpha_j (:849-855), and the next iteration's row-i candidates.
template <class XT>
__global__ __launch_bounds__(SMO_B) void smo_update_kernel(SmoArgs a, int parity)
{
    const XT* __restrict__ xT = (const XT*)a.xT;
    __shared__ double xs[SMO_DP];
    __shared__ SmoPickI shi[SMO_B / 64];
    __shared__ SmoPickJ shj[SMO_B / 64];
    const int tid = threadIdx.x;
    if (a.state->done) return;
    const double* alpha = a.alpha[parity];
    const double* G = a.G[parity];
    SmoPickJ pj{SMO_INF, -SMO_INF, -1, 0};
    for (int b = tid; b < a.nb; b += SMO_B) {
        const SmoPickJ q = a.part_j[b];
        if (later_min(q.obj, q.j, pj.obj, pj.j)) { pj.obj = q.obj; pj.j = q.j; }
        pj.g2 = fmax(pj.g2, q.g2);
    }
    pj = reduce_pick_j(pj, shj);
    const int i = a.state->i, j = pj.j;
    const double gmax = a.state->gmax;
    if (i < 0 || j < 0 || __dadd_rn(gmax, pj.g2) < a.eps) {
        if (blockIdx.x == 0 && tid == 0) { a.state->done = 1; a.state->final_parity = parity; }
        return;
    }
    // y_i == y_j branch with C_i = C_j = 1
    double ai = alpha[i], aj = alpha[j];
    const double old_ai = ai, old_aj = aj;
    {
        const double qij = (double)a.Qi[j];
        double quad = __dsub_rn(2.0, __dmul_rn(2.0, qij));
        if (quad <= 0.0) quad = SMO_TAU;
        const double delta = __ddiv_rn(__dsub_rn(G[i], G[j]), quad);
        const double sum = __dadd_rn(ai, aj);
        ai = __dsub_rn(ai, delta);
        aj = __dadd_rn(aj, delta);
        if (sum > 1.0) {
            if (ai > 1.0) { ai = 1.0; aj = __dsub_rn(sum, 1.0); }
        } else {
            if (aj < 0.0) { aj = 0.0; ai = sum; }
        }
        if (sum > 1.0) {
            if (aj > 1.0) { aj = 1.0; ai = __dsub_rn(sum, 1.0); }
        } else {
            if (ai < 0.0) { ai = 0.0; aj = sum; }
        }
    }
    const double dai = __dsub_rn(ai, old_ai), daj = __dsub_rn(aj, old_aj);
    if (tid < a.D) xs[tid] = (double)xT[(size_t)tid * a.ld + j];
    __syncthreads();
    const double xsqj = a.xsq[j];
    const int k = blockIdx.x * SMO_B + tid;
    SmoPickI pi{-SMO_INF, -1, 0};
    if (k < a.n) {
        double dot = 0.0;
        for (int d = 0; d < a.D; ++d) dot = fma((double)xT[(size_t)d * a.ld + k], xs[d], dot);
        const float qj = rbf_q(a.gamma, xsqj, a.xsq[k], dot);
        const double gn = __dadd_rn(G[k], __dadd_rn(__dmul_rn((double)a.Qi[k], dai), __dmul_rn((double)qj, daj)));
        const double an = k == i ? ai : (k == j ? aj : alpha[k]);
        a.G[parity ^ 1][k] = gn;
        a.alpha[parity ^ 1][k] = an;
        if (an < 1.0) { pi.v = -gn; pi.i = k; }                       // !is_upper_bound(k)
    }
    pi = reduce_pick_i(pi, shi);
    if (tid == 0) a.part_i[blockIdx.x] = pi;
    if (blockIdx.x == 0 && tid == 0) a.state->n_iter += 1;
}

}  // namespace
}  // namespace cs

// ---- C ABI ----------------------------------------------------------------------------------
using namespace cs;

struct cs_fit {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DevBuf feat, keys, sel, nanc, cen, scl, xs, mean, part, scat, comps, mproj, proj;
    DevBuf x, xT, xsq, alpha0, alpha1, G0, G1, Qi, part_i, part_j, state;
    double last_ms = 0.0;
};

int cs_fit_create(int device_id, cs_fit** out)
{
    if (!out) return fail(CS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int rc = require_gfx950(device_id);
    if (rc) return rc;
    cs_fit* f = new cs_fit();
    f->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&f->ev0);
    if (e == hipSuccess) e = hipEventCreate(&f->ev1);
    if (e != hipSuccess) {
        delete f;
        return fail(CS_ERR_HIP, "stream/event creation failed: %s", hipGetErrorString(e));
    }
    *out = f;
    return CS_OK;
}

int cs_fit_wait_stream(cs_fit* f, void* hip_stream)
{
    if (!f) return fail(CS_ERR_INVALID, "fit handle is NULL");
    HIPCHK(hipSetDevice(f->device));
    return wait_on_stream(f->stream, hip_stream);
}

void cs_fit_free(cs_fit* f)
{
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->ev0) (void)hipEventDestroy(f->ev0);
    if (f->ev1) (void)hipEventDestroy(f->ev1);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
}

int cs_fit_last_ms(const cs_fit* f, double* device_ms)
{
    if (!f || !device_ms) return fail(CS_ERR_INVALID, "NULL argument");
    *device_ms = f->last_ms;
    return CS_OK;
}

namespace {

struct Timer {
    cs_fit* f;
    explicit Timer(cs_fit* f_) : f(f_) { (void)hipEventRecord(f->ev0, f->stream); }
    int stop()
    {
        HIPCHK(hipEventRecord(f->ev1, f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, f->ev0, f->ev1));
        f->last_ms = ms;
        return CS_OK;
    }
};

// features on the device: the caller's pointer, or a staged copy
int stage_features(cs_fit* f, const float* features, int64_t n, int F, int kind, const float** dev)
{
    if (kind == CS_MEM_DEVICE) { *dev = features; return CS_OK; }
    const size_t bytes = (size_t)n * F * sizeof(float);
    int rc = f->feat.ensure(bytes);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(f->feat.p, features, bytes, hipMemcpyHostToDevice, f->stream));
    *dev = f->feat.as<float>();
    return CS_OK;
}

int check_features(const cs_fit* f, const float* features, int64_t n, int F, int kind)
{
    if (!f) return fail(CS_ERR_INVALID, "handle is NULL");
    if (!features) return fail(CS_ERR_INVALID, "features is NULL");
    if (kind != CS_MEM_HOST && kind != CS_MEM_DEVICE) return fail(CS_ERR_INVALID, "memory kind must be CS_MEM_HOST or CS_MEM_DEVICE");
    if (n < 1 || n > (1ll << 30)) return fail(CS_ERR_INVALID, "n=%lld: need 1 .. 2^30 training cells", (long long)n);
    if (F < 1 || F > (1 << 20)) return fail(CS_ERR_INVALID, "n_features=%d out of range (1 .. 2^20)", F);
    return CS_OK;
}

// numpy's linear-interpolation percentile of a sorted float32 sample, from the two neighbouring order
// statistics (numpy/lib/_function_base_impl.py: _QuantileMethods['linear'], _get_indexes, _get_gamma, _lerp); float64 result
struct QuantilePos { int64_t lo, hi; double gamma; };
QuantilePos quantile_pos(int64_t n, double q)
{
    const double vi = (double)(n - 1) * q;                                       // _QuantileMethods['linear']
    QuantilePos p;
    const double fl = std::floor(vi);
    p.gamma = vi - fl;
    p.lo = (int64_t)fl;
    p.hi = p.lo + 1;
    if (vi >= (double)(n - 1)) { p.lo = p.hi = n - 1; }
    if (vi < 0) { p.lo = p.hi = 0; }
    return p;
}
double lerp_f32(float a, float b, double t)
{
    const float diff = b - a;
    volatile double prod = (double)diff * t;                                     // volatile: no contraction into an FMA
    double r = (double)a + prod;
    if (t >= 0.5) {
        volatile double prod2 = (double)diff * (1.0 - t);
        r = (double)b - prod2;
    }
    return r;
}

}  // namespace

int cs_fit_scaler(cs_fit* f, const float* features, int64_t n, int32_t n_features, int kind, float* center, double* scale)
{
    int rc = check_features(f, features, n, n_features, kind);
    if (rc) return rc;
    if (!center || !scale) return fail(CS_ERR_INVALID, "NULL output");
    HIPCHK(hipSetDevice(f->device));
    const int F = n_features;
    const float* d_feat;
    if ((rc = stage_features(f, features, n, F, kind, &d_feat))) return rc;
    const long ld = (long)((n + 63) / 64 * 64);
    if ((rc = f->keys.ensure((size_t)F * ld * sizeof(unsigned)))) return rc;
    if ((rc = f->sel.ensure((size_t)F * SEL_T * sizeof(float)))) return rc;
    if ((rc = f->nanc.ensure(sizeof(int)))) return rc;

    const QuantilePos q25 = quantile_pos(n, 0.25), q75 = quantile_pos(n, 0.75);   // np.true_divide([25, 75], 100): exact
    SelRanks rk;
    rk.r[0] = (unsigned)((n - 1) / 2); rk.r[1] = (unsigned)(n / 2);
    rk.r[2] = (unsigned)q25.lo; rk.r[3] = (unsigned)q25.hi;
    rk.r[4] = (unsigned)q75.lo; rk.r[5] = (unsigned)q75.hi;

    Timer tm(f);
    HIPCHK(hipMemsetAsync(f->nanc.p, 0, sizeof(int), f->stream));
    hipLaunchKernelGGL(keys_transpose_kernel, dim3((unsigned)((n + 31) / 32), (unsigned)((F + 31) / 32)), dim3(256), 0, f->stream,
                       d_feat, (long)n, F, ld, f->keys.as<unsigned>(), f->nanc.as<int>());
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(column_select_kernel, dim3((unsigned)F), dim3(SEL_THREADS), 0, f->stream, f->keys.as<unsigned>(), (long)n, ld,
                       rk, f->sel.as<float>());
    HIPCHK(hipGetLastError());
    std::vector<float> sel((size_t)F * SEL_T);
    int nans = 0;
    HIPCHK(hipMemcpyAsync(sel.data(), f->sel.p, sel.size() * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipMemcpyAsync(&nans, f->nanc.p, sizeof(int), hipMemcpyDeviceToHost, f->stream));
    if ((rc = tm.stop())) return rc;
    if (nans) return fail(CS_ERR_UNSUPPORTED, "features hold %d NaNs: nanmedian / nanpercentile semantics are not implemented", nans);

    for (int c = 0; c < F; ++c) {
        const float* s = &sel[(size_t)c * SEL_T];
        // np.nanmedian: the middle element, or mean of the two middle ones in float32 ((a + b) / 2)
        if (n & 1) center[c] = s[0];
        else { volatile float sum = s[0] + s[1]; center[c] = sum / 2.0f; }
        const double lo = lerp_f32(s[2], s[3], q25.gamma), hi = lerp_f32(s[4], s[5], q75.gamma);
        double sc = hi - lo;                                                     // _data.py:1672
        if (sc < 10.0 * std::numeric_limits<double>::epsilon()) sc = 1.0;       // _handle_zeros_in_scale, _data.py:114-123
        scale[c] = sc;
    }
    return CS_OK;
}

int cs_fit_pca_moments(cs_fit* f, const float* features, int64_t n, int32_t n_features, int kind, const float* center,
                       const double* scale, float* mean, double* scatter)
{
    int rc = check_features(f, features, n, n_features, kind);
    if (rc) return rc;
    if (!center || !scale || !mean || !scatter) return fail(CS_ERR_INVALID, "NULL argument");
    const int F = n_features;
    if (F % COV_T || F > 8192)
        return fail(CS_ERR_UNSUPPORTED, "n_features=%d: the moment kernel needs a multiple of %d up to 8192", F, COV_T);
    HIPCHK(hipSetDevice(f->device));
    const float* d_feat;
    if ((rc = stage_features(f, features, n, F, kind, &d_feat))) return rc;
    const int T = F / COV_T, tiles = T * (T + 1) / 2;
    int ksplit = std::max(1, std::min(16, (int)(2048 / tiles)));
    ksplit = (int)std::min<int64_t>(ksplit, (n + 3) / 4);
    const size_t FF = (size_t)F * F;
    if ((rc = f->cen.ensure(sizeof(float) * F)) || (rc = f->scl.ensure(sizeof(double) * F)) || (rc = f->mean.ensure(sizeof(float) * F)) ||
        (rc = f->xs.ensure((size_t)n * F * sizeof(float))) || (rc = f->part.ensure(FF * ksplit * sizeof(double))) ||
        (rc = f->scat.ensure(FF * sizeof(double))))
        return rc;
    HIPCHK(hipMemcpyAsync(f->cen.p, center, sizeof(float) * F, hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->scl.p, scale, sizeof(double) * F, hipMemcpyHostToDevice, f->stream));
    Timer tm(f);
    const long total = (long)n * F;
    const unsigned eg = (unsigned)std::min<long>((total + 255) / 256, 256L * 64);
    hipLaunchKernelGGL(scale_kernel, dim3(eg), dim3(256), 0, f->stream, d_feat, f->cen.as<float>(), f->scl.as<double>(), total, F,
                       f->xs.as<float>());
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(column_mean_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, f->stream, f->xs.as<float>(), (long)n, F,
                       f->mean.as<float>());
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(center_kernel, dim3(eg), dim3(256), 0, f->stream, f->xs.as<float>(), f->mean.as<float>(), total, F);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)tiles, (unsigned)ksplit), dim3(256), 0, f->stream, f->xs.as<float>(), (long)n, F,
                       ksplit, f->part.as<double>());
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(scatter_merge_kernel, dim3((unsigned)std::min<size_t>((FF + 255) / 256, 256 * 64)), dim3(256), 0, f->stream,
                       f->part.as<double>(), F, ksplit, f->scat.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(mean, f->mean.p, sizeof(float) * F, hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipMemcpyAsync(scatter, f->scat.p, FF * sizeof(double), hipMemcpyDeviceToHost, f->stream));
    return tm.stop();
}

int cs_fit_project(cs_fit* f, const float* features, int64_t n, int32_t n_features, int kind, const float* center,
                   const double* scale, const float* components, const float* mean_proj, int32_t n_components, float* out)
{
    int rc = check_features(f, features, n, n_features, kind);
    if (rc) return rc;
    if (!center || !scale || !components || !mean_proj || !out) return fail(CS_ERR_INVALID, "NULL argument");
    if (n_components < 1 || n_components > 128) return fail(CS_ERR_INVALID, "n_components=%d (1..128)", n_components);
    HIPCHK(hipSetDevice(f->device));
    const int F = n_features, C = n_components;
    const float* d_feat;
    if ((rc = stage_features(f, features, n, F, kind, &d_feat))) return rc;
    const int fpad = (F + 511) / 512 * 512, cpad = (C + 15) / 16 * 16;
    std::vector<float> cp((size_t)cpad * fpad, 0.0f);
    for (int c = 0; c < C; ++c) memcpy(&cp[(size_t)c * fpad], components + (size_t)c * F, sizeof(float) * F);
    if ((rc = f->cen.ensure(sizeof(float) * F)) || (rc = f->scl.ensure(sizeof(double) * F)) ||
        (rc = f->comps.ensure(cp.size() * sizeof(float))) || (rc = f->mproj.ensure(sizeof(float) * C)) ||
        (rc = f->proj.ensure((size_t)n * C * sizeof(float))))
        return rc;
    HIPCHK(hipMemcpyAsync(f->cen.p, center, sizeof(float) * F, hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->scl.p, scale, sizeof(double) * F, hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->comps.p, cp.data(), cp.size() * sizeof(float), hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->mproj.p, mean_proj, sizeof(float) * C, hipMemcpyHostToDevice, f->stream));
    Timer tm(f);
    HIPCHK(launch_scaler_pca(d_feat, f->cen.as<float>(), f->scl.as<double>(), f->comps.as<float>(), f->mproj.as<float>(), F, fpad, C,
                             cpad, f->proj.as<float>(), n, f->stream));
    HIPCHK(hipMemcpyAsync(out, f->proj.p, (size_t)n * C * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    return tm.stop();
}

int cs_fit_ocsvm(cs_fit* f, const double* x, int64_t n, int32_t n_components, double gamma, double nu, double eps,
                 int64_t max_iter, double* alpha, double* rho, double* obj, int64_t* n_iter, int32_t* status)
{
    if (!f) return fail(CS_ERR_INVALID, "handle is NULL");
    if (!x || !alpha || !rho) return fail(CS_ERR_INVALID, "NULL argument");
    if (n < 1 || n > (1 << 28)) return fail(CS_ERR_INVALID, "n=%lld: need 1 .. 2^28 training points", (long long)n);
    const int D = n_components;
    if (D < 1 || D > SMO_DP) return fail(CS_ERR_INVALID, "n_components=%d (1..%d)", D, SMO_DP);
    if (!(nu > 0.0 && nu <= 1.0)) return fail(CS_ERR_INVALID, "nu=%g: need 0 < nu <= 1 (libsvm's check)", nu);
    if (!(gamma >= 0.0) || !(eps > 0.0)) return fail(CS_ERR_INVALID, "gamma=%g / eps=%g", gamma, eps);
    HIPCHK(hipSetDevice(f->device));
    const int N = (int)n;
    const long ld = (long)((n + 63) / 64 * 64);
    const int nb = (N + SMO_B - 1) / SMO_B;
    int rc;
    if ((rc = f->x.ensure((size_t)n * D * 8)) || (rc = f->xT.ensure((size_t)D * ld * 8)) || (rc = f->xsq.ensure((size_t)n * 8)) ||
        (rc = f->alpha0.ensure((size_t)n * 8)) || (rc = f->alpha1.ensure((size_t)n * 8)) || (rc = f->G0.ensure((size_t)n * 8)) ||
        (rc = f->G1.ensure((size_t)n * 8)) || (rc = f->Qi.ensure((size_t)n * 4)) || (rc = f->part_i.ensure((size_t)nb * sizeof(SmoPickI))) ||
        (rc = f->part_j.ensure((size_t)nb * sizeof(SmoPickJ))) || (rc = f->state.ensure(sizeof(SmoState))))
        return rc;

    // PCA output is float32 cast to double (sklearn svm/_base.py:190): then every value is a float and the
    // transposed copy can be stored as such without changing a single product
    bool as_float = true;
    for (size_t e = 0, ne = (size_t)n * D; e < ne; ++e) {
        if (!std::isfinite(x[e])) return fail(CS_ERR_INVALID, "x holds a non-finite value at element %zu (sklearn's check_array rejects it)", e);
        as_float = as_float && (double)(float)x[e] == x[e];
    }

    // solve_one_class, svm.cpp:1718-1735: nu_l accumulated point by point, the first points filled to C = 1
    std::vector<double> a0((size_t)n, 0.0);
    int n0 = 0;
    {
        double nu_l = 0.0;
        for (int64_t i = 0; i < n; ++i) { volatile double t = 1.0 * nu; nu_l += t; }
        int64_t i = 0;
        while (nu_l > 0 && i < n) {
            a0[(size_t)i] = std::min(1.0, nu_l);
            nu_l -= a0[(size_t)i];
            ++i;
        }
        n0 = (int)i;
    }
    HIPCHK(hipMemcpyAsync(f->x.p, x, (size_t)n * D * 8, hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemcpyAsync(f->alpha0.p, a0.data(), (size_t)n * 8, hipMemcpyHostToDevice, f->stream));
    HIPCHK(hipMemsetAsync(f->state.p, 0, sizeof(SmoState), f->stream));
    Timer tm(f);
    SmoArgs a{};
    a.xT = f->xT.p; a.xsq = f->xsq.as<double>(); a.ld = ld; a.n = N; a.D = D; a.nb = nb;
    a.gamma = gamma; a.eps = eps; a.max_iter = max_iter;
    a.alpha[0] = f->alpha0.as<double>(); a.alpha[1] = f->alpha1.as<double>();
    a.G[0] = f->G0.as<double>(); a.G[1] = f->G1.as<double>();
    a.Qi = f->Qi.as<float>(); a.part_i = f->part_i.as<SmoPickI>(); a.part_j = f->part_j.as<SmoPickJ>();
    a.state = f->state.as<SmoState>();
    if (as_float) {
        hipLaunchKernelGGL(smo_prepare_kernel<float>, dim3((unsigned)nb), dim3(256), 0, f->stream, f->x.as<double>(), N, D, ld,
                           f->xT.as<float>(), f->xsq.as<double>());
        hipLaunchKernelGGL(smo_init_kernel<float>, dim3((unsigned)nb), dim3(SMO_B), 0, f->stream, a, n0);
    } else {
        hipLaunchKernelGGL(smo_prepare_kernel<double>, dim3((unsigned)nb), dim3(256), 0, f->stream, f->x.as<double>(), N, D, ld,
                           f->xT.as<double>(), f->xsq.as<double>());
        hipLaunchKernelGGL(smo_init_kernel<double>, dim3((unsigned)nb), dim3(SMO_B), 0, f->stream, a, n0);
    }
    HIPCHK(hipGetLastError());

    const long long hard_cap = 200ll * 1000 * 1000;                   // sklearn's default max_iter = -1 is unbounded
    const int batch = 256;
    SmoState st{};
    long long issued = 0;
    int parity = 0;
    while (true) {
        for (int it = 0; it < batch; ++it) {
            if (as_float) {
                hipLaunchKernelGGL(smo_row_i_kernel<float>, dim3((unsigned)nb), dim3(SMO_B), 0, f->stream, a, parity);
                hipLaunchKernelGGL(smo_update_kernel<float>, dim3((unsigned)nb), dim3(SMO_B), 0, f->stream, a, parity);
            } else {
                hipLaunchKernelGGL(smo_row_i_kernel<double>, dim3((unsigned)nb), dim3(SMO_B), 0, f->stream, a, parity);
                hipLaunchKernelGGL(smo_update_kernel<double>, dim3((unsigned)nb), dim3(SMO_B), 0, f->stream, a, parity);
            }
            parity ^= 1;
        }
        HIPCHK(hipGetLastError());
        issued += batch;
        HIPCHK(hipMemcpyAsync(&st, f->state.p, sizeof(SmoState), hipMemcpyDeviceToHost, f->stream));
        HIPCHK(hipStreamSynchronize(f->stream));
        if (st.done) break;
        if (issued >= hard_cap) return fail(CS_ERR_UNSUPPORTED, "SMO did not converge within %lld iterations", hard_cap);
    }
    std::vector<double> G((size_t)n);
    HIPCHK(hipMemcpyAsync(alpha, a.alpha[st.final_parity], (size_t)n * 8, hipMemcpyDeviceToHost, f->stream));
    HIPCHK(hipMemcpyAsync(G.data(), a.G[st.final_parity], (size_t)n * 8, hipMemcpyDeviceToHost, f->stream));
    if ((rc = tm.stop())) return rc;

    // calculate_rho, svm.cpp:1123-1160 (y = +1) and the objective, :901-907 (p = 0)
    {
        int64_t nr_free = 0;
        double ub = SMO_INF, lb = -SMO_INF, sum_free = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            const double yG = G[(size_t)i];
            if (alpha[i] >= 1.0) lb = std::max(lb, yG);
            else if (alpha[i] <= 0.0) ub = std::min(ub, yG);
            else { ++nr_free; sum_free += yG; }
        }
        *rho = nr_free > 0 ? sum_free / (double)nr_free : (ub + lb) / 2;
        if (obj) {
            double v = 0.0;
            for (int64_t i = 0; i < n; ++i) { volatile double t = alpha[i] * (G[(size_t)i] + 0.0); v += t; }
            *obj = v / 2;
        }
    }
    if (n_iter) *n_iter = st.n_iter;
    if (status) *status = st.done == 2 ? 1 : 0;                       // 1: stopped at max_iter (libsvm's solve_timed_out)
    return CS_OK;
}
