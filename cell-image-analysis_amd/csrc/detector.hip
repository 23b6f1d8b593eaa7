// detector.hip -- the scoring tail of compute_anomaly_scores (improved_detection.py:134-153)
// on device, so a cell's 18 bytes of results are the only thing that leaves the GPU:
//   scaler.transform  (x - center_) / scale_        sklearn _data.py:1715-1718
//   pca.transform     x @ components_.T - mean_proj sklearn _base.py:147-155
//   OneClassSVM       sum_i a_i exp(-g |x - sv_i|^2) - rho   libsvm svm.cpp:461-476, 2818-2838
//   negation / labels improved_detection.py:138-150
// plus the per-cell mean of the squared / absolute error partial sums (:126-127) and the
// counter-based synthetic crop generator used by the benchmark and the parity tests.
#include "common.hpp"

#include <cstring>

namespace cs {

namespace {

// ---------------------------------------------------------------- scaler + PCA
// A small exact-fp32 MFMA GEMM: 16 cells x cpad components per workgroup, K = features.
// The scaler is fused into the staging of the A operand; the division is done in double
// and rounded once, as numpy does for float32 /= float64.
constexpr int PCA_MT = 4;                         // 16-cell tiles per workgroup
constexpr int PCA_CELLS = 16 * PCA_MT;
constexpr int PCA_KC = 256;                       // features per LDS chunk
constexpr int PCA_LD = PCA_KC + 4;                // padded row (floats)
constexpr int PCA_LDS = PCA_CELLS * PCA_LD * 4;

__global__ __launch_bounds__(256) void scaler_pca_kernel(
    const float* __restrict__ feat, const float* __restrict__ center, const double* __restrict__ scale,
    const float* __restrict__ comps, const float* __restrict__ mean_proj, int F, int fpad, int C,
    int cpad, float* __restrict__ out, long n)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = (float*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const long cell0 = (long)blockIdx.x * PCA_CELLS;
    const int ntiles = cpad / 16;   // <= 8: wave w owns component tiles w and w + 4

    f32x4 acc[2][PCA_MT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < PCA_MT; ++m) acc[t][m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // Staging: PCA_KC == blockDim, so thread `tid` owns feature k0 + tid of all PCA_CELLS cells of the workgroup: its
    // center / scale are per-chunk constants, the loads of one cell are a coalesced row, the LDS column is its own.
    // The raw features of chunk k0 + PCA_KC are loaded (unconditionally, clamped) before the MFMAs of chunk k0 and
    // scaled + written after them, so the load latency hides behind the matrix phase.
    static_assert(PCA_KC == 256, "one feature column per thread");
    float raw[PCA_CELLS];
    auto issue = [&](int k0) {
        const int gk = k0 + tid < F ? k0 + tid : F - 1;
#pragma unroll
        for (int c = 0; c < PCA_CELLS; ++c) {
            const long cell = cell0 + c < n ? cell0 + c : n - 1;
            raw[c] = feat[cell * F + gk];
        }
    };
    auto stash = [&](int k0) {
        const int gk = k0 + tid;
        const bool kok = gk < F;
        const float ctr = center[kok ? gk : 0];
        const double scl = scale[kok ? gk : 0];
#pragma unroll
        for (int c = 0; c < PCA_CELLS; ++c) {
            const float t = raw[c] - ctr;
            const float v = (float)((double)t / scl);   // (a reciprocal multiply + tie test, exact, measured slower: 7.3 vs 6.0 ms per 1 M cells)
            xs[c * PCA_LD + tid] = (kok && cell0 + c < n) ? v : 0.0f;
        }
    };
    issue(0);
    for (int k0 = 0; k0 < fpad; k0 += PCA_KC) {
        stash(k0);
        __syncthreads();
        if (k0 + PCA_KC < fpad) issue(k0 + PCA_KC);
        // B fragments (components_, L2-resident) come straight from global memory: the loads of K step ks + 1 are issued
        // before the MFMAs of step ks (rows of the zero-padded table exist for every tile < cpad / 16; a tile beyond
        // ntiles re-reads tile 0 and is never used)
        const float* bp[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tile = wave + 4 * t < ntiles ? wave + 4 * t : 0;
            bp[t] = comps + (size_t)(tile * 16 + li) * fpad + k0 + kq * 4;
        }
        f32x4 b[2] = {*(const f32x4*)bp[0], *(const f32x4*)bp[1]};
#pragma unroll 2
        for (int ks = 0; ks < PCA_KC / 16; ++ks) {
            f32x4 a[PCA_MT];
#pragma unroll
            for (int m = 0; m < PCA_MT; ++m)
                a[m] = *(const f32x4*)(xs + (m * 16 + li) * PCA_LD + ks * 16 + kq * 4);
            const int kn = ks + 1 < PCA_KC / 16 ? ks + 1 : ks;
            const f32x4 nb[2] = {*(const f32x4*)(bp[0] + kn * 16), *(const f32x4*)(bp[1] + kn * 16)};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (wave + 4 * t < ntiles) {  // wave-uniform
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int m = 0; m < PCA_MT; ++m)
                            acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][j], b[t][j], acc[t][m], 0, 0, 0);
                }
            }
            b[0] = nb[0];
            b[1] = nb[1];
        }
        __syncthreads();
    }
    // D[row = 4 kq + r (cell)][col = li (component)]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int comp = (wave + 4 * t) * 16 + li;
        if (wave + 4 * t < ntiles && comp < C) {
            const float mp = mean_proj[comp];
#pragma unroll
            for (int m = 0; m < PCA_MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long cell = cell0 + m * 16 + 4 * kq + r;
                    if (cell < n) out[cell * C + comp] = acc[t][m][r] - mp;
                }
        }
    }
}

// ---------------------------------------------------------------- scaler + PCA on the bf16 matrix pipe
// The same GEMM with the fp32 contraction as six bf16 products (conv45_bf16x3.hip has the algebra and the hardware check):
// the scaled feature -- still (f - center) / scale in double, rounded once to fp32 -- is split into three bf16 terms when it
// is staged, components_ is split on the host (pack_pca_bf16x3).  A one-hot component row reproduces the scaled feature
// bit for bit (x1 + x2 + x3 = x exactly, every partial sum representable): the scaler test holds unchanged.
// Staging: thread = (pair of adjacent features, 16 of the workgroup's 64 cells); a pair packs into one dword per plane, so a
// cell's row of a plane is written as 64 consecutive dwords.  Row stride 288 B = twice an odd number of 16-byte slots.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int PX_KC = 128;                        // features per LDS chunk
constexpr int PX_ROW = PX_KC * 2 + 32;            // bytes per cell row of a plane
constexpr int PX_PLANE = PCA_CELLS * PX_ROW;      // 18,432 B
constexpr int PX_LDS = 3 * PX_PLANE;              // 55,296 B

// The K = fpad features are summed in DET_RANGES fixed ranges of whole chunks: a range's products accumulate from zero in the
// MFMA accumulators, the range sums are then added in ascending order (fp32).  SPLIT = false: one workgroup walks all ranges of
// its 64 cells.  SPLIT = true: workgroup (x, j) computes range j alone and leaves its sum in part[j][cell][C]; pca_split_sum_kernel
// adds the ranges in the same order -- the SAME arithmetic, so a cell's components do not depend on how many cells it is screened
// with.  The split form is for small calls (the reference screens one sample of 1e2 .. 1e4 cells per call, improved_detection.py:199):
// at 128 cells the unsplit kernel is two workgroups walking 2,048 features one chunk after the other (94 us).
constexpr int DET_RANGES = 8;

template <bool SPLIT>
__global__ __launch_bounds__(256) void scaler_pca_x3_kernel(
    const float* __restrict__ feat, const float* __restrict__ center, const double* __restrict__ scale,
    const bf16x8* __restrict__ comps /* pack_pca_bf16x3 */, const float* __restrict__ mean_proj, int F, int fpad, int C,
    int cpad, float* __restrict__ out /* SPLIT: part [DET_RANGES][n][C] */, long n)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const long cell0 = (long)blockIdx.x * PCA_CELLS;
    const int ntiles = cpad / 16;   // <= 8: wave w owns component tiles w and w + 4
    const int nkb = fpad / 32;

    f32x4 acc[2][PCA_MT], tot[2][PCA_MT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < PCA_MT; ++m) acc[t][m] = tot[t][m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int nch = fpad / PX_KC;                // chunks; range j = chunks [nch j / R, nch (j + 1) / R)
    int rj = SPLIT ? (int)blockIdx.y : 0;
    const int kbeg = SPLIT ? (nch * rj / DET_RANGES) * PX_KC : 0;
    const int kend = SPLIT ? (nch * (rj + 1) / DET_RANGES) * PX_KC : fpad;

    const int fp = tid & 63, cg = tid >> 6;      // feature pair of the chunk, 16-cell group
    float raw[16][2];
    auto issue = [&](int k0) {
        const int g0 = k0 + 2 * fp < F ? k0 + 2 * fp : F - 1, g1 = k0 + 2 * fp + 1 < F ? k0 + 2 * fp + 1 : F - 1;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const long cell = cell0 + cg * 16 + c < n ? cell0 + cg * 16 + c : n - 1;
            raw[c][0] = feat[cell * F + g0];
            raw[c][1] = feat[cell * F + g1];
        }
    };
    auto stash = [&](int k0) {
        const int gk = k0 + 2 * fp;
        const bool ok0 = gk < F, ok1 = gk + 1 < F;
        const float ctr0 = center[ok0 ? gk : 0], ctr1 = center[ok1 ? gk + 1 : 0];
        const double scl0 = scale[ok0 ? gk : 0], scl1 = scale[ok1 ? gk + 1 : 0];
        char* d = smem + (cg * 16) * PX_ROW + fp * 4;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const bool cok = cell0 + cg * 16 + c < n;
            float v[2];
            v[0] = (ok0 && cok) ? (float)((double)(raw[c][0] - ctr0) / scl0) : 0.0f;
            v[1] = (ok1 && cok) ? (float)((double)(raw[c][1] - ctr1) / scl1) : 0.0f;
            bf16x2 h1, h2, h3;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const __bf16 a1 = (__bf16)v[j];
                const float r1 = v[j] - (float)a1;
                const __bf16 a2 = (__bf16)r1;
                const float r2 = r1 - (float)a2;
                h1[j] = a1; h2[j] = a2; h3[j] = (__bf16)r2;
            }
            *(bf16x2*)(d + c * PX_ROW) = h1;
            *(bf16x2*)(d + c * PX_ROW + PX_PLANE) = h2;
            *(bf16x2*)(d + c * PX_ROW + 2 * PX_PLANE) = h3;
        }
    };
    // B fragments: [tile][k block][plane][lane][8]; a tile beyond ntiles re-reads tile 0 and is never used
    const bf16x8* bp[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int tile = wave + 4 * t < ntiles ? wave + 4 * t : 0;
        bp[t] = comps + (size_t)tile * nkb * 3 * 64 + lane;
    }
    auto load_b = [&](int kb, bf16x8 (&b)[2][3]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[t][p] = bp[t][((size_t)kb * 3 + p) * 64];
    };
    bf16x8 bn[2][3];
    if (kbeg < kend) {
        issue(kbeg);
        load_b(kbeg / 32, bn);
    }
    for (int k0 = kbeg; k0 < kend; k0 += PX_KC) {
        stash(k0);
        __syncthreads();
        if (k0 + PX_KC < kend) issue(k0 + PX_KC);
#pragma unroll
        for (int ks = 0; ks < PX_KC / 32; ++ks) {
            bf16x8 b[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) b[t][p] = bn[t][p];
            const int kb = k0 / 32 + ks;
            if (kb + 1 < kend / 32) load_b(kb + 1, bn);
#pragma unroll
            for (int m = 0; m < PCA_MT; ++m) {
                const char* ap = smem + (m * 16 + li) * PX_ROW + ks * 64 + kq * 16;
                const bf16x8 a1 = *(const bf16x8*)ap;
                const bf16x8 a2 = *(const bf16x8*)(ap + PX_PLANE);
                const bf16x8 a3 = *(const bf16x8*)(ap + 2 * PX_PLANE);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (wave + 4 * t < ntiles) {  // wave-uniform
                        acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b[t][2], acc[t][m], 0, 0, 0);
                        acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b[t][1], acc[t][m], 0, 0, 0);
                        acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b[t][0], acc[t][m], 0, 0, 0);
                        acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b[t][1], acc[t][m], 0, 0, 0);
                        acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b[t][0], acc[t][m], 0, 0, 0);
                        acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b[t][0], acc[t][m], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
        // the end of a range (every empty range in between adds an exact zero): its sum joins the total
        while (rj < DET_RANGES && (nch * (rj + 1) / DET_RANGES) * PX_KC <= k0 + PX_KC) {
            if ((nch * (rj + 1) / DET_RANGES) * PX_KC == k0 + PX_KC) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int m = 0; m < PCA_MT; ++m) { tot[t][m] += acc[t][m]; acc[t][m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
            }
            ++rj;
            if (SPLIT) break;
        }
    }
    // D[row = 4 kq + r (cell)][col = li (component)]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int comp = (wave + 4 * t) * 16 + li;
        if (wave + 4 * t < ntiles && comp < C) {
            const float mp = SPLIT ? 0.0f : mean_proj[comp];
#pragma unroll
            for (int m = 0; m < PCA_MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long cell = cell0 + m * 16 + 4 * kq + r;
                    if (cell < n) {
                        if constexpr (SPLIT) out[((size_t)blockIdx.y * n + cell) * C + comp] = tot[t][m][r];
                        else out[cell * C + comp] = tot[t][m][r] - mp;
                    }
                }
        }
    }
}

// out[cell][c] = ((P_0 + P_1) + ... + P_7) - mean_proj[c]: the order scaler_pca_x3_kernel<false> adds its ranges in
__global__ void pca_split_sum_kernel(const float* __restrict__ part, const float* __restrict__ mean_proj, int C, float* __restrict__ out, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * C) return;
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < DET_RANGES; ++j) v += part[(size_t)j * n * C + i];
    out[i] = v - mean_proj[i % C];
}

// ---------------------------------------------------------------- one-class SVM decision (fp64 MFMA)
// ||x - s||^2 = ||x||^2 + ||s||^2 - 2 x.s: the cross terms of 16 cells x 16 support vectors are one
// v_mfma_f64_16x16x4_f64 chain over the D components (libsvm evaluates the kernel in double; so does
// this, the expansion costs ~1e-16 * (||x||^2 + ||s||^2) absolute on the distance, far inside the 1e-9
// tolerance on the decision).  A wave keeps the A fragments of two 16-cell tiles resident (2 x KS doubles);
// the workgroup's eight waves (256 cells) share each block of 16 support vectors, staged once in LDS
// (double buffered, one barrier per block) together with the block's ||s||^2 and dual coefficients;
// exp() and the coefficient-weighted sum run on the accumulator layout (lane = support vector, register
// = cell) and are reduced over the 16 lanes of a row at the end, in fixed order.
// f64 MFMA maps: A[row = l & 15][k = l >> 4], B[k = l >> 4][col = l & 15], D[row = (l >> 4) + 4 reg][col = l & 15].
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int SVMM_WAVES = 8, SVMM_TILES = 2, SVMM_CELLS = SVMM_WAVES * SVMM_TILES * 16;

// The support-vector blocks are summed in DET_RANGES fixed ranges: per lane over the blocks of a range, over the 16 lanes of a row
// in a fixed tree, and the range sums in ascending order.  SPLIT = false: one workgroup walks all ranges of its 256 cells.  SPLIT =
// true: workgroup (x, j) evaluates range j alone and leaves its sum in part[j][cell]; svm_split_sum_kernel adds the ranges in the
// same order and subtracts rho -- the same arithmetic (see scaler_pca_x3_kernel).  For small calls: at 128 cells the unsplit
// kernel is ONE workgroup walking ~27 blocks of double-precision exp() one after the other (167 us per detector).
// one or two detectors per launch (blockIdx.z picks): the split form runs both side by side
struct SvmArgs {
    const double* svT[2];       // [D][nsv_pad]
    const double* svn[2];       // [nsv_pad] ||s||^2
    const double* coef[2];      // [nsv_pad]
    int nsv_pad[2];
    double gamma[2], rho[2];
};

template <int KS, bool SPLIT>   // K steps of 4 components: D <= 4 KS
__global__ __launch_bounds__(64 * SVMM_WAVES, 1) void ocsvm_mfma_kernel(
    const float* __restrict__ pca, int D, SvmArgs sa, double* __restrict__ dec /* SPLIT: part [detector][DET_RANGES][n] */, long n)
{
    const int det = (int)blockIdx.z;
    const double* __restrict__ svT = sa.svT[det];
    const double* __restrict__ svn = sa.svn[det];
    const double* __restrict__ coef = sa.coef[det];
    const int nsv_pad = sa.nsv_pad[det];
    const double gamma = sa.gamma[det], rho = sa.rho[det];
    constexpr int ROWS = 4 * KS + 2;                     // components (zero padded) + ||s||^2 row + coef row
    __shared__ double sb[2][ROWS][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const long cell0 = (long)blockIdx.x * SVMM_CELLS + wave * (SVMM_TILES * 16);

    double xa[SVMM_TILES][KS];
    double rown[SVMM_TILES][4];
#pragma unroll
    for (int t = 0; t < SVMM_TILES; ++t) {
        long cell = cell0 + t * 16 + li;
        if (cell >= n) cell = n - 1;                     // tail rows recompute the last cell, never stored
        double nx = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int d = 4 * s + kq;
            const double v = d < D ? (double)pca[cell * D + d] : 0.0;
            xa[t][s] = v;
            nx = fma(v, v, nx);
        }
        nx += __shfl_xor(nx, 16);
        nx += __shfl_xor(nx, 32);                        // every lane: ||x||^2 of cell li of the tile
#pragma unroll
        for (int r = 0; r < 4; ++r) rown[t][r] = __shfl(nx, kq + 4 * r);   // D row (kq + 4 r) <- lane li' = kq + 4 r
    }

    // staging of one block: ROWS x 16 doubles, thread e -> (row, col)
    constexpr int NST = (ROWS * 16 + 64 * SVMM_WAVES - 1) / (64 * SVMM_WAVES);
    auto fetch = [&](int blk, double v[NST]) {
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int e = tid + 64 * SVMM_WAVES * j, row = e >> 4, col = e & 15;
            const size_t sidx = (size_t)blk * 16 + col;
            double x = 0.0;
            if (row < D) x = svT[(size_t)row * nsv_pad + sidx];
            else if (row == 4 * KS) x = svn[sidx];
            else if (row == 4 * KS + 1) x = coef[sidx];
            v[j] = x;
        }
    };
    auto stash = [&](int buf, const double v[NST]) {
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int e = tid + 64 * SVMM_WAVES * j;
            if (e < ROWS * 16) sb[buf][e >> 4][e & 15] = v[j];
        }
    };
    const int nblk = nsv_pad / 16;
    int rj = SPLIT ? (int)blockIdx.y : 0;
    const int bbeg = SPLIT ? nblk * rj / DET_RANGES : 0, bend = SPLIT ? nblk * (rj + 1) / DET_RANGES : nblk;
    if (bbeg < bend) {
        double v[NST];
        fetch(bbeg, v);
        stash(bbeg & 1, v);
    }
    __syncthreads();

    double sum[SVMM_TILES][4], tot[SVMM_TILES][4];
#pragma unroll
    for (int t = 0; t < SVMM_TILES; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[t][r] = tot[t][r] = 0.0;
    for (int blk = bbeg; blk < bend; ++blk) {
        const int cur = blk & 1;
        double nv[NST];
        if (blk + 1 < bend) fetch(blk + 1, nv);
        f64x4 acc[SVMM_TILES];
#pragma unroll
        for (int t = 0; t < SVMM_TILES; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const double b = sb[cur][4 * s + kq][li];
#pragma unroll
            for (int t = 0; t < SVMM_TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[t][s], b, acc[t], 0, 0, 0);
        }
        const double sn = sb[cur][4 * KS][li], cf = sb[cur][4 * KS + 1][li];
#pragma unroll
        for (int t = 0; t < SVMM_TILES; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double d2 = fma(-2.0, acc[t][r], rown[t][r] + sn);
                sum[t][r] = fma(cf, exp(-gamma * d2), sum[t][r]);
            }
        if (blk + 1 < bend) stash(cur ^ 1, nv);
        __syncthreads();
        // the end of a range: its 16 lane sums in a fixed tree, the result added to the total (an empty range adds an exact zero)
        while (rj < DET_RANGES && nblk * (rj + 1) / DET_RANGES <= blk + 1) {
            if (nblk * (rj + 1) / DET_RANGES == blk + 1) {
#pragma unroll
                for (int t = 0; t < SVMM_TILES; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double v = sum[t][r];
                        v += __shfl_xor(v, 1);
                        v += __shfl_xor(v, 2);
                        v += __shfl_xor(v, 4);
                        v += __shfl_xor(v, 8);
                        tot[t][r] += v;
                        sum[t][r] = 0.0;
                    }
            }
            ++rj;
            if (SPLIT) break;
        }
    }
#pragma unroll
    for (int t = 0; t < SVMM_TILES; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long cell = cell0 + t * 16 + kq + 4 * r;
            if (li == 0 && cell < n) {
                if constexpr (SPLIT) dec[((size_t)det * DET_RANGES + blockIdx.y) * n + cell] = tot[t][r];
                else dec[cell] = tot[t][r] - rho;
            }
        }
}

// dec[cell] = ((P_0 + P_1) + ... + P_7) - rho: the order ocsvm_mfma_kernel<KS, false> adds its ranges in
__global__ void svm_split_sum_kernel(const double* __restrict__ part, double rho0, double rho1, double* __restrict__ dec0,
                                     double* __restrict__ dec1, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int det = (int)blockIdx.y;
    if (i >= n) return;
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < DET_RANGES; ++j) v += part[((size_t)det * DET_RANGES + j) * n + i];
    if (det == 0) dec0[i] = v - rho0;
    else dec1[i] = v - rho1;
}


// ---------------------------------------------------------------- finalize
__global__ void finalize_kernel(const float* __restrict__ errpart, int nparts, int npix,
                                const double* __restrict__ dec_c, const double* __restrict__ dec_m,
                                float* __restrict__ mse, float* __restrict__ mae,
                                double* __restrict__ score_c, double* __restrict__ score_m,
                                int8_t* __restrict__ pred_c, int8_t* __restrict__ pred_m, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // A crop with a NaN / Inf pixel: its error sums are NaN / Inf (what NumPy's mean((X - R)**2) gives, improved_detection.py:126-127);
    // its scores are reported as NaN and both flags as -1 (the reference raises in pca.transform: "Input X contains NaN").  The
    // convs saw that pixel as 0 (conv12_fused.hip), so no other cell is touched.
    bool bad = false;
    if (errpart) {
        float s2 = 0.0f, s1 = 0.0f;
        for (int p = 0; p < nparts; ++p) {
            s2 += errpart[(i * nparts + p) * 2 + 0];
            s1 += errpart[(i * nparts + p) * 2 + 1];
        }
        if (mse) mse[i] = s2 / (float)npix;
        if (mae) mae[i] = s1 / (float)npix;
        bad = !(fabsf(s2) <= 3.402823466e38f && fabsf(s1) <= 3.402823466e38f);
    }
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    if (dec_c) {
        const double d = bad ? qnan : dec_c[i];
        if (score_c) score_c[i] = -d;                 // improved_detection.py:149
        if (pred_c) pred_c[i] = (d > 0) ? 1 : -1;     // svm.cpp:2838
    }
    if (dec_m) {
        const double d = bad ? qnan : dec_m[i];
        if (score_m) score_m[i] = -d;
        if (pred_m) pred_m[i] = (d > 0) ? 1 : -1;
    }
}

// ---------------------------------------------------------------- synthetic crops
// Must stay bit-identical to oracle/cae_oracle.c:orc_hash24.
__device__ __forceinline__ uint32_t hash24(uint64_t seed, uint64_t cell, uint32_t pix)
{
    uint64_t z = seed + cell * 0x9E3779B97F4A7C15ULL + (uint64_t)pix * 0xD1B54A32D192ED03ULL;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (uint32_t)(z >> 40);
}

__global__ void synth_kernel(uint64_t seed, long first_cell, long n, int npix, float* __restrict__ out)
{
    const long total = n * npix;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long c = i / npix;
        const int p = (int)(i % npix);
        out[i] = (float)hash24(seed, (uint64_t)(first_cell + c), (uint32_t)p) * (1.0f / 16777216.0f);
    }
}

}  // namespace

hipError_t launch_scaler_pca(const float* feat, const float* center, const double* scale,
                             const float* comps_pad, const float* mean_proj, int F, int fpad, int C,
                             int cpad, float* pca_out, int64_t n_cells, hipStream_t stream)
{
    if (n_cells <= 0) return hipSuccess;
    if (cpad % 16 || cpad > 128 || fpad % PCA_KC) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((n_cells + PCA_CELLS - 1) / PCA_CELLS);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)scaler_pca_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PCA_LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(scaler_pca_kernel, dim3(grid), dim3(256), PCA_LDS, stream, feat, center, scale,
                       comps_pad, mean_proj, F, fpad, C, cpad, pca_out, (long)n_cells);
    return hipGetLastError();
}

// components_ (zero padded [cpad][fpad] fp32) as three bf16 planes in scaler_pca_x3_kernel's B order:
// [tile = comp / 16][k block = f / 32][plane][lane = 16 kq + li][8]: element j = plane of comps[16 tile + li][32 block + 8 kq + j]
size_t pack_pca_bf16x3(const float* comps_pad, int cpad, int fpad, uint16_t* dst)
{
    const size_t n = (size_t)cpad * fpad * 3;
    if (!dst) return n;
    auto rne = [](float x) -> uint16_t {
        uint32_t u;
        memcpy(&u, &x, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    auto val = [](uint16_t h) -> float {
        const uint32_t u = (uint32_t)h << 16;
        float x;
        memcpy(&x, &u, 4);
        return x;
    };
    const int nkb = fpad / 32;
    for (int c = 0; c < cpad; ++c)
        for (int f = 0; f < fpad; ++f) {
            const float v = comps_pad[(size_t)c * fpad + f];
            const uint16_t w1 = rne(v);
            const float r1 = v - val(w1);
            const uint16_t w2 = rne(r1);
            const float r2 = r1 - val(w2);
            const uint16_t pl[3] = {w1, w2, rne(r2)};
            const int tile = c >> 4, li = c & 15, kb = f >> 5, kq = (f >> 3) & 3, j = f & 7;
            for (int p = 0; p < 3; ++p) dst[((((size_t)tile * nkb + kb) * 3 + p) * 64 + kq * 16 + li) * 8 + j] = pl[p];
        }
    return n;
}

size_t det_split_ws_bytes(int C)
{
    const size_t pca = (size_t)DET_RANGES * DET_SPLIT_MAX_CELLS * (size_t)C * sizeof(float);
    const size_t svm = (size_t)2 * DET_RANGES * DET_SPLIT_MAX_CELLS * sizeof(double);
    return pca > svm ? pca : svm;
}

hipError_t launch_scaler_pca_x3(const float* feat, const float* center, const double* scale, const uint16_t* comps_planes,
                                const float* mean_proj, int F, int fpad, int C, int cpad, float* pca_out, int64_t n_cells,
                                hipStream_t stream, void* split_ws)
{
    if (n_cells <= 0) return hipSuccess;
    if (cpad % 16 || cpad > 128 || fpad % PX_KC) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((n_cells + PCA_CELLS - 1) / PCA_CELLS);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)scaler_pca_x3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PX_LDS);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)scaler_pca_x3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PX_LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (split_ws && n_cells <= DET_SPLIT_MAX_CELLS) {       // a small call: the ranges side by side, then their sum in the same order
        hipLaunchKernelGGL(scaler_pca_x3_kernel<true>, dim3(grid, DET_RANGES), dim3(256), PX_LDS, stream, feat, center, scale,
                           (const bf16x8*)comps_planes, mean_proj, F, fpad, C, cpad, (float*)split_ws, (long)n_cells);
        const long tot = (long)n_cells * C;
        hipLaunchKernelGGL(pca_split_sum_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, (const float*)split_ws, mean_proj,
                           C, pca_out, (long)n_cells);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(scaler_pca_x3_kernel<false>, dim3(grid), dim3(256), PX_LDS, stream, feat, center, scale, (const bf16x8*)comps_planes,
                       mean_proj, F, fpad, C, cpad, pca_out, (long)n_cells);
    return hipGetLastError();
}

template <bool SPLIT>
static void ocsvm_launch(const float* pca, int D, const SvmArgs& sa, double* out, int64_t n_cells, dim3 grid, hipStream_t stream)
{
    if (D <= 100)   // the reference's n_components (CAE_improved_modeltrain.py:412) whenever N_train > 100
        hipLaunchKernelGGL((ocsvm_mfma_kernel<25, SPLIT>), grid, dim3(64 * SVMM_WAVES), 0, stream, pca, D, sa, out, (long)n_cells);
    else
        hipLaunchKernelGGL((ocsvm_mfma_kernel<32, SPLIT>), grid, dim3(64 * SVMM_WAVES), 0, stream, pca, D, sa, out, (long)n_cells);
}

hipError_t launch_ocsvm(const float* pca, int D, const double* svT, const double* svn, const double* coef, int nsv_pad, double gamma,
                        double rho, double* dec, int64_t n_cells, hipStream_t stream, void* split_ws)
{
    if (n_cells <= 0) return hipSuccess;
    if (!svT || !svn || !coef || D < 1 || D > 128 || nsv_pad % 16) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((n_cells + SVMM_CELLS - 1) / SVMM_CELLS);
    SvmArgs sa{{svT, svT}, {svn, svn}, {coef, coef}, {nsv_pad, nsv_pad}, {gamma, gamma}, {rho, rho}};
    if (split_ws && n_cells <= DET_SPLIT_MAX_CELLS) {
        ocsvm_launch<true>(pca, D, sa, (double*)split_ws, n_cells, dim3(grid, DET_RANGES, 1), stream);
        hipLaunchKernelGGL(svm_split_sum_kernel, dim3((unsigned)((n_cells + 255) / 256), 1), dim3(256), 0, stream, (const double*)split_ws, rho, rho,
                           dec, dec, (long)n_cells);
        return hipGetLastError();
    }
    ocsvm_launch<false>(pca, D, sa, dec, n_cells, dim3(grid), stream);
    return hipGetLastError();
}

// both detectors of a SMALL call in one launch (and one sum): n_cells <= DET_SPLIT_MAX_CELLS, split_ws = det_split_ws_bytes(C) bytes
hipError_t launch_ocsvm_pair_split(const float* pca, int D, const double* const svT[2], const double* const svn[2],
                                   const double* const coef[2], const int nsv_pad[2], const double gamma[2], const double rho[2],
                                   double* dec0, double* dec1, int64_t n_cells, hipStream_t stream, void* split_ws)
{
    if (n_cells <= 0) return hipSuccess;
    if (!split_ws || n_cells > DET_SPLIT_MAX_CELLS || D < 1 || D > 128 || nsv_pad[0] % 16 || nsv_pad[1] % 16) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((n_cells + SVMM_CELLS - 1) / SVMM_CELLS);
    SvmArgs sa{{svT[0], svT[1]}, {svn[0], svn[1]}, {coef[0], coef[1]}, {nsv_pad[0], nsv_pad[1]}, {gamma[0], gamma[1]}, {rho[0], rho[1]}};
    ocsvm_launch<true>(pca, D, sa, (double*)split_ws, n_cells, dim3(grid, DET_RANGES, 2), stream);
    hipLaunchKernelGGL(svm_split_sum_kernel, dim3((unsigned)((n_cells + 255) / 256), 2), dim3(256), 0, stream, (const double*)split_ws, rho[0], rho[1],
                       dec0, dec1, (long)n_cells);
    return hipGetLastError();
}

hipError_t launch_finalize(const float* errpart, int nparts, int npix, const double* dec_c,
                           const double* dec_m, float* mse, float* mae, double* score_c,
                           double* score_m, int8_t* pred_c, int8_t* pred_m, int64_t n_cells,
                           hipStream_t stream)
{
    if (n_cells <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((n_cells + 255) / 256);
    hipLaunchKernelGGL(finalize_kernel, dim3(grid), dim3(256), 0, stream, errpart, nparts, npix, dec_c,
                       dec_m, mse, mae, score_c, score_m, pred_c, pred_m, (long)n_cells);
    return hipGetLastError();
}

hipError_t launch_synth(uint64_t seed, int64_t first_cell, int64_t n, int npix, float* out,
                        hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const long total = (long)n * npix;
    long blocks = (total + 255) / 256;
    if (blocks > 256L * 32) blocks = 256L * 32;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, seed, (long)first_cell,
                       (long)n, npix, out);
    return hipGetLastError();
}

}  // namespace cs
