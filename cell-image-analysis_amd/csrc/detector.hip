// detector.hip -- the scoring tail of compute_anomaly_scores (improved_detection.py:134-153)
// on device, so a cell's 18 bytes of results are the only thing that leaves the GPU:
//   scaler.transform  (x - center_) / scale_        sklearn _data.py:1715-1718
//   pca.transform     x @ components_.T - mean_proj sklearn _base.py:147-155
//   OneClassSVM       sum_i a_i exp(-g |x - sv_i|^2) - rho   libsvm svm.cpp:461-476, 2818-2838
//   negation / labels improved_detection.py:138-150
// plus the per-cell mean of the squared / absolute error partial sums (:126-127) and the
// counter-based synthetic crop generator used by the benchmark and the parity tests.
#include "common.hpp"

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

    for (int k0 = 0; k0 < fpad; k0 += PCA_KC) {
        for (int idx = tid; idx < PCA_CELLS * PCA_KC; idx += 256) {
            const int c = idx / PCA_KC, k = idx % PCA_KC;
            const int gk = k0 + k;
            const long cell = cell0 + c;
            float v = 0.0f;
            if (cell < n && gk < F) {
                const float t = feat[cell * F + gk] - center[gk];
                v = (float)((double)t / scale[gk]);
            }
            xs[c * PCA_LD + k] = v;
        }
        __syncthreads();
#pragma unroll 2
        for (int ks = 0; ks < PCA_KC / 16; ++ks) {
            f32x4 a[PCA_MT];
#pragma unroll
            for (int m = 0; m < PCA_MT; ++m)
                a[m] = *(const f32x4*)(xs + (m * 16 + li) * PCA_LD + ks * 16 + kq * 4);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int tile = wave + 4 * t;  // wave-uniform
                if (tile < ntiles) {
                    const f32x4 b = *(const f32x4*)(comps + (size_t)(tile * 16 + li) * fpad + k0 + ks * 16 + kq * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int m = 0; m < PCA_MT; ++m)
                            acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][j], b[j], acc[t][m], 0, 0, 0);
                }
            }
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

// ---------------------------------------------------------------- one-class SVM decision
// fp64 throughout (libsvm is double).  Lane <-> cell: a lane keeps its cell's D coordinates as
// doubles in registers; support vectors are wave-uniform and arrive through scalar loads, so
// the inner loop is two fp64 VALU instructions per (cell, sv, coordinate) and nothing else.
// The eight waves of a workgroup share 64 cells and each walk an eighth of the support
// vectors; their partial sums are added in wave order (deterministic).
constexpr int SVMR_WAVES = 8;   // waves per workgroup = ways the support vectors are split

template <int D>
__global__ __launch_bounds__(64 * SVMR_WAVES, 2) void ocsvm_reg_kernel(
    const float* __restrict__ pca, const double* __restrict__ sv /* [nsv_pad][D], zero padded */,
    const double* __restrict__ coef /* [nsv_pad], zero padded */, int nsv, double gamma, double rho,
    double* __restrict__ dec, long n)
{
    __shared__ double part[SVMR_WAVES][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long cell = (long)blockIdx.x * 64 + lane;
    double x[D];
    {
        const long src = cell < n ? cell : n - 1;   // tail lanes recompute the last cell, never stored
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = (double)pca[src * D + d];
    }
    // support vectors are walked two at a time (rows 2j, 2j+1 of the zero-padded table: a padded
    // row has coef 0), each distance split into two independent fma chains: four chains in
    // flight hide the fp64 fma latency and let the two rows' scalar loads overlap.
    const int npair = (nsv + 1) / 2;
    const int per = (npair + SVMR_WAVES - 1) / SVMR_WAVES;
    const int j0 = wave * per, j1 = (j0 + per < npair) ? j0 + per : npair;
    double sum = 0.0;
    for (int j = j0; j < j1; ++j) {
        const double* __restrict__ s0 = sv + (size_t)(2 * j) * D;   // wave-uniform -> scalar loads
        const double* __restrict__ s1 = s0 + D;
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
        for (int d = 0; d < D; d += 2) {
            const double e0 = x[d] - s0[d], e1 = x[d + 1] - s0[d + 1];
            const double f0 = x[d] - s1[d], f1 = x[d + 1] - s1[d + 1];
            a0 = fma(e0, e0, a0); a1 = fma(e1, e1, a1);
            b0 = fma(f0, f0, b0); b1 = fma(f1, f1, b1);
        }
        sum += coef[2 * j] * exp(-gamma * (a0 + a1));
        sum += coef[2 * j + 1] * exp(-gamma * (b0 + b1));
    }
    part[wave][lane] = sum;
    __syncthreads();
    if (wave == 0 && cell < n) {
        double t = part[0][lane];
#pragma unroll
        for (int w = 1; w < SVMR_WAVES; ++w) t += part[w][lane];
        dec[cell] = t - rho;
    }
}

// Generic-D fallback (D <= 128): lane <-> support vector, 16 cells per workgroup in LDS.
constexpr int SVM_CELLS = 16;
constexpr int SVM_MAXD = 128;

__global__ __launch_bounds__(256) void ocsvm_kernel(
    const float* __restrict__ pca, int D, const double* __restrict__ svT,
    const double* __restrict__ coef, int nsv_pad, double gamma, double rho,
    double* __restrict__ dec, long n)
{
    __shared__ double xs[SVM_CELLS * SVM_MAXD];
    __shared__ double red[SVM_CELLS][4];
    const int tid = threadIdx.x;
    const long cell0 = (long)blockIdx.x * SVM_CELLS;
    for (int idx = tid; idx < SVM_CELLS * D; idx += 256) {
        const int c = idx / D, d = idx % D;
        const long cell = cell0 + c;
        xs[c * D + d] = (cell < n) ? (double)pca[cell * D + d] : 0.0;
    }
    __syncthreads();

    double part[SVM_CELLS];
#pragma unroll
    for (int c = 0; c < SVM_CELLS; ++c) part[c] = 0.0;

    for (int i = tid; i < nsv_pad; i += 256) {
        double d2[SVM_CELLS];
#pragma unroll
        for (int c = 0; c < SVM_CELLS; ++c) d2[c] = 0.0;
        for (int d = 0; d < D; ++d) {
            const double s = svT[(size_t)d * nsv_pad + i];
#pragma unroll
            for (int c = 0; c < SVM_CELLS; ++c) {
                const double diff = xs[c * D + d] - s;
                d2[c] = fma(diff, diff, d2[c]);
            }
        }
        const double a = coef[i];
#pragma unroll
        for (int c = 0; c < SVM_CELLS; ++c) part[c] += a * exp(-gamma * d2[c]);
    }

#pragma unroll
    for (int c = 0; c < SVM_CELLS; ++c) {
        double v = part[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((tid & 63) == 0) red[c][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < SVM_CELLS) {
        const long cell = cell0 + tid;
        if (cell < n) dec[cell] = ((red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3])) - rho;
    }
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
    if (errpart) {
        float s2 = 0.0f, s1 = 0.0f;
        for (int p = 0; p < nparts; ++p) {
            s2 += errpart[(i * nparts + p) * 2 + 0];
            s1 += errpart[(i * nparts + p) * 2 + 1];
        }
        if (mse) mse[i] = s2 / (float)npix;
        if (mae) mae[i] = s1 / (float)npix;
    }
    if (dec_c) {
        const double d = dec_c[i];
        if (score_c) score_c[i] = -d;                 // improved_detection.py:149
        if (pred_c) pred_c[i] = (d > 0) ? 1 : -1;     // svm.cpp:2838
    }
    if (dec_m) {
        const double d = dec_m[i];
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

hipError_t launch_ocsvm(const float* pca, int D, const double* sv, const double* svT, const double* coef,
                        int nsv, int nsv_pad, double gamma, double rho, double* dec, int64_t n_cells,
                        hipStream_t stream)
{
    if (n_cells <= 0) return hipSuccess;
    if (D == 100) {   // the reference's n_components (CAE_improved_modeltrain.py:412) whenever N_train > 100
        const unsigned grid = (unsigned)((n_cells + 63) / 64);
        hipLaunchKernelGGL(ocsvm_reg_kernel<100>, dim3(grid), dim3(64 * SVMR_WAVES), 0, stream, pca, sv, coef, nsv, gamma, rho,
                           dec, (long)n_cells);
        return hipGetLastError();
    }
    if (D > SVM_MAXD || nsv_pad % 256) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((n_cells + SVM_CELLS - 1) / SVM_CELLS);
    hipLaunchKernelGGL(ocsvm_kernel, dim3(grid), dim3(256), 0, stream, pca, D, svT, coef, nsv_pad, gamma,
                       rho, dec, (long)n_cells);
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
