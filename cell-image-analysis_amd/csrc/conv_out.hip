// conv_out.hip -- the output conv of the autoencoder fused with the reconstruction error:
// UpSampling2D(2x2) -> Conv2D(1, 3x3, 'same', sigmoid)          CAE_improved_modeltrain.py:214-216
// -> sum (x - r)^2 and sum |x - r| per cell                      improved_detection.py:126-127
// The reconstruction itself is never written to HBM unless a debug pointer asks for it.
//
// cout = 1, so this is a per-pixel 288-term dot product, not a GEMM: a VALU kernel.
// One thread owns one pixel of the stored (32x32) a6 grid = a 2x2 block of output pixels.
// Nearest upsampling makes the 3x3 taps of output (2y+a, 2x+b) fall on only 2x2 stored
// pixels, rows y+a-1..y+a and columns x+b-1..x+b, several taps sharing one pixel; the taps
// that share a pixel are pre-summed on the host into 16 effective weight vectors
// W_eff[a][b][ry][rx][cin] (an exact algebraic identity; it changes only the order of fp32
// roundings), so an output costs 4x32 FMAs instead of 9x32.  The zero padding of the
// upsampled grid coincides with the zero halo of the stored grid.  The weight indices are
// wave-uniform, so W_eff is fetched with scalar loads into SGPRs.
#include "common.hpp"

namespace cs {

namespace {
constexpr int C7_CIN = 32;
constexpr int C7_HS = 32, C7_WS = 32;        // stored a6 size
constexpr int C7_SR = 8;                     // a6 rows per workgroup (16 output rows)
constexpr int C7_R = C7_SR + 2, C7_WP = C7_WS + 2, C7_PS = C7_CIN + 4;
constexpr int C7_LDS = C7_R * C7_WP * C7_PS * 4;
constexpr int C7_NSTRIP = C7_HS / C7_SR;     // 4 partial sums per cell


template <bool TRAIN>
__global__ __launch_bounds__(256, 2) void conv7_err_kernel(
    const float* __restrict__ a6, const float* __restrict__ x, const float* __restrict__ weff /*[16][32]*/,
    const float* __restrict__ b7p, float* __restrict__ errpart, float* __restrict__ recon, long n_cells,
    float* __restrict__ dz, float* __restrict__ dzsum_part)
{
    const float b7 = b7p[0];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float (*red)[4] = (float (*)[4])(smem + C7_LDS);  // 2 (3: TRAIN) x 4 wave sums, after the strip
    const int tid = threadIdx.x;
    const long total = n_cells * C7_NSTRIP;
    // The kernel is HBM-bound (it streams a6 once, 172 KB/cell with the row halo), so the strip of
    // item i+1 is loaded into registers before the arithmetic of item i and written to LDS after
    // it: loads stay in flight the whole time instead of only between items.
    // Staging: a thread owns one interior 16-byte element (pixel tid >> 3, channel quad tid & 7) of every staged row
    // -- 32 x 8 = 256 per row -- with constant offsets; the halo columns are zeroed once and never rewritten, rows
    // outside the image are loaded from a clamped row and zeroed at the LDS write.  (VALU instructions are on this
    // kernel's critical path: the div/mod index arithmetic of an element-wise mapping was a quarter of them.)
    static_assert(C7_WS * (C7_CIN / 4) == 256, "one interior element per thread and row");
    constexpr int NLD = C7_R;
    const int goff = (tid >> 3) * C7_CIN + (tid & 7) * 4;
    const int loff = (((tid >> 3) + 1) * C7_PS + (tid & 7) * 4) * 4;      // bytes inside a strip row
    f32x4 stg[NLD];
    auto issue = [&](long item) {
        const long cell = item / C7_NSTRIP;
        const int y0 = (int)(item % C7_NSTRIP) * C7_SR;
        const float* src = a6 + (size_t)cell * C7_HS * C7_WS * C7_CIN + goff;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            int sy = y0 - 1 + k;
            sy = sy < 0 ? 0 : (sy > C7_HS - 1 ? C7_HS - 1 : sy);
            stg[k] = *(const f32x4*)(src + sy * (C7_WS * C7_CIN));
        }
    };
    for (int i = tid; i < C7_LDS / 16; i += 256) ((f32x4*)smem)[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if ((long)blockIdx.x < total) issue(blockIdx.x);
    __syncthreads();
    for (long item = blockIdx.x; item < total; item += gridDim.x) {
        const long cell = item / C7_NSTRIP;
        const int strip = (int)(item % C7_NSTRIP);
        const int y0 = strip * C7_SR;  // first a6 row of the strip

#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int sy = y0 - 1 + k;
            f32x4 v = stg[k];
            if (sy < 0 || sy >= C7_HS) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            *(f32x4*)(smem + k * (C7_WP * C7_PS * 4) + loff) = v;
        }
        __syncthreads();
        if (item + gridDim.x < total) issue(item + gridDim.x);

        const int ly = tid >> 5, lx = tid & 31;  // a6 pixel (y0 + ly, lx)
        float acc[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll 1
        for (int q = 0; q < C7_CIN / 4; ++q) {
            f32x4 nb[3][3];
#pragma unroll
            for (int ry = 0; ry < 3; ++ry)
#pragma unroll
                for (int rx = 0; rx < 3; ++rx)
                    nb[ry][rx] = *(const f32x4*)(smem + (((ly + ry) * C7_WP + lx + rx) * C7_PS + q * 4) * 4);
            // output (2y+a, 2x+b) reads stored pixels (y+a-1+ry, x+b-1+rx), ry,rx in {0,1}
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int ry = 0; ry < 2; ++ry)
#pragma unroll
                        for (int rx = 0; rx < 2; ++rx) {
                            const int e = ((a * 2 + b) * 2 + ry) * 2 + rx;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[a][b] = fmaf(nb[a + ry][b + rx][j], weff[e * C7_CIN + q * 4 + j], acc[a][b]);
                        }
        }

        // sigmoid, error terms
        float s2 = 0.0f, s1 = 0.0f, sz = 0.0f;
        const float kz = 2.0f / (float)(n_cells * 64 * 64);        // TRAIN: d mean((out - x)^2) / d out
        const int Y = 2 * (y0 + ly), X = 2 * lx;
        const float* xc = x + (size_t)cell * 64 * 64;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float v = acc[a][b] + b7;
                const float r = 1.0f / (1.0f + expf(-v));
                const float xv = xc[(Y + a) * 64 + X + b];
                const float d = xv - r;
                s2 = fmaf(d, d, s2);
                s1 += fabsf(d);
                if (recon) recon[(size_t)cell * 64 * 64 + (Y + a) * 64 + X + b] = r;
                if constexpr (TRAIN) {
                    const float g = kz * (r - xv) * r * (1.0f - r);     // loss_dz_kernel's expression
                    dz[(size_t)cell * 64 * 64 + (Y + a) * 64 + X + b] = g;
                    sz += g;
                }
            }
        // deterministic block reduction: wave shuffle tree, then 4 wave sums in order
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s2 += __shfl_down(s2, off, 64);
            s1 += __shfl_down(s1, off, 64);
            if constexpr (TRAIN) sz += __shfl_down(sz, off, 64);
        }
        if ((tid & 63) == 0) { red[0][tid >> 6] = s2; red[1][tid >> 6] = s1; if constexpr (TRAIN) red[2][tid >> 6] = sz; }
        __syncthreads();
        if (tid == 0) {
            errpart[(cell * C7_NSTRIP + strip) * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            errpart[(cell * C7_NSTRIP + strip) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
            if constexpr (TRAIN) dzsum_part[cell * C7_NSTRIP + strip] = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
        }
        __syncthreads();
    }
}
}  // namespace

// W_eff[a][b][ry][rx][ci] = sum of the taps (dy,dx) whose upsampled source pixel is stored
// pixel (y+a-1+ry, x+b-1+rx):  a=0: ry=0 <- dy=-1, ry=1 <- dy in {0,+1};  a=1: ry=0 <- dy in {-1,0}, ry=1 <- dy=+1.
void conv7_effective_weights(const float* w7_hwio, float* weff /*[16][32]*/)
{
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int ry = 0; ry < 2; ++ry)
                for (int rx = 0; rx < 2; ++rx)
                    for (int ci = 0; ci < C7_CIN; ++ci) {
                        float sum = 0.0f;
                        for (int dy = -1; dy <= 1; ++dy)
                            for (int dx = -1; dx <= 1; ++dx)
                                if (((a + dy) >> 1) + 1 == a + ry && ((b + dx) >> 1) + 1 == b + rx)
                                    sum += w7_hwio[((dy + 1) * 3 + (dx + 1)) * C7_CIN + ci];  // HWIO, cout = 1
                        weff[((((a * 2 + b) * 2 + ry) * 2 + rx)) * C7_CIN + ci] = sum;
                    }
}

template <bool TRAIN>
static hipError_t launch_conv7_err_t(const float* a6, const float* x, const float* weff_dev, const float* b7_dev, float* errpart,
                                     float* recon, float* dz, float* dzsum_part, int64_t n_cells, hipStream_t stream)
{
    constexpr int LDS = C7_LDS + 48;
    static int resident = 0;   // persistent grid = what the chip holds at once (see conv_mfma.hip)
    if (!resident) {
        hipError_t e = hipFuncSetAttribute((const void*)conv7_err_kernel<TRAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        int dev = 0, cus = 0, per_cu = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv7_err_kernel<TRAIN>, 256, LDS);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        resident = cus * per_cu;
    }
    const long total = (long)n_cells * C7_NSTRIP;
    if (total <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(total < resident ? total : resident);
    hipLaunchKernelGGL(conv7_err_kernel<TRAIN>, dim3(grid), dim3(256), LDS, stream, a6, x, weff_dev, b7_dev, errpart, recon,
                       (long)n_cells, dz, dzsum_part);
    return hipGetLastError();
}

hipError_t launch_conv7_err(const float* a6, const float* x, const float* weff_dev, const float* b7_dev,
                            float* errpart, float* recon, int64_t n_cells, hipStream_t stream)
{
    return launch_conv7_err_t<false>(a6, x, weff_dev, b7_dev, errpart, recon, nullptr, nullptr, n_cells, stream);
}

hipError_t launch_conv7_err_train(const float* a6, const float* x, const float* weff_dev, const float* b7_dev, float* errpart,
                                  float* recon, float* dz, float* dzsum_part, int64_t n_cells, hipStream_t stream)
{
    return launch_conv7_err_t<true>(a6, x, weff_dev, b7_dev, errpart, recon, dz, dzsum_part, n_cells, stream);
}

}  // namespace cs
