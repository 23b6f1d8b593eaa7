// conv_generic.hip -- the conv layers of a NON-reference autoencoder of the same layer grammar
// (create_improved_autoencoder(input_shape) is generic in its input size, CAE_improved_modeltrain.py:184;
// BASELINE.json configs[4] / SURVEY.md Appendix A.2: 128x128 crops, channels 32-64-128 | 128-64-32-1).
//
// Same arithmetic as conv_mfma.hip -- 3x3 'same' conv as an exact-fp32 MFMA implicit GEMM
// (v_mfma_f32_16x16x4_f32: M = 16 consecutive pixels of a conv row, N = 16 output channels, K = 9 taps x cin),
// bias -> ReLU -> BatchNorm(x*s+t) [-> 2x2 max-pool], or bias -> sigmoid for the last conv; UpSampling2D
// is a >>1 in the input address -- but with run-time shapes: the weights are NOT register resident (cin
// up to 128+ gives K = 1152+), each B fragment is read once from the HWIO kernel (L2 resident) and reused
// for every tile of the strip, whose accumulators (<= 16 tiles) stay in registers.  The reference
// architecture never takes this path (it has its own tuned kernels); this one trades speed for shape
// freedom: any grid with W % 16 == 0 and W <= 128, any cin in {1, 4k}, any cout.
#include "common.hpp"

namespace cs {

namespace {

constexpr int GEN_SR = 2;   // conv rows per work item (a pool window's two rows)

struct GenArgs {
    const float* in;     // stored input [n][Hs][Ws][cin]  (Hs = H/2 when ups)
    const float* w;      // HWIO [3][3][cin][cout]
    const float* ep;     // [3][cout]: bias, bn scale, bn shift  (scale/shift unused for EPI sigmoid)
    float* out;          // [n][Ho][Wo][cout]
    long n;
    int H, W, cin, cout; // conv grid
    int ups, epi;        // epi: GEN_EPI_*
    int ps;              // LDS pixel stride in floats
};

// TPS = tiles per strip = GEN_SR * W / 16
template <int TPS>
__global__ __launch_bounds__(256) void conv_generic_kernel(GenArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float strip[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int H = g.H, W = g.W, cin = g.cin, cout = g.cout, ps = g.ps;
    const int Hs = g.ups ? H / 2 : H, Ws = g.ups ? W / 2 : W;
    const int R = g.ups ? 3 : 4, WP = Ws + 2;
    constexpr int TPR = TPS / GEN_SR;                     // tiles per conv row
    const int nstrip = H / GEN_SR;
    const int ncb = (cout + 63) / 64;                     // blocks of 4 slices

    const long items = g.n * nstrip * ncb;
    for (long item = blockIdx.x; item < items; item += gridDim.x) {
        const int cb = (int)(item % ncb);
        const long cs_ = item / ncb;
        const int y0 = (int)(cs_ % nstrip) * GEN_SR;
        const long cell = cs_ / nstrip;
        const float* src = g.in + (size_t)cell * Hs * Ws * cin;
        const int ybase = g.ups ? (y0 / 2 - 1) : (y0 - 1);

        __syncthreads();                                  // previous item's readers are done
        if (cin % 4 == 0) {
            const int c4n = cin / 4;
            for (int e = tid; e < R * WP * c4n; e += 256) {
                const int c4 = e % c4n, pix = e / c4n;
                const int r = pix / WP, c = pix - r * WP;
                const int sy = ybase + r, sx = c - 1;
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) v = *(const f32x4*)(src + ((size_t)sy * Ws + sx) * cin + 4 * c4);
                *(f32x4*)(strip + pix * ps + 4 * c4) = v;
            }
        } else {
            for (int e = tid; e < R * WP * cin; e += 256) {
                const int ci = e % cin, pix = e / cin;
                const int r = pix / WP, c = pix - r * WP;
                const int sy = ybase + r, sx = c - 1;
                float v = 0.0f;
                if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) v = src[((size_t)sy * Ws + sx) * cin + ci];
                strip[pix * ps + ci] = v;
            }
        }
        __syncthreads();

        // waves = (16-channel slice) x (tile group): with fewer than 4 slices left in this block of 64
        // channels the spare waves split the strip's tile columns instead of idling (cout = 32: 2 x 2,
        // cout = 1: 1 x 4); a tile and the one below it stay in the same wave (pooling)
        const int nsl_blk = min(4, (cout - cb * 64 + 15) / 16);
        const int nmg = nsl_blk >= 3 ? 1 : (nsl_blk == 2 ? 2 : 4);
        const int slice = wave % (4 / nmg), mg = wave / (4 / nmg);
        const int co = (cb * 4 + slice) * 16 + li;
        const bool live = (cb * 4 + slice) * 16 < cout;   // wave-uniform: this slice exists
        auto mine = [&](int t) { return ((t % TPR) & (nmg - 1)) == mg; };   // wave-uniform
        if (live) {
            f32x4 acc[TPS];
#pragma unroll
            for (int t = 0; t < TPS; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (cin % 16 == 0) {
                // K walked as (tap, 16-channel block, j): lane (li, kq) reads the 4 channels 16 q + 4 kq + j of its
                // pixel with one ds_read_b128 and uses them in 4 successive MFMAs (MFMA j contracts the channels
                // {16 q + 4 kq' + j}); the matching B rows are w[tap][16 q + 4 kq + j][co].  Per tap the tiles' LDS
                // offsets are computed once; the next block's B values are fetched while this block's MFMAs run.
                const int nq = cin / 16;
                const float* wl = g.w + co;
                const bool cok = co < cout;
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                    int base[TPS];
#pragma unroll
                    for (int t = 0; t < TPS; ++t) {
                        const int py = t / TPR, px = (t % TPR) * 16 + li;
                        int r, c;
                        if (g.ups) {
                            r = ((y0 + py + dy) >> 1) - ybase;
                            c = ((px + dx) >> 1) + 1;
                        } else {
                            r = py + dy + 1;
                            c = px + dx + 1;
                        }
                        base[t] = (r * WP + c) * ps + 4 * kq;
                    }
                    const float* wt = wl + (size_t)(tap * cin + 4 * kq) * cout;
                    float bn[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) bn[j] = cok ? wt[(size_t)j * cout] : 0.0f;
                    for (int q = 0; q < nq; ++q) {
                        float b[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) b[j] = bn[j];
                        if (q + 1 < nq) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) bn[j] = cok ? wt[(size_t)(16 * (q + 1) + j) * cout] : 0.0f;
                        }
#pragma unroll
                        for (int t = 0; t < TPS; ++t) {
                            if (!mine(t)) continue;
                            const f32x4 a = *(const f32x4*)(strip + base[t] + 16 * q);
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[t], 0, 0, 0);
                        }
                    }
                }
            } else {
            const int ksteps = (9 * cin + 3) / 4;         // cin == 1: 9 -> 3 steps, zero padded
            for (int s = 0; s < ksteps; ++s) {
                const int k = 4 * s + kq;                 // this lane's K index: k = tap * cin + ci
                const bool kv = k < 9 * cin;
                const int tap = kv ? k / cin : 0, ci = kv ? k - tap * cin : 0;
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                const float b = (kv && co < cout) ? g.w[((size_t)tap * cin + ci) * cout + co] : 0.0f;
#pragma unroll
                for (int t = 0; t < TPS; ++t) {
                    if (!mine(t)) continue;
                    const int py = t / TPR, px = (t % TPR) * 16 + li;
                    int r, c;
                    if (g.ups) {
                        r = ((y0 + py + dy) >> 1) - ybase;
                        c = ((px + dx) >> 1) + 1;
                    } else {
                        r = py + dy + 1;
                        c = px + dx + 1;
                    }
                    const float a = kv ? strip[(r * WP + c) * ps + ci] : 0.0f;
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
                }
            }
            }
            // D: lane = channel li of the slice, registers = pixels 4 kq .. 4 kq + 3 of the tile
            if (co < cout) {
                const float bias = g.epi == GEN_EPI_PLAIN ? 0.0f : g.ep[co];
                if (g.epi == GEN_EPI_RELU || g.epi == GEN_EPI_PLAIN) {
                    const bool relu = g.epi == GEN_EPI_RELU;
                    float* o = g.out + ((size_t)cell * H + y0) * W * cout;
#pragma unroll
                    for (int t = 0; t < TPS; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (!mine(t)) continue;
                            const int py = t / TPR, px = (t % TPR) * 16 + 4 * kq + r;
                            const float z = acc[t][r] + bias;
                            o[((size_t)py * W + px) * cout + co] = relu ? fmaxf(z, 0.0f) : z;
                        }
                } else if (g.epi == GEN_EPI_SIGMOID) {
                    float* o = g.out + ((size_t)cell * H + y0) * W * cout;
#pragma unroll
                    for (int t = 0; t < TPS; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (!mine(t)) continue;
                            const int py = t / TPR, px = (t % TPR) * 16 + 4 * kq + r;
                            const float z = acc[t][r] + bias;
                            o[((size_t)py * W + px) * cout + co] = 1.0f / (1.0f + expf(-z));
                        }
                } else {
                    const float bns = g.ep[cout + co], bnt = g.ep[2 * cout + co];
                    auto post = [&](float v) { v += bias; v = fmaxf(v, 0.0f); return fmaf(v, bns, bnt); };
                    if (g.epi == GEN_EPI_BN) {
                        float* o = g.out + ((size_t)cell * H + y0) * W * cout;
#pragma unroll
                        for (int t = 0; t < TPS; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if (!mine(t)) continue;
                                const int py = t / TPR, px = (t % TPR) * 16 + 4 * kq + r;
                                o[((size_t)py * W + px) * cout + co] = post(acc[t][r]);
                            }
                    } else {   // GEN_EPI_BN_POOL: rows y0, y0+1 are tiles t and t + TPR
                        float* o = g.out + ((size_t)cell * (H / 2) + y0 / 2) * (W / 2) * cout;
#pragma unroll
                        for (int t = 0; t < TPR; ++t)
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                if (!mine(t)) continue;
                                const float m0 = fmaxf(post(acc[t][2 * h]), post(acc[t][2 * h + 1]));
                                const float m1 = fmaxf(post(acc[t + TPR][2 * h]), post(acc[t + TPR][2 * h + 1]));
                                o[(size_t)(t * 8 + 2 * kq + h) * cout + co] = fmaxf(m0, m1);
                            }
                    }
                }
            }
        }
    }
}

// per-cell squared / absolute error partial sums of a reconstruction: errpart[n][4][2], wave w of the
// workgroup owns partial w (fixed order, deterministic), as conv7_err_kernel lays them out
__global__ __launch_bounds__(256) void recon_err_kernel(const float* __restrict__ recon, const float* __restrict__ x, int npix,
                                                        float* __restrict__ errpart)
{
    const long cell = blockIdx.x;
    const float* r = recon + (size_t)cell * npix;
    const float* t = x + (size_t)cell * npix;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (npix + 3) / 4;
    float s2 = 0.0f, s1 = 0.0f;
    for (int p = wave * per + lane; p < min((wave + 1) * per, npix); p += 64) {
        const float d = t[p] - r[p];
        s2 += d * d;
        s1 += fabsf(d);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s2 += __shfl_xor(s2, m);
        s1 += __shfl_xor(s1, m);
    }
    if (lane == 0) {
        errpart[(cell * 4 + wave) * 2 + 0] = s2;
        errpart[(cell * 4 + wave) * 2 + 1] = s1;
    }
}

}  // namespace

int conv_generic_supported(int H, int W, int cin, int cout, char* why, size_t why_len)
{
    const char* msg = nullptr;
    if (W % 16 != 0 || W < 16 || W > 128 || H % 2 != 0) msg = "conv grid must have W in {16,32,...,128} (multiple of 16) and even H";
    else if (!(cin == 1 || cin % 4 == 0)) msg = "cin must be 1 or a multiple of 4";
    else if (cout < 1) msg = "cout must be positive";
    else if ((size_t)4 * (W + 2) * (cin + 4) * sizeof(float) > 160 * 1024) msg = "input strip exceeds the 160 KB LDS";
    if (msg) {
        if (why) snprintf(why, why_len, "%s (grid %dx%d, cin %d, cout %d)", msg, H, W, cin, cout);
        return 0;
    }
    return 1;
}

hipError_t launch_conv_generic(const float* in, const float* w_hwio, const float* ep, float* out, int64_t n, int H, int W, int cin,
                               int cout, int ups, int epi, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    if (!conv_generic_supported(H, W, cin, cout, nullptr, 0)) return hipErrorInvalidValue;
    GenArgs g;
    g.in = in; g.w = w_hwio; g.ep = ep; g.out = out; g.n = n; g.H = H; g.W = W; g.cin = cin; g.cout = cout;
    g.ups = ups; g.epi = epi;
    g.ps = cin == 1 ? 1 : cin + 4;                        // odd number of 16-B slots per pixel
    const int Ws = ups ? W / 2 : W, R = ups ? 3 : 4;
    const size_t lds = (size_t)R * (Ws + 2) * g.ps * sizeof(float);
    const long items = (long)n * (H / GEN_SR) * ((cout + 63) / 64);
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = (unsigned)(items < (long)cus * 8 ? items : (long)cus * 8);
    const int tps = GEN_SR * W / 16;
    hipError_t e = hipSuccess;
#define GEN_LAUNCH(T)                                                                                                     \
    do {                                                                                                                  \
        e = hipFuncSetAttribute((const void*)conv_generic_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) hipLaunchKernelGGL(conv_generic_kernel<T>, dim3(grid), dim3(256), lds, stream, g);           \
    } while (0)
    switch (tps) {
        case 2: GEN_LAUNCH(2); break;
        case 4: GEN_LAUNCH(4); break;
        case 6: GEN_LAUNCH(6); break;
        case 8: GEN_LAUNCH(8); break;
        case 10: GEN_LAUNCH(10); break;
        case 12: GEN_LAUNCH(12); break;
        case 14: GEN_LAUNCH(14); break;
        case 16: GEN_LAUNCH(16); break;
        default: return hipErrorInvalidValue;
    }
#undef GEN_LAUNCH
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

hipError_t launch_recon_err(const float* recon, const float* x, int64_t n, int npix, float* errpart, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(recon_err_kernel, dim3((unsigned)n), dim3(256), 0, stream, recon, x, npix, errpart);
    return hipGetLastError();
}

}  // namespace cs
