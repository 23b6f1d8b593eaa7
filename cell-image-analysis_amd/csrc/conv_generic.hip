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

#include <cstdlib>

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

// ---- version 2 of the MFMA path (cin % 16 == 0, W in {16, 32, 64, 128}) -----------------------------------------------------
// The first kernel stages two conv rows per item and lets every wave re-fetch its B fragments for 2 x W / 16 tiles: at cin = 128
// that is one L2 load per four MFMAs and 0.27 of the matrix peak.  Here an item is a strip of SR rows chosen so that a wave owns
// TPW = 8 or 16 tiles (all of them live accumulators): a B fragment fetched once feeds 8-16 MFMAs, and a 512-thread workgroup
// (8 waves = nslw 16-filter slices x nmg tile groups) keeps two waves on every SIMD.  Tiles are handed out in vertical pairs
// (rows 2 rp, 2 rp + 1 of one 16-pixel column block) so that the 2x2 max-pool stays inside a wave.
struct Gen2Args {
    const float* in; const float* w; const float* ep; float* out;
    long n;
    int H, W, cin, cout, ups, epi, ps;
    int SR, nmg, nslw;          // strip rows, tile groups, slices per workgroup pass (nmg * nslw = 8)
};

template <int TPW>
__global__ __launch_bounds__(512, 2) void conv_generic2_kernel(Gen2Args g)
{
    extern __shared__ __attribute__((aligned(16))) float strip[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int H = g.H, W = g.W, cin = g.cin, cout = g.cout, ps = g.ps, SR = g.SR;
    const int Hs = g.ups ? H / 2 : H, Ws = g.ups ? W / 2 : W;
    const int R = g.ups ? SR / 2 + 2 : SR + 2, WP = Ws + 2;
    const int TPR = W / 16;                               // tiles per conv row (1, 2, 4, 8)
    const int nstrip = H / SR;
    const int cpb = g.nslw * 16;                          // filters per workgroup pass
    const int ncb = (cout + cpb - 1) / cpb;
    const int slice = wave % g.nslw, mg = wave / g.nslw;
    constexpr int PPW = TPW / 2;                          // vertical tile pairs per wave

    const long items = g.n * nstrip * ncb;
    for (long item = blockIdx.x; item < items; item += gridDim.x) {
        const int cb = (int)(item % ncb);
        const long cs_ = item / ncb;
        const int y0 = (int)(cs_ % nstrip) * SR;
        const long cell = cs_ / nstrip;
        const float* src = g.in + (size_t)cell * Hs * Ws * cin;
        const int ybase = g.ups ? (y0 / 2 - 1) : (y0 - 1);

        __syncthreads();                                  // previous item's readers are done
        {
            const int c4n = cin / 4;
            for (int e = tid; e < R * WP * c4n; e += 512) {
                const int c4 = e % c4n, pix = e / c4n;
                const int r = pix / WP, c = pix - r * WP;
                const int sy = ybase + r, sx = c - 1;
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) v = *(const f32x4*)(src + ((size_t)sy * Ws + sx) * cin + 4 * c4);
                *(f32x4*)(strip + pix * ps + 4 * c4) = v;
            }
        }
        __syncthreads();

        const int co = cb * cpb + slice * 16 + li;
        const bool live = cb * cpb + slice * 16 < cout;   // wave-uniform: this slice exists
        if (!live) continue;
        const bool cok = co < cout;
        // this wave's tiles: pair i -> pair index pi = mg + nmg * i of the strip = (row pair pi / TPR, column block pi % TPR);
        // tile 2 i is its upper row, 2 i + 1 the lower
        f32x4 acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        int tpy[PPW], tpx[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pi = mg + g.nmg * i;
            tpy[i] = 2 * (pi / TPR);
            tpx[i] = (pi % TPR) * 16;
        }
        const int nq = cin / 16;
        const float* wl = g.w + co;
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            int base[TPW];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int py = tpy[t >> 1] + (t & 1), px = tpx[t >> 1] + li;
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
                // one LDS read ahead of the MFMAs that consume it, pinned: without the pins the scheduler hoists all TPW reads
                // (4 VGPRs each) above the first MFMA and the 16-tile form spills
                f32x4 a = *(const f32x4*)(strip + base[0] + 16 * q);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    f32x4 an = a;
                    if (t + 1 < TPW) an = *(const f32x4*)(strip + base[t + 1] + 16 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[t], 0, 0, 0);
                    a = an;
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
            }
        }
        // D: lane = channel li of the slice, registers = pixels 4 kq .. 4 kq + 3 of the tile
        if (cok) {
            const float bias = g.epi == GEN_EPI_PLAIN ? 0.0f : g.ep[co];
            if (g.epi == GEN_EPI_BN_POOL) {
                const float bns = g.ep[cout + co], bnt = g.ep[2 * cout + co];
                auto post = [&](float v) { v += bias; v = fmaxf(v, 0.0f); return fmaf(v, bns, bnt); };
                float* o = g.out + ((size_t)cell * (H / 2) + y0 / 2) * (W / 2) * cout + co;
#pragma unroll
                for (int i = 0; i < PPW; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float m0 = fmaxf(post(acc[2 * i][2 * h]), post(acc[2 * i][2 * h + 1]));
                        const float m1 = fmaxf(post(acc[2 * i + 1][2 * h]), post(acc[2 * i + 1][2 * h + 1]));
                        o[((size_t)(tpy[i] / 2) * (W / 2) + tpx[i] / 2 + 2 * kq + h) * cout] = fmaxf(m0, m1);
                    }
            } else {
                const float bns = g.epi == GEN_EPI_BN ? g.ep[cout + co] : 1.0f, bnt = g.epi == GEN_EPI_BN ? g.ep[2 * cout + co] : 0.0f;
                float* o = g.out + ((size_t)cell * H + y0) * W * cout + co;
#pragma unroll
                for (int t = 0; t < TPW; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int py = tpy[t >> 1] + (t & 1), px = tpx[t >> 1] + 4 * kq + r;
                        const float z = acc[t][r] + bias;
                        float v;
                        if (g.epi == GEN_EPI_BN) v = fmaf(fmaxf(z, 0.0f), bns, bnt);
                        else if (g.epi == GEN_EPI_RELU) v = fmaxf(z, 0.0f);
                        else if (g.epi == GEN_EPI_SIGMOID) v = 1.0f / (1.0f + expf(-z));
                        else v = z;
                        o[((size_t)py * W + px) * cout] = v;
                    }
            }
        }
    }
}

// ---- the 1-filter last conv behind an UpSampling2D, folded: a thread owns one STORED pixel = a 2x2 block of outputs ---------
// The four outputs read the same 3x3 stored neighbourhood; with the effective 2x2 kernels of the four phases (pack_generic_folded
// with cout = 1: [phase][tap][cin]) a block costs 9 x cin / 4 activation reads + 16 x cin multiply-adds instead of 4 x 9 x cin.
__global__ __launch_bounds__(256) void conv_last_folded_kernel(GenArgs g, int SRS /* stored rows per strip */)
{
    extern __shared__ __attribute__((aligned(16))) float strip[];
    const int tid = threadIdx.x;
    const int H = g.H, W = g.W, cin = g.cin, ps = g.ps;
    const int Hs = H / 2, Ws = W / 2;
    const int R = SRS + 2, WP = Ws + 2;
    float* wl = strip + R * WP * ps;                      // W_eff [4 phases][4 taps][cin]
    for (int e = tid; e < 16 * cin; e += 256) wl[e] = g.w[e];
    const float bias = g.ep[0];
    const int nstrip = Hs / SRS, c4n = cin / 4;
    for (long item = blockIdx.x; item < g.n * nstrip; item += gridDim.x) {
        const int ys0 = (int)(item % nstrip) * SRS;
        const long cell = item / nstrip;
        const float* src = g.in + (size_t)cell * Hs * Ws * cin;
        __syncthreads();
        for (int e = tid; e < R * WP * c4n; e += 256) {
            const int c4 = e % c4n, pix = e / c4n;
            const int r = pix / WP, c = pix - r * WP;
            const int sy = ys0 - 1 + r, sx = c - 1;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) v = *(const f32x4*)(src + ((size_t)sy * Ws + sx) * cin + 4 * c4);
            *(f32x4*)(strip + pix * ps + 4 * c4) = v;
        }
        __syncthreads();
        for (int p = tid; p < SRS * Ws; p += 256) {
            const int ys = p / Ws, xs = p - ys * Ws;
            f32x4 s[4];                                   // one 4-lane partial sum per phase
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) s[ph] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            for (int c4 = 0; c4 < c4n; ++c4) {
                f32x4 nb[3][3];                           // the 3x3 stored neighbourhood, these 4 channels
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) nb[i][j] = *(const f32x4*)(strip + ((ys + i) * WP + xs + j) * ps + 4 * c4);
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) {
                    const int a = ph >> 1, b = ph & 1;
#pragma unroll
                    for (int tap = 0; tap < 4; ++tap) {
                        const f32x4 wv = *(const f32x4*)(wl + (ph * 4 + tap) * cin + 4 * c4);
                        const f32x4 av = nb[a + (tap >> 1)][b + (tap & 1)];
#pragma unroll
                        for (int k = 0; k < 4; ++k) s[ph][k] = fmaf(av[k], wv[k], s[ph][k]);
                    }
                }
            }
            float* o = g.out + ((size_t)cell * H + 2 * (ys0 + ys)) * W + 2 * xs;
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const float z = ((s[ph][0] + s[ph][1]) + (s[ph][2] + s[ph][3])) + bias;
                o[(size_t)(ph >> 1) * W + (ph & 1)] = 1.0f / (1.0f + expf(-z));
            }
        }
    }
}

// ---- upsample-fed convs with the upsample folded into four 2x2-tap phase convs (4/9 of the multiply-adds) ---------------------
// Nearest x2 upsampling makes the 3x3 taps of output pixel (2y+a, 2x+b) land on only 2x2 stored pixels, so the conv splits into
// four output phases (a,b), each a 2x2-tap conv over the STORED grid with the taps that share a stored pixel pre-summed on the
// host (pack_generic_folded; same algebra as conv_mfma.hip's FOLD form, only the order of fp32 roundings changes).  Tiles are
// phase-pure: 16 stored pixels of a stored row, outputs at stride 2.  A wave owns ALL tiles of the strip for 4 / nmg phases
// (the tile groups of version 2 become phase groups), so a B fragment still feeds TPW MFMAs.
template <int TPW>
__global__ __launch_bounds__(512, 2) void conv_generic2f_kernel(Gen2Args g)
{
    extern __shared__ __attribute__((aligned(16))) float strip[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int H = g.H, W = g.W, cin = g.cin, cout = g.cout, ps = g.ps, SR = g.SR;
    const int Hs = H / 2, Ws = W / 2;
    const int R = SR / 2 + 2, WP = Ws + 2;
    const int TPRs = Ws / 16;                             // tiles per stored row
    const int nstrip = H / SR;
    const int cpb = g.nslw * 16, ncb = (cout + cpb - 1) / cpb;
    const int slice = wave % g.nslw, mg = wave / g.nslw;
    const int nph = 4 / g.nmg;                            // phases per wave

    const long items = g.n * nstrip * ncb;
    for (long item = blockIdx.x; item < items; item += gridDim.x) {
        const int cb = (int)(item % ncb);
        const long cs_ = item / ncb;
        const int y0 = (int)(cs_ % nstrip) * SR;
        const long cell = cs_ / nstrip;
        const float* src = g.in + (size_t)cell * Hs * Ws * cin;
        const int ybase = y0 / 2 - 1;

        __syncthreads();
        {
            const int c4n = cin / 4;
            for (int e = tid; e < R * WP * c4n; e += 512) {
                const int c4 = e % c4n, pix = e / c4n;
                const int r = pix / WP, c = pix - r * WP;
                const int sy = ybase + r, sx = c - 1;
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) v = *(const f32x4*)(src + ((size_t)sy * Ws + sx) * cin + 4 * c4);
                *(f32x4*)(strip + pix * ps + 4 * c4) = v;
            }
        }
        __syncthreads();

        const int co = cb * cpb + slice * 16 + li;
        if (cb * cpb + slice * 16 >= cout) continue;      // wave-uniform
        const bool cok = co < cout;
        const int nq = cin / 16;
        const float bias = cok ? g.ep[co] : 0.0f;
        const float bns = (cok && g.epi == GEN_EPI_BN) ? g.ep[cout + co] : 1.0f, bnt = (cok && g.epi == GEN_EPI_BN) ? g.ep[2 * cout + co] : 0.0f;
        for (int pi = 0; pi < nph; ++pi) {
            const int phase = mg * nph + pi, a = phase >> 1, b = phase & 1;
            f32x4 acc[TPW];
#pragma unroll
            for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            for (int tap = 0; tap < 4; ++tap) {
                const int ry = tap >> 1, rx = tap & 1;
                int base[TPW];
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const int ys = t / TPRs, xs = (t % TPRs) * 16 + li;
                    // LDS row 0 is stored row y0/2 - 1: stored (ys + a - 1 + ry, xs + b - 1 + rx) -> LDS (ys + a + ry, xs + b + rx)
                    base[t] = ((ys + a + ry) * WP + xs + b + rx) * ps + 4 * kq;
                }
                const float* wt = g.w + ((size_t)(phase * 4 + tap) * cin + 4 * kq) * cout + co;
                float bn[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bn[j] = cok ? wt[(size_t)j * cout] : 0.0f;
                for (int q = 0; q < nq; ++q) {
                    float bb[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) bb[j] = bn[j];
                    if (q + 1 < nq) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) bn[j] = cok ? wt[(size_t)(16 * (q + 1) + j) * cout] : 0.0f;
                    }
                    f32x4 av = *(const f32x4*)(strip + base[0] + 16 * q);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        f32x4 an = av;
                        if (t + 1 < TPW) an = *(const f32x4*)(strip + base[t + 1] + 16 * q);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bb[j], acc[t], 0, 0, 0);
                        av = an;
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    }
                }
            }
            if (cok) {
                float* o = g.out + ((size_t)cell * H + y0 + a) * W * cout + (size_t)b * cout + co;
#pragma unroll
                for (int t = 0; t < TPW; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ys = t / TPRs, xs = (t % TPRs) * 16 + 4 * kq + r;
                        const float z = fmaxf(acc[t][r] + bias, 0.0f);
                        o[((size_t)(2 * ys) * W + 2 * xs) * cout] = g.epi == GEN_EPI_BN ? fmaf(z, bns, bnt) : z;
                    }
            }
        }
    }
}

// ---- the first conv (cin = 1): K = 9 taps padded to 12 = three MFMAs per tile ------------------------------------------------
// Same tile ownership as version 2; the A operand is one LDS dword per lane and MFMA (lane (pixel li, k = 4 s + kq) reads tap k of its
// pixel; the padded taps k >= 9 carry zero weights and read tap 0), the three B fragments of the wave's slice stay in registers for
// the whole launch.  The first kernel spent ~15 address instructions per MFMA on this layer (16 % of the 128x128 variant's time
// for 1.3 % of its multiply-adds).
template <int TPW>
__global__ __launch_bounds__(512, 2) void conv_generic_c1_kernel(Gen2Args g)
{
    extern __shared__ __attribute__((aligned(16))) float strip[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int H = g.H, W = g.W, cout = g.cout, SR = g.SR;
    const int R = SR + 2, WP = W + 2;
    const int TPR = W / 16;
    const int nstrip = H / SR;
    const int cpb = g.nslw * 16, ncb = (cout + cpb - 1) / cpb;
    const int slice = wave % g.nslw, mg = wave / g.nslw;
    constexpr int PPW = TPW / 2;
    int toff[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int k = 4 * s + kq, kk = k < 9 ? k : 0;
        toff[s] = (kk / 3) * WP + (kk % 3) + li;
    }
    const long items = g.n * nstrip * ncb;
    for (long item = blockIdx.x; item < items; item += gridDim.x) {
        const int cb = (int)(item % ncb);
        const long cs_ = item / ncb;
        const int y0 = (int)(cs_ % nstrip) * SR;
        const long cell = cs_ / nstrip;
        const float* src = g.in + (size_t)cell * H * W;
        __syncthreads();
        for (int e = tid; e < R * WP; e += 512) {
            const int r = e / WP, c = e - r * WP;
            const int sy = y0 - 1 + r, sx = c - 1;
            strip[e] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? src[(size_t)sy * W + sx] : 0.0f;
        }
        __syncthreads();
        const int co = cb * cpb + slice * 16 + li;
        if (cb * cpb + slice * 16 >= cout) continue;      // wave-uniform: this slice does not exist
        const bool cok = co < cout;
        float B[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int k = 4 * s + kq;
            B[s] = (cok && k < 9) ? g.w[(size_t)k * cout + co] : 0.0f;
        }
        f32x4 acc[TPW];
        int tpy[PPW], tpx[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pi = mg + g.nmg * i;
            tpy[i] = 2 * (pi / TPR);
            tpx[i] = (pi % TPR) * 16;
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const float* a = strip + (tpy[t >> 1] + (t & 1)) * WP + tpx[t >> 1];
            f32x4 c = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s = 0; s < 3; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[toff[s]], B[s], c, 0, 0, 0);
            acc[t] = c;
        }
        if (cok) {
            const float bias = g.epi == GEN_EPI_PLAIN ? 0.0f : g.ep[co];
            if (g.epi == GEN_EPI_BN_POOL) {
                const float bns = g.ep[cout + co], bnt = g.ep[2 * cout + co];
                auto post = [&](float v) { v += bias; v = fmaxf(v, 0.0f); return fmaf(v, bns, bnt); };
                float* o = g.out + ((size_t)cell * (H / 2) + y0 / 2) * (W / 2) * cout + co;
#pragma unroll
                for (int i = 0; i < PPW; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float m0 = fmaxf(post(acc[2 * i][2 * h]), post(acc[2 * i][2 * h + 1]));
                        const float m1 = fmaxf(post(acc[2 * i + 1][2 * h]), post(acc[2 * i + 1][2 * h + 1]));
                        o[((size_t)(tpy[i] / 2) * (W / 2) + tpx[i] / 2 + 2 * kq + h) * cout] = fmaxf(m0, m1);
                    }
            } else {
                const float bns = g.epi == GEN_EPI_BN ? g.ep[cout + co] : 1.0f, bnt = g.epi == GEN_EPI_BN ? g.ep[2 * cout + co] : 0.0f;
                float* o = g.out + ((size_t)cell * H + y0) * W * cout + co;
#pragma unroll
                for (int t = 0; t < TPW; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int py = tpy[t >> 1] + (t & 1), px = tpx[t >> 1] + 4 * kq + r;
                        const float z = acc[t][r] + bias;
                        float v;
                        if (g.epi == GEN_EPI_BN) v = fmaf(fmaxf(z, 0.0f), bns, bnt);
                        else if (g.epi == GEN_EPI_RELU) v = fmaxf(z, 0.0f);
                        else if (g.epi == GEN_EPI_SIGMOID) v = 1.0f / (1.0f + expf(-z));
                        else v = z;
                        o[((size_t)py * W + px) * cout] = v;
                    }
            }
        }
    }
}

// ---- the 1-filter last conv (bias -> sigmoid) on the vector ALU ---------------------------------------------------------------
// cout = 1 makes the MFMA form waste 15 of 16 output columns (the padded last conv of the 128x128 variant cost as much matrix time
// as a 64-filter layer).  A thread owns one output pixel: 9 x cin multiply-adds from the staged strip, the kernel read as LDS
// broadcasts.  Same k order as the MFMA form is not needed for parity: the reconstruction bar is 1e-5 absolute on a sigmoid.
__global__ __launch_bounds__(256) void conv_last_generic_kernel(GenArgs g, int SR)
{
    extern __shared__ __attribute__((aligned(16))) float strip[];
    const int tid = threadIdx.x;
    const int H = g.H, W = g.W, cin = g.cin, ps = g.ps;
    const int Hs = g.ups ? H / 2 : H, Ws = g.ups ? W / 2 : W;
    const int R = g.ups ? SR / 2 + 2 : SR + 2, WP = Ws + 2;
    float* wl = strip + R * WP * ps;                      // the kernel [9][cin]
    for (int e = tid; e < 9 * cin; e += 256) wl[e] = g.w[e];
    const float bias = g.ep[0];
    const int nstrip = H / SR, c4n = cin / 4;
    for (long item = blockIdx.x; item < g.n * nstrip; item += gridDim.x) {
        const int y0 = (int)(item % nstrip) * SR;
        const long cell = item / nstrip;
        const float* src = g.in + (size_t)cell * Hs * Ws * cin;
        const int ybase = g.ups ? (y0 / 2 - 1) : (y0 - 1);
        __syncthreads();
        for (int e = tid; e < R * WP * c4n; e += 256) {
            const int c4 = e % c4n, pix = e / c4n;
            const int r = pix / WP, c = pix - r * WP;
            const int sy = ybase + r, sx = c - 1;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) v = *(const f32x4*)(src + ((size_t)sy * Ws + sx) * cin + 4 * c4);
            *(f32x4*)(strip + pix * ps + 4 * c4) = v;
        }
        __syncthreads();
        for (int p = tid; p < SR * W; p += 256) {
            const int py = p / W, px = p - py * W;
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                int r, c;
                if (g.ups) { r = ((y0 + py + dy) >> 1) - ybase; c = ((px + dx) >> 1) + 1; }
                else { r = py + dy + 1; c = px + dx + 1; }
                const float* a = strip + (r * WP + c) * ps;
                const float* wt = wl + tap * cin;
                for (int c4 = 0; c4 < c4n; ++c4) {
                    const f32x4 av = *(const f32x4*)(a + 4 * c4), wv = *(const f32x4*)(wt + 4 * c4);
                    s0 = fmaf(av[0], wv[0], s0); s1 = fmaf(av[1], wv[1], s1); s2 = fmaf(av[2], wv[2], s2); s3 = fmaf(av[3], wv[3], s3);
                }
            }
            const float z = ((s0 + s1) + (s2 + s3)) + bias;
            g.out[((size_t)cell * H + y0 + py) * W + px] = 1.0f / (1.0f + expf(-z));
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

// version-2 plan for a layer: SR rows per strip, nmg tile groups x nslw slices = 8 waves, TPW tiles per wave
static bool gen2_plan(int H, int W, int cin, int cout, int ups, int* SR, int* nmg, int* nslw, int* tpw, size_t* lds)
{
    if (!(cin % 16 == 0 || (cin == 1 && !ups)) || !(W == 16 || W == 32 || W == 64 || W == 128) || cout < 16) return false;
    const int slices = (cout + 15) / 16;
    int ns = 1;
    while (ns * 2 <= slices && ns < 8) ns *= 2;           // largest power of two <= min(slices, 8)
    const int mg = 8 / ns, TPR = W / 16, Ws = ups ? W / 2 : W, ps = cin == 1 ? 1 : cin + 4;
    for (int t = 16; t >= 4; t /= 2) {                    // tiles per wave: as many as registers and LDS allow
        const int pairs = (t / 2) * mg;                   // vertical tile pairs per strip
        if (pairs % TPR) continue;
        const int sr = 2 * pairs / TPR;
        if (sr < 2 || H % sr) continue;
        const int R = ups ? sr / 2 + 2 : sr + 2;
        const size_t bytes = (size_t)R * (Ws + 2) * ps * sizeof(float);
        if (bytes > 100 * 1024) continue;
        *SR = sr; *nmg = mg; *nslw = ns; *tpw = t; *lds = bytes;
        return true;
    }
    return false;
}

// folded-upsample plan: nmg in {1, 2, 4} phase groups, every wave owns all TPW = (SR / 2) (Ws / 16) tiles of the strip
static bool gen2f_plan(int H, int W, int cin, int cout, int* SR, int* nmg, int* nslw, int* tpw, size_t* lds)
{
    const int Ws = W / 2;
    if (cin % 16 != 0 || !(Ws == 16 || Ws == 32 || Ws == 64) || cout < 32) return false;
    const int slices = (cout + 15) / 16;
    int ns = 2;
    while (ns * 2 <= slices && ns < 8) ns *= 2;           // 2, 4 or 8 slices per pass -> 4, 2 or 1 phase groups
    const int mg = 8 / ns, TPRs = Ws / 16, ps = cin + 4;
    for (int t = 16; t >= 4; t /= 2) {
        if (t % TPRs) continue;
        const int srs = t / TPRs;                         // stored rows per strip
        if (srs < 1 || (H / 2) % srs) continue;
        const size_t bytes = (size_t)(srs + 2) * (Ws + 2) * ps * sizeof(float);
        if (bytes > 100 * 1024) continue;
        *SR = 2 * srs; *nmg = mg; *nslw = ns; *tpw = t; *lds = bytes;
        return true;
    }
    return false;
}

// Effective 2x2 kernels per output phase: W_eff[a][b][ry][rx] = sum of W[dy][dx] over the taps with ((a+dy)>>1)+1 == a+ry and
// ((b+dx)>>1)+1 == b+rx, summed in fp32 in (dy,dx) order (as pack_conv_fragments_folded does).  dst: [4 phases][4 taps][cin][cout].
size_t pack_generic_folded(int cin, int cout, const float* hwio, float* dst)
{
    const size_t total = (size_t)16 * cin * cout;
    if (!dst) return total;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int tap = 0; tap < 4; ++tap) {
                const int ry = tap >> 1, rx = tap & 1;
                for (int ci = 0; ci < cin; ++ci)
                    for (int co = 0; co < cout; ++co) {
                        float sum = 0.0f;
                        for (int dy = -1; dy <= 1; ++dy)
                            for (int dx = -1; dx <= 1; ++dx)
                                if (((a + dy) >> 1) + 1 == a + ry && ((b + dx) >> 1) + 1 == b + rx)
                                    sum += hwio[((size_t)((dy + 1) * 3 + (dx + 1)) * cin + ci) * cout + co];
                        dst[((size_t)((a * 2 + b) * 4 + tap) * cin + ci) * cout + co] = sum;
                    }
            }
    return total;
}

int conv_generic_folds(int H, int W, int cin, int cout)
{
    int SR, nmg, nslw, tpw;
    size_t lds;
    return gen2f_plan(H, W, cin, cout, &SR, &nmg, &nslw, &tpw, &lds) ? 1 : 0;
}

hipError_t launch_conv_generic(const float* in, const float* w_hwio, const float* ep, float* out, int64_t n, int H, int W, int cin,
                               int cout, int ups, int epi, hipStream_t stream, const float* w_folded)
{
    if (n <= 0) return hipSuccess;
    if (!conv_generic_supported(H, W, cin, cout, nullptr, 0)) return hipErrorInvalidValue;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    GenArgs g;
    g.in = in; g.w = w_hwio; g.ep = ep; g.out = out; g.n = n; g.H = H; g.W = W; g.cin = cin; g.cout = cout;
    g.ups = ups; g.epi = epi;
    g.ps = cin == 1 ? 1 : cin + 4;                        // odd number of 16-B slots per pixel
    hipError_t e = hipSuccess;
    if (cout == 1 && epi == GEN_EPI_SIGMOID && ups && w_folded && cin % 4 == 0 && cin >= 4) {
        const int Hs = H / 2, Ws = W / 2;
        const int SRS = Hs % 4 == 0 ? 4 : (Hs % 2 == 0 ? 2 : 1);
        const size_t lds = ((size_t)(SRS + 2) * (Ws + 2) * g.ps + 16 * cin) * sizeof(float);
        if (lds <= 64 * 1024) {
            const long items = (long)n * (Hs / SRS);
            const unsigned grid = (unsigned)(items < (long)cus * 8 ? items : (long)cus * 8);
            GenArgs gf = g;
            gf.w = w_folded;
            e = hipFuncSetAttribute((const void*)conv_last_folded_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(conv_last_folded_kernel, dim3(grid), dim3(256), lds, stream, gf, SRS);
            return hipGetLastError();
        }
    }
    if (cout == 1 && epi == GEN_EPI_SIGMOID && cin % 4 == 0 && cin >= 4) {
        // the 1-filter last conv on the vector ALU: strips of 4 conv rows
        const int SR = H % 4 == 0 ? 4 : 2, Ws = ups ? W / 2 : W, R = ups ? SR / 2 + 2 : SR + 2;
        const size_t lds = ((size_t)R * (Ws + 2) * g.ps + 9 * cin) * sizeof(float);
        if (lds <= 64 * 1024) {
            const long items = (long)n * (H / SR);
            const unsigned grid = (unsigned)(items < (long)cus * 8 ? items : (long)cus * 8);
            e = hipFuncSetAttribute((const void*)conv_last_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(conv_last_generic_kernel, dim3(grid), dim3(256), lds, stream, g, SR);
            return hipGetLastError();
        }
    }
    int SR = 0, nmg = 0, nslw = 0, tpw = 0;
    size_t lds2 = 0;
    if (ups && w_folded && (epi == GEN_EPI_BN || epi == GEN_EPI_RELU) && gen2f_plan(H, W, cin, cout, &SR, &nmg, &nslw, &tpw, &lds2)) {
        Gen2Args a;
        a.in = in; a.w = w_folded; a.ep = ep; a.out = out; a.n = n; a.H = H; a.W = W; a.cin = cin; a.cout = cout; a.ups = 1; a.epi = epi;
        a.ps = g.ps; a.SR = SR; a.nmg = nmg; a.nslw = nslw;
        const long items = (long)n * (H / SR) * ((cout + nslw * 16 - 1) / (nslw * 16));
        const int per_cu = lds2 <= 76 * 1024 ? 2 : 1;
        const unsigned grid = (unsigned)(items < (long)cus * per_cu ? items : (long)cus * per_cu);
#define GEN2F_LAUNCH(T)                                                                                                        \
    do {                                                                                                                       \
        e = hipFuncSetAttribute((const void*)conv_generic2f_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
        if (e == hipSuccess) hipLaunchKernelGGL(conv_generic2f_kernel<T>, dim3(grid), dim3(512), lds2, stream, a);             \
    } while (0)
        switch (tpw) {
            case 4: GEN2F_LAUNCH(4); break;
            case 8: GEN2F_LAUNCH(8); break;
            case 16: GEN2F_LAUNCH(16); break;
            default: return hipErrorInvalidValue;
        }
#undef GEN2F_LAUNCH
        if (e != hipSuccess) return e;
        return hipGetLastError();
    }
    if (gen2_plan(H, W, cin, cout, ups, &SR, &nmg, &nslw, &tpw, &lds2)) {
        Gen2Args a;
        a.in = in; a.w = w_hwio; a.ep = ep; a.out = out; a.n = n; a.H = H; a.W = W; a.cin = cin; a.cout = cout; a.ups = ups; a.epi = epi;
        a.ps = g.ps; a.SR = SR; a.nmg = nmg; a.nslw = nslw;
        const long items = (long)n * (H / SR) * ((cout + nslw * 16 - 1) / (nslw * 16));
        const int per_cu = lds2 <= 76 * 1024 ? 2 : 1;
        const unsigned grid = (unsigned)(items < (long)cus * per_cu ? items : (long)cus * per_cu);
#define GEN2_LAUNCH(T)                                                                                                        \
    do {                                                                                                                      \
        e = hipFuncSetAttribute((const void*)conv_generic2_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
        if (e == hipSuccess) hipLaunchKernelGGL(conv_generic2_kernel<T>, dim3(grid), dim3(512), lds2, stream, a);             \
    } while (0)
#define GENC1_LAUNCH(T)                                                                                                        \
    do {                                                                                                                       \
        e = hipFuncSetAttribute((const void*)conv_generic_c1_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
        if (e == hipSuccess) hipLaunchKernelGGL(conv_generic_c1_kernel<T>, dim3(grid), dim3(512), lds2, stream, a);             \
    } while (0)
        if (cin == 1) {
            switch (tpw) {
                case 4: GENC1_LAUNCH(4); break;
                case 8: GENC1_LAUNCH(8); break;
                case 16: GENC1_LAUNCH(16); break;
                default: return hipErrorInvalidValue;
            }
        } else
        switch (tpw) {
            case 4: GEN2_LAUNCH(4); break;
            case 8: GEN2_LAUNCH(8); break;
            case 16: GEN2_LAUNCH(16); break;
            default: return hipErrorInvalidValue;
        }
#undef GENC1_LAUNCH
#undef GEN2_LAUNCH
        if (e != hipSuccess) return e;
        return hipGetLastError();
    }
    const int Ws = ups ? W / 2 : W, R = ups ? 3 : 4;
    const size_t lds = (size_t)R * (Ws + 2) * g.ps * sizeof(float);
    const long items = (long)n * (H / GEN_SR) * ((cout + 63) / 64);
    const unsigned grid = (unsigned)(items < (long)cus * 8 ? items : (long)cus * 8);
    const int tps = GEN_SR * W / 16;
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
