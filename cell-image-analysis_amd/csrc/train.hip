// train.hip -- kernels of one `autoencoder.fit` step (CAE_improved_modeltrain.py:223-227, 286-293)
// that are not convolutions over the forward template: BatchNormalization in training mode
// (batch statistics, moving-average update, backward), max-pool routing, MSE loss gradient,
// weight gradients (MFMA GEMM over pixels), partial-sum reduction, Adam, fragment re-packing.
// Reductions over the batch are two-level and summed in a fixed order: results are
// deterministic run to run (no float atomics anywhere).
#include "common.hpp"

namespace cs {

namespace {

// ============================================================== BatchNormalization, forward
// r: [P][C] relu output (P = batch*H*W pixels).  Each workgroup reduces a contiguous pixel range
// to {count, mean, M2} per channel (two passes over its range: exact local mean first), the
// consumers merge the G partials with Chan's formula in double.
// (a thread owns four adjacent channels of every L-th pixel of the range: 16-byte loads -- these kernels stream a tensor once
// and are bound by the bytes they keep in flight, not by arithmetic)
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ r, long P, int C,
                                                               float* __restrict__ part /*[G][3][C]*/)
{
    __shared__ float red[1024];
    __shared__ float meanc[256];
    const int tid = threadIdx.x;
    const int C4 = C >> 2, c4 = tid % C4, lane = tid / C4, L = 256 / C4;
    const long p0 = (P * blockIdx.x) / gridDim.x, p1 = (P * (blockIdx.x + 1)) / gridDim.x;
    const f32x4* __restrict__ r4 = (const f32x4*)r;
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
    for (long p = p0 + lane; p < p1; p += L) s += r4[p * C4 + c4];
#pragma unroll
    for (int j = 0; j < 4; ++j) red[lane * C + 4 * c4 + j] = s[j];
    __syncthreads();
    if (tid < C) {
        float t = 0.0f;
        for (int l = 0; l < L; ++l) t += red[l * C + tid];
        meanc[tid] = t / (float)(p1 - p0);
    }
    __syncthreads();
    const f32x4 mu = {meanc[4 * c4], meanc[4 * c4 + 1], meanc[4 * c4 + 2], meanc[4 * c4 + 3]};
    f32x4 m2 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
    for (long p = p0 + lane; p < p1; p += L) { const f32x4 d = r4[p * C4 + c4] - mu; m2 += d * d; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) red[lane * C + 4 * c4 + j] = m2[j];
    __syncthreads();
    if (tid < C) {
        float t = 0.0f;
        for (int l = 0; l < L; ++l) t += red[l * C + tid];
        float* o = part + (size_t)blockIdx.x * 3 * C;
        o[tid] = (float)(p1 - p0);
        o[C + tid] = meanc[tid];
        o[2 * C + tid] = t;
    }
}

// One workgroup merges the G partials of a layer in a fixed two-level order (8 groups of
// consecutive partials per channel, then the 8 group results), publishes {mean, inv} for
// bn_apply / the backward pass and updates the moving statistics
// (moving = moving*momentum + batch*(1-momentum)).  Merging inside every workgroup of
// bn_apply cost 20 us per layer (G dependent L2 round trips on each CU).
constexpr int BNF_THREADS = 256;    // one workgroup per channel: each thread merges G / 256 partials, then a fixed-order tree

struct ChanStat { double n, mu, M2; };
__device__ __forceinline__ void chan_merge(ChanStat& a, double nb, double mb, double M2b)
{
    if (nb > 0.0) {                                   // an empty partial would make nb / tot = 0 / 0
        const double tot = a.n + nb, delta = mb - a.mu;
        a.mu += delta * (nb / tot);
        a.M2 += M2b + delta * delta * (a.n * nb / tot);
        a.n = tot;
    }
}

// Workgroup c merges the G partials of channel c with Chan's formula in double: thread t takes the consecutive partials
// [G t / 256, G (t + 1) / 256) in order, then eight tree levels combine neighbours in a fixed order (deterministic).  The
// previous single-workgroup form ran ~80 dependent double divisions per thread (9.4 us per layer, 56 us per step).
__global__ __launch_bounds__(BNF_THREADS) void bn_stats_final_kernel(const float* __restrict__ part, int G, int C, float eps,
                                                                     float momentum, float* __restrict__ mov_mean,
                                                                     float* __restrict__ mov_var, float* __restrict__ stats)
{
    __shared__ double sn[BNF_THREADS], smu[BNF_THREADS], sM2[BNF_THREADS];
    const int tid = threadIdx.x, c = blockIdx.x;
    const int g0 = (G * tid) / BNF_THREADS, g1 = (G * (tid + 1)) / BNF_THREADS;
    ChanStat a{0.0, 0.0, 0.0};
    for (int g = g0; g < g1; ++g) {
        const float* o = part + (size_t)g * 3 * C;
        chan_merge(a, o[c], o[C + c], o[2 * C + c]);
    }
    sn[tid] = a.n; smu[tid] = a.mu; sM2[tid] = a.M2;
    __syncthreads();
    for (int half = BNF_THREADS / 2; half >= 1; half >>= 1) {
        // level: thread t < half combines the results of the contiguous partial ranges [2t] and [2t+1] of the level below
        ChanStat m{0.0, 0.0, 0.0};
        if (tid < half) {
            m = ChanStat{sn[2 * tid], smu[2 * tid], sM2[2 * tid]};
            chan_merge(m, sn[2 * tid + 1], smu[2 * tid + 1], sM2[2 * tid + 1]);
        }
        __syncthreads();
        if (tid < half) { sn[tid] = m.n; smu[tid] = m.mu; sM2[tid] = m.M2; }
        __syncthreads();
    }
    if (tid == 0) {
        const float fm = (float)smu[0], fv = (float)(sM2[0] / sn[0]);   // biased variance, as Keras normalises with
        stats[c] = fm;
        stats[C + c] = 1.0f / sqrtf(fv + eps);
        mov_mean[c] = mov_mean[c] * momentum + fm * (1.0f - momentum);
        mov_var[c] = mov_var[c] * momentum + fv * (1.0f - momentum);
    }
}

// The same merge, leaving the merged {count, mean, M2} of every channel as ONE partial [3][C] (floats, the layout of `part`):
// a rank's contribution to a synchronised BatchNormalization -- the triples of all ranks are then merged by
// bn_stats_final_kernel exactly as it merges workgroups' partials.
__global__ __launch_bounds__(BNF_THREADS) void bn_stats_merge_kernel(const float* __restrict__ part, int G, int C, float* __restrict__ out)
{
    __shared__ double sn[BNF_THREADS], smu[BNF_THREADS], sM2[BNF_THREADS];
    const int tid = threadIdx.x, c = blockIdx.x;
    const int g0 = (G * tid) / BNF_THREADS, g1 = (G * (tid + 1)) / BNF_THREADS;
    ChanStat a{0.0, 0.0, 0.0};
    for (int g = g0; g < g1; ++g) {
        const float* o = part + (size_t)g * 3 * C;
        chan_merge(a, o[c], o[C + c], o[2 * C + c]);
    }
    sn[tid] = a.n; smu[tid] = a.mu; sM2[tid] = a.M2;
    __syncthreads();
    for (int half = BNF_THREADS / 2; half >= 1; half >>= 1) {
        ChanStat m{0.0, 0.0, 0.0};
        if (tid < half) {
            m = ChanStat{sn[2 * tid], smu[2 * tid], sM2[2 * tid]};
            chan_merge(m, sn[2 * tid + 1], smu[2 * tid + 1], sM2[2 * tid + 1]);
        }
        __syncthreads();
        if (tid < half) { sn[tid] = m.n; smu[tid] = m.mu; sM2[tid] = m.M2; }
        __syncthreads();
    }
    if (tid == 0) { out[c] = (float)sn[0]; out[C + c] = (float)smu[0]; out[2 * C + c] = (float)sM2[0]; }
}

// y = gamma*(r-mean)*inv + beta, then 2x2 max-pool (encoder) or identity (decoder).
__global__ __launch_bounds__(256) void bn_apply_kernel(
    const float* __restrict__ r, int C, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ stats /*[2][C]*/, float* __restrict__ a, long N, int H, int W, int pool)
{
    __shared__ float sg[256], sb[256];
    const int tid = threadIdx.x;
    if (tid < C) {
        const float g = gamma[tid] * stats[C + tid];
        sg[tid] = g;                                  // y = r*g + b
        sb[tid] = beta[tid] - stats[tid] * g;
    }
    __syncthreads();
    // H, W, C are powers of two and batch*H*W*C < 2^31: 32-bit shift/mask indexing; a thread owns four adjacent channels
    const int C4 = C >> 2, lC4 = __ffs(C4) - 1, lW = __ffs(W) - 1, lH = __ffs(H) - 1;
    const int lWo = pool ? lW - 1 : lW, lHo = pool ? lH - 1 : lH;
    const int total4 = (int)(N << (lHo + lWo + lC4));
    const f32x4* __restrict__ r4 = (const f32x4*)r;
    f32x4* __restrict__ a4 = (f32x4*)a;
#pragma unroll 2
    for (int o = blockIdx.x * 256 + tid; o < total4; o += gridDim.x * 256) {
        const int c4 = o & (C4 - 1);
        const int pix = o >> lC4;
        const f32x4 g = *(const f32x4*)&sg[4 * c4], b = *(const f32x4*)&sb[4 * c4];
        auto bn = [&](const f32x4& v) { return f32x4{fmaf(v[0], g[0], b[0]), fmaf(v[1], g[1], b[1]), fmaf(v[2], g[2], b[2]), fmaf(v[3], g[3], b[3])}; };
        if (pool) {
            const int xo = pix & ((1 << lWo) - 1), yo = (pix >> lWo) & ((1 << lHo) - 1), n = pix >> (lWo + lHo);
            const f32x4* p = r4 + ((((size_t)n << lH) + 2 * yo) << lW) * C4 + (size_t)(2 * xo) * C4 + c4;
            const f32x4 y00 = bn(p[0]), y01 = bn(p[C4]), y10 = bn(p[(size_t)W * C4]), y11 = bn(p[(size_t)W * C4 + C4]);
            f32x4 m;
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = fmaxf(fmaxf(y00[j], y01[j]), fmaxf(y10[j], y11[j]));
            a4[o] = m;
        } else {
            a4[o] = bn(r4[o]);
        }
    }
}

// ============================================================== loss
// dz7 = dL/dz of the sigmoid conv: L = mean((out-y)^2) over all B*H*W elements (loss='mse').
__global__ __launch_bounds__(256) void loss_dz_kernel(const float* __restrict__ out, const float* __restrict__ y,
                                                      long total, float* __restrict__ dz,
                                                      float* __restrict__ dzsum_part /*[G]*/, const float* __restrict__ errpart,
                                                      long nparts, float* __restrict__ out2)
{
    if (errpart && blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {   // the batch's loss / mae, same order as loss_scalar_kernel
        double s2 = 0.0, s1 = 0.0;
        for (long i = 0; i < nparts; ++i) { s2 += errpart[2 * i]; s1 += errpart[2 * i + 1]; }
        out2[0] = (float)(s2 / (double)total);
        out2[1] = (float)(s1 / (double)total);
    }
    __shared__ float red[4];
    const float k = 2.0f / (float)total;
    float s = 0.0f;
#pragma unroll 4
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float o = out[i];
        const float d = k * (o - y[i]) * o * (1.0f - o);
        dz[i] = d;
        s += d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) dzsum_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// loss / mae scalars from the conv7 error partial sums (errpart: [B][4][2]); a stand-alone launch only where no loss_dz follows
__global__ void loss_scalar_kernel(const float* __restrict__ errpart, long nparts, long nelem, float* __restrict__ out2)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s2 = 0.0, s1 = 0.0;
    for (long i = 0; i < nparts; ++i) { s2 += errpart[2 * i]; s1 += errpart[2 * i + 1]; }
    out2[0] = (float)(s2 / (double)nelem);
    out2[1] = (float)(s1 / (double)nelem);
}

// ============================================================== BatchNormalization, backward
// Gradient arriving at the BN output: `da` at pooled resolution for encoder layers (MaxPooling2D
// routes it to the arg-max of each 2x2 window of y = BN(r), first maximum in (dy,dx) order),
// at full resolution for decoder layers.
// A thread owns four adjacent channels (16-byte loads) of every L-th (pooled) pixel of its workgroup's range.
struct Win4 { f32x4 dy[4]; f32x4 xh[4]; f32x4 r[4]; };

__device__ __forceinline__ void window4(const f32x4* __restrict__ da4, const f32x4* __restrict__ r4, int n, int yo, int xo, int c4,
                                        int H, int W, int C4, const f32x4& mean, const f32x4& inv, const f32x4& gam,
                                        const f32x4& bet, Win4& w)
{
    const f32x4* p = r4 + (((size_t)n * H + 2 * yo) * W + 2 * xo) * C4 + c4;
    w.r[0] = p[0]; w.r[1] = p[C4]; w.r[2] = p[(long)W * C4]; w.r[3] = p[(long)W * C4 + C4];
    const f32x4 g = da4[(((size_t)n * (H / 2) + yo) * (W / 2) + xo) * C4 + c4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float best = -INFINITY;
        int arg = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (w.r[j][e] - mean[e]) * inv[e];
            w.xh[j][e] = xh;
            const float y = fmaf(xh, gam[e], bet[e]);
            if (y > best) { best = y; arg = j; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) w.dy[j][e] = (j == arg) ? g[e] : 0.0f;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const float* __restrict__ da, const float* __restrict__ r, const float* __restrict__ stats,
    const float* __restrict__ gamma, const float* __restrict__ beta, long N, int H, int W, int C, int pool,
    float* __restrict__ part /*[G][2][C]*/)
{
    __shared__ double red[2][1024];
    const int tid = threadIdx.x;
    const int C4 = C >> 2, c4 = tid % C4, lane = tid / C4, L = 256 / C4;
    const f32x4 mean = *(const f32x4*)(stats + 4 * c4), inv = *(const f32x4*)(stats + C + 4 * c4);
    const f32x4 gam = *(const f32x4*)(gamma + 4 * c4), bet = *(const f32x4*)(beta + 4 * c4);
    const f32x4* __restrict__ r4 = (const f32x4*)r;
    const f32x4* __restrict__ da4 = (const f32x4*)da;
    double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
    if (pool) {
        const int lWo = __ffs(W) - 2, lHo = __ffs(H) - 2;
        const int P = (int)(N << (lWo + lHo));
        const int p0 = (int)(((long)P * blockIdx.x) / gridDim.x), p1 = (int)(((long)P * (blockIdx.x + 1)) / gridDim.x);
        for (int p = p0 + lane; p < p1; p += L) {
            const int xo = p & ((1 << lWo) - 1), yo = (p >> lWo) & ((1 << lHo) - 1), n = p >> (lWo + lHo);
            Win4 w;
            window4(da4, r4, n, yo, xo, c4, H, W, C4, mean, inv, gam, bet, w);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) { s0[e] += w.dy[j][e]; s1[e] += (double)w.dy[j][e] * w.xh[j][e]; }
        }
    } else {
        const long P = N * H * W;
        const long p0 = (P * blockIdx.x) / gridDim.x, p1 = (P * (blockIdx.x + 1)) / gridDim.x;
#pragma unroll 4
        for (long p = p0 + lane; p < p1; p += L) {
            const f32x4 dy = da4[p * C4 + c4];
            const f32x4 rv = r4[p * C4 + c4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (rv[e] - mean[e]) * inv[e];
                s0[e] += dy[e];
                s1[e] += (double)dy[e] * xh;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][lane * C + 4 * c4 + e] = s0[e]; red[1][lane * C + 4 * c4 + e] = s1[e]; }
    __syncthreads();
    if (tid < C) {
        double t0 = 0.0, t1 = 0.0;
        for (int l = 0; l < L; ++l) { t0 += red[0][l * C + tid]; t1 += red[1][l * C + tid]; }
        part[(size_t)blockIdx.x * 2 * C + tid] = (float)t0;
        part[(size_t)blockIdx.x * 2 * C + C + tid] = (float)t1;
    }
}

// Sums the per-workgroup {sum dy, sum dy*xhat} in a fixed two-level order; writes the means used
// by bn_bwd_dz and the parameter gradients dbeta = sum dy, dgamma = sum dy*xhat.
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* __restrict__ part, int G, int C, double nred,
                                                           float* __restrict__ sums /*[2][C] means*/,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    __shared__ double s0[256], s1[256];
    const int tid = threadIdx.x, c = tid % C, grp = tid / C, NG = 256 / C;
    const int g0 = (G * grp) / NG, g1 = (G * (grp + 1)) / NG;
    double t0 = 0.0, t1 = 0.0;
#pragma unroll 4
    for (int g = g0; g < g1; ++g) { t0 += part[(size_t)g * 2 * C + c]; t1 += part[(size_t)g * 2 * C + C + c]; }
    s0[tid] = t0; s1[tid] = t1;
    __syncthreads();
    if (tid < C) {
        t0 = 0.0; t1 = 0.0;
        for (int k = 0; k < NG; ++k) { t0 += s0[k * C + tid]; t1 += s1[k * C + tid]; }
        sums[tid] = (float)(t0 / nred);
        sums[C + tid] = (float)(t1 / nred);
        dbeta[tid] = (float)t0;
        dgamma[tid] = (float)t1;
    }
}

// dz = relu'(r) * gamma*inv * (dy - mean(dy) - xhat*mean(dy*xhat)); also sum(dz) per channel
// (the conv bias gradient) and, from block 0, dgamma = sum(dy*xhat), dbeta = sum(dy).
__global__ __launch_bounds__(256) void bn_bwd_dz_kernel(
    const float* __restrict__ da, const float* __restrict__ r, const float* __restrict__ stats,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ sums /*[2][C]*/,
    long N, int H, int W, int C, int pool, float* __restrict__ dz, float* __restrict__ dzsum_part /*[Gz][C]*/)
{
    __shared__ double red[1024];
    const int tid = threadIdx.x;
    const int C4 = C >> 2, c4 = tid % C4, lane = tid / C4, L = 256 / C4;
    const f32x4 mean = *(const f32x4*)(stats + 4 * c4), inv = *(const f32x4*)(stats + C + 4 * c4);
    const f32x4 gam = *(const f32x4*)(gamma + 4 * c4), bet = *(const f32x4*)(beta + 4 * c4);
    const f32x4 mdy = *(const f32x4*)(sums + 4 * c4), mdx = *(const f32x4*)(sums + C + 4 * c4);
    const f32x4 k = gam * inv;
    const f32x4* __restrict__ r4 = (const f32x4*)r;
    const f32x4* __restrict__ da4 = (const f32x4*)da;
    f32x4* __restrict__ dz4 = (f32x4*)dz;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (pool) {
        const int lWo = __ffs(W) - 2, lHo = __ffs(H) - 2;
        const int P = (int)(N << (lWo + lHo));
        const int p0 = (int)(((long)P * blockIdx.x) / gridDim.x), p1 = (int)(((long)P * (blockIdx.x + 1)) / gridDim.x);
        for (int p = p0 + lane; p < p1; p += L) {
            const int xo = p & ((1 << lWo) - 1), yo = (p >> lWo) & ((1 << lHo) - 1), n = p >> (lWo + lHo);
            Win4 w;
            window4(da4, r4, n, yo, xo, c4, H, W, C4, mean, inv, gam, bet, w);
            f32x4* o = dz4 + (((size_t)n * H + 2 * yo) * W + 2 * xo) * C4 + c4;
            const long offs[4] = {0, C4, (long)W * C4, (long)W * C4 + C4};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dr = k[e] * (w.dy[j][e] - mdy[e] - w.xh[j][e] * mdx[e]);
                    v[e] = w.r[j][e] > 0.0f ? dr : 0.0f;
                    s[e] += v[e];
                }
                o[offs[j]] = v;
            }
        }
    } else {
        const long P = N * H * W;
        const long p0 = (P * blockIdx.x) / gridDim.x, p1 = (P * (blockIdx.x + 1)) / gridDim.x;
#pragma unroll 4
        for (long p = p0 + lane; p < p1; p += L) {
            const f32x4 rv = r4[p * C4 + c4], dy = da4[p * C4 + c4];
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (rv[e] - mean[e]) * inv[e];
                const float dr = k[e] * (dy[e] - mdy[e] - xh * mdx[e]);
                v[e] = rv[e] > 0.0f ? dr : 0.0f;
                s[e] += v[e];
            }
            dz4[p * C4 + c4] = v;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[lane * C + 4 * c4 + e] = s[e];
    __syncthreads();
    if (tid < C) {
        double t = 0.0;
        for (int l = 0; l < L; ++l) t += red[l * C + tid];
        dzsum_part[(size_t)blockIdx.x * C + tid] = (float)t;
    }
}

// ============================================================== weight gradients
// dW[tap][ci][co] = sum over (cell, y, x) X[y+dy][x+dx][ci] * dZ[y][x][co]   (X zero padded, and
// read through >>1 when an UpSampling2D precedes the conv): an exact-fp32 MFMA GEMM with
// M = (tap, ci), N = co, K = pixels.  A workgroup stages an input strip and the matching dZ
// strip in LDS, every wave owns the M tiles {w, w+4, ...} x all N tiles and keeps their
// accumulators in registers across all the items of its persistent loop; partial sums go to
// part[workgroup][M][N] and are reduced in workgroup order by reduce_all_kernel.
template <int H_, int W_, int CIN_, int COUT_, bool UPS_, int SR_>
struct WgCfg {
    static constexpr int H = H_, W = W_, CIN = CIN_, COUT = COUT_, SR = SR_;
    static constexpr bool UPS = UPS_;
    static constexpr int HS = UPS ? H / 2 : H, WS = UPS ? W / 2 : W;
    static constexpr int R = UPS ? SR / 2 + 2 : SR + 2, WP = WS + 2;
    static constexpr int PS = CIN + 16, DZP = COUT + 16;           // padded pixel strides (floats)
    static constexpr int XS_BYTES = R * WP * PS * 4, DZ_BYTES = SR * W * DZP * 4;
    static constexpr int LDS_BYTES = XS_BYTES + DZ_BYTES;
    static constexpr int KQ = CIN / 16, MT = 9 * KQ, NT = COUT / 16, MTW = (MT + 3) / 4;
    static constexpr int NSTRIP = H / SR;
    static constexpr int M = 9 * CIN;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(W % 4 == 0 && CIN % 16 == 0 && COUT % 16 == 0 && H % SR == 0 && SR % 2 == 0, "shape");
};

template <class C>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(const float* __restrict__ xin, const float* __restrict__ dz,
                                                         float* __restrict__ part, long n_cells)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = (float*)smem;
    float* dzs = (float*)(smem + C::XS_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    f32x4 acc[C::MTW][C::NT];
#pragma unroll
    for (int mi = 0; mi < C::MTW; ++mi)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) acc[mi][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const long total = n_cells * C::NSTRIP;
    for (long item = blockIdx.x; item < total; item += gridDim.x) {
        const long cell = item / C::NSTRIP;
        const int y0 = (int)(item % C::NSTRIP) * C::SR;
        // stage the input strip (zero halo) and the dZ strip
        {
            constexpr int C4 = C::CIN / 4, TOT = C::R * C::WP * C4;
            const int ybase = C::UPS ? (y0 / 2 - 1) : (y0 - 1);
            const float* src = xin + (size_t)cell * C::HS * C::WS * C::CIN;
#pragma unroll 4
            for (int idx = tid; idx < TOT; idx += 256) {
                const int pix = idx / C4, c4 = idx % C4;
                const int r = pix / C::WP, c = pix % C::WP;
                const int sy = ybase + r, sx = c - 1;
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (sy >= 0 && sy < C::HS && sx >= 0 && sx < C::WS)
                    v = *(const f32x4*)(src + ((size_t)sy * C::WS + sx) * C::CIN + c4 * 4);
                *(f32x4*)(xs + pix * C::PS + c4 * 4) = v;
            }
            constexpr int D4 = C::COUT / 4, DTOT = C::SR * C::W * D4;
            const float* dsrc = dz + ((size_t)cell * C::H + y0) * C::W * C::COUT;
#pragma unroll 4
            for (int idx = tid; idx < DTOT; idx += 256) {
                const int pix = idx / D4, c4 = idx % D4;
                *(f32x4*)(dzs + pix * C::DZP + c4 * 4) = *(const f32x4*)(dsrc + (size_t)pix * C::COUT + c4 * 4);
            }
        }
        __syncthreads();
        for (int y = 0; y < C::SR; ++y)
            for (int xq = 0; xq < C::W / 4; ++xq) {
                const int x = 4 * xq + kq;   // this lane's pixel of the 4-pixel K step
                float b[C::NT];
#pragma unroll
                for (int nt = 0; nt < C::NT; ++nt) b[nt] = dzs[(y * C::W + x) * C::DZP + 16 * nt + li];
#pragma unroll
                for (int mi = 0; mi < C::MTW; ++mi) {
                    const int mt = wave + 4 * mi;   // wave-uniform
                    if (mt < C::MT) {
                        const int tap = mt / C::KQ, cb = mt % C::KQ;
                        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                        int pr, pc;
                        if constexpr (C::UPS) { pr = ((y + dy) >> 1) + 1; pc = ((x + dx) >> 1) + 1; }
                        else { pr = y + dy + 1; pc = x + dx + 1; }
                        const float a = xs[(pr * C::WP + pc) * C::PS + 16 * cb + li];
#pragma unroll
                        for (int nt = 0; nt < C::NT; ++nt)
                            acc[mi][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nt], acc[mi][nt], 0, 0, 0);
                    }
                }
            }
        __syncthreads();
    }
    // D[row = 4 kq + r -> m][col = li -> n]
    float* o = part + (size_t)blockIdx.x * C::M * C::COUT;
#pragma unroll
    for (int mi = 0; mi < C::MTW; ++mi) {
        const int mt = wave + 4 * mi;
        if (mt < C::MT) {
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    o[(size_t)(mt * 16 + 4 * kq + r) * C::COUT + nt * 16 + li] = acc[mi][nt][r];
        }
    }
}

//                    H   W  CIN COUT UPS    SR
using WgL2 = WgCfg<32, 32, 32, 64, false,  4>;
using WgL3 = WgCfg<16, 16, 64, 32, false,  4>;
using WgL4 = WgCfg< 8,  8, 32, 32, false,  4>;
using WgL5 = WgCfg<16, 16, 32, 64, true,   4>;
using WgL6 = WgCfg<32, 32, 64, 32, true,   4>;

template <class C>
hipError_t launch_wg(const float* xin, const float* dz, float* part, int64_t n_cells, int max_parts, int* nparts,
                     hipStream_t stream)
{
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad_mfma_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const long total = (long)n_cells * C::NSTRIP;
    const int grid = (int)(total < max_parts ? total : max_parts);
    *nparts = grid;
    hipLaunchKernelGGL(wgrad_mfma_kernel<C>, dim3(grid), dim3(256), C::LDS_BYTES, stream, xin, dz, part, (long)n_cells);
    return hipGetLastError();
}

// Strip heights of the two edge-layer weight-gradient kernels: at batch 32 a workgroup per 8 conv rows gives 256
// workgroups (TRAIN_MAX_PARTS); these kernels are one global load per 9 FMAs, so they live on parallelism.
constexpr int WGF_SR = 8;     // conv1: conv rows per item
constexpr int WGL_SRS = 4;    // conv7: stored a6 rows per item (8 output rows)

// conv1 (cin = 1): dW[tap][co] = sum x[y+dy][x+dx] * dz[y][x][co].  Thread = (co, pixel lane).
__global__ __launch_bounds__(256) void wgrad_first_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                          float* __restrict__ part /*[G][9][32]*/, long n_cells)
{
    constexpr int H = 64, W = 64, CO = 32, SR = WGF_SR, WP = W + 2, R = SR + 2;
    __shared__ float xs[R * WP];
    __shared__ float red[256];
    const int tid = threadIdx.x, co = tid & 31, lane = tid >> 5;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.0f;
    const long total = n_cells * (H / SR);
    for (long item = blockIdx.x; item < total; item += gridDim.x) {
        const long cell = item / (H / SR);
        const int y0 = (int)(item % (H / SR)) * SR;
        for (int idx = tid; idx < R * WP; idx += 256) {
            const int r = idx / WP, c = idx % WP;
            const int sy = y0 - 1 + r, sx = c - 1;
            xs[idx] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? x[(cell * H + sy) * W + sx] : 0.0f;
        }
        __syncthreads();
#pragma unroll 4
        for (int p = lane; p < SR * W; p += 8) {
            const int y = p / W, xx = p % W;
            const float d = dz[((cell * H + y0 + y) * W + xx) * CO + co];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = fmaf(xs[(y + t / 3) * WP + xx + t % 3], d, acc[t]);
        }
        __syncthreads();
    }
    for (int t = 0; t < 9; ++t) {
        red[tid] = acc[t];
        __syncthreads();
        if (tid < 32) {
            float s = 0.0f;
            for (int l = 0; l < 8; ++l) s += red[l * 32 + tid];
            part[((size_t)blockIdx.x * 9 + t) * CO + tid] = s;
        }
        __syncthreads();
    }
}

// conv7 (cout = 1, reads up(a6)): dW[tap][ci] = sum a6[(Y+dy)>>1][(X+dx)>>1][ci] * dz7[Y][X].
__global__ __launch_bounds__(256) void wgrad_last_kernel(const float* __restrict__ a6, const float* __restrict__ dz7,
                                                         float* __restrict__ part /*[G][9][32]*/, long n_cells)
{
    constexpr int HS = 32, WS = 32, CI = 32, SRS = WGL_SRS, WP = WS + 2, R = SRS + 2, PS = CI + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* as = (float*)smem;                 // [R][WP][PS]
    float* red = as + R * WP * PS;            // [256]
    const int tid = threadIdx.x, ci = tid & 31, lane = tid >> 5;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.0f;
    const long total = n_cells * (HS / SRS);
    for (long item = blockIdx.x; item < total; item += gridDim.x) {
        const long cell = item / (HS / SRS);
        const int ys0 = (int)(item % (HS / SRS)) * SRS;   // first stored row
        for (int idx = tid; idx < R * WP * CI; idx += 256) {
            const int c = idx % CI, pix = idx / CI;
            const int r = pix / WP, cc = pix % WP;
            const int sy = ys0 - 1 + r, sx = cc - 1;
            as[pix * PS + c] = (sy >= 0 && sy < HS && sx >= 0 && sx < WS) ? a6[((cell * HS + sy) * WS + sx) * CI + c] : 0.0f;
        }
        __syncthreads();
        const int Y0 = 2 * ys0;
#pragma unroll 4
        for (int p = lane; p < 2 * SRS * 64; p += 8) {
            const int Yl = p / 64, X = p % 64;      // output pixel (Y0 + Yl, X)
            const float d = dz7[(cell * 64 + Y0 + Yl) * 64 + X];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                const int pr = ((Yl + dy) >> 1) + 1, pc = ((X + dx) >> 1) + 1;
                acc[t] = fmaf(as[(pr * WP + pc) * PS + ci], d, acc[t]);
            }
        }
        __syncthreads();
    }
    for (int t = 0; t < 9; ++t) {
        red[tid] = acc[t];
        __syncthreads();
        if (tid < 32) {
            float s = 0.0f;
            for (int l = 0; l < 8; ++l) s += red[l * 32 + tid];
            part[((size_t)blockIdx.x * 9 + t) * CI + tid] = s;
        }
        __syncthreads();
    }
}

// ============================================================== reduce / Adam / packing
// errpart != NULL: thread 0 also leaves the batch's {loss, mae} in out2 (the order of loss_scalar_kernel).
// Four consecutive elements of one descriptor are summed by FOUR threads (16-byte loads; thread q takes the q-th quarter of the
// partials, the four sums are added in the order ((q0 + q1) + q2) + q3): the pass streams ~60 MB of partial sums at batch 32 at the
// end of the step's critical path and is bound by the loads it keeps in flight.  A descriptor's tail shorter than four, or one
// whose rows are not 16-byte aligned, is summed element by element in partial order.  Deterministic either way.
__global__ __launch_bounds__(256) void reduce_all_kernel(const ReduceDesc* __restrict__ descs, int ndesc, float* __restrict__ flat_grad,
                                                         const float* __restrict__ errpart, long nparts, long nelem, float* __restrict__ out2)
{
    __shared__ f32x4 red[3][64];
    const int gl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + gl;
    if (i == 0 && q == 0 && errpart) {
        double s2 = 0.0, s1 = 0.0;
        for (long k = 0; k < nparts; ++k) { s2 += errpart[2 * k]; s1 += errpart[2 * k + 1]; }
        out2[0] = (float)(s2 / (double)nelem);
        out2[1] = (float)(s1 / (double)nelem);
    }
    long base = 0;                                  // in groups of four
    bool vec = false, mine = false;
    long e = 0;
    ReduceDesc D{};
    for (int d = 0; d < ndesc; ++d) {
        const long ng = (descs[d].len + 3) >> 2;
        if (i < base + ng) {
            D = descs[d];
            e = (i - base) << 2;
            mine = true;
            vec = e + 4 <= D.len && (D.stride & 3) == 0 && ((((size_t)D.src) | ((size_t)(flat_grad + D.dst))) & 15) == 0;
            break;
        }
        base += ng;
    }
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
    if (mine && vec) {
        const int p0 = D.nparts * q / 4, p1 = D.nparts * (q + 1) / 4;
#pragma unroll 8
        for (int p = p0; p < p1; ++p) s += *(const f32x4*)(D.src + (size_t)p * D.stride + e);
    }
    if (q > 0) red[q - 1][gl] = s;
    __syncthreads();
    if (q != 0 || !mine) return;
    if (vec) {
        *(f32x4*)(flat_grad + D.dst + e) = ((s + red[0][gl]) + red[1][gl]) + red[2][gl];
    } else {
        for (long k = e; k < e + 4 && k < D.len; ++k) {
            float t = 0.0f;
            for (int p = 0; p < D.nparts; ++p) t += D.src[(size_t)p * D.stride + k];
            flat_grad[D.dst + k] = t;
        }
    }
}

// Keras Adam: w -= alpha * m / (sqrt(v) + eps), alpha = lr*sqrt(1-b2^t)/(1-b1^t) computed on the host and passed by value
// (alpha_dev, when not NULL, overrides it from device memory).  macc (optional): the running sums Keras keeps over an epoch --
// {sum of batch losses, sum of batch MAEs, batches} in double, advanced by the step's {loss, mae} in batch_scal -- so that an
// epoch of steps needs no host round trip (cs_train_step_async / cs_train_read_metrics).
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, const float* __restrict__ alpha_dev, float alpha_val, float b1, float b2, float eps,
                            const float* __restrict__ batch_scal, double* __restrict__ macc)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && macc) {
        macc[0] += (double)batch_scal[0];
        macc[1] += (double)batch_scal[1];
        macc[2] += 1.0;
    }
    if (i >= n) return;
    const float alpha = alpha_dev ? *alpha_dev : alpha_val;
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = v[i] + (gi * gi - v[i]) * (1.0f - b2);
    m[i] = mi; v[i] = vi;
    p[i] -= (mi * alpha) / (sqrtf(vi) + eps);
}

// B-operand fragments of conv_mfma_kernel from an HWIO kernel in device memory.
// transposed = 1 builds the backward-data kernel: W'[tap][ci'][co'] = W[8-tap][co'][ci'].
__global__ void pack_frag_kernel(const float* __restrict__ hwio, int cin, int cout, int transposed,
                                 float* __restrict__ dst)
{
    const int ecin = transposed ? cout : cin, ecout = transposed ? cin : cout;   // effective conv shape
    const int kq_n = ecin / 16, nb = (ecin == 1) ? 3 : 9 * kq_n * 4;
    const int total = (ecout / 16) * nb * 64;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int lane = i & 63, s = (i >> 6) % nb, nsl = (i >> 6) / nb;
    const int li = lane & 15, kq = lane >> 4, co = nsl * 16 + li;
    int tap, ci;
    bool valid = true;
    if (ecin == 1) { tap = 4 * s + kq; ci = 0; valid = tap < 9; }
    else { const int j = s & 3, q = (s >> 2) % kq_n; tap = (s >> 2) / kq_n; ci = 16 * q + 4 * kq + j; }
    float v = 0.0f;
    if (valid) v = transposed ? hwio[((size_t)(8 - tap) * cin + co) * cout + ci] : hwio[((size_t)tap * cin + ci) * cout + co];
    dst[i] = v;
}

// All operand packs of one training step in ONE launch (12 fragment packs + W_eff of conv7): after every Adam
// update the MFMA B fragments of the six forward convs and of the six backward-data convs are rebuilt from the
// HWIO parameters; as 13 launches of ~2.5 us that was 5 % of a batch-32 step.
__global__ void pack_all_kernel(PackTable tab)
{
    int job = 0, b = blockIdx.x;
    while (job < tab.n - 1 && b >= tab.job[job].blocks) { b -= tab.job[job].blocks; ++job; }
    const PackJob J = tab.job[job];
    const int i = b * blockDim.x + threadIdx.x;
    if (J.transposed == 2) {                                  // conv7's effective weights
        if (i >= 16 * 32) return;
        const int ci = i & 31, e = i >> 5;
        const int rx = e & 1, ry = (e >> 1) & 1, bb = (e >> 2) & 1, a = (e >> 3) & 1;
        float sum = 0.0f;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx)
                if (((a + dy) >> 1) + 1 == a + ry && ((bb + dx) >> 1) + 1 == bb + rx) sum += J.src[((dy + 1) * 3 + (dx + 1)) * 32 + ci];
        J.dst[i] = sum;
        return;
    }
    const int cin = J.cin, cout = J.cout, transposed = J.transposed;
    const int ecin = transposed ? cout : cin, ecout = transposed ? cin : cout;
    const int kq_n = ecin / 16, nb = (ecin == 1) ? 3 : 9 * kq_n * 4;
    const int total = (ecout / 16) * nb * 64;
    if (i >= total) return;
    const int lane = i & 63, s = (i >> 6) % nb, nsl = (i >> 6) / nb;
    const int li = lane & 15, kq = lane >> 4, co = nsl * 16 + li;
    int tap, ci;
    bool valid = true;
    if (ecin == 1) { tap = 4 * s + kq; ci = 0; valid = tap < 9; }
    else { const int j = s & 3, q = (s >> 2) % kq_n; tap = (s >> 2) / kq_n; ci = 16 * q + 4 * kq + j; }
    float v = 0.0f;
    if (valid) v = transposed ? J.src[((size_t)(8 - tap) * cin + co) * cout + ci] : J.src[((size_t)tap * cin + ci) * cout + co];
    J.dst[i] = v;
}

// W_eff[a][b][ry][rx][ci] of conv7_err_kernel from the HWIO (3,3,32,1) kernel in device memory.
__global__ void pack_w7eff_kernel(const float* __restrict__ w7, float* __restrict__ weff)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 16 * 32) return;
    const int ci = i & 31, e = i >> 5;
    const int rx = e & 1, ry = (e >> 1) & 1, b = (e >> 2) & 1, a = (e >> 3) & 1;
    float sum = 0.0f;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx)
            if (((a + dy) >> 1) + 1 == a + ry && ((b + dx) >> 1) + 1 == b + rx) sum += w7[((dy + 1) * 3 + (dx + 1)) * 32 + ci];
    weff[i] = sum;
}

// {bias, s, t} epilogue constants of the inference kernels from trainable params + moving stats.
__global__ void pack_ep_kernel(const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mov_mean, const float* __restrict__ mov_var, float eps, int C,
                               float* __restrict__ ep)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = gamma[c] / sqrtf(mov_var[c] + eps);
    ep[c] = bias[c];
    ep[C + c] = s;
    ep[2 * C + c] = beta[c] - mov_mean[c] * s;
}

}  // namespace

// ============================================================== launchers
// the streaming BatchNormalization kernels give a thread four adjacent channels: C a multiple of 4 that divides 1,024
static bool bn_vec_ok(int C) { return C >= 4 && C <= 256 && C % 4 == 0 && 1024 % C == 0; }
static int stat_grid(long P) { long g = P / 128; if (g < 1) g = 1; if (g > BN_MAX_PARTS) g = BN_MAX_PARTS; return (int)g; }

hipError_t launch_bn_stats(const float* r, long P, int C, float* part, int* G, hipStream_t s)
{
    if (!bn_vec_ok(C)) return hipErrorInvalidValue;
    *G = stat_grid(P);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(*G), dim3(256), 0, s, r, P, C, part);
    return hipGetLastError();
}

hipError_t launch_bn_stats_final(const float* part, int G, int C, float eps, float momentum, float* mov_mean,
                                 float* mov_var, float* stats, hipStream_t s)
{
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((unsigned)C), dim3(BNF_THREADS), 0, s, part, G, C, eps, momentum, mov_mean, mov_var, stats);
    return hipGetLastError();
}

hipError_t launch_bn_stats_merge(const float* part, int G, int C, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(bn_stats_merge_kernel, dim3((unsigned)C), dim3(BNF_THREADS), 0, s, part, G, C, out);
    return hipGetLastError();
}

hipError_t launch_bn_apply(const float* r, int C, const float* gamma, const float* beta, const float* stats, float* a,
                           long N, int H, int W, int pool, hipStream_t s)
{
    const long total = N * (pool ? H / 2 : H) * (pool ? W / 2 : W) * C;
    if (total >= (1L << 31) || !bn_vec_ok(C)) return hipErrorInvalidValue;
    long g = (total / 4 + 511) / 512; if (g > 2048) g = 2048; if (g < 1) g = 1;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)g), dim3(256), 0, s, r, C, gamma, beta, stats, a, N, H, W, pool);
    return hipGetLastError();
}

hipError_t launch_loss_dz(const float* out, const float* y, long total, float* dz, float* dzsum_part, int* G, hipStream_t s,
                          const float* errpart, long nparts, float* out2)
{
    long g = (total + 4095) / 4096; if (g < 1) g = 1; if (g > TRAIN_MAX_PARTS) g = TRAIN_MAX_PARTS;
    *G = (int)g;
    hipLaunchKernelGGL(loss_dz_kernel, dim3(*G), dim3(256), 0, s, out, y, total, dz, dzsum_part, errpart, nparts, out2);
    return hipGetLastError();
}

hipError_t launch_loss_scalar(const float* errpart, long nparts, long nelem, float* out2, hipStream_t s)
{
    hipLaunchKernelGGL(loss_scalar_kernel, dim3(1), dim3(64), 0, s, errpart, nparts, nelem, out2);
    return hipGetLastError();
}

hipError_t launch_bn_bwd_reduce(const float* da, const float* r, const float* stats, const float* gamma, const float* beta,
                                long N, int H, int W, int C, int pool, float* part, int* G, hipStream_t s)
{
    if (!bn_vec_ok(C)) return hipErrorInvalidValue;
    *G = stat_grid(N * (pool ? H / 2 : H) * (pool ? W / 2 : W));
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(*G), dim3(256), 0, s, da, r, stats, gamma, beta, N, H, W, C, pool, part);
    return hipGetLastError();
}

hipError_t launch_bn_bwd_final(const float* part, int G, int C, double nred, float* sums, float* dgamma, float* dbeta,
                               hipStream_t s)
{
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(1), dim3(256), 0, s, part, G, C, nred, sums, dgamma, dbeta);
    return hipGetLastError();
}

hipError_t launch_bn_bwd_dz(const float* da, const float* r, const float* stats, const float* gamma, const float* beta,
                            const float* sums, long N, int H, int W, int C, int pool, float* dz, float* dzsum_part,
                            int* Gz, hipStream_t s)
{
    if (!bn_vec_ok(C)) return hipErrorInvalidValue;
    *Gz = stat_grid(N * (pool ? H / 2 : H) * (pool ? W / 2 : W));
    hipLaunchKernelGGL(bn_bwd_dz_kernel, dim3(*Gz), dim3(256), 0, s, da, r, stats, gamma, beta, sums, N, H, W, C, pool, dz,
                       dzsum_part);
    return hipGetLastError();
}

hipError_t launch_wgrad(int layer, const float* xin, const float* dz, float* part, int64_t n_cells, int* nparts,
                        hipStream_t s)
{
    const int mp = TRAIN_MAX_PARTS;
    switch (layer) {
        case 0: {
            const long total = n_cells * (64 / WGF_SR);
            *nparts = (int)(total < mp ? total : mp);
            hipLaunchKernelGGL(wgrad_first_kernel, dim3(*nparts), dim3(256), 0, s, xin, dz, part, (long)n_cells);
            return hipGetLastError();
        }
        case 1: return launch_wg<WgL2>(xin, dz, part, n_cells, mp, nparts, s);
        case 2: return launch_wg<WgL3>(xin, dz, part, n_cells, mp, nparts, s);
        case 3: return launch_wg<WgL4>(xin, dz, part, n_cells, mp, nparts, s);
        case 4: return launch_wg<WgL5>(xin, dz, part, n_cells, mp, nparts, s);
        case 5: return launch_wg<WgL6>(xin, dz, part, n_cells, mp, nparts, s);
        case 6: {
            const long total = n_cells * (32 / WGL_SRS);
            *nparts = (int)(total < mp ? total : mp);
            const int lds = ((WGL_SRS + 2) * 34 * 33 + 256) * 4;
            hipLaunchKernelGGL(wgrad_last_kernel, dim3(*nparts), dim3(256), lds, s, xin, dz, part, (long)n_cells);
            return hipGetLastError();
        }
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_reduce_all(const ReduceDesc* descs_dev, int ndesc, long total_len, float* flat_grad, hipStream_t s,
                             const float* errpart, long nparts, long nelem, float* out2)
{
    // four threads per group of four elements; every descriptor rounds up to whole groups (ndesc extra groups at most)
    const long groups = total_len / 4 + ndesc + 1;
    hipLaunchKernelGGL(reduce_all_kernel, dim3((unsigned)((groups + 63) / 64)), dim3(256), 0, s, descs_dev, ndesc, flat_grad, errpart, nparts, nelem, out2);
    return hipGetLastError();
}

hipError_t launch_adam(float* p, const float* g, float* m, float* v, long n, const float* alpha, float b1, float b2, float eps,
                       hipStream_t s, float alpha_val, const float* batch_scal, double* macc)
{
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, alpha, alpha_val, b1, b2, eps,
                       batch_scal, macc);
    return hipGetLastError();
}

hipError_t launch_pack_frag(const float* hwio, int cin, int cout, int transposed, float* dst, hipStream_t s)
{
    const int ecin = transposed ? cout : cin, ecout = transposed ? cin : cout;
    const int nb = (ecin == 1) ? 3 : 9 * (ecin / 16) * 4;
    const int total = (ecout / 16) * nb * 64;
    hipLaunchKernelGGL(pack_frag_kernel, dim3((total + 255) / 256), dim3(256), 0, s, hwio, cin, cout, transposed, dst);
    return hipGetLastError();
}

hipError_t launch_pack_all(PackTable& tab, hipStream_t s)
{
    int blocks = 0;
    for (int k = 0; k < tab.n; ++k) {
        PackJob& J = tab.job[k];
        int total;
        if (J.transposed == 2) total = 16 * 32;
        else {
            const int ecin = J.transposed ? J.cout : J.cin, ecout = J.transposed ? J.cin : J.cout;
            total = (ecout / 16) * ((ecin == 1) ? 3 : 9 * (ecin / 16) * 4) * 64;
        }
        J.blocks = (total + 255) / 256;
        blocks += J.blocks;
    }
    hipLaunchKernelGGL(pack_all_kernel, dim3(blocks), dim3(256), 0, s, tab);
    return hipGetLastError();
}

hipError_t launch_pack_w7eff(const float* w7, float* weff, hipStream_t s)
{
    hipLaunchKernelGGL(pack_w7eff_kernel, dim3(2), dim3(256), 0, s, w7, weff);
    return hipGetLastError();
}

hipError_t launch_pack_ep(const float* bias, const float* gamma, const float* beta, const float* mov_mean,
                          const float* mov_var, float eps, int C, float* ep, hipStream_t s)
{
    hipLaunchKernelGGL(pack_ep_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, bias, gamma, beta, mov_mean, mov_var, eps, C, ep);
    return hipGetLastError();
}

// ---- training augmentation ---------------------------------------------------------------
// ImageDataGenerator.apply_transform for one 64x64 image (CAE_improved_modeltrain.py:246-254, :287):
// scipy.ndimage.affine_transform(order=1, mode='nearest') at mat @ (r, c) + offset, then the flips.
// Coordinates and interpolation in fp64 as SciPy does; the host builds mat/offset in double exactly as
// Keras's apply_affine_transform would (cellscreen/augment.py), so the kernel only resamples.
__global__ __launch_bounds__(256) void augment_kernel(const float* __restrict__ in, const cs_aug_affine* __restrict__ tf,
                                                      float* __restrict__ out, int H, int W)
{
    const cs_aug_affine a = tf[blockIdx.x];
    const float* src = in + (size_t)blockIdx.x * H * W;
    float* dst = out + (size_t)blockIdx.x * H * W;
    for (int p = threadIdx.x; p < H * W; p += 256) {
        const int r = p / W, c = p - r * W;
        float v;
        if (a.identity) {
            v = src[p];
        } else {
            double rr = a.m[0] * (double)r + a.m[1] * (double)c + a.off[0];
            double cc = a.m[2] * (double)r + a.m[3] * (double)c + a.off[1];
            rr = fmin(fmax(rr, 0.0), (double)(H - 1));
            cc = fmin(fmax(cc, 0.0), (double)(W - 1));
            const int r0 = min((int)floor(rr), H - 2), c0 = min((int)floor(cc), W - 2);
            const double fr = rr - (double)r0, fc = cc - (double)c0;
            const double x00 = src[r0 * W + c0], x01 = src[r0 * W + c0 + 1];
            const double x10 = src[(r0 + 1) * W + c0], x11 = src[(r0 + 1) * W + c0 + 1];
            v = (float)((1.0 - fr) * ((1.0 - fc) * x00 + fc * x01) + fr * ((1.0 - fc) * x10 + fc * x11));
        }
        const int ro = a.flip_v ? H - 1 - r : r, co = a.flip_h ? W - 1 - c : c;
        dst[ro * W + co] = v;
    }
}

// cs_train_fit_step: block b gathers crop idx[b] of the resident training set, writes it unchanged as the target and resampled
// (the arithmetic of augment_kernel) as the input.  tf / idx live in pinned host memory (32 x 60 bytes per step).
__global__ __launch_bounds__(256) void fit_gather_kernel(const float* __restrict__ train, const cs_aug_affine* __restrict__ tf,
                                                         const int* __restrict__ idx, float* __restrict__ x_out,
                                                         float* __restrict__ y_out, int H, int W, int has_tf)
{
    const float* src = train + (size_t)idx[blockIdx.x] * H * W;
    float* dx = x_out + (size_t)blockIdx.x * H * W;
    float* dy = y_out + (size_t)blockIdx.x * H * W;
    cs_aug_affine a;
    a.identity = 1; a.flip_h = 0; a.flip_v = 0;
    if (has_tf) a = tf[blockIdx.x];
    for (int p = threadIdx.x; p < H * W; p += 256) {
        const int r = p / W, c = p - r * W;
        const float s0 = src[p];
        dy[p] = s0;
        float v;
        if (a.identity) {
            v = s0;
        } else {
            double rr = a.m[0] * (double)r + a.m[1] * (double)c + a.off[0];
            double cc = a.m[2] * (double)r + a.m[3] * (double)c + a.off[1];
            rr = fmin(fmax(rr, 0.0), (double)(H - 1));
            cc = fmin(fmax(cc, 0.0), (double)(W - 1));
            const int r0 = min((int)floor(rr), H - 2), c0 = min((int)floor(cc), W - 2);
            const double fr = rr - (double)r0, fc = cc - (double)c0;
            const double x00 = src[r0 * W + c0], x01 = src[r0 * W + c0 + 1];
            const double x10 = src[(r0 + 1) * W + c0], x11 = src[(r0 + 1) * W + c0 + 1];
            v = (float)((1.0 - fr) * ((1.0 - fc) * x00 + fc * x01) + fr * ((1.0 - fc) * x10 + fc * x11));
        }
        const int ro = a.flip_v ? H - 1 - r : r, co = a.flip_h ? W - 1 - c : c;
        dx[ro * W + co] = v;
    }
}

hipError_t launch_fit_gather(const float* train, const cs_aug_affine* tf, const int* idx, float* x_out, float* y_out, int64_t n, int H,
                             int W, bool has_tf, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fit_gather_kernel, dim3((unsigned)n), dim3(256), 0, s, train, tf, idx, x_out, y_out, H, W, has_tf ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_augment(const float* in, const cs_aug_affine* tf_dev, float* out, int64_t n, int H, int W, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)n), dim3(256), 0, s, in, tf_dev, out, H, W);
    return hipGetLastError();
}

}  // namespace cs
