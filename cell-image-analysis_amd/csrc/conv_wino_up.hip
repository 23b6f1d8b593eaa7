// conv_wino_up.hip -- conv5 and conv6 of the reference autoencoder (UpSampling2D(2x2) -> Conv2D 32->64 /
// 64->32, 3x3 same, relu -> BatchNorm; CAE_improved_modeltrain.py:206-213) as four Winograd F(2x2, 2x2)
// phase convs each.
//
// Nearest x2 upsampling makes the 3x3 taps of output pixel (2Y+a, 2X+b) land on only 2x2 STORED pixels,
// so the layer is four phase convs (a,b) with 2x2 effective taps over the stored 16x16 grid
// (W_eff[a][b][ry][rx] = the taps that share a stored pixel, pre-summed: the "folded" form of
// conv_mfma.hip, 4/9 of the direct MACs).  A 2x2-tap conv producing a 2x2 block of outputs is
// Winograd F(2x2, 2x2): 9 multiplies instead of 16 per channel pair, with
//     B^T = [1 -1 0; 0 1 0; 0 1 -1]   G = [1 0; 1 1; 0 1]   A^T = [1 1 0; 0 1 -1]
// (entries 0, +-1 only: the transforms add no rounding beyond one fp32 add each), so the layer
// executes 9/16 x 4/9 = 1/4 of its direct multiply-adds.
//
//   V_p = B^T d_p B     3x3 patch of phase p = (a,b): stored rows 2ty+a-1 .. +2, cols 2tx+b-1 .. +2
//   M_p,xi = V_p,xi U_p,xi   9 products [16 tiles x 64] x [64 x 16] per phase and 16-channel slice   (fp32 MFMA)
//   Y_p = A^T M_p A     the 2x2 stored pixels of the tile -> output pixels (2(2ty+u)+a, 2(2tx+v)+b)
//
// A workgroup is 8 waves = 4 phases x 2 halves of the output slices; a wave keeps U_p for its slices
// resident (conv6: 9 xi x 16 K steps x 1 slice, conv5: 9 x 8 x 2 slices = 144 VGPRs either way), transforms
// its phase's patches in registers (each lane: its tile's 3x3 patch for its 4 channels), issues 144 MFMAs
// per 16-tile group and finishes its own outputs: no cross-wave traffic, one barrier per group (strip
// double buffer).  conv5's waves handle their two slices one after the other and repeat the (cheap)
// transform, so only nine accumulators are live at a time.
#include "common.hpp"

#include <type_traits>

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace cs {

namespace {

// HS x WS: stored input grid (the output grid is twice that); a 16-tile group is 16 / (WS/2) tile rows.
template <int HS_, int WS_, int CIN_, int COUT_>
struct WUCfg {
    static constexpr int HS = HS_, WS = WS_, CIN = CIN_, COUT = COUT_;
    static constexpr int HO = 2 * HS, WO = 2 * WS;
    static constexpr int TW = WS / 2;                          // tiles per tile row (8 | 4)
    static constexpr int SR = 2 * (16 / TW);                   // stored rows per group (4 | 8)
    static constexpr int R = SR + 2, WP = WS + 2;
    static constexpr int PS = CIN + 4;                         // odd number of 16-B slots per pixel
    static constexpr int STRIP = R * WP * PS * 4;              // bytes, double buffered
    static constexpr int LDS = 2 * STRIP;
    static constexpr int NGRP = HS / SR;                       // groups per cell (4 | 1)
    static constexpr int NQ = CIN / 16, KS = CIN / 4;
    static constexpr int NSW = COUT / 32;                      // output slices per wave (8 waves = 4 phases x 2)
    static constexpr int NB = 9 * KS * NSW;                    // B registers (144 for both layers)
    static constexpr int C4 = CIN / 4, TOT = R * WP * C4;
    static constexpr int THREADS = 512;
    // staging: a thread owns one 16-byte element of the strip's INTERIOR columns per pass (the halo columns are zeroed
    // once at kernel start and never rewritten), rows RPP at a time -- no divisions, the only run-time test is the row
    static constexpr int EPR = WS * C4;                        // interior elements per staged row (256 | 64)
    static constexpr int RPP = THREADS / EPR;                  // rows per pass (2 | 8)
    static constexpr int NLD = (R + RPP - 1) / RPP;            // passes = loads per thread per strip (3 | 2)
    static_assert(THREADS % EPR == 0 && (EPR & (EPR - 1)) == 0 && (C4 & (C4 - 1)) == 0, "row-wise staging");
    static_assert(NB == 144 && LDS <= 160 * 1024 && HS % SR == 0 && (NSW == 1 || NSW == 2), "layer does not fit this design");
};
using WUL6 = WUCfg<16, 16, 64, 32>;    // conv6: a5 16x16x64 -> a6 32x32x32
using WUL5 = WUCfg<8, 8, 32, 64>;      // conv5: a4 8x8x32  -> a5 16x16x64

// fp32 MFMA and VALU instructions never execute together on a SIMD (PMC: SQ_VALU_MFMA_COEXEC_CYCLES = 0; a
// kernel's time is 32 x MFMAs + 4 x VALU instructions per SIMD), so the index arithmetic of the staging is on the
// critical path: it is reduced to per-thread constants + one row test.
template <class C>
__device__ __forceinline__ f32x4 wu_load(const float* __restrict__ cellp, int y0, int j, int rsub, int goff)
{
    int sy = y0 - 1 + rsub + C::RPP * j;
    sy = sy < 0 ? 0 : (sy > C::HS - 1 ? C::HS - 1 : sy);          // clamped: the load is unconditional
    return *(const f32x4*)(cellp + sy * (C::WS * C::CIN) + goff);
}
template <class C>
__device__ __forceinline__ void wu_store(float* strip, int y0, int j, int rsub, int loff, f32x4 v)
{
    const int r = rsub + C::RPP * j;
    const int sy = y0 - 1 + r;
    if (sy < 0 || sy >= C::HS) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};  // halo rows of the image
    if (C::R % C::RPP == 0 || r < C::R) *(f32x4*)(strip + r * (C::WP * C::PS) + loff) = v;
}

// DIAG: diagnostic build, s_memtime stamps summed per wave: [0] next-strip load issue, [1] transforms + MFMAs,
// [2] output transform + epilogue + stores, [3] next-strip LDS writes, [4] barrier.  Never used for results or timing.
__device__ __forceinline__ unsigned long long wu_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

template <class C, bool DIAG>
__global__ __launch_bounds__(C::THREADS, 2) void conv_wino_up_kernel(const float* __restrict__ in, const float* __restrict__ ufrag,
                                                                     const float* __restrict__ ep /* [3][cout] */,
                                                                     float* __restrict__ out, long n_cells,
                                                                     unsigned long long* __restrict__ diag)
{
    unsigned long long dg[5] = {0, 0, 0, 0, 0}, dt = 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = wave >> 1, wsl = wave & 1;                      // phase (a,b); which half of the output slices
    const int pa = ph >> 1, pb = ph & 1;
    const int li = lane & 15, kq = lane >> 4;

    float B[C::NB];
#pragma unroll
    for (int s = 0; s < C::NB; ++s) B[s] = ufrag[((size_t)wave * C::NB + s) * 64 + lane];
    float bias[C::NSW], bns[C::NSW], bnt[C::NSW];
#pragma unroll
    for (int k = 0; k < C::NSW; ++k) {
        const int co = (wsl * C::NSW + k) * 16 + li;
        bias[k] = ep[co]; bns[k] = ep[C::COUT + co]; bnt[k] = ep[2 * C::COUT + co];
        asm volatile("" : "+v"(bias[k]), "+v"(bns[k]), "+v"(bnt[k]));   // touch before the loop (see conv_mfma.hip)
    }
#pragma unroll
    for (int s = 0; s < C::NB; ++s) asm volatile("" : "+v"(B[s]));

    // top-left of this lane's 3x3 patch inside the strip (strip row 0 = stored row y0 - 1, col 0 = -1)
    const int trow = li / C::TW, tcol = li % C::TW;
    const int poff = ((2 * trow + pa) * C::WP + 2 * tcol + pb) * C::PS + 4 * kq;

    const long total = n_cells * C::NGRP;
    const long first = blockIdx.x;
    if (first >= total) return;
    // staging constants of this thread: interior element e of a row, row rsub of a pass
    const int se = tid & (C::EPR - 1), rsub = tid / C::EPR;
    const int spx = se / C::C4, sc4 = se & (C::C4 - 1);
    const int goff = spx * C::CIN + sc4 * 4;                       // inside a stored row
    const int loff = (spx + 1) * C::PS + sc4 * 4;                  // inside a strip row (+1: halo column)
    auto cell_ptr = [&](long cell) { return in + (size_t)cell * C::HS * C::WS * C::CIN; };
    for (int i = tid; i < 2 * C::STRIP / 16; i += C::THREADS) ((f32x4*)smem)[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    __syncthreads();
#pragma unroll
    for (int j = 0; j < C::NLD; ++j)
        wu_store<C>((float*)smem, (int)(first % C::NGRP) * C::SR, j, rsub, loff,
                    wu_load<C>(cell_ptr(first / C::NGRP), (int)(first % C::NGRP) * C::SR, j, rsub, goff));
    __syncthreads();

    int buf = 0;
    for (long item = first; item < total; item += gridDim.x) {
        const long cell = item / C::NGRP;
        const int grp = (int)(item % C::NGRP);
        const long nitem = item + gridDim.x;
        const bool has_next = nitem < total;
        const float* strip = (const float*)(smem + buf * C::STRIP);
        float* nstrip = (float*)(smem + (buf ^ 1) * C::STRIP);

        if constexpr (DIAG) dt = wu_stamp();
        f32x4 stg[C::NLD];
        if (has_next) {
#pragma unroll
            for (int j = 0; j < C::NLD; ++j) stg[j] = wu_load<C>(cell_ptr(nitem / C::NGRP), (int)(nitem % C::NGRP) * C::SR, j, rsub, goff);
        }
        if constexpr (DIAG) { const unsigned long long t = wu_stamp(); dg[0] += t - dt; dt = t; }

        const float* d0 = strip + poff;
#pragma unroll
        for (int k = 0; k < C::NSW; ++k) {         // conv5: the wave's two slices in turn (the transform is repeated)
            f32x4 acc[9];
#pragma unroll
            for (int x = 0; x < 9; ++x) acc[x] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            // the centre point M[1][1] enters all four outputs of Y = A^T M A with coefficient +1: starting its accumulator
            // at the bias adds the bias to every output (one rounding earlier in the chain, four adds per register fewer)
            acc[4] = f32x4{bias[k], bias[k], bias[k], bias[k]};
#pragma unroll
            for (int q = 0; q < C::NQ; ++q) {
                // W = B^T d (rows), then V[r] = W[r] B (columns), one row of V at a time.  JW channels of the
                // lane's four at a time: conv6 (144 weight + 36 accumulator VGPRs) transforms two channels per
                // pass (8-byte patch reads) to stay inside the 256-register budget of two waves per SIMD
                constexpr int JW = 4;
                typedef float fvec __attribute__((ext_vector_type(JW)));
#pragma unroll
                for (int jh = 0; jh < 4 / JW; ++jh) {
                    // keep the scheduler from hoisting every pass's patch reads to the top of the group (spills)
                    __builtin_amdgcn_sched_barrier(0);
                    fvec w[3][3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const fvec p0 = *(const fvec*)(d0 + (0 * C::WP + c) * C::PS + 16 * q + JW * jh);
                        const fvec p1 = *(const fvec*)(d0 + (1 * C::WP + c) * C::PS + 16 * q + JW * jh);
                        const fvec p2 = *(const fvec*)(d0 + (2 * C::WP + c) * C::PS + 16 * q + JW * jh);
                        w[0][c] = p0 - p1;
                        w[1][c] = p1;
                        w[2][c] = p1 - p2;
                    }
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const fvec v[3] = {w[r][0] - w[r][1], w[r][1], w[r][1] - w[r][2]};
#pragma unroll
                        for (int j = 0; j < JW; ++j)
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                acc[3 * r + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[c][j], B[(k * 9 + 3 * r + c) * C::KS + 4 * q + JW * jh + j],
                                                                                     acc[3 * r + c], 0, 0, 0);
                    }
                }
            }
            if constexpr (DIAG) {
                asm volatile("" ::"v"(acc[0][0]), "v"(acc[8][3]));
                const unsigned long long t = wu_stamp(); dg[1] += t - dt; dt = t;
            }
            // Y = A^T M A per tile register, bias -> relu -> BN, scatter to the phase's output pixels
            const int co = (wsl * C::NSW + k) * 16 + li;
            auto post = [&](float v) { v = fmaxf(v, 0.0f); return fmaf(v, bns[k], bnt[k]); };   // bias: already in M[1][1]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m00 = acc[0][r], m01 = acc[1][r], m02 = acc[2][r];
                const float m10 = acc[3][r], m11 = acc[4][r], m12 = acc[5][r];
                const float m20 = acc[6][r], m21 = acc[7][r], m22 = acc[8][r];
                const float t00 = m00 + m10, t01 = m01 + m11, t02 = m02 + m12;     // A^T M, row u = 0
                const float t10 = m10 - m20, t11 = m11 - m21, t12 = m12 - m22;     //        row u = 1
                const float y00 = t00 + t01, y01 = t01 - t02, y10 = t10 + t11, y11 = t11 - t12;
                const int t = 4 * kq + r;                                           // tile of the group (MFMA D row)
                const int Y = grp * C::SR + 2 * (t / C::TW), X = 2 * (t % C::TW);   // stored pixel (u = v = 0) of the tile
                float* o = out + (((size_t)cell * C::HO + 2 * Y + pa) * C::WO + 2 * X + pb) * C::COUT + co;
                o[0] = post(y00);                                                   // (u,v) = (0,0)
                o[2 * C::COUT] = post(y01);                                         // (0,1): output column + 2
                o[(size_t)2 * C::WO * C::COUT] = post(y10);                         // (1,0): output row + 2
                o[(size_t)2 * C::WO * C::COUT + 2 * C::COUT] = post(y11);
            }
            if constexpr (DIAG) { const unsigned long long t = wu_stamp(); dg[2] += t - dt; dt = t; }
        }
        if (has_next) {
#pragma unroll
            for (int j = 0; j < C::NLD; ++j) wu_store<C>(nstrip, (int)(nitem % C::NGRP) * C::SR, j, rsub, loff, stg[j]);
        }
        if constexpr (DIAG) { const unsigned long long t = wu_stamp(); dg[3] += t - dt; dt = t; }
        __syncthreads();   // this strip fully read; the next strip complete in the other buffer
        if constexpr (DIAG) { const unsigned long long t = wu_stamp(); dg[4] += t - dt; dt = t; }
        buf ^= 1;
    }
    if constexpr (DIAG) {
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) diag[((size_t)blockIdx.x * 8 + wave) * 5 + k] = dg[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// conv6 + conv7 + reconstruction error in one kernel (the screening path, which needs neither a6 nor the
// reconstruction in HBM).  The conv6 part is the kernel above with a cell-major loop (a workgroup takes a whole
// cell, its four 8-row groups in order); instead of storing a6, the epilogue leaves the group's 8 x 32 x 32 block
// in LDS.  conv7 over the upsampled a6 is, per output phase (a,b), a 2x2-tap conv over STORED a6 pixels
// (conv_out.hip: W_eff[a][b][ry][rx][c]); contracting the 32 channels first,
//     T[y][x][n] = sum_c a6[y][x][c] W_eff[n][c],   n = ((a 2 + b) 2 + ry) 2 + rx          (one MFMA chain per 16 pixels)
//     out(2y+a, 2x+b) = b7 + sum_{ry,rx} T[y+a-1+ry][x+b-1+rx][n(a,b,ry,rx)]                (4 LDS reads per output)
// turns it into a 32 -> 16 GEMM on the matrix pipe plus a gather; T lives in a 16-row ring in LDS (a new a6 row y
// completes output rows 2y-1 and 2y; the last row, 63, is finished after the fourth group against the zero
// padding).  Sigmoid, (x - r)^2 and |x - r| accumulate per thread over the whole cell and leave as 8 per-wave
// partial sums per cell, in fixed order.  Two barriers per group: after the a6 block is written, after T and the
// next input strip are written.  One workgroup per CU (130 KB of LDS; measured: conv6 alone runs as fast with one
// workgroup per CU as with two, the SIMDs are MFMA/VALU-bound either way).
struct F67 {
    using C = WUL6;
    static constexpr int PA = 36;                              // a6 pixel stride in LDS (floats): 9 16-byte slots
    static constexpr int A6_BYTES = 8 * 32 * PA * 4;
    static constexpr int TWD = 34, TSLOTS = 16;                // T row: 32 columns + zero halo; 16-row ring
    static constexpr int T_BYTES = TSLOTS * TWD * 16 * 4;
    static constexpr int W_BYTES = 16 * 36 * 4;                // W_eff [n][36], read back as two 16-byte values per lane and group
    static constexpr int LDS = C::LDS + A6_BYTES + T_BYTES + W_BYTES;
    static constexpr int NPARTS = 8;                           // error partial sums per cell (one per wave)
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

// 1 / (1 + e^-v) on the transcendental unit: v_exp_f32 (2^x, 1 ulp) and v_rcp_f32 (1 ulp) -- about 2e-7 relative on the
// reconstruction, against the 1e-5 the error sums are held to; a dozen instructions fewer per pixel than expf + IEEE divide
// on a path where every VALU instruction is on the critical path.
__device__ __forceinline__ float f67_sigmoid(float v)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}

// DIAG stamps: [0] load issue, [1] transforms + MFMAs, [2] output transform + a6 block to LDS + barrier, [3] T + next strip
// to LDS + barrier, [4] gather, sigmoid, error terms.
template <bool DIAG>
__global__ __launch_bounds__(WUL6::THREADS, 1) void conv67_fused_kernel(
    const float* __restrict__ in /* a5 */, const float* __restrict__ ufrag, const float* __restrict__ ep /* [3][32] */,
    const float* __restrict__ x /* crops [n][64][64] */, const float* __restrict__ weff /* [16][32] */,
    const float* __restrict__ b7p, float* __restrict__ errpart /* [n][8][2] */, long n_cells,
    unsigned long long* __restrict__ diag)
{
    using C = WUL6;
    unsigned long long dg[5] = {0, 0, 0, 0, 0}, dt = 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* a6s = (float*)(smem + C::LDS);
    float* tb = (float*)(smem + C::LDS + F67::A6_BYTES);
    float* wl = (float*)(smem + C::LDS + F67::A6_BYTES + F67::T_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = wave >> 1, wsl = wave & 1;
    const int pa = ph >> 1, pb = ph & 1;
    const int li = lane & 15, kq = lane >> 4;

    float B[C::NB];
#pragma unroll
    for (int s = 0; s < C::NB; ++s) B[s] = ufrag[((size_t)wave * C::NB + s) * 64 + lane];
    const int co = wsl * 16 + li;
    float bias = ep[co], bns = ep[C::COUT + co], bnt = ep[2 * C::COUT + co];
    float b7 = b7p[0];
    asm volatile("" : "+v"(bias), "+v"(bns), "+v"(bnt), "+v"(b7));
#pragma unroll
    for (int s = 0; s < C::NB; ++s) asm volatile("" : "+v"(B[s]));

    const int trow = li / C::TW, tcol = li % C::TW;
    const int poff = ((2 * trow + pa) * C::WP + 2 * tcol + pb) * C::PS + 4 * kq;
    const long first = blockIdx.x;
    if (first >= n_cells) return;
    const int se = tid & (C::EPR - 1), rsub = tid / C::EPR;
    const int spx = se / C::C4, sc4 = se & (C::C4 - 1);
    const int goff = spx * C::CIN + sc4 * 4;
    const int loff = (spx + 1) * C::PS + sc4 * 4;
    auto cell_ptr = [&](long cell) { return in + (size_t)cell * C::HS * C::WS * C::CIN; };
    for (int i = tid; i < F67::LDS / 16; i += C::THREADS) ((f32x4*)smem)[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    __syncthreads();
    wl[(tid >> 5) * 36 + (tid & 31)] = weff[tid];                  // [n][c], 512 floats, padded rows
#pragma unroll
    for (int j = 0; j < C::NLD; ++j) wu_store<C>((float*)smem, 0, j, rsub, loff, wu_load<C>(cell_ptr(first), 0, j, rsub, goff));
    __syncthreads();

    int buf = 0;
    for (long cell = first; cell < n_cells; cell += gridDim.x) {
        const float* xc = x + (size_t)cell * 64 * 64;
        float s2 = 0.0f, s1 = 0.0f;
#pragma unroll 1
        for (int grp = 0; grp < C::NGRP; ++grp) {
            const long ncell = grp < C::NGRP - 1 ? cell : cell + gridDim.x;
            const int ngrp = grp < C::NGRP - 1 ? grp + 1 : 0;
            const bool has_next = ncell < n_cells;
            const float* strip = (const float*)(smem + buf * C::STRIP);
            float* nstrip = (float*)(smem + (buf ^ 1) * C::STRIP);

            if constexpr (DIAG) dt = wu_stamp();
            f32x4 stg[C::NLD];
            if (has_next) {
#pragma unroll
                for (int j = 0; j < C::NLD; ++j) stg[j] = wu_load<C>(cell_ptr(ncell), ngrp * C::SR, j, rsub, goff);
            }
            // the crop pixels this thread's outputs are compared with (unconditional, clamped loads)
            float xv[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = wave + 8 * h, o = 2 * (8 * grp + (k >> 1)) - 1 + (k & 1);
                xv[h] = xc[(o < 0 ? 0 : o) * 64 + lane];
            }
            float xtail = 0.0f;
            if (grp == C::NGRP - 1 && wave == 0) xtail = xc[63 * 64 + lane];

            if constexpr (DIAG) { const unsigned long long t = wu_stamp(); dg[0] += t - dt; dt = t; }
            // ---- conv6: transforms + MFMAs, as in conv_wino_up_kernel
            const float* d0 = strip + poff;
            f32x4 acc[9];
#pragma unroll
            for (int xx = 0; xx < 9; ++xx) acc[xx] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            acc[4] = f32x4{bias, bias, bias, bias};                 // bias through M[1][1], as in conv_wino_up_kernel
#pragma unroll
            for (int q = 0; q < C::NQ; ++q) {
                __builtin_amdgcn_sched_barrier(0);
                f32x4 w[3][3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x4 p0 = *(const f32x4*)(d0 + (0 * C::WP + c) * C::PS + 16 * q);
                    const f32x4 p1 = *(const f32x4*)(d0 + (1 * C::WP + c) * C::PS + 16 * q);
                    const f32x4 p2 = *(const f32x4*)(d0 + (2 * C::WP + c) * C::PS + 16 * q);
                    w[0][c] = p0 - p1;
                    w[1][c] = p1;
                    w[2][c] = p1 - p2;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const f32x4 v[3] = {w[r][0] - w[r][1], w[r][1], w[r][1] - w[r][2]};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            acc[3 * r + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[c][j], B[(3 * r + c) * C::KS + 4 * q + j], acc[3 * r + c], 0, 0, 0);
                }
            }
            if constexpr (DIAG) {
                asm volatile("" ::"v"(acc[0][0]), "v"(acc[8][3]));
                const unsigned long long t = wu_stamp(); dg[1] += t - dt; dt = t;
            }
            // ---- Y = A^T M A, bias -> relu -> BN; the group's a6 block goes to LDS: local row 2 (2 (t / 8) + u) + a
            auto post = [&](float v) { v = fmaxf(v, 0.0f); return fmaf(v, bns, bnt); };
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m00 = acc[0][r], m01 = acc[1][r], m02 = acc[2][r];
                const float m10 = acc[3][r], m11 = acc[4][r], m12 = acc[5][r];
                const float m20 = acc[6][r], m21 = acc[7][r], m22 = acc[8][r];
                const float t00 = m00 + m10, t01 = m01 + m11, t02 = m02 + m12;
                const float t10 = m10 - m20, t11 = m11 - m21, t12 = m12 - m22;
                const float y00 = t00 + t01, y01 = t01 - t02, y10 = t10 + t11, y11 = t11 - t12;
                const int t = 4 * kq + r;
                const int lr = 4 * (t / C::TW) + pa, lc = 4 * (t % C::TW) + pb;        // (u,v) = (0,0)
                float* o = a6s + (lr * 32 + lc) * F67::PA + co;
                o[0] = post(y00);
                o[2 * F67::PA] = post(y01);                                            // column + 2
                o[2 * 32 * F67::PA] = post(y10);                                       // row + 2
                o[(2 * 32 + 2) * F67::PA] = post(y11);
            }
            __syncthreads();
            if constexpr (DIAG) { const unsigned long long t = wu_stamp(); dg[2] += t - dt; dt = t; }
            // ---- T = a6 W_eff^T for local row `wave`, 16 pixels per MFMA chain; K order: channel 8 kq + s
            float wc[8];                                                                 // W_eff[n = li][c = 8 kq + s]
            *(f32x4*)&wc[0] = *(const f32x4*)(wl + li * 36 + kq * 8);
            *(f32x4*)&wc[4] = *(const f32x4*)(wl + li * 36 + kq * 8 + 4);
            {
                const float* ap = a6s + ((wave * 32 + li) * F67::PA + kq * 8);
                const f32x4 a00 = *(const f32x4*)ap, a01 = *(const f32x4*)(ap + 4);
                const f32x4 a10 = *(const f32x4*)(ap + 16 * F67::PA), a11 = *(const f32x4*)(ap + 16 * F67::PA + 4);
                f32x4 t0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, t1 = t0;                     // two independent chains, interleaved
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    t0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a00[s], wc[s], t0, 0, 0, 0);
                    t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a10[s], wc[s], t1, 0, 0, 0);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    t0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a01[s], wc[4 + s], t0, 0, 0, 0);
                    t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a11[s], wc[4 + s], t1, 0, 0, 0);
                }
                float* tw = tb + ((((8 * grp + wave) & (F67::TSLOTS - 1)) * F67::TWD) + 4 * kq + 1) * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) { tw[r * 16] = t0[r]; tw[(16 + r) * 16] = t1[r]; }
            }
            if (has_next) {
#pragma unroll
                for (int j = 0; j < C::NLD; ++j) wu_store<C>(nstrip, ngrp * C::SR, j, rsub, loff, stg[j]);
            }
            __syncthreads();
            if constexpr (DIAG) { const unsigned long long t = wu_stamp(); dg[3] += t - dt; dt = t; }
            // ---- gather: new a6 row y finishes output rows 2y - 1 (phase a = 1 of row y - 1) and 2y (a = 0)
            const int px = lane & 1, xh = (lane >> 1) + px;                              // halo column of rx = 0
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = wave + 8 * h;                                              // wave-uniform
                const int y = 8 * grp + (k >> 1), e = k & 1;
                if (2 * y - 1 + e < 0) continue;
                const int nb = ((1 - e) * 2 + px) * 4;
                const float* ra = tb + (((y - 1) & (F67::TSLOTS - 1)) * F67::TWD + xh) * 16 + nb;
                const float* rb = tb + ((y & (F67::TSLOTS - 1)) * F67::TWD + xh) * 16 + nb;
                float ta0 = ra[0], ta1 = ra[16 + 1];
                const float tb0 = rb[2], tb1 = rb[16 + 3];
                if (y == 0) { ta0 = 0.0f; ta1 = 0.0f; }                                  // row -1: zero padding
                const float v = ((ta0 + ta1) + (tb0 + tb1)) + b7;
                const float rr = f67_sigmoid(v);
                const float d = xv[h] - rr;
                s2 = fmaf(d, d, s2);
                s1 += fabsf(d);
            }
            if (grp == C::NGRP - 1 && wave == 0) {                                      // output row 63: a6 row 31 and the padding
                const int nb = (2 + px) * 4;
                const float* ra = tb + ((31 & (F67::TSLOTS - 1)) * F67::TWD + xh) * 16 + nb;
                const float v = (ra[0] + ra[16 + 1]) + b7;
                const float rr = f67_sigmoid(v);
                const float d = xtail - rr;
                s2 = fmaf(d, d, s2);
                s1 += fabsf(d);
            }
            if constexpr (DIAG) {
                asm volatile("" ::"v"(s2), "v"(s1));
                const unsigned long long t = wu_stamp(); dg[4] += t - dt; dt = t;
            }
            buf ^= 1;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s2 += __shfl_down(s2, off, 64);
            s1 += __shfl_down(s1, off, 64);
        }
        if (lane == 0) {
            errpart[((size_t)cell * F67::NPARTS + wave) * 2 + 0] = s2;
            errpart[((size_t)cell * F67::NPARTS + wave) * 2 + 1] = s1;
        }
    }
    if constexpr (DIAG) {
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) diag[((size_t)blockIdx.x * 8 + wave) * 5 + k] = dg[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same fused kernel with conv6's contraction on the 16-bit matrix pipe.  conv6 runs in its FOLDED-DIRECT form here -- four
// 2x2-tap phase convs over the stored grid, 16/9 of the Winograd form's multiply-adds -- because a Winograd input transform
// would have to be split again for every transformed value (more VALU instructions than the MFMAs they feed at 32 filters),
// whereas the stored pixels are split ONCE when the strip is staged.
//   strip: the group's 4 stored rows + halo as [row 6][pixel 18][hi: 64 ch | lo | 32 B] fp16 -- twice an odd number of 16-byte
//          slots per pixel, so every ds_read_b128 lane group lands on 16 distinct slots; single-buffered: the next strip is
//          written between the two barriers of a group, when no wave reads the current one any more;
//   wave = (phase, 16-filter half): B = 4 taps x 2 channel blocks x 2 planes x 4 VGPRs = 64 registers; a tile is one stored
//          row of the group (16 pixels), its A fragment of one (tap, block) two ds_read_b128 at immediate offsets;
//   everything after the a6 block (T = a6 W_eff^T on fp32 MFMAs, the T ring, gather, sigmoid, error sums) is the code above.
// T ring, n-major: [ring row 16][n-block row 4][n in block 4][x 34 -> 36]; which n-block sits where, the one-float shift of the column-phase-1
// blocks and the pitches that keep every access of this stretch off shared banks are described at F67H and where T is written.
struct F67T {
    static constexpr int TW = 36;
    static constexpr int T_BYTES = F67::TSLOTS * 16 * TW * 4;  // 36,864 B
};

// The arithmetic: a TWO-term fp16 split (three products).  fp16 carries 11 significant bits, so x = hi + lo (hi = fp16(x), lo = fp16(x - hi)) holds 22, and
//     x w ~ hi_x hi_w + (hi_x lo_w + lo_x hi_w)
// with every partial product exact in the MFMA's fp32 accumulator (v_mfma_f32_16x16x32_f16, the bf16 instruction's rate): HALF
// the matrix instructions, two thirds of the weight registers and LDS bytes, a 3-instruction split per value instead of 5.5.
// What it needs that bf16 does not is range (fp16: 2^-14 .. 65504): operands carry exact power-of-two scales.
//   weights   S_w = the power of two that puts max|W_eff| in [2^14, 2^15), applied on the host (pack_conv6_f16x2);
//   a5        a scale per staged STRIP (6 stored rows): S_a puts the strip's max|a5| in [2^14, 2^15).  The max is taken where
//             the strip is staged (registers -> DPP row max -> one LDS atomic max per 16 lanes, read back after the barrier that
//             is there anyway), so it depends on the cell's own data only: results do not depend on what else is in the batch;
//   undo      acc * 2^-(e_a + e_w) in the epilogue's fma (exact: a power of two).
// With the scale this tight the residuals need no scale of their own: lo is a normal fp16 for every |x| above 2^-18 of the
// strip's maximum and loses at most 2^-40 of that maximum below (MI355X: the f16 MFMA takes subnormal inputs as they are and
// its products are exact -- tools/microbench/fp16_mfma_probe.hip, profiles/r03_fp16_mfma_probe.json).  One MFMA carries ONE
// magnitude (hi hi | hi lo, lo hi on a second accumulator), as in the bf16 kernels.  Error against the fp64 oracle:
// tests/study_split_fp16.py (CPU emulation: a6 1.8e-7 of its range; the fp32 MFMA chain 7.2e-7) and the unchanged -m gpu bars.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct F67H {
    static constexpr int PLB = 2 * 64;                         // bytes of one plane of a pixel: 64 channels
    static constexpr int PXB = 2 * PLB + 32;                   // 288 B per staged pixel = twice an odd number of 16-byte slots
    static constexpr int ROWB = 18 * PXB;
    static constexpr int STRIP = 6 * ROWB;                     // 31,104 B
    static constexpr int TW = F67T::TW;
    static constexpr int T_BYTES = F67T::T_BYTES;
    // a6 block, W_eff rows: pitches of 40 floats; a pixel's channels 0-15 / 16-31 swap places when bit 3 of its column is set.  With
    // these every LDS access of the T stretch is conflict-free under the bank rules of gfx950 (tools/lds_bank_model.py; the previous
    // layout -- 36 / 36, no swap -- cost 2,816 extra LDS cycles per cell, exactly SQ_LDS_BANK_CONFLICT / cells of profiles/r04_b):
    //   a6 block   ds_write_b32, 32-lane groups over 32 banks: lanes (filter li, pixel 8 kq + ..): kq = 0 / 1 on opposite halves;
    //   T operand  ds_read_b128 of lane (pixel 4 (li & 3) + (li >> 2), channels 4 kq .. +3 | 16 + 4 kq .. +3);
    //   W_eff      ds_read_b128 of lane (row li, the same channels).
    static constexpr int PA = 40;
    static constexpr int A6_BYTES = 8 * 32 * PA * 4;           // 40,960 B
    static constexpr int WP = 40;
    static constexpr int W_BYTES = 16 * WP * 4;
    static constexpr int OFF_MAX = STRIP + A6_BYTES + T_BYTES + W_BYTES;   // two words: the strip maxima (alternating)
    static constexpr int LDS = OFF_MAX + 16;
    static_assert(LDS <= 160 * 1024 && STRIP % 16 == 0, "LDS budget");
};

// max|v| over the 16 lanes of a DPP row (quad swaps, then row rotations by 4 and 8)
__device__ __forceinline__ unsigned int h2_rowmax(unsigned int m)
{
    unsigned int o;
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, true);  m = m > o ? m : o;     // quad_perm [1,0,3,2]
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, true);  m = m > o ? m : o;     // quad_perm [2,3,0,1]
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0x124, 0xF, 0xF, true); m = m > o ? m : o;     // row_ror:4
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xF, 0xF, true); m = m > o ? m : o;     // row_ror:8
    return m;
}

// the power of two S that puts a maximum with float bits `mbits` (>= 0) into [2^14, 2^15), and 1 / S.  The exponent is clamped
// so that both stay normal floats: a strip whose maximum is below 2^-87 (or zero) is scaled by 2^101 -- its values then sit in
// fp16's lowest binades or vanish, 2^-87 of anything the next layer can see.
__device__ __forceinline__ void h2_scale(unsigned int mbits, float& S, float& invS)
{
    int E = (int)((mbits >> 23) & 0xffu);
    E = E < 40 ? 40 : (E > 254 ? 254 : E);
    S = __builtin_bit_cast(float, (unsigned int)(268 - E) << 23);          // 2^(14 - (E - 127))
    invS = __builtin_bit_cast(float, (unsigned int)(E - 14) << 23);
}

__device__ __forceinline__ void h2_split4(const f32x4& x, float S, f16x4& hi, f16x4& lo)
{
    const f32x4 v = x * S;
    hi = __builtin_convertvector(v, f16x4);
    const f32x4 r = v - __builtin_convertvector(hi, f32x4);            // exact in fp32
    lo = __builtin_convertvector(r, f16x4);
}

__device__ __forceinline__ void h2_store(char* strip, int j, int rsub, int loffb, const f32x4& v, float S)
{
    const int r = rsub + WUL6::RPP * j;
    f16x4 hi, lo;
    h2_split4(v, S, hi, lo);
    char* d = strip + r * F67H::ROWB + loffb;
    *(f16x4*)d = hi;
    *(f16x4*)(d + F67H::PLB) = lo;
}

// the staged value of pass j (zero for the halo rows of the image) and its contribution to the strip's max|.|
__device__ __forceinline__ f32x4 h2_prep(int y0, int j, int rsub, f32x4 v, unsigned int& mx)
{
    const int sy = y0 - 1 + rsub + WUL6::RPP * j;
    if (sy < 0 || sy >= WUL6::HS) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // (scalars first: __builtin_bit_cast of a vector ELEMENT expression reads element 0 whatever the index -- clang 19, ROCm 7.2)
    const float a = v[0], b = v[1], c = v[2], d = v[3];
    const unsigned int u = __builtin_bit_cast(unsigned int, fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d))));
    mx = mx > u ? mx : u;
    return v;
}

// DIAG stamps as conv67_x3_kernel's.
template <bool DIAG>
__global__ __launch_bounds__(WUL6::THREADS, 2) void conv67_h2_kernel(
    const float* __restrict__ in /* a5 */, const f16x8* __restrict__ wfrag, const float* __restrict__ ep /* [3][32] */,
    const float* __restrict__ x /* crops [n][64][64] */, const float* __restrict__ weff /* [16][32] */,
    const float* __restrict__ b7p, float* __restrict__ errpart /* [n][8][2] */, long n_cells, float inv_sw,
    unsigned long long* __restrict__ diag)
{
    using C = WUL6;
    static_assert(C::NLD == 3 && C::RPP == 2 && C::R == 6 && C::EPR == 256, "row-wise staging constants");
    unsigned long long dg[5] = {0, 0, 0, 0, 0}, dt = 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* a6s = (float*)(smem + F67H::STRIP);
    float* tb = (float*)(smem + F67H::STRIP + F67H::A6_BYTES);
    float* wl = (float*)(smem + F67H::STRIP + F67H::A6_BYTES + F67H::T_BYTES);
    unsigned int* mxw = (unsigned int*)(smem + F67H::OFF_MAX);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = wave >> 1, wsl = wave & 1;
    const int pa = ph >> 1, pb = ph & 1;
    const int li = lane & 15, kq = lane >> 4;

    f16x8 B[4][2][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int p = 0; p < 2; ++p) B[t][k][p] = wfrag[(((wave * 4 + t) * 2 + k) * 2 + p) * 64 + lane];
    const int co = wsl * 16 + li;
    const int cosw = co ^ ((kq & 1) << 4);                         // where filter co of this lane's a6 pixels lives (F67H)
    const float bias = ep[co], bns = ep[C::COUT + co], bnt = ep[2 * C::COUT + co];
    const float b7 = b7p[0];

    const long first = blockIdx.x;
    if (first >= n_cells) return;
    const int se = tid & (C::EPR - 1), rsub = tid / C::EPR;
    const int spx = se / C::C4, sc4 = se & (C::C4 - 1);
    const int goff = spx * C::CIN + sc4 * 4;
    const int loffb = (spx + 1) * F67H::PXB + sc4 * 8;             // +1: halo column (zeroed once, never rewritten)
    auto cell_ptr = [&](long cell) { return in + (size_t)cell * C::HS * C::WS * C::CIN; };
    for (int i = tid; i < F67H::LDS / 16; i += C::THREADS) ((f32x4*)smem)[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    __syncthreads();
    wl[(tid >> 5) * F67H::WP + (tid & 31)] = weff[tid];            // [n][c], 512 floats, padded rows
    float unscale;                                                 // 1 / (S_a S_w) of the strip the MFMAs read
    {
        f32x4 stg[C::NLD];
        unsigned int mx = 0;
#pragma unroll
        for (int j = 0; j < C::NLD; ++j) stg[j] = h2_prep(0, j, rsub, wu_load<C>(cell_ptr(first), 0, j, rsub, goff), mx);
        mx = h2_rowmax(mx);
        if (li == 0) atomicMax(&mxw[0], mx);
        __syncthreads();
        float S, invS;
        h2_scale(mxw[0], S, invS);
        unscale = invS * inv_sw;
#pragma unroll
        for (int j = 0; j < C::NLD; ++j) h2_store(smem, j, rsub, loffb, stg[j], S);
    }
    __syncthreads();
    if (tid == 0) mxw[0] = 0;                                      // the next use of word 0 is two strips away, behind barriers
    // A operand: stored (t + a - 1 + ry, xs + b - 1 + rx) of the group = staged (t + a + ry, xs + b + rx); lane = (xs, 8 channels at 8 kq)
    const char* abase = smem + (pa * 18 + li + pb) * F67H::PXB + kq * 16;

    int strip_no = 0;                                              // strips staged so far by this workgroup: word (strip_no & 1) collects the next maximum
    for (long cell = first; cell < n_cells; cell += gridDim.x) {
        const float* xc = x + (size_t)cell * 64 * 64;
        float s2 = 0.0f, s1 = 0.0f;
#pragma unroll 1
        for (int grp = 0; grp < C::NGRP; ++grp) {
            const long ncell = grp < C::NGRP - 1 ? cell : cell + gridDim.x;
            const int ngrp = grp < C::NGRP - 1 ? grp + 1 : 0;
            const bool has_next = ncell < n_cells;
            ++strip_no;
            unsigned int* const mword = mxw + (strip_no & 1);

            if constexpr (DIAG) dt = wu_stamp();
            f32x4 stg[C::NLD];
            if (has_next) {
#pragma unroll
                for (int j = 0; j < C::NLD; ++j) stg[j] = wu_load<C>(cell_ptr(ncell), ngrp * C::SR, j, rsub, goff);
            }
            // the crop pixels this thread's outputs are compared with (unconditional, clamped loads)
            float xv[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = wave + 8 * h, o = 2 * (8 * grp + (k >> 1)) - 1 + (k & 1);
                xv[h] = xc[(o < 0 ? 0 : o) * 64 + lane];
            }
            float xtail = 0.0f;
            if (grp == C::NGRP - 1 && wave == 0) xtail = xc[63 * 64 + lane];

            if constexpr (DIAG) { const unsigned long long t = wu_stamp(); dg[0] += t - dt; dt = t; }
            // ---- conv6, folded direct, three fp16 products per (tile, tap, 32-channel block); the large products and the two
            // cross terms on separate accumulators
            f32x4 ah[4], al[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) ah[t] = al[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            // walked by STAGED fragment as in conv67_x3_kernel: the fragment at staged row sr (column shift rx, block k) is tap
            // (0, rx) of tile sr and tap (1, rx) of tile sr - 1
            auto rd = [&](int f, f16x8 (&a)[2]) {
                const int sr = f >> 2, rx = (f >> 1) & 1, k = f & 1;
                const char* p = abase + sr * F67H::ROWB + rx * F67H::PXB + k * 64;
                a[0] = *(const f16x8*)p;
                a[1] = *(const f16x8*)(p + F67H::PLB);
            };
            auto mac3 = [&](const f16x8 (&a)[2], const f16x8 (&b)[2], f32x4& dh, f32x4& dl) {
                dl = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], dl, 0, 0, 0);
                dh = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], dh, 0, 0, 0);
                dl = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], dl, 0, 0, 0);
            };
            {
                f16x8 a[2];
                rd(0, a);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int f = 0; f < 20; ++f) {
                    const int sr = f >> 2, rx = (f >> 1) & 1, k = f & 1;
                    f16x8 an[2] = {a[0], a[1]};
                    if (f + 1 < 20) rd(f + 1, an);
                    if (sr < 4) mac3(a, B[rx][k], ah[sr], al[sr]);
                    if (sr > 0) mac3(a, B[2 + rx][k], ah[sr - 1], al[sr - 1]);
                    a[0] = an[0]; a[1] = an[1];
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if (sr > 0 && sr < 4) __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                }
            }
            if constexpr (DIAG) {
                asm volatile("" ::"v"(ah[0][0]), "v"(al[3][3]));
                const unsigned long long t = wu_stamp(); dg[1] += t - dt; dt = t;
            }
            // ---- undo the scales, bias -> relu -> BN; D row 4 kq + r = stored pixel xs of stored row t -> a6 block row 2 t + a, column 2 xs + b
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = fmaxf(fmaf(ah[t][r] + al[t][r], unscale, bias), 0.0f);
                    a6s[((2 * t + pa) * 32 + 2 * (4 * kq + r) + pb) * F67H::PA + cosw] = fmaf(v, bns, bnt);     // column 8 kq + 2 r + pb: bit 3 = kq & 1
                }
            // the next strip's values are here by now (loaded at the top of the group): its max|.| into this strip's word
            if (has_next) {
                unsigned int mx = 0;
#pragma unroll
                for (int j = 0; j < C::NLD; ++j) stg[j] = h2_prep(ngrp * C::SR, j, rsub, stg[j], mx);
                mx = h2_rowmax(mx);
                if (li == 0) atomicMax(mword, mx);
            }
            __syncthreads();
            if constexpr (DIAG) { const unsigned long long t = wu_stamp(); dg[2] += t - dt; dt = t; }
            // ---- T^T = W_eff a6^T for local row `wave`: the 16 transform rows n are the MFMA's rows, 16 pixels its columns, two chains
            // (pixels x and 16 + x).  K order: step s of lane kq is channel 4 kq + (s & 3) + 16 (s >> 2).  D then has row 4 kq + r in
            // register r of lane (pixel, kq): a ds_write_b32 of one r puts 16 neighbouring pixels of one T row on neighbouring floats.
            // T ring rows: the four n-blocks {e = 1 | 0} x {column phase px = 0 | 1} sit at block rows {0, 1, 3, 2}, and the rows of the
            // px = 1 blocks are stored one float to the left (their reader never wants the left halo): lanes kq = 0, 1 write
            // blocks (0, 3), lanes 2, 3 blocks (1, 2) -- 48 and 16 floats apart mod 32 -- and the gather's even / odd lanes read
            // blocks 16 floats apart: both conflict-free (with [n][x] rows in natural order one of the two was always 2-way).
            {
                const int wrow = 4 * ((li >> 3) | (((li >> 2) & 1) << 1)) + (li & 3);       // A row i = li stands for n = 4 blk(i >> 2) + (i & 3), blk = (0, 2, 1, 3)
                float wc[8];                                                                // W_eff[n][c = 4 kq + (s & 3) + 16 (s >> 2)]
                *(f32x4*)&wc[0] = *(const f32x4*)(wl + wrow * F67H::WP + 4 * kq);
                *(f32x4*)&wc[4] = *(const f32x4*)(wl + wrow * F67H::WP + 4 * kq + 16);
                const int pix = 4 * (li & 3) + (li >> 2), sw = ((li >> 1) & 1) << 4;        // bit 3 of pix (and of 16 + pix) is bit 1 of li
                const float* ap = a6s + (wave * 32 + pix) * F67H::PA + 4 * kq;
                const f32x4 a00 = *(const f32x4*)(ap + sw), a01 = *(const f32x4*)(ap + (sw ^ 16));
                const f32x4 a10 = *(const f32x4*)(ap + 16 * F67H::PA + sw), a11 = *(const f32x4*)(ap + 16 * F67H::PA + (sw ^ 16));
                f32x4 t0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, t1 = t0;                     // two independent chains, interleaved
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    t0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[s], a00[s], t0, 0, 0, 0);
                    t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[s], a10[s], t1, 0, 0, 0);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    t0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[4 + s], a01[s], t0, 0, 0, 0);
                    t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[4 + s], a11[s], t1, 0, 0, 0);
                }
                // D rows of lane kq: n = 4 blk(kq) + r, n-block blk(kq) = (e, px) blocks (B0, B2, B1, B3) -> ring block rows (0, 3, 1, 2)
                const int rblk = kq == 0 ? 0 : (kq == 1 ? 3 : (kq == 2 ? 1 : 2));
                float* tw = tb + ((((8 * grp + wave) & (F67::TSLOTS - 1)) * 16) + 4 * rblk) * F67H::TW + pix + 1 - (kq >> 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) { tw[r * F67H::TW] = t0[r]; tw[r * F67H::TW + 16] = t1[r]; }
            }
            if (has_next) {      // every wave is past the first barrier: nobody reads the current strip any more, and the word holds the maximum
                float S, invS;
                h2_scale(*mword, S, invS);
                unscale = invS * inv_sw;
#pragma unroll
                for (int j = 0; j < C::NLD; ++j) h2_store(smem, j, rsub, loffb, stg[j], S);
            }
            __syncthreads();
            if (tid == 0) *mword = 0;            // read by everyone before the barrier above; its next atomics come two groups later
            if constexpr (DIAG) { const unsigned long long t = wu_stamp(); dg[3] += t - dt; dt = t; }
            // ---- gather: new a6 row y finishes output rows 2y - 1 (phase a = 1 of row y - 1) and 2y (a = 0)
            // n-block (e, px) sits at ring block row {0, 1, 3, 2}[(1 - e) 2 + px]; the px = 1 blocks are stored one float to the left, so
            // the halo column of rx = 0 is float lane >> 1 in both
            const int px = lane & 1, xh = lane >> 1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = wave + 8 * h;                                              // wave-uniform
                const int y = 8 * grp + (k >> 1), e = k & 1;
                if (2 * y - 1 + e < 0) continue;
                const int nb = (e ? px : 3 - px) * 4;
                const float* ra = tb + ((((y - 1) & (F67::TSLOTS - 1)) * 16) + nb) * F67H::TW + xh;
                const float* rb = tb + (((y & (F67::TSLOTS - 1)) * 16) + nb) * F67H::TW + xh;
                float ta0 = ra[0], ta1 = ra[F67H::TW + 1];
                const float tb0 = rb[2 * F67H::TW], tb1 = rb[3 * F67H::TW + 1];
                if (y == 0) { ta0 = 0.0f; ta1 = 0.0f; }                                  // row -1: zero padding
                const float v = ((ta0 + ta1) + (tb0 + tb1)) + b7;
                const float rr = f67_sigmoid(v);
                const float d = xv[h] - rr;
                s2 = fmaf(d, d, s2);
                s1 += fabsf(d);
            }
            if (grp == C::NGRP - 1 && wave == 0) {                                      // output row 63: a6 row 31 and the padding
                const int nb = (3 - px) * 4;                                             // n-blocks (e = 0, px)
                const float* ra = tb + (((31 & (F67::TSLOTS - 1)) * 16) + nb) * F67H::TW + xh;
                const float v = (ra[0] + ra[F67H::TW + 1]) + b7;
                const float rr = f67_sigmoid(v);
                const float d = xtail - rr;
                s2 = fmaf(d, d, s2);
                s1 += fabsf(d);
            }
            if constexpr (DIAG) {
                asm volatile("" ::"v"(s2), "v"(s1));
                const unsigned long long t = wu_stamp(); dg[4] += t - dt; dt = t;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s2 += __shfl_down(s2, off, 64);
            s1 += __shfl_down(s1, off, 64);
        }
        if (lane == 0) {
            errpart[((size_t)cell * F67::NPARTS + wave) * 2 + 0] = s2;
            errpart[((size_t)cell * F67::NPARTS + wave) * 2 + 1] = s1;
        }
    }
    if constexpr (DIAG) {
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 5; ++k) diag[((size_t)blockIdx.x * 8 + wave) * 5 + k] = dg[k];
        }
    }
}

// U = G W_eff G^T per (phase, cin, cout), evaluated in double and rounded once.
// Layout [wave = phase * 2 + half][(k * 9 + xi) * KS + 4 q + j][lane]:
//   U[xi = 3 r + c][ci = 16 q + 4 kq + j][co = 16 (half * NSW + k) + li].
template <class C>
size_t pack_frags(const float* hwio, float* dst)
{
    const size_t total = (size_t)8 * C::NB * 64;
    if (!dst) return total;
    static const double G[3][2] = {{1, 0}, {1, 1}, {0, 1}};
    for (int ph = 0; ph < 4; ++ph)
        for (int half = 0; half < 2; ++half)
            for (int k = 0; k < C::NSW; ++k)
                for (int xi = 0; xi < 9; ++xi)
                    for (int kk = 0; kk < C::KS; ++kk)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int a = ph >> 1, b = ph & 1, li = lane & 15, kq = lane >> 4, q = kk >> 2, j = kk & 3;
                            const int ci = 16 * q + 4 * kq + j, co = 16 * (half * C::NSW + k) + li, r = xi / 3, c = xi % 3;
                            double u = 0.0;
                            for (int ry = 0; ry < 2; ++ry)
                                for (int rx = 0; rx < 2; ++rx) {
                                    double weff = 0.0;   // taps (dy,dx) whose upsampled source is stored pixel (a-1+ry, b-1+rx)
                                    for (int dy = -1; dy <= 1; ++dy)
                                        for (int dx = -1; dx <= 1; ++dx)
                                            if (((a + dy) >> 1) == a + ry - 1 && ((b + dx) >> 1) == b + rx - 1)
                                                weff += (double)hwio[((size_t)((dy + 1) * 3 + (dx + 1)) * C::CIN + ci) * C::COUT + co];
                                    u += G[r][ry] * weff * G[c][rx];
                                }
                            dst[((size_t)(ph * 2 + half) * C::NB + (k * 9 + xi) * C::KS + kk) * 64 + lane] = (float)u;
                        }
    return total;
}

unsigned long long* g_wu_diag[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
int g_wu_diag_blocks[6] = {0, 0, 0, 0, 0, 0};

template <class C>
hipError_t launch(int layer, const float* in, const float* ufrag, const float* ep, float* out, int64_t n_cells, hipStream_t stream)
{
    static int resident = 0;
    static const bool diag = getenv("CS_WINO_DIAG") != nullptr;
    constexpr int LDSB = C::LDS;
    if (!resident) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_wino_up_kernel<C, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)conv_wino_up_kernel<C, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        if (e != hipSuccess) return e;
        int dev = 0, cus = 0, per_cu = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv_wino_up_kernel<C, false>, C::THREADS, LDSB);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        resident = cus * per_cu;
        if (diag) {
            if ((e = hipMalloc(&g_wu_diag[layer], (size_t)resident * 40 * sizeof(unsigned long long))) != hipSuccess) return e;
            g_wu_diag_blocks[layer] = resident;
        }
    }
    const long total = (long)n_cells * C::NGRP;
    if (total <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(total < resident ? total : resident);
    if (diag)
        hipLaunchKernelGGL((conv_wino_up_kernel<C, true>), dim3(grid), dim3(C::THREADS), LDSB, stream, in, ufrag, ep, out, (long)n_cells,
                           g_wu_diag[layer]);
    else
        hipLaunchKernelGGL((conv_wino_up_kernel<C, false>), dim3(grid), dim3(C::THREADS), LDSB, stream, in, ufrag, ep, out, (long)n_cells,
                           (unsigned long long*)nullptr);
    return hipGetLastError();
}

}  // namespace

int conv67_fused_nparts() { return F67::NPARTS; }

hipError_t launch_conv67_fused(const float* a5, const float* ufrag, const float* ep, const float* x, const float* weff_dev,
                               const float* b7_dev, float* errpart, int64_t n_cells, hipStream_t stream)
{
    static int cus = 0;
    static const bool diag = getenv("CS_WINO_DIAG") != nullptr;
    if (!cus) {
        hipError_t e = hipFuncSetAttribute((const void*)conv67_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, F67::LDS);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)conv67_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, F67::LDS);
        if (e != hipSuccess) return e;
        int dev = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        if (diag && !g_wu_diag[3]) {                               // slot 3 of the diagnostic table: the fused kernel
            if ((e = hipMalloc(&g_wu_diag[3], (size_t)cus * 40 * sizeof(unsigned long long))) != hipSuccess) return e;
            g_wu_diag_blocks[3] = cus;
        }
    }
    if (n_cells <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(n_cells < cus ? n_cells : cus);
    if (diag)
        hipLaunchKernelGGL(conv67_fused_kernel<true>, dim3(grid), dim3(WUL6::THREADS), F67::LDS, stream, a5, ufrag, ep, x, weff_dev, b7_dev,
                           errpart, (long)n_cells, g_wu_diag[3]);
    else
        hipLaunchKernelGGL(conv67_fused_kernel<false>, dim3(grid), dim3(WUL6::THREADS), F67::LDS, stream, a5, ufrag, ep, x, weff_dev, b7_dev,
                           errpart, (long)n_cells, (unsigned long long*)nullptr);
    return hipGetLastError();
}

// helpers of every fp16-split packer: the power of two that puts max|w| into [2^14, 2^15), and one value's two fp16 terms
float f16x2_weight_scale(const float* w, size_t n)
{
    float m = 0.0f;
    for (size_t i = 0; i < n; ++i) m = fmaxf(m, fabsf(w[i]));
    if (!(m > 0.0f) || !std::isfinite(m)) return 1.0f;
    int e;
    frexpf(m, &e);                       // m = f 2^e, f in [0.5, 1)
    return ldexpf(1.0f, 15 - e);         // S m = f 2^15 in [2^14, 2^15)
}

// one value -> its two fp16 terms (bit patterns): hi = fp16(S w), lo = fp16(S w - hi)
void f16x2_split(float w, float S, uint16_t& hi, uint16_t& lo)
{
    const float v = w * S;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    memcpy(&hi, &h, 2);
    memcpy(&lo, &l, 2);
}

// conv6's folded weights (pack_generic_folded(64, 32, ...): [phase 4][tap 4][cin 64][cout 32]) as two fp16 planes in conv67_h2_kernel's
// order: [wave = phase * 2 + half][tap][block 2][plane 2][lane 64][8]: element j = plane of S_w W_eff[phase][tap][32 block + 8 kq + j][16 half + li];
// *inv_sw = 1 / S_w
size_t pack_conv6_f16x2(const float* weff, uint16_t* dst, float* inv_sw)
{
    const size_t n = (size_t)8 * 4 * 2 * 2 * 64 * 8;
    if (!dst) return n;
    const float S = f16x2_weight_scale(weff, (size_t)16 * 64 * 32);
    if (inv_sw) *inv_sw = 1.0f / S;
    for (int w = 0; w < 8; ++w)
        for (int t = 0; t < 4; ++t)
            for (int k = 0; k < 2; ++k)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int ph = w >> 1, half = w & 1, li = l & 15, kq = l >> 4;
                        const float v = weff[((size_t)(ph * 4 + t) * 64 + 32 * k + 8 * kq + j) * 32 + 16 * half + li];
                        uint16_t pl[2];
                        f16x2_split(v, S, pl[0], pl[1]);
                        for (int p = 0; p < 2; ++p) dst[((((((size_t)w * 4 + t) * 2 + k) * 2 + p) * 64) + l) * 8 + j] = pl[p];
                    }
    return n;
}

hipError_t launch_conv67_h2(const float* a5, const uint16_t* wplanes, float inv_sw, const float* ep, const float* x, const float* weff_dev,
                            const float* b7_dev, float* errpart, int64_t n_cells, hipStream_t stream)
{
    static int cus = 0;
    static const bool diag = getenv("CS_WINO_DIAG") != nullptr;
    if (!cus) {
        hipError_t e = hipFuncSetAttribute((const void*)conv67_h2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, F67H::LDS);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)conv67_h2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, F67H::LDS);
        if (e != hipSuccess) return e;

        int dev = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        if (diag && !g_wu_diag[3]) {                               // slot 3 of the diagnostic table: the fused kernel (any form)
            if ((e = hipMalloc(&g_wu_diag[3], (size_t)cus * 40 * sizeof(unsigned long long))) != hipSuccess) return e;
            g_wu_diag_blocks[3] = cus;
        }
    }
    if (n_cells <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(n_cells < cus ? n_cells : cus);
    if (diag)
        hipLaunchKernelGGL(conv67_h2_kernel<true>, dim3(grid), dim3(WUL6::THREADS), F67H::LDS, stream, a5, (const f16x8*)wplanes, ep, x,
                           weff_dev, b7_dev, errpart, (long)n_cells, inv_sw, g_wu_diag[3]);
    else
        hipLaunchKernelGGL(conv67_h2_kernel<false>, dim3(grid), dim3(WUL6::THREADS), F67H::LDS, stream, a5, (const f16x8*)wplanes, ep, x,
                           weff_dev, b7_dev, errpart, (long)n_cells, inv_sw, (unsigned long long*)nullptr);
    return hipGetLastError();
}

size_t pack_wino_up_fragments(int layer, const float* hwio, float* dst)
{
    return layer == 5 ? pack_frags<WUL6>(hwio, dst) : pack_frags<WUL5>(hwio, dst);
}

hipError_t launch_conv_wino_up(int layer, const float* in, const float* ufrag, const float* ep, float* out, int64_t n_cells,
                               hipStream_t stream)
{
    if (layer == 5) return launch<WUL6>(layer, in, ufrag, ep, out, n_cells, stream);
    if (layer == 4) return launch<WUL5>(layer, in, ufrag, ep, out, n_cells, stream);
    return hipErrorInvalidValue;
}

}  // namespace cs

// Diagnostic only (CS_WINO_DIAG=1): per-wave phase cycles of the LAST launch of `layer` (4 or 5), averaged over waves.
extern "C" int cs_debug_wino_up_diag(int layer, double out5[5])
{
    using namespace cs;
    if (layer < 3 || layer > 5 || !g_wu_diag[layer]) return -1;    // 3: the fused conv6 + conv7 kernel
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    const size_t n = (size_t)g_wu_diag_blocks[layer] * 40;
    unsigned long long* h = new unsigned long long[n];
    if (hipMemcpy(h, g_wu_diag[layer], n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) { delete[] h; return -3; }
    for (int k = 0; k < 5; ++k) out5[k] = 0.0;
    for (size_t i = 0; i < n; ++i) out5[i % 5] += (double)h[i];
    for (int k = 0; k < 5; ++k) out5[k] /= (double)(n / 5);
    delete[] h;
    return 0;
}
