// conv_generic_x3.hip -- the run-time-shaped conv of conv_generic.hip (version 2 and its folded-upsample form) with the
// fp32 contraction carried by the bf16 matrix pipe, for the inference path of NON-reference architectures
// (create_improved_autoencoder(input_shape), CAE_improved_modeltrain.py:184; BASELINE.json configs[4]).
//
// Every fp32 operand is split into three bf16 terms (x = x1 + x2 + x3 to 2^-24; conv45_bf16x3.hip has the algebra and
// the hardware check of the technique) and a product is taken as six v_mfma_f32_16x16x32_bf16 partial products, each
// exact in the fp32 accumulator: six 16-cycle instructions per 32 channels against eight 32-cycle
// v_mfma_f32_16x16x4_f32, and the bf16 instruction leaves half of its issue slots to the VALU, LDS and memory
// instructions around it.
//   * activations: split ONCE per staged element (5.5 VALU instructions per value) into three bf16 planes kept side by
//     side in the staged pixel ([x1: cin][x2: cin][x3: cin] + 32 B: 6 cin + 32 B is twice an odd number of 16-byte
//     slots, which puts the 16 lanes of every ds_read_b128 lane group on 16 distinct slots -- enumerated); a tile's A
//     fragment of one (tap, 32-channel block) is one ds_read_b128 per plane, all at immediate offsets from one
//     address register per tile (cin is a template parameter: 32, 64 or 128 -- at 256 no strip of four tile rows fits the LDS);
//   * weights: split on the host when the model is loaded (pack_generic_bf16x3), read from L2 as one 16-byte load
//     per lane and plane and (tap, block), reused for every tile the wave owns (TPW = 4 or 8 live accumulators).
// Tile ownership, strips, epilogues and the folded upsample (four 2x2-tap phase convs on the stored grid, 4/9 of the
// multiply-adds) are those of conv_generic.hip.  Training keeps the fp32 kernels (its weights change every step).
#include "common.hpp"

#include <cstdlib>
#include <cstring>

namespace cs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

// ---- H2: the same kernels with the contraction as a TWO-term fp16 split (three products; DESIGN.md 3h, conv_wino_up.hip has the
// algebra and the hardware facts).  Weights carry a per-layer power-of-two scale (pack_generic_f16x2, inv_sw undoes it); a staged
// strip is scaled by the power of two that puts ITS max|x| into [2^14, 2^15): the strip is staged as fp32 first (the maximum has to
// be known before anything is split), then split IN PLACE -- a pixel's fp32 values and its [hi plane | lo plane] are the same
// 4 cin bytes, and one wave owns a pixel (all of a wave's reads of an instruction precede its writes), so the second pass needs
// no barrier of its own beyond the one that publishes the maximum.
__device__ __forceinline__ unsigned int g3h_rowmax(unsigned int m)
{
    unsigned int o;
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, true);  m = m > o ? m : o;     // quad_perm [1,0,3,2]
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, true);  m = m > o ? m : o;     // quad_perm [2,3,0,1]
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0x124, 0xF, 0xF, true); m = m > o ? m : o;     // row_ror:4
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xF, 0xF, true); m = m > o ? m : o;     // row_ror:8
    return m;
}
__device__ __forceinline__ void g3h_scale(unsigned int mbits, float& S, float& invS)
{
    int E = (int)((mbits >> 23) & 0xffu);
    E = E < 40 ? 40 : (E > 254 ? 254 : E);
    S = __builtin_bit_cast(float, (unsigned int)(268 - E) << 23);          // 2^(14 - (E - 127))
    invS = __builtin_bit_cast(float, (unsigned int)(E - 14) << 23);
}

struct Gen3Args {
    const float* in;       // stored input [n][Hs][Ws][cin]  (Hs = H/2 for the folded form)
    const uint16_t* w;     // pack_generic_bf16x3: [step = tap * cin/32 + block][plane][cout_pad][kq][8] bf16 (H2: two fp16 planes)
    float inv_sw;          // H2: 1 / the weights' scale
    const float* ep;       // [3][cout]
    float* out;
    long n;
    int H, W, cin, cout, epi;
    int SR, nmg, nslw;     // strip rows, tile (phase) groups, slices per workgroup pass (nmg * nslw = 8)
};

__device__ __forceinline__ void split4(const f32x4& v, bf16x4& h1, bf16x4& h2, bf16x4& h3)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 a1 = (__bf16)v[j];
        const float r1 = v[j] - (float)a1;
        const __bf16 a2 = (__bf16)r1;
        const float r2 = r1 - (float)a2;
        h1[j] = a1;
        h2[j] = a2;
        h3[j] = (__bf16)r2;
    }
}

// acc += a * b over one 32-channel block, a and b as three bf16 planes: the six products down to 2^-16
__device__ __forceinline__ f32x4 mac6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 acc)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

template <int CIN, int TPW, bool FOLD, bool H2 = false>
__global__ __launch_bounds__(512, 2) void conv_generic_x3_kernel(Gen3Args g)
{
    constexpr int NPL = H2 ? 2 : 3;                       // planes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    constexpr int cin = CIN;
    const int H = g.H, W = g.W, cout = g.cout, SR = g.SR;
    const int Hs = FOLD ? H / 2 : H, Ws = FOLD ? W / 2 : W;
    const int R = FOLD ? SR / 2 + 2 : SR + 2, WP = Ws + 2;
    constexpr int psb = 2 * NPL * CIN + 32;               // bytes per staged pixel: the planes + pad (twice an odd number of 16-byte slots)
    constexpr int PB = 2 * CIN;                           // byte offset of a plane inside the pixel
    constexpr int nkb = CIN / 32;
    unsigned int* const mxw = (unsigned int*)(smem + (size_t)R * WP * psb);      // H2: the strip's max|x| (one word behind the strip)
    float us = 1.0f;                                      // H2: 1 / (strip scale x weight scale), applied with the bias
    if constexpr (H2) { if (tid == 0) *mxw = 0; }
    const int nstrip = H / SR;
    const int cpb = g.nslw * 16, ncb = (cout + cpb - 1) / cpb;
    const int coutp = (cout + 15) & ~15;
    const int slice = wave % g.nslw, mg = wave / g.nslw;
    const size_t wstep = (size_t)coutp * 4;               // bf16x8 elements per (step, plane)

    const long items = g.n * nstrip * ncb;
    for (long item = blockIdx.x; item < items; item += gridDim.x) {
        const int cb = (int)(item % ncb);
        const long cs_ = item / ncb;
        const int y0 = (int)(cs_ % nstrip) * SR;
        const long cell = cs_ / nstrip;
        const float* src = g.in + (size_t)cell * Hs * Ws * cin;
        const int ybase = FOLD ? (y0 / 2 - 1) : (y0 - 1);

        __syncthreads();                                  // previous item's readers are done
        if constexpr (H2) {
            constexpr int c4n = CIN / 4;
            float am = 0.0f;
            for (int e = tid; e < R * WP * c4n; e += 512) {
                const int c4 = e % c4n, pix = e / c4n;
                const int r = pix / WP, c = pix - r * WP;
                const int sy = ybase + r, sx = c - 1;
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) v = *(const f32x4*)(src + ((size_t)sy * Ws + sx) * cin + 4 * c4);
                *(f32x4*)(smem + pix * psb + c4 * 16) = v;                 // fp32 for now: the same bytes become [hi | lo] below
                const float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];   // (scalars: see conv45_bf16x3.hip, h2_absmax4)
                am = fmaxf(am, fmaxf(fmaxf(fabsf(a0), fabsf(a1)), fmaxf(fabsf(a2), fabsf(a3))));
            }
            const unsigned int m = g3h_rowmax(__builtin_bit_cast(unsigned int, am));
            if (li == 0) atomicMax(mxw, m);
            __syncthreads();
            float S, invS;
            g3h_scale(*mxw, S, invS);
            us = invS * g.inv_sw;
            // in place, one wave per pixel group: lane = (pixel of the group, channel quad)
            constexpr int ppw = 64 / c4n;                 // pixels per wave and pass (2, 4 or 8)
            const int c4 = lane % c4n, pl = lane / c4n;
            for (int p0 = wave * ppw; p0 < R * WP; p0 += 8 * ppw) {
                const int pix = p0 + pl;
                if (pix < R * WP) {
                    char* base = smem + pix * psb;
                    const f32x4 v = *(const f32x4*)(base + c4 * 16) * S;
                    const f16x4 hi = __builtin_convertvector(v, f16x4);
                    const f32x4 rr = v - __builtin_convertvector(hi, f32x4);      // exact in fp32
                    *(f16x4*)(base + c4 * 8) = hi;
                    *(f16x4*)(base + PB + c4 * 8) = __builtin_convertvector(rr, f16x4);
                }
            }
        } else {
            constexpr int c4n = CIN / 4;
            for (int e = tid; e < R * WP * c4n; e += 512) {
                const int c4 = e % c4n, pix = e / c4n;
                const int r = pix / WP, c = pix - r * WP;
                const int sy = ybase + r, sx = c - 1;
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) v = *(const f32x4*)(src + ((size_t)sy * Ws + sx) * cin + 4 * c4);
                bf16x4 h1, h2, h3;
                split4(v, h1, h2, h3);
                char* d = smem + pix * psb + c4 * 8;
                *(bf16x4*)d = h1;
                *(bf16x4*)(d + PB) = h2;
                *(bf16x4*)(d + 2 * PB) = h3;
            }
        }
        __syncthreads();
        if constexpr (H2) { if (tid == 0) *mxw = 0; }     // read by everyone before the barrier above; the next atomics are behind the next item's first barrier

        const int cbase = cb * cpb + slice * 16;
        if (cbase >= cout) continue;                      // wave-uniform: this slice does not exist
        const int co = cbase + li;
        const bool cok = co < cout;
        // 16-byte fragments of 8 sixteen-bit values (bf16 or fp16: the loads do not care; the MFMA builtin is chosen by H2)
        const bf16x8* wl = (const bf16x8*)g.w + (size_t)co * 4 + kq;      // + (step * NPL + plane) * wstep
        auto load_b = [&](int step, bf16x8 (&b)[3]) {
#pragma unroll
            for (int p = 0; p < NPL; ++p) b[p] = wl[((size_t)step * NPL + p) * wstep];
        };
        auto read_a = [&](int off, bf16x8 (&a)[3]) {
            a[0] = *(const bf16x8*)(smem + off);
            a[1] = *(const bf16x8*)(smem + off + PB);
            if constexpr (!H2) a[2] = *(const bf16x8*)(smem + off + 2 * PB);
        };
        auto mac = [&](const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 acc) {
            if constexpr (H2) {       // planes: [0] hi, [1] lo; one magnitude per instruction
                const f16x8 ah = __builtin_bit_cast(f16x8, a[0]), al = __builtin_bit_cast(f16x8, a[1]);
                const f16x8 bh = __builtin_bit_cast(f16x8, b[0]), bl = __builtin_bit_cast(f16x8, b[1]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
                return acc;
            } else {
                return mac6(a, b, acc);
            }
        };
        // one (tap, block) step over the wave's TPW tiles: the next tile's plane reads are issued ahead of the MFMAs of the
        // current one and pinned there (unpinned, the scheduler hoists every tile's reads above the first MFMA)
        auto step_tiles = [&](const int (&base)[TPW], int kb, const bf16x8 (&b)[3], f32x4 (&acc)[TPW]) {
            bf16x8 a[3];
            read_a(base[0] + kb * 64, a);
            __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                bf16x8 an[3] = {a[0], a[1], a[2]};
                if (t + 1 < TPW) read_a(base[t + 1] + kb * 64, an);
                acc[t] = mac(a, b, acc[t]);
                a[0] = an[0]; a[1] = an[1]; a[2] = an[2];
                __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, H2 ? 3 : 6, 0);
            }
        };

        if constexpr (!FOLD) {
            constexpr int PPW = TPW / 2;                  // vertical tile pairs per wave
            const int TPR = W / 16;
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
            bf16x8 bn[3];
            load_b(0, bn);
            const int nstep = 9 * nkb;
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap % 3;     // staged coordinates: +1 halo, -1 tap
                int base[TPW];
#pragma unroll
                for (int t = 0; t < TPW; ++t)
                    base[t] = ((tpy[t >> 1] + (t & 1) + dy) * WP + tpx[t >> 1] + li + dx) * psb + kq * 16;
#pragma unroll 1
                for (int kb = 0; kb < nkb; ++kb) {
                    bf16x8 b[3] = {bn[0], bn[1], bn[2]};
                    const int s = tap * nkb + kb;
                    if (s + 1 < nstep) load_b(s + 1, bn);
                    step_tiles(base, kb, b, acc);
                }
            }
            // D: lane = channel li of the slice, registers = pixels 4 kq .. 4 kq + 3 of the tile
            if (cok) {
                const float bias = g.epi == GEN_EPI_PLAIN ? 0.0f : g.ep[co];
                if (g.epi == GEN_EPI_BN_POOL) {
                    const float bns = g.ep[cout + co], bnt = g.ep[2 * cout + co];
                    auto post = [&](float v) { v = fmaf(v, us, bias); v = fmaxf(v, 0.0f); return fmaf(v, bns, bnt); };     // us = 1 unless H2
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
                            const float z = fmaf(acc[t][r], us, bias);
                            float v;
                            if (g.epi == GEN_EPI_BN) v = fmaf(fmaxf(z, 0.0f), bns, bnt);
                            else if (g.epi == GEN_EPI_RELU) v = fmaxf(z, 0.0f);
                            else if (g.epi == GEN_EPI_SIGMOID) v = 1.0f / (1.0f + expf(-z));
                            else v = z;
                            o[((size_t)py * W + px) * cout] = v;
                        }
                }
            }
        } else {
            // folded upsample: the wave owns ALL TPW stored-pixel tiles of the strip for 4 / nmg output phases
            const int TPRs = Ws / 16;
            const int nph = 4 / g.nmg;
            const float bias = cok ? g.ep[co] : 0.0f;
            const float bns = (cok && g.epi == GEN_EPI_BN) ? g.ep[cout + co] : 1.0f, bnt = (cok && g.epi == GEN_EPI_BN) ? g.ep[2 * cout + co] : 0.0f;
#pragma unroll 1
            for (int pi = 0; pi < nph; ++pi) {
                const int phase = mg * nph + pi, a = phase >> 1, b = phase & 1;
                f32x4 acc[TPW];
#pragma unroll
                for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                bf16x8 bn[3];
                load_b(phase * 4 * nkb, bn);
#pragma unroll 1
                for (int tap = 0; tap < 4; ++tap) {
                    const int ry = tap >> 1, rx = tap & 1;
                    int base[TPW];
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        const int ys = t / TPRs, xs = (t % TPRs) * 16 + li;
                        // staged row 0 is stored row y0/2 - 1: stored (ys + a - 1 + ry, xs + b - 1 + rx) -> staged (ys + a + ry, xs + b + rx)
                        base[t] = ((ys + a + ry) * WP + xs + b + rx) * psb + kq * 16;
                    }
#pragma unroll 1
                    for (int kb = 0; kb < nkb; ++kb) {
                        bf16x8 bb[3] = {bn[0], bn[1], bn[2]};
                        const int s = tap * nkb + kb;
                        if (s + 1 < 4 * nkb) load_b(phase * 4 * nkb + s + 1, bn);
                        step_tiles(base, kb, bb, acc);
                    }
                }
                if (cok) {
                    float* o = g.out + ((size_t)cell * H + y0 + a) * W * cout + (size_t)b * cout + co;
#pragma unroll
                    for (int t = 0; t < TPW; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ys = t / TPRs, xs = (t % TPRs) * 16 + 4 * kq + r;
                            const float z = fmaxf(fmaf(acc[t][r], us, bias), 0.0f);
                            o[((size_t)(2 * ys) * W + 2 * xs) * cout] = g.epi == GEN_EPI_BN ? fmaf(z, bns, bnt) : z;
                        }
                }
            }
        }
    }
}

// ---- the 1-filter last conv behind an UpSampling2D (cout = 1, sigmoid) on the matrix pipe ---------------------------------------
// conv_generic.hip runs it on the vector ALU (16 x cin multiply-adds per stored pixel).  As in the reference graph's fused
// conv6 + conv7 kernel, contracting the channels FIRST turns it into a GEMM with a full N: with the folded kernels
// W_eff[n = (a 2 + b) 4 + (ry 2 + rx)][c] (pack_generic_folded with cout = 1),
//     T[pixel][n] = sum_c a[pixel][c] W_eff[n][c]                  one six-MFMA chain per 16 staged pixels and 32 channels
//     out(2y + a, 2x + b) = sigmoid(b7 + sum_{ry,rx} T[(y + a - 1 + ry, x + b - 1 + rx)][n(a, b, ry, rx)])      4 LDS reads per output
// The strip (SRS stored rows + halo, flattened: T is per pixel, so tiles may wrap rows) is staged as three bf16 planes; T lives in
// LDS n-major with a pitch = 4 (mod 8) floats, which keeps the gather's 32 lanes (16 columns x 2 column phases) on 31 banks.
template <int CIN>
__global__ __launch_bounds__(512) void conv_last_x3_kernel(const float* __restrict__ in, const bf16x8* __restrict__ wpl,
                                                           const float* __restrict__ ep, float* __restrict__ out, long n, int H, int W,
                                                           int SRS, int NT /* 16-pixel tiles per strip */, int P /* T pitch */)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    constexpr int psb = 6 * CIN + 32, PB = 2 * CIN, nkb = CIN / 32, c4n = CIN / 4;
    const int Hs = H / 2, Ws = W / 2, R = SRS + 2, WP = Ws + 2, NPX = R * WP;
    float* const T = (float*)(smem + (size_t)NT * 16 * psb);
    bf16x8 B[nkb][3];
#pragma unroll
    for (int kb = 0; kb < nkb; ++kb)
#pragma unroll
        for (int p = 0; p < 3; ++p) B[kb][p] = wpl[(kb * 3 + p) * 64 + lane];
    const float bias = ep[0];
    for (int i = tid; i < NT * 16 * psb / 16; i += 512) ((f32x4*)smem)[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};   // pad pixels stay zero
    const int nstrip = Hs / SRS;
    for (long item = blockIdx.x; item < n * nstrip; item += gridDim.x) {
        const int ys0 = (int)(item % nstrip) * SRS;
        const long cell = item / nstrip;
        const float* src = in + (size_t)cell * Hs * Ws * CIN;
        __syncthreads();                                  // the previous item's gather is done with T; its tiles with the strip
        for (int e = tid; e < NPX * c4n; e += 512) {
            const int c4 = e % c4n, pix = e / c4n;
            const int r = pix / WP, c = pix - r * WP;
            const int sy = ys0 - 1 + r, sx = c - 1;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) v = *(const f32x4*)(src + ((size_t)sy * Ws + sx) * CIN + 4 * c4);
            bf16x4 h1, h2, h3;
            split4(v, h1, h2, h3);
            char* d = smem + pix * psb + c4 * 8;
            *(bf16x4*)d = h1;
            *(bf16x4*)(d + PB) = h2;
            *(bf16x4*)(d + 2 * PB) = h3;
        }
        __syncthreads();
        for (int t = wave; t < NT; t += 8) {
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            const char* ap = smem + (16 * t + li) * psb + kq * 16;
#pragma unroll
            for (int kb = 0; kb < nkb; ++kb) {
                const bf16x8 a[3] = {*(const bf16x8*)(ap + kb * 64), *(const bf16x8*)(ap + kb * 64 + PB), *(const bf16x8*)(ap + kb * 64 + 2 * PB)};
                acc = mac6(a, B[kb], acc);
            }
            // D: lane = n, registers = pixels 16 t + 4 kq + r
            *(f32x4*)(T + li * P + 16 * t + 4 * kq) = acc;
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * SRS * W; idx += 512) {
            const int Yl = idx / W, X = idx - Yl * W;
            const int ys = Yl >> 1, a = Yl & 1, xs = X >> 1, b = X & 1;
            // stored (ys0 + ys + a - 1 + ry, xs + b - 1 + rx) = staged (ys + a + ry, xs + b + rx); n = ((a 2 + b) 2 + ry) 2 + rx
            const float* t0 = T + ((a * 2 + b) * 4) * P + (ys + a) * WP + xs + b;
            const float s0 = t0[0] + t0[P + 1];
            const float s1 = t0[2 * P + WP] + t0[3 * P + WP + 1];
            const float z = (s0 + s1) + bias;
            out[((size_t)cell * H + 2 * ys0 + Yl) * W + X] = 1.0f / (1.0f + expf(-z));
        }
    }
}

// pack_generic_bf16x3 on the device (training: the weights change every step).  One thread per (tap, block, filter, channel).
__global__ void pack_generic_bf16x3_kernel(const float* __restrict__ w, int ntaps, int cin, int cout, uint16_t* __restrict__ dst)
{
    const int coutp = (cout + 15) & ~15, nkb = cin / 32;
    const long total = (long)ntaps * nkb * coutp * 32;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i & 31);
        const int co = (int)((i >> 5) % coutp);
        const long tk = (i >> 5) / coutp;                       // tap * nkb + block
        const int kb = (int)(tk % nkb), t = (int)(tk / nkb);
        const float v = co < cout ? w[((size_t)t * cin + 32 * kb + k) * cout + co] : 0.0f;
        const __bf16 a1 = (__bf16)v;
        const float r1 = v - (float)a1;
        const __bf16 a2 = (__bf16)r1;
        const __bf16 a3 = (__bf16)(r1 - (float)a2);
        const size_t base = ((size_t)tk * 3 * coutp + co) * 32 + k;
        dst[base] = __builtin_bit_cast(uint16_t, a1);
        dst[base + (size_t)coutp * 32] = __builtin_bit_cast(uint16_t, a2);
        dst[base + (size_t)2 * coutp * 32] = __builtin_bit_cast(uint16_t, a3);
    }
}

uint16_t bf16_rne(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
float bf16_val(uint16_t h)
{
    const uint32_t u = (uint32_t)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}

constexpr size_t X3_MAX_LDS = 150 * 1024;

// plain form: SR rows per strip, nmg tile groups x nslw slices = 8 waves, TPW tiles per wave (conv_generic.hip's version-2 plan
// with the three-plane strip's bytes)
bool x3_plan(int H, int W, int cin, int cout, int* SR, int* nmg, int* nslw, int* tpw, size_t* lds)
{
    if (!(cin == 32 || cin == 64 || cin == 128) || !(W == 16 || W == 32 || W == 64 || W == 128) || cout < 16) return false;
    const int slices = (cout + 15) / 16;
    int ns = 1;
    while (ns * 2 <= slices && ns < 8) ns *= 2;
    const int mg = 8 / ns, TPR = W / 16, psb = 6 * cin + 32;
    for (int t = 8; t >= 4; t /= 2) {                     // 16 live accumulators + the three-plane fragments spill at 256 VGPRs
        const int pairs = (t / 2) * mg;
        if (pairs % TPR) continue;
        const int sr = 2 * pairs / TPR;
        if (sr < 2 || H % sr) continue;
        const size_t bytes = (size_t)(sr + 2) * (W + 2) * psb;
        if (bytes > X3_MAX_LDS) continue;
        *SR = sr; *nmg = mg; *nslw = ns; *tpw = t; *lds = bytes;
        return true;
    }
    return false;
}

// folded form: nmg in {1, 2, 4} phase groups, every wave owns all TPW = (SR / 2) (Ws / 16) tiles of the strip
bool x3f_plan(int H, int W, int cin, int cout, int* SR, int* nmg, int* nslw, int* tpw, size_t* lds)
{
    const int Ws = W / 2;
    if (!(cin == 32 || cin == 64 || cin == 128) || !(Ws == 16 || Ws == 32 || Ws == 64) || cout < 32) return false;
    const int slices = (cout + 15) / 16;
    int ns = 2;
    while (ns * 2 <= slices && ns < 8) ns *= 2;
    const int mg = 8 / ns, TPRs = Ws / 16, psb = 6 * cin + 32;
    for (int t = 8; t >= 4; t /= 2) {
        if (t % TPRs) continue;
        const int srs = t / TPRs;
        if (srs < 1 || (H / 2) % srs) continue;
        const size_t bytes = (size_t)(srs + 2) * (Ws + 2) * psb;
        if (bytes > X3_MAX_LDS) continue;
        *SR = 2 * srs; *nmg = mg; *nslw = ns; *tpw = t; *lds = bytes;
        return true;
    }
    return false;
}

}  // namespace

int conv_generic_x3_takes(int H, int W, int cin, int cout, int ups)
{
    int SR, nmg, nslw, tpw;
    size_t lds;
    return (ups ? x3f_plan(H, W, cin, cout, &SR, &nmg, &nslw, &tpw, &lds) : x3_plan(H, W, cin, cout, &SR, &nmg, &nslw, &tpw, &lds)) ? 1 : 0;
}

// w: [ntaps][cin][cout] fp32 (the HWIO kernel with ntaps = 9, or pack_generic_folded's effective kernels with ntaps = 16).
// dst: [step = tap * cin/32 + block][plane][cout_pad][kq][8]: element j = plane of w[tap][32 block + 8 kq + j][co], zero for co >= cout
size_t pack_generic_bf16x3(int ntaps, int cin, int cout, const float* w, uint16_t* dst)
{
    const int coutp = (cout + 15) & ~15, nkb = cin / 32;
    const size_t total = (size_t)ntaps * nkb * 3 * coutp * 32;
    if (!dst) return total;
    for (int t = 0; t < ntaps; ++t)
        for (int kb = 0; kb < nkb; ++kb)
            for (int co = 0; co < coutp; ++co)
                for (int k = 0; k < 32; ++k) {
                    const float v = co < cout ? w[((size_t)t * cin + 32 * kb + k) * cout + co] : 0.0f;
                    const uint16_t w1 = bf16_rne(v);
                    const float r1 = v - bf16_val(w1);
                    const uint16_t w2 = bf16_rne(r1);
                    const float r2 = r1 - bf16_val(w2);
                    const uint16_t pl[3] = {w1, w2, bf16_rne(r2)};
                    for (int p = 0; p < 3; ++p) dst[((((size_t)t * nkb + kb) * 3 + p) * coutp + co) * 32 + k] = pl[p];
                }
    return total;
}

// the same layout with TWO fp16 planes of S_w w (S_w: the power of two that puts max|w| of the layer into [2^14, 2^15)):
// dst: [step][plane 2][cout_pad][kq][8]; *inv_sw = 1 / S_w
size_t pack_generic_f16x2(int ntaps, int cin, int cout, const float* w, uint16_t* dst, float* inv_sw)
{
    const int coutp = (cout + 15) & ~15, nkb = cin / 32;
    const size_t total = (size_t)ntaps * nkb * 2 * coutp * 32;
    if (!dst) return total;
    const float S = f16x2_weight_scale(w, (size_t)ntaps * cin * cout);
    *inv_sw = 1.0f / S;
    for (int t = 0; t < ntaps; ++t)
        for (int kb = 0; kb < nkb; ++kb)
            for (int co = 0; co < coutp; ++co)
                for (int k = 0; k < 32; ++k) {
                    const float v = co < cout ? w[((size_t)t * cin + 32 * kb + k) * cout + co] : 0.0f;
                    uint16_t pl[2];
                    f16x2_split(v, S, pl[0], pl[1]);
                    for (int p = 0; p < 2; ++p) dst[((((size_t)t * nkb + kb) * 2 + p) * coutp + co) * 32 + k] = pl[p];
                }
    return total;
}

// inv_sw != 0: wplanes = pack_generic_f16x2's planes, the contraction as a two-term fp16 split (same plans, a smaller strip)
hipError_t launch_conv_generic_x3(const float* in, const uint16_t* wplanes, const float* ep, float* out, int64_t n, int H, int W, int cin,
                                  int cout, int ups, int epi, hipStream_t stream, float inv_sw)
{
    if (n <= 0) return hipSuccess;
    int SR = 0, nmg = 0, nslw = 0, tpw = 0;
    size_t lds = 0;
    const bool ok = ups ? x3f_plan(H, W, cin, cout, &SR, &nmg, &nslw, &tpw, &lds) : x3_plan(H, W, cin, cout, &SR, &nmg, &nslw, &tpw, &lds);
    if (!ok || (ups && !(epi == GEN_EPI_BN || epi == GEN_EPI_RELU))) return hipErrorInvalidValue;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    Gen3Args a;
    a.in = in; a.w = wplanes; a.ep = ep; a.out = out; a.n = n; a.H = H; a.W = W; a.cin = cin; a.cout = cout; a.epi = epi;
    a.SR = SR; a.nmg = nmg; a.nslw = nslw; a.inv_sw = inv_sw;
    const bool h2 = inv_sw != 0.0f;
    if (h2) {   // two planes per pixel + the word that collects the strip's maximum
        const int Ws = ups ? W / 2 : W, R = ups ? SR / 2 + 2 : SR + 2;
        lds = (size_t)R * (Ws + 2) * (4 * cin + 32) + 16;
    }
    const long items = (long)n * (H / SR) * ((cout + nslw * 16 - 1) / (nslw * 16));
    const int per_cu = (tpw <= 8 && lds <= 76 * 1024) ? 2 : 1;
    const unsigned grid = (unsigned)(items < (long)cus * per_cu ? items : (long)cus * per_cu);
    hipError_t e = hipSuccess;
#define X3_LAUNCH(C, T, F)                                                                                                          \
    do {                                                                                                                            \
        if (h2) {                                                                                                                   \
            e = hipFuncSetAttribute((const void*)conv_generic_x3_kernel<C, T, F, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e == hipSuccess) hipLaunchKernelGGL((conv_generic_x3_kernel<C, T, F, true>), dim3(grid), dim3(512), lds, stream, a); \
        } else {                                                                                                                    \
            e = hipFuncSetAttribute((const void*)conv_generic_x3_kernel<C, T, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e == hipSuccess) hipLaunchKernelGGL((conv_generic_x3_kernel<C, T, F>), dim3(grid), dim3(512), lds, stream, a);       \
        }                                                                                                                           \
    } while (0)
#define X3_TPW(C, F)                                       \
    switch (tpw) {                                         \
        case 4: X3_LAUNCH(C, 4, F); break;                 \
        case 8: X3_LAUNCH(C, 8, F); break;                 \
        default: return hipErrorInvalidValue;              \
    }
#define X3_CIN(F)                                          \
    switch (cin) {                                         \
        case 32: X3_TPW(32, F); break;                     \
        case 64: X3_TPW(64, F); break;                     \
        case 128: X3_TPW(128, F); break;                   \
        default: return hipErrorInvalidValue;              \
    }
    if (ups) { X3_CIN(true); } else { X3_CIN(false); }
#undef X3_CIN
#undef X3_TPW
#undef X3_LAUNCH
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

// the last conv: plan (stored rows per strip, tiles, T pitch, LDS bytes) or false
static bool last_x3_plan(int H, int W, int cin, int* SRS, int* NT, int* P, size_t* lds)
{
    if (!(cin == 32 || cin == 64) || H % 2 || W % 2) return false;
    const int Hs = H / 2, Ws = W / 2, psb = 6 * cin + 32;
    for (int srs = 8; srs >= 1; srs /= 2) {
        if (Hs % srs) continue;
        const int npx = (srs + 2) * (Ws + 2), nt = (npx + 15) / 16, p = ((nt * 16 + 7) & ~7) + 4;
        const size_t bytes = (size_t)nt * 16 * psb + (size_t)16 * p * sizeof(float);
        if (bytes > 150 * 1024) continue;
        *SRS = srs; *NT = nt; *P = p; *lds = bytes;
        return true;
    }
    return false;
}

int conv_last_x3_takes(int H, int W, int cin)
{
    int SRS, NT, P;
    size_t lds;
    return last_x3_plan(H, W, cin, &SRS, &NT, &P, &lds) ? 1 : 0;
}

// weff: pack_generic_folded(cin, 1, ...) = W_eff[n = phase 4 + tap][c]; dst: [block][plane][lane = 16 kq + n][8]
size_t pack_last_bf16x3(int cin, const float* weff, uint16_t* dst)
{
    const int nkb = cin / 32;
    const size_t total = (size_t)nkb * 3 * 64 * 8;
    if (!dst) return total;
    for (int kb = 0; kb < nkb; ++kb)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int nn = l & 15, kq = l >> 4;
                const float v = weff[(size_t)nn * cin + 32 * kb + 8 * kq + j];
                const uint16_t w1 = bf16_rne(v);
                const float r1 = v - bf16_val(w1);
                const uint16_t w2 = bf16_rne(r1);
                const uint16_t pl[3] = {w1, w2, bf16_rne(r1 - bf16_val(w2))};
                for (int p = 0; p < 3; ++p) dst[(((size_t)kb * 3 + p) * 64 + l) * 8 + j] = pl[p];
            }
    return total;
}

hipError_t launch_conv_last_x3(const float* in, const uint16_t* wplanes, const float* ep, float* out, int64_t n, int H, int W, int cin,
                               hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    int SRS = 0, NT = 0, P = 0;
    size_t lds = 0;
    if (!last_x3_plan(H, W, cin, &SRS, &NT, &P, &lds)) return hipErrorInvalidValue;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long items = (long)n * ((H / 2) / SRS);
    const int per_cu = lds <= 76 * 1024 ? 2 : 1;
    const unsigned grid = (unsigned)(items < (long)cus * per_cu ? items : (long)cus * per_cu);
    hipError_t e;
    if (cin == 32) {
        if ((e = hipFuncSetAttribute((const void*)conv_last_x3_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(conv_last_x3_kernel<32>, dim3(grid), dim3(512), lds, stream, in, (const bf16x8*)wplanes, ep, out, (long)n, H, W, SRS, NT, P);
    } else {
        if ((e = hipFuncSetAttribute((const void*)conv_last_x3_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(conv_last_x3_kernel<64>, dim3(grid), dim3(512), lds, stream, in, (const bf16x8*)wplanes, ep, out, (long)n, H, W, SRS, NT, P);
    }
    return hipGetLastError();
}

hipError_t launch_pack_generic_bf16x3(const float* w, int ntaps, int cin, int cout, uint16_t* dst, hipStream_t stream)
{
    const long total = (long)ntaps * (cin / 32) * ((cout + 15) & ~15) * 32;
    if (total <= 0) return hipSuccess;
    long grid = (total + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(pack_generic_bf16x3_kernel, dim3((unsigned)grid), dim3(256), 0, stream, w, ntaps, cin, cout, dst);
    return hipGetLastError();
}

}  // namespace cs
