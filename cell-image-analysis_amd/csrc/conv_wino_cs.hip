// conv_wino_cs.hip -- Winograd F(2x2, 3x3) for the two encoder convs that sit in front of a 2x2
// max-pool, with the 4x4 transform domain split by COLUMN across the four waves of a workgroup:
//   conv2: Conv2D 32->64 on the 32x32 grid  (CAE_improved_modeltrain.py:195-197)
//   conv3: Conv2D 64->32 on the 16x16 grid  (:199-201)  -> the 8x8x32 `encoded` tensor
//
//   V = B^T d B       input transform of each 4x4 patch d (stride 2)
//   M_xi = V_xi U_xi   16 products [16 tiles x cin] x [cin x cout], xi = 4 r + c          (fp32 MFMA)
//   Y = A^T M A       2x2 outputs per tile = the max-pool window -> one value per tile and channel
//
// Wave c owns column c of the transform domain: it derives V[:, c] for ALL input channels (one
// quarter of the transform, no redundancy between waves), keeps U[4r + c] for all output channels
// in 128 VGPRs (4 r x cout/16 slices x cin/4 K steps), runs the 4 x (cout/16) x (cin/4) = 128 MFMAs
// of a 16-tile group and folds rows (s = A^T M, in registers).  The column fold Y = s A crosses
// waves: every wave leaves its s in LDS (32 / 16 KB), one barrier, then wave w finishes its share of
// (slice, tile) pairs: bias -> relu -> BN -> 2x2 max -> store.  Compared with the previous kernel
// (every wave transformed every column for its own output slice) the LDS patch reads and transform
// VALU work drop 4x; the price is the exchange and a second barrier per group.
#include "common.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace cs {

namespace {

template <int H_, int W_, int CIN_, int COUT_>
struct WinoCfg {
    static constexpr int H = H_, W = W_, CIN = CIN_, COUT = COUT_;
    static constexpr int TW = W / 2;                     // tiles per tile row (16 | 8)
    static constexpr int TR = 16 / TW;                   // tile rows per 16-tile group (1 | 2)
    static constexpr int SR = 2 * TR;                    // conv rows per group
    static constexpr int R = SR + 2, WP = W + 2;         // staged rows / cols incl. halo
    // padded pixel stride: tiles step 2 pixels, so an ODD number of 16-B slots per pixel spreads the
    // 16 tiles x 4 channel quads of a ds_read_b128 over distinct slots
    static constexpr int PS = CIN + 4;
    static constexpr int STRIP = R * WP * PS * 4;        // bytes, double buffered
    static constexpr int NQ = CIN / 16, NS = COUT / 16, KS = CIN / 4;
    static constexpr int NB = 4 * NS * KS;               // 128 for both layers
    static constexpr int XCH = 4 * NS * 2 * 64 * 16;     // bytes: [wave c][slice][s0|s1][lane] f32x4
    static constexpr int LDS = 2 * STRIP + XCH;
    static constexpr int NGRP = H / SR;                  // groups per cell
    static constexpr int C4 = CIN / 4, TOT = R * WP * C4;
    // staging: a thread owns one 16-byte element of the strip's INTERIOR columns per staged row (the halo columns are
    // zeroed once at kernel start and never rewritten): W * C4 = 256 elements per row for both layers, no divisions
    static constexpr int NLD = R;                        // per-thread 16-B loads per strip (one per row)
    static_assert(W * C4 == 256, "one interior element per thread and row");
    static constexpr bool QOUTER = NQ > NS;              // which of V / accumulators is kept whole
    static constexpr int FR = NS;                        // tile registers a wave finishes (4*NS units / 4 waves)
    static_assert(NB == 128 && LDS <= 80 * 1024 && TW * TR == 16 && (NS == 2 || NS == 4), "layer does not fit this design");
};
using WinoL2 = WinoCfg<32, 32, 32, 64>;
using WinoL3 = WinoCfg<16, 16, 64, 32>;

// Strip staging.  The load is UNCONDITIONAL (rows outside the image read a clamped in-range row) and the zero
// padding is applied when the value is written to LDS: a load under a divergent branch makes the compiler wait for
// it (vmcnt(0)) right where it is issued, which serialises the prefetch.  fp32 MFMA and VALU instructions never
// execute together on a SIMD (PMC: SQ_VALU_MFMA_COEXEC_CYCLES = 0), so the index arithmetic is per-thread constants
// (goff / loff) + one row test.
template <class C>
__device__ __forceinline__ f32x4 wn_load(const float* __restrict__ cellp, int y0, int r, int goff)
{
    int sy = y0 - 1 + r;
    sy = sy < 0 ? 0 : (sy > C::H - 1 ? C::H - 1 : sy);
    return *(const f32x4*)(cellp + sy * (C::W * C::CIN) + goff);
}
template <class C>
__device__ __forceinline__ void wn_store(float* strip, int y0, int r, int loff, f32x4 v)
{
    const int sy = y0 - 1 + r;
    if (sy < 0 || sy >= C::H) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    *(f32x4*)(strip + r * (C::WP * C::PS) + loff) = v;
}

// DIAG: diagnostic build that stamps s_memtime at the phase boundaries of every group and sums the
// differences per wave: [0] prefetch issue + transform + MFMA + row fold + exchange write, [1] next-strip
// LDS writes, [2] wait at barrier A, [3] column fold + epilogue + stores, [4] wait at barrier B, [5] the
// part of [0] spent issuing the next strip's global loads.  Never used for results or timing.
__device__ __forceinline__ unsigned long long wcs_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

template <class C, bool DIAG>
__global__ __launch_bounds__(256, 2) void conv_wino_cs_kernel(const float* __restrict__ in, const float* __restrict__ ufrag,
                                                             const float* __restrict__ ep /* [3][cout] */,
                                                             float* __restrict__ out, long n_cells,
                                                             unsigned long long* __restrict__ diag)
{
    unsigned long long dg[6] = {0, 0, 0, 0, 0, 0}, dt = 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const xch = (float*)(smem + 2 * C::STRIP);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave = transform-domain column
    const int li = lane & 15, kq = lane >> 4;

    float B[C::NB];
#pragma unroll
    for (int s = 0; s < C::NB; ++s) B[s] = ufrag[((size_t)wc * C::NB + s) * 64 + lane];

    // V[:, c] = B^T (d B[:, c]);  d B[:, c] = d[:, ca] + sg * d[:, cb]
    const int ca = wc == 0 ? 0 : (wc == 2 ? 2 : 1);
    const int cb = wc == 2 ? 1 : (wc == 3 ? 3 : 2);
    const float sg = wc == 1 ? 1.0f : -1.0f;
    // patch of tile li inside the strip, this lane's channel quad
    const int trow = li / C::TW, tcol = li % C::TW;
    const int poff = ((2 * trow) * C::WP + 2 * tcol) * C::PS + 4 * kq;

    // the (slice, tile register) share this wave finishes
    const int fs = wc % C::NS, rbase = (wc / C::NS) * C::FR;
    const int co = fs * 16 + li;
    float bias = ep[co], bns = ep[C::COUT + co], bnt = ep[2 * C::COUT + co];
    // Touch the epilogue constants here: their first use would otherwise be inside the loop, where the
    // compiler's wait for them (vmcnt(0), every iteration) would also wait for the strip prefetch.
    asm volatile("" : "+v"(bias), "+v"(bns), "+v"(bnt));
#pragma unroll
    for (int s = 0; s < C::NB; ++s) asm volatile("" : "+v"(B[s]));

    const long total = n_cells * C::NGRP;
    const long first = blockIdx.x;
    if (first >= total) return;
    const int spx = tid / C::C4, sc4 = tid % C::C4;                 // this thread's interior element of a staged row
    const int goff = spx * C::CIN + sc4 * 4, loff = (spx + 1) * C::PS + sc4 * 4;
    auto cell_ptr = [&](long cell) { return in + (size_t)cell * C::H * C::W * C::CIN; };
    for (int i = tid; i < 2 * C::STRIP / 16; i += 256) ((f32x4*)smem)[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    __syncthreads();
#pragma unroll
    for (int r = 0; r < C::R; ++r)
        wn_store<C>((float*)smem, (int)(first % C::NGRP) * C::SR, r, loff, wn_load<C>(cell_ptr(first / C::NGRP), (int)(first % C::NGRP) * C::SR, r, goff));
    __syncthreads();

    f32x4 stg[C::NLD];

    int buf = 0;
    for (long item = first; item < total; item += gridDim.x) {
        const long cell = item / C::NGRP;
        const int grp = (int)(item % C::NGRP);
        const long nitem = item + gridDim.x;
        const bool has_next = nitem < total;
        const float* strip = (const float*)(smem + buf * C::STRIP);
        float* nstrip = (float*)(smem + (buf ^ 1) * C::STRIP);

        if constexpr (DIAG) dt = wcs_stamp();
        if (has_next) {
#pragma unroll
            for (int j = 0; j < C::NLD; ++j) stg[j] = wn_load<C>(cell_ptr(nitem / C::NGRP), (int)(nitem % C::NGRP) * C::SR, j, goff);
        }
        if constexpr (DIAG) { const unsigned long long t = wcs_stamp(); dg[5] += t - dt; }
        const float* da_p = strip + poff + ca * C::PS;
        const float* db_p = strip + poff + cb * C::PS;
        auto transform = [&](int q, f32x4 v[4]) {
            f32x4 w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 da = *(const f32x4*)(da_p + i * C::WP * C::PS + 16 * q);
                const f32x4 db = *(const f32x4*)(db_p + i * C::WP * C::PS + 16 * q);
                w[i] = da + sg * db;
            }
            v[0] = w[0] - w[2]; v[1] = w[1] + w[2]; v[2] = w[2] - w[1]; v[3] = w[1] - w[3];
        };
        // row fold s = A^T M of one slice, left in LDS for the column fold
        auto fold_store = [&](int s, const f32x4 acc[4]) {
            f32x4 s0, s1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s0[r] = (acc[0][r] + acc[1][r]) + acc[2][r];
                s1[r] = (acc[1][r] - acc[2][r]) - acc[3][r];
            }
            float* x = xch + ((size_t)((wc * C::NS + s) * 2) * 64 + lane) * 4;
            *(f32x4*)x = s0;
            *(f32x4*)(x + 64 * 4) = s1;
        };

        if constexpr (!C::QOUTER) {
            // V for every channel group first (NQ x 16 VGPRs), then one output slice at a time
            f32x4 V[C::NQ][4];
#pragma unroll
            for (int q = 0; q < C::NQ; ++q) transform(q, V[q]);
#pragma unroll
            for (int s = 0; s < C::NS; ++s) {
                f32x4 acc[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int q = 0; q < C::NQ; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[q][r][j], B[(r * C::NS + s) * C::KS + 4 * q + j], acc[r], 0, 0, 0);
                fold_store(s, acc);
            }
        } else {
            // one channel group at a time (16 VGPRs of V), accumulators of every slice live
            f32x4 acc[C::NS][4];
#pragma unroll
            for (int s = 0; s < C::NS; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[s][r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < C::NQ; ++q) {
                f32x4 v[4];
                transform(q, v);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int s = 0; s < C::NS; ++s)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[s][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[r][j], B[(r * C::NS + s) * C::KS + 4 * q + j], acc[s][r], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < C::NS; ++s) fold_store(s, acc[s]);
        }
        if constexpr (DIAG) { const unsigned long long t = wcs_stamp(); dg[0] += t - dt; dt = t; }
        if (has_next) {
#pragma unroll
            for (int j = 0; j < C::NLD; ++j) wn_store<C>(nstrip, (int)(nitem % C::NGRP) * C::SR, j, loff, stg[j]);
        }
        if constexpr (DIAG) { const unsigned long long t = wcs_stamp(); dg[1] += t - dt; dt = t; }
        __syncthreads();   // s of all four columns in LDS; this strip fully read; next strip complete
        if constexpr (DIAG) { const unsigned long long t = wcs_stamp(); dg[2] += t - dt; dt = t; }
        // column fold Y = s A for this wave's (slice, tiles), then bias -> relu -> BN -> 2x2 max
        f32x4 t0[4], t1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* x = xch + ((size_t)((c * C::NS + fs) * 2) * 64 + lane) * 4;
            t0[c] = *(const f32x4*)x;
            t1[c] = *(const f32x4*)(x + 64 * 4);
        }
        const f32x4 y00 = (t0[0] + t0[1]) + t0[2], y01 = (t0[1] - t0[2]) - t0[3];
        const f32x4 y10 = (t1[0] + t1[1]) + t1[2], y11 = (t1[1] - t1[2]) - t1[3];
        auto post = [&](float v) { v += bias; v = fmaxf(v, 0.0f); return fmaf(v, bns, bnt); };
#pragma unroll
        for (int rr = 0; rr < C::FR; ++rr) {
            float a, b, c2, d;
            if constexpr (C::FR == 4) {
                a = y00[rr]; b = y01[rr]; c2 = y10[rr]; d = y11[rr];
            } else {   // FR == 2: registers rbase + rr with rbase in {0, 2} (wave-uniform)
                a = rbase ? y00[2 + rr] : y00[rr]; b = rbase ? y01[2 + rr] : y01[rr];
                c2 = rbase ? y10[2 + rr] : y10[rr]; d = rbase ? y11[2 + rr] : y11[rr];
            }
            // the epilogue map is monotone (direction = sign of the BN scale): pool the raw values, map once
            const float mx = fmaxf(fmaxf(a, b), fmaxf(c2, d)), mn = fminf(fminf(a, b), fminf(c2, d));
            const float res = post(bns >= 0.0f ? mx : mn);
            const int t = 4 * kq + rbase + rr;                        // tile of the group (MFMA D row)
            const int ty = grp * C::TR + t / C::TW, tx = t % C::TW;
            out[(((size_t)cell * (C::H / 2) + ty) * (C::W / 2) + tx) * C::COUT + co] = res;
        }
        if constexpr (DIAG) { const unsigned long long t = wcs_stamp(); dg[3] += t - dt; dt = t; }
        __syncthreads();   // exchange area free again
        if constexpr (DIAG) { const unsigned long long t = wcs_stamp(); dg[4] += t - dt; dt = t; }
        buf ^= 1;
    }
    if constexpr (DIAG) {
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) diag[((size_t)blockIdx.x * 4 + wc) * 6 + k] = dg[k];
        }
    }
}

// ---- conv3 with the Winograd-domain contraction on the 16-bit matrix pipe: the staged-strip layout -----------------------------
// Same decomposition as conv_wino_cs_kernel<WinoL3> (wave c = column c of the 4x4 transform domain, V[:, c] derived in
// registers, row fold in registers, column fold through LDS).  A lane (tile li, kq) owns channels {4 kq .. +3} and
// {16 + 4 kq .. +3} of a 32-channel block: two ds_read_b128 64 B apart, which with a 24-float pad per staged row puts every
// lane group on 16 distinct 16-byte slots (enumerated).
struct W3X {
    using C = WinoL3;
    static constexpr int ROWP = C::WP * C::PS + 24;                 // floats per staged row
    static constexpr int STRIP = C::R * ROWP * 4;                   // 29,952 B
    static constexpr int NKB = C::CIN / 32;
};

// ---- conv3 with the Winograd-domain contraction as a TWO-term fp16 split (three products) ---------------------------------
// conv_wino_up.hip (conv67_h2_kernel) has the algebra and the hardware facts; tests/study_split_fp16.py measures conv3 in this form
// at 2.6e-7 of the feature range (the fp32 MFMA chain: 5.2e-7).  768 matrix
// instructions per cell, a 3-instruction split per transformed value, and U as two fp16 planes in 128 VGPRs -- which lets TWO
// workgroups share a CU (two waves per SIMD: one workgroup's barriers and split bursts under the other's MFMAs).
//   scale   |V| <= 4 max|p2| (B^T of F(2,3) has absolute row sums 2): S puts 4 max|p2| of the STRIP into [2^14, 2^15); the
//           maximum is taken where the next strip is staged (registers -> DPP row max -> LDS atomic max, read behind the
//           barrier that is there anyway).  The strip itself stays fp32 in LDS (the transform runs in fp32); V S is split in
//           registers; 1 / (S S_w) is applied with the bias in the epilogue's fma.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct W3H {
    using C = WinoL3;
    static constexpr int OFF_MAX = 2 * W3X::STRIP + C::XCH;        // two words: strip maxima, alternating
    static constexpr int LDS = OFF_MAX + 16;
    static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
};

__device__ __forceinline__ unsigned int w3h_rowmax(unsigned int m)
{
    unsigned int o;
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, true);  m = m > o ? m : o;     // quad_perm [1,0,3,2]
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, true);  m = m > o ? m : o;     // quad_perm [2,3,0,1]
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0x124, 0xF, 0xF, true); m = m > o ? m : o;     // row_ror:4
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xF, 0xF, true); m = m > o ? m : o;     // row_ror:8
    return m;
}
// S = the power of two that puts 4 x (a maximum with float bits mbits) into [2^14, 2^15), and 1 / S (exponents clamped so that
// both stay normal floats)
__device__ __forceinline__ void w3h_scale(unsigned int mbits, float& S, float& invS)
{
    int E = (int)((mbits >> 23) & 0xffu) + 2;
    E = E < 40 ? 40 : (E > 254 ? 254 : E);
    S = __builtin_bit_cast(float, (unsigned int)(268 - E) << 23);
    invS = __builtin_bit_cast(float, (unsigned int)(E - 14) << 23);
}
// Eight scaled values -> their hi and lo fragments as ONE block of 12 instructions (v_cvt_pk_f16_f32 per pair, then the residuals as
// v_fma_mix{lo,hi}_f16: exact in fp32, one rounding); the convert / convert back / subtract / convert form compiled to ~ 20 and the
// kernel ran 2.3 % slower (24.7 vs 24.1 ms per 1 M cells; per-VALUE asm blocks had measured slower in round 3: common.hpp).  -0
// residuals come out +0.  The results feed MFMAs directly, which the hazard recogniser cannot see through the asm: the block ends with the two wait
// states a VALU write -> MFMA read needs.
__device__ __forceinline__ void w3h_split8(const f32x4& lo4, const f32x4& hi4, float S, f16x8& ah, f16x8& al)
{
    const f32x4 a = lo4 * S, b = hi4 * S;
    const float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    unsigned int h0, h1, h2, h3, l0, l1, l2, l3;
    asm("v_cvt_pk_f16_f32 %0, %8, %9\n\t"
        "v_cvt_pk_f16_f32 %1, %10, %11\n\t"
        "v_cvt_pk_f16_f32 %2, %12, %13\n\t"
        "v_cvt_pk_f16_f32 %3, %14, %15\n\t"
        "v_fma_mixlo_f16 %4, %8, 1.0, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %5, %10, 1.0, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %6, %12, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %7, %14, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %4, %9, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %5, %11, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %6, %13, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %7, %15, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 1"
        : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
    ah = __builtin_bit_cast(f16x8, u32x4{h0, h1, h2, h3});
    al = __builtin_bit_cast(f16x8, u32x4{l0, l1, l2, l3});
}
__global__ __launch_bounds__(256, 2) void conv3_wino_h2_kernel(const float* __restrict__ in, const f16x8* __restrict__ ufrag,
                                                              const float* __restrict__ ep /* [3][32] */, float* __restrict__ out,
                                                              long n_cells, float inv_sw)
{
    using C = WinoL3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const xch = (float*)(smem + 2 * W3X::STRIP);
    unsigned int* const mxw = (unsigned int*)(smem + W3H::OFF_MAX);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave = transform-domain column
    const int li = lane & 15, kq = lane >> 4;

    f16x8 B[4][C::NS][W3X::NKB][2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < C::NS; ++s)
#pragma unroll
            for (int kb = 0; kb < W3X::NKB; ++kb)
#pragma unroll
                for (int p = 0; p < 2; ++p) B[r][s][kb][p] = ufrag[(((((size_t)wc * 4 + r) * C::NS + s) * W3X::NKB + kb) * 2 + p) * 64 + lane];

    const int ca = wc == 0 ? 0 : (wc == 2 ? 2 : 1);
    const int cb = wc == 2 ? 1 : (wc == 3 ? 3 : 2);
    const float sg = wc == 1 ? 1.0f : -1.0f;
    const int trow = li / C::TW, tcol = li % C::TW;
    const int poff = (2 * trow) * W3X::ROWP + (2 * tcol) * C::PS + 4 * kq;

    const int fs = wc % C::NS, rbase = (wc / C::NS) * C::FR;
    const int co = fs * 16 + li;
    const float bias = ep[co], bns = ep[C::COUT + co], bnt = ep[2 * C::COUT + co];

    const long total = n_cells * C::NGRP;
    const long first = blockIdx.x;
    if (first >= total) return;
    const int spx = tid / C::C4, sc4 = tid % C::C4;
    const int goff = spx * C::CIN + sc4 * 4, loff = (spx + 1) * C::PS + sc4 * 4;
    auto cell_ptr = [&](long cell) { return in + (size_t)cell * C::H * C::W * C::CIN; };
    // the staged value of row r (zero outside the image) and its share of the strip's max|.|
    auto st_prep = [&](int y0, int r, f32x4 v, unsigned int& mx) {
        const int sy = y0 - 1 + r;
        if (sy < 0 || sy >= C::H) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const float a = v[0], b = v[1], c = v[2], d = v[3];         // scalars first (see conv45_bf16x3.hip, h2_absmax4)
        const unsigned int u = __builtin_bit_cast(unsigned int, fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d))));
        mx = mx > u ? mx : u;
        return v;
    };
    for (int i = tid; i < W3H::LDS / 16; i += 256) ((f32x4*)smem)[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    __syncthreads();
    float S, invS;
    {
        unsigned int mx = 0;
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            const f32x4 v = st_prep((int)(first % C::NGRP) * C::SR, r, wn_load<C>(cell_ptr(first / C::NGRP), (int)(first % C::NGRP) * C::SR, r, goff), mx);
            *(f32x4*)((float*)smem + r * W3X::ROWP + loff) = v;
        }
        mx = w3h_rowmax(mx);
        if (li == 0) atomicMax(&mxw[0], mx);
        __syncthreads();
        w3h_scale(mxw[0], S, invS);
        __syncthreads();
        if (tid == 0) mxw[0] = 0;
    }

    f32x4 stg[C::NLD];
    int buf = 0;
    int k = 0;                                                     // strips staged so far: word (k + 1) & 1 collects the next maximum
    for (long item = first; item < total; item += gridDim.x, ++k) {
        const long cell = item / C::NGRP;
        const int grp = (int)(item % C::NGRP);
        const long nitem = item + gridDim.x;
        const bool has_next = nitem < total;
        const float* strip = (const float*)(smem + buf * W3X::STRIP);
        float* nstrip = (float*)(smem + (buf ^ 1) * W3X::STRIP);
        unsigned int* const mword = mxw + ((k + 1) & 1);
        const float unscale = invS * inv_sw;
        if (has_next) {
#pragma unroll
            for (int j = 0; j < C::NLD; ++j) stg[j] = wn_load<C>(cell_ptr(nitem / C::NGRP), (int)(nitem % C::NGRP) * C::SR, j, goff);
        }
        const float* da_p = strip + poff + ca * C::PS;
        const float* db_p = strip + poff + cb * C::PS;
        f32x4 acc[C::NS][4];
#pragma unroll
        for (int s = 0; s < C::NS; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[s][r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kb = 0; kb < W3X::NKB; ++kb) {
            // V[:, c] of the lane's 8 channels: halves h = channels 32 kb + 16 h + 4 kq .. +3
            f32x4 v[4][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 da = *(const f32x4*)(da_p + i * W3X::ROWP + 32 * kb + 16 * h);
                    const f32x4 db = *(const f32x4*)(db_p + i * W3X::ROWP + 32 * kb + 16 * h);
                    w[i] = da + sg * db;
                }
                v[0][h] = w[0] - w[2]; v[1][h] = w[1] + w[2]; v[2][h] = w[2] - w[1]; v[3][h] = w[1] - w[3];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f16x8 ah, al;
                w3h_split8(v[r][0], v[r][1], S, ah, al);
#pragma unroll
                for (int s = 0; s < C::NS; ++s) {
                    f32x4 d = acc[s][r];
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[r][s][kb][1], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, B[r][s][kb][0], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[r][s][kb][0], d, 0, 0, 0);
                    acc[s][r] = d;
                }
            }
        }
        // row fold s = A^T M of each slice, left in LDS for the column fold
#pragma unroll
        for (int s = 0; s < C::NS; ++s) {
            f32x4 s0, s1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s0[r] = (acc[s][0][r] + acc[s][1][r]) + acc[s][2][r];
                s1[r] = (acc[s][1][r] - acc[s][2][r]) - acc[s][3][r];
            }
            float* x = xch + ((size_t)((wc * C::NS + s) * 2) * 64 + lane) * 4;
            *(f32x4*)x = s0;
            *(f32x4*)(x + 64 * 4) = s1;
        }
        if (has_next) {
            unsigned int mx = 0;
#pragma unroll
            for (int j = 0; j < C::NLD; ++j) {
                stg[j] = st_prep((int)(nitem % C::NGRP) * C::SR, j, stg[j], mx);
                *(f32x4*)(nstrip + j * W3X::ROWP + loff) = stg[j];
            }
            mx = w3h_rowmax(mx);
            if (li == 0) atomicMax(mword, mx);
        }
        __syncthreads();   // s of all four columns in LDS; this strip fully read; next strip and its maximum complete
        if (has_next) w3h_scale(*mword, S, invS);
        f32x4 t0[4], t1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* x = xch + ((size_t)((c * C::NS + fs) * 2) * 64 + lane) * 4;
            t0[c] = *(const f32x4*)x;
            t1[c] = *(const f32x4*)(x + 64 * 4);
        }
        const f32x4 y00 = (t0[0] + t0[1]) + t0[2], y01 = (t0[1] - t0[2]) - t0[3];
        const f32x4 y10 = (t1[0] + t1[1]) + t1[2], y11 = (t1[1] - t1[2]) - t1[3];
        auto post = [&](float v) { v = fmaxf(fmaf(v, unscale, bias), 0.0f); return fmaf(v, bns, bnt); };
#pragma unroll
        for (int rr = 0; rr < C::FR; ++rr) {
            const float a = rbase ? y00[2 + rr] : y00[rr], b = rbase ? y01[2 + rr] : y01[rr];
            const float c2 = rbase ? y10[2 + rr] : y10[rr], d = rbase ? y11[2 + rr] : y11[rr];
            const float mx = fmaxf(fmaxf(a, b), fmaxf(c2, d)), mn = fminf(fminf(a, b), fminf(c2, d));
            const float res = post(bns >= 0.0f ? mx : mn);
            const int t = 4 * kq + rbase + rr;
            const int ty = grp * C::TR + t / C::TW, tx = t % C::TW;
            out[(((size_t)cell * (C::H / 2) + ty) * (C::W / 2) + tx) * C::COUT + co] = res;
        }
        __syncthreads();   // exchange area free again; every thread has read this strip's maximum
        if (tid == 0) *mword = 0;
        buf ^= 1;
    }
}

// ---- conv2 with a ring of staged rows --------------------------------------------------------------
// Consecutive 16-tile groups of a cell (one tile row = 2 conv rows each) share two of their four staged rows.
// The strip double buffer above reloads all four; here a workgroup walks whole cells and keeps the staged rows
// in an 8-slot ring (same 39 KB): each group loads only the two rows the next group adds (three + a zero row
// when the next group starts a new cell), one 16-byte load per thread and row.  Halves conv2's reads (255 ->
// ~130 KB per cell) and the load-issue / LDS-write cycles of the prefetch.  Sequence rows: q = 0..33 per cell,
// q = 0 and q = 33 are the zero halo rows; running position P = cell_iter * 34 + q lives in slot P & 7; the
// halo COLUMNS of every slot are zeroed once and never written again.
constexpr int RG_SLOTS = 8;

__global__ __launch_bounds__(256, 2) void conv2_wino_ring_kernel(const float* __restrict__ in, const float* __restrict__ ufrag,
                                                                const float* __restrict__ ep, float* __restrict__ out, long n_cells)
{
    using C = WinoL2;
    constexpr int ROWF = C::WP * C::PS;                              // floats per ring slot
    constexpr int RING = RG_SLOTS * ROWF * 4;                        // bytes
    static_assert(RING == 2 * C::STRIP && C::TR == 1 && C::W * C::C4 == 256, "ring geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const ring = (float*)smem;
    float* const xch = (float*)(smem + RING);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wc = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    float B[C::NB];
#pragma unroll
    for (int s = 0; s < C::NB; ++s) B[s] = ufrag[((size_t)wc * C::NB + s) * 64 + lane];
    const int ca = wc == 0 ? 0 : (wc == 2 ? 2 : 1);
    const int cb = wc == 2 ? 1 : (wc == 3 ? 3 : 2);
    const float sg = wc == 1 ? 1.0f : -1.0f;
    const int pcol = (2 * li) * C::PS + 4 * kq;                      // tile li's first patch column, this lane's channel quad
    const int fs = wc, co = fs * 16 + li;                            // NS == 4: wave w finishes slice w
    float bias = ep[co], bns = ep[C::COUT + co], bnt = ep[2 * C::COUT + co];
    asm volatile("" : "+v"(bias), "+v"(bns), "+v"(bnt));
#pragma unroll
    for (int s = 0; s < C::NB; ++s) asm volatile("" : "+v"(B[s]));

    // this thread's element of a staged row: pixel tid >> 3, channel quad tid & 7
    const int spx = tid >> 3, sc4 = tid & 7;
    const int soff = (spx + 1) * C::PS + sc4 * 4;                    // +1: halo column
    const int goff = spx * C::CIN + sc4 * 4;
    auto row_ptr = [&](long cell, int y) { return in + ((size_t)cell * C::H + y) * C::W * C::CIN + goff; };

    const long my_cells = (n_cells - blockIdx.x + gridDim.x - 1) / gridDim.x;   // cells blockIdx, +grid, ...
    if (my_cells <= 0) return;
    for (int idx = tid; idx < RG_SLOTS * ROWF / 4; idx += 256) ((f32x4*)ring)[idx] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    __syncthreads();
    // first cell: q = 1..3 (rows 0..2) into slots 1..3; q = 0 stays zero
#pragma unroll
    for (int q = 1; q <= 3; ++q) *(f32x4*)(ring + q * ROWF + soff) = *(const f32x4*)row_ptr(blockIdx.x, q - 1);
    __syncthreads();

    // the first channel group's patch reads of a group are issued before barrier B of the previous group, so
    // their latency hides behind the epilogue instead of heading the transform
    f32x4 pa[4], pb[4];
    auto patch_q0 = [&](long Pn) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ro = (int)((Pn + i) & (RG_SLOTS - 1)) * ROWF;
            pa[i] = *(const f32x4*)(ring + pcol + ca * C::PS + ro);
            pb[i] = *(const f32x4*)(ring + pcol + cb * C::PS + ro);
        }
    };
    patch_q0(0);
    const long n_items = my_cells * C::NGRP;
    for (long it = 0; it < n_items; ++it) {
        const long ci = it / C::NGRP;
        const int grp = (int)(it % C::NGRP);
        const long cell = blockIdx.x + ci * gridDim.x;
        const long P = ci * 34 + 2 * grp;                            // running position of the group's first row
        const bool has_next = it + 1 < n_items;
        const bool new_cell = grp == C::NGRP - 1;                    // the next group starts the next cell

        // rows the next group adds: same cell q = 2 grp + 4, + 5 (y = 2 grp + 3, + 4; y = 32 is the zero row),
        // or the next cell's q = 0..3 (zero row, y = 0..2)
        f32x4 stg[3];
        if (has_next) {
            if (!new_cell) {
                const int y = 2 * grp + 3;
                stg[0] = *(const f32x4*)row_ptr(cell, y);
                stg[1] = *(const f32x4*)row_ptr(cell, y + 1 < C::H ? y + 1 : y);     // clamped; zeroed at the write if y + 1 == H
            } else {
                const long nc = cell + gridDim.x;
#pragma unroll
                for (int j = 0; j < 3; ++j) stg[j] = *(const f32x4*)row_ptr(nc, j);
            }
        }

        int roff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) roff[i] = (int)((P + i) & (RG_SLOTS - 1)) * ROWF;
        const float* da_p = ring + pcol + ca * C::PS;
        const float* db_p = ring + pcol + cb * C::PS;
        f32x4 V[C::NQ][4];
#pragma unroll
        for (int q = 0; q < C::NQ; ++q) {
            f32x4 w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (q == 0) {
                    w[i] = pa[i] + sg * pb[i];
                } else {
                    const f32x4 da = *(const f32x4*)(da_p + roff[i] + 16 * q);
                    const f32x4 db = *(const f32x4*)(db_p + roff[i] + 16 * q);
                    w[i] = da + sg * db;
                }
            }
            V[q][0] = w[0] - w[2]; V[q][1] = w[1] + w[2]; V[q][2] = w[2] - w[1]; V[q][3] = w[1] - w[3];
        }
#pragma unroll
        for (int s = 0; s < C::NS; ++s) {
            f32x4 acc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < C::NQ; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[q][r][j], B[(r * C::NS + s) * C::KS + 4 * q + j], acc[r], 0, 0, 0);
            f32x4 s0, s1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s0[r] = (acc[0][r] + acc[1][r]) + acc[2][r];
                s1[r] = (acc[1][r] - acc[2][r]) - acc[3][r];
            }
            float* x = xch + ((size_t)((wc * C::NS + s) * 2) * 64 + lane) * 4;
            *(f32x4*)x = s0;
            *(f32x4*)(x + 64 * 4) = s1;
        }
        if (has_next) {
            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
            if (!new_cell) {
                const int y = 2 * grp + 3;
                *(f32x4*)(ring + (int)((P + 4) & (RG_SLOTS - 1)) * ROWF + soff) = stg[0];
                *(f32x4*)(ring + (int)((P + 5) & (RG_SLOTS - 1)) * ROWF + soff) = (y + 1 < C::H) ? stg[1] : zero;
            } else {
                *(f32x4*)(ring + (int)((P + 4) & (RG_SLOTS - 1)) * ROWF + soff) = zero;        // q = 0 of the next cell
#pragma unroll
                for (int j = 0; j < 3; ++j) *(f32x4*)(ring + (int)((P + 5 + j) & (RG_SLOTS - 1)) * ROWF + soff) = stg[j];
            }
        }
        __syncthreads();   // s of all four columns in LDS; the next group's rows complete

        f32x4 t0[4], t1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* x = xch + ((size_t)((c * C::NS + fs) * 2) * 64 + lane) * 4;
            t0[c] = *(const f32x4*)x;
            t1[c] = *(const f32x4*)(x + 64 * 4);
        }
        const f32x4 y00 = (t0[0] + t0[1]) + t0[2], y01 = (t0[1] - t0[2]) - t0[3];
        const f32x4 y10 = (t1[0] + t1[1]) + t1[2], y11 = (t1[1] - t1[2]) - t1[3];
        if (has_next) patch_q0(new_cell ? P + 4 : P + 2);            // the next group's first row
        auto post = [&](float v) { v += bias; v = fmaxf(v, 0.0f); return fmaf(v, bns, bnt); };
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const float a = y00[rr], b = y01[rr], c2 = y10[rr], d = y11[rr];
            const float mx = fmaxf(fmaxf(a, b), fmaxf(c2, d)), mn = fminf(fminf(a, b), fminf(c2, d));
            const float res = post(bns >= 0.0f ? mx : mn);
            const int t = 4 * kq + rr;
            out[(((size_t)cell * (C::H / 2) + grp) * (C::W / 2) + t) * C::COUT + co] = res;
        }
        __syncthreads();   // exchange area free again
    }
}

template <class C>
size_t pack_frags(const float* hwio, float* dst)
{
    // U = G g G^T per (cin, cout), evaluated in double and rounded once; B fragments
    // [column c][(r * NS + s) * KS + 4 q + j][lane] = U[xi = 4 r + c][ci = 16 q + 4 kq + j][co = 16 s + li]
    const size_t total = (size_t)4 * C::NB * 64;
    if (!dst) return total;
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            for (int s = 0; s < C::NS; ++s)
                for (int kk = 0; kk < C::KS; ++kk)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int li = lane & 15, kq = lane >> 4, q = kk >> 2, j = kk & 3;
                        const int ci = 16 * q + 4 * kq + j, co = 16 * s + li;
                        double u = 0.0;   // U[r][c] = sum_{a,b} G[r][a] g[a][b] G[c][b]
                        for (int a = 0; a < 3; ++a)
                            for (int b = 0; b < 3; ++b)
                                u += G[r][a] * (double)hwio[((size_t)(a * 3 + b) * C::CIN + ci) * C::COUT + co] * G[c][b];
                        dst[((size_t)c * C::NB + (r * C::NS + s) * C::KS + kk) * 64 + lane] = (float)u;
                    }
    return total;
}

unsigned long long* g_diag[3] = {nullptr, nullptr, nullptr};
int g_diag_blocks[3] = {0, 0, 0};

template <class C>
hipError_t launch(int layer, const float* in, const float* ufrag, const float* ep, float* out, int64_t n_cells, hipStream_t stream)
{
    static int resident = 0;
    static const bool diag = getenv("CS_WINO_DIAG") != nullptr;
    if (!resident) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_wino_cs_kernel<C, false>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)conv_wino_cs_kernel<C, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return e;
        int dev = 0, cus = 0, per_cu = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv_wino_cs_kernel<C, false>, 256, C::LDS);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        resident = cus * per_cu;
        if (diag) {
            if ((e = hipMalloc(&g_diag[layer], (size_t)resident * 24 * sizeof(unsigned long long))) != hipSuccess) return e;
            g_diag_blocks[layer] = resident;
        }
    }
    const long total = (long)n_cells * C::NGRP;
    if (total <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(total < resident ? total : resident);
    if (diag)
        hipLaunchKernelGGL((conv_wino_cs_kernel<C, true>), dim3(grid), dim3(256), C::LDS, stream, in, ufrag, ep, out, (long)n_cells,
                           g_diag[layer]);
    else
        hipLaunchKernelGGL((conv_wino_cs_kernel<C, false>), dim3(grid), dim3(256), C::LDS, stream, in, ufrag, ep, out, (long)n_cells,
                           (unsigned long long*)nullptr);
    return hipGetLastError();
}

}  // namespace

size_t pack_wino_cs_fragments(int layer, const float* hwio, float* dst)
{
    return layer == 1 ? pack_frags<WinoL2>(hwio, dst) : pack_frags<WinoL3>(hwio, dst);
}

// conv3's U = G g G^T (double, rounded once to fp32) as two fp16 planes of S_w U in conv3_wino_h2_kernel's order:
// [wave c][r][slice][channel block][plane 2][lane][8]: element j of lane (li, kq) = plane of U[4 r + c][ci][16 slice + li] with
// ci = 32 block + (j < 4 ? 4 kq + j : 16 + 4 kq + j - 4); *inv_sw = 1 / S_w
size_t pack_wino3_h2(const float* hwio /* [3][3][64][32] */, uint16_t* dst, float* inv_sw)
{
    using C = WinoL3;
    const size_t total = (size_t)4 * 4 * C::NS * W3X::NKB * 2 * 64 * 8;
    if (!dst) return total;
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> U((size_t)16 * C::CIN * C::COUT);
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            for (int ci = 0; ci < C::CIN; ++ci)
                for (int co = 0; co < C::COUT; ++co) {
                    double u = 0.0;
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) u += G[r][a] * (double)hwio[((size_t)(a * 3 + b) * C::CIN + ci) * C::COUT + co] * G[c][b];
                    U[((size_t)(r * 4 + c) * C::CIN + ci) * C::COUT + co] = (float)u;
                }
    const float S = f16x2_weight_scale(U.data(), U.size());
    if (inv_sw) *inv_sw = 1.0f / S;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            for (int s = 0; s < C::NS; ++s)
                for (int kb = 0; kb < W3X::NKB; ++kb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int li = lane & 15, kq = lane >> 4, co = 16 * s + li;
                            const int ci = 32 * kb + (j < 4 ? 4 * kq + j : 16 + 4 * kq + j - 4);
                            uint16_t pl[2];
                            f16x2_split(U[((size_t)(r * 4 + c) * C::CIN + ci) * C::COUT + co], S, pl[0], pl[1]);
                            for (int p = 0; p < 2; ++p)
                                dst[((((((size_t)c * 4 + r) * C::NS + s) * W3X::NKB + kb) * 2 + p) * 64 + lane) * 8 + j] = pl[p];
                        }
    return total;
}

hipError_t launch_conv3_wino_h2(const float* in, const uint16_t* uplanes, float inv_sw, const float* ep, float* out, int64_t n_cells,
                                hipStream_t stream)
{
    static int resident = 0;
    if (!resident) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3_wino_h2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W3H::LDS);
        if (e != hipSuccess) return e;
        int dev = 0, cus = 0, per_cu = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv3_wino_h2_kernel, 256, W3H::LDS)) != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        resident = cus * per_cu;
    }
    const long total = (long)n_cells * WinoL3::NGRP;
    if (total <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(total < resident ? total : resident);
    hipLaunchKernelGGL(conv3_wino_h2_kernel, dim3(grid), dim3(256), W3H::LDS, stream, in, (const f16x8*)uplanes, ep, out, (long)n_cells, inv_sw);
    return hipGetLastError();
}

hipError_t launch_conv_wino_cs(int layer, const float* in, const float* ufrag, const float* ep, float* out, int64_t n_cells,
                               hipStream_t stream)
{
    static const bool no_ring = getenv("CS_WINO_DIAG") != nullptr;       // tools/wino_diag.py stamps the strip kernel
    if (layer == 1 && !no_ring) {
        static int resident = 0;
        constexpr int lds = 2 * WinoL2::STRIP + WinoL2::XCH;
        if (!resident) {
            hipError_t e = hipFuncSetAttribute((const void*)conv2_wino_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return e;
            int dev = 0, cus = 0, per_cu = 0;
            if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
            if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
            if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv2_wino_ring_kernel, 256, lds)) != hipSuccess) return e;
            if (per_cu < 1) per_cu = 1;
            resident = cus * per_cu;
        }
        if (n_cells <= 0) return hipSuccess;
        const unsigned grid = (unsigned)(n_cells < resident ? n_cells : resident);
        hipLaunchKernelGGL(conv2_wino_ring_kernel, dim3(grid), dim3(256), lds, stream, in, ufrag, ep, out, (long)n_cells);
        return hipGetLastError();
    }
    if (layer == 1) return launch<WinoL2>(layer, in, ufrag, ep, out, n_cells, stream);
    if (layer == 2) return launch<WinoL3>(layer, in, ufrag, ep, out, n_cells, stream);
    return hipErrorInvalidValue;
}

}  // namespace cs

// Diagnostic only (CS_WINO_DIAG=1): per-wave phase cycles of the LAST launch of `layer`, averaged over waves.
extern "C" int cs_debug_wino_cs_diag(int layer, double out6[6])
{
    using namespace cs;
    if (layer < 1 || layer > 2 || !g_diag[layer]) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    const size_t n = (size_t)g_diag_blocks[layer] * 24;
    unsigned long long* h = new unsigned long long[n];
    if (hipMemcpy(h, g_diag[layer], n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) { delete[] h; return -3; }
    for (int k = 0; k < 6; ++k) out6[k] = 0.0;
    for (size_t i = 0; i < n; ++i) out6[i % 6] += (double)h[i];
    for (int k = 0; k < 6; ++k) out6[k] /= (double)(n / 6);
    delete[] h;
    return 0;
}
