// conv_mfma.hip -- 3x3 'same' conv + bias + ReLU + BatchNorm(inference) [+ 2x2 max-pool]
// for conv layers 1..6 of the reference autoencoder (CAE_improved_modeltrain.py:191-214)
// as an exact-fp32 MFMA implicit GEMM on gfx950.
//
// Mapping (v_mfma_f32_16x16x4_f32: D[16 pixels][16 couts] += A[16 pixels][4 k] * B[4 k][16 couts]):
//   M = output pixels (a tile is 16 consecutive pixels of the conv grid),
//   N = output channels (a wave owns one 16-channel slice),
//   K = 9 taps x cin, walked in (tap, 16-channel block, 4 x 4-channel quads) order.
// The B operand (the wave's [K x 16] weight slice, 72 or 144 values per lane) stays in
// VGPRs for the lifetime of the persistent workgroup: weights never touch LDS, so LDS
// only holds the activation strip and several workgroups share a CU, overlapping one
// workgroup's staging/epilogue with another's MFMAs.
// The A operand is read from an NHWC strip staged in LDS with a zero halo; the pixel
// stride is padded by 16 B so the 16 pixels of a ds_read_b128 land on distinct 16-B
// bank slots.  For the decoder layers the strip is kept at the stored (half) resolution
// and UpSampling2D(2x2, nearest) is a `>> 1` in the A-operand address.
// Accumulation is a k-ordered fp32 fma chain (the MFMA's exact semantics), one rounding
// per product, bias added after the sum -- same numerics class as the fp32 oracle.
#include "common.hpp"

#include <cstdlib>
#include <type_traits>

namespace cs {

// Strip staging modes (how the next strip's HBM/L2 latency is hidden behind the MFMA loop):
//   STAGE_PF  the whole next strip is loaded into registers before the MFMA loop of the current
//             item and written to LDS after it (small strips / kernels with registers to spare);
//   STAGE_DB  two LDS strip buffers; every tile-pair iteration loads a slice of the NEXT strip at
//             its top and writes it to the other buffer at its bottom (a few registers in flight,
//             one barrier per item) -- for the register-heavy layers (cin = 64: 144 weight VGPRs).
// Measured on MI355X (round 1): with the strip staged synchronously between items the
// co-resident workgroups did not hide it (conv3 lost 23 %, conv2/conv6 8 % of the matrix pipe
// to it); a start-up stagger of the workgroups and static per-slot s_setprio changed nothing;
// conv2 needs 3 waves/SIMD (it has half the MFMAs per tile pair of the cin = 64 layers), which
// is why it uses STAGE_PF on 4-row strips rather than STAGE_DB.
enum { STAGE_PF = 1, STAGE_DB = 2 };

// Epilogues:
//   EPI_BN       bias -> ReLU -> BN(x*s+t)                      inference, decoder layers
//   EPI_BN_POOL  bias -> ReLU -> BN -> 2x2 max-pool              inference, encoder layers
//   EPI_RELU     bias -> ReLU (BatchNormalization in training mode needs batch statistics of
//                this tensor first, so BN/pool run in their own kernels)   training forward
//   EPI_PLAIN    nothing                                         training: gradient wrt the conv input
//   EPI_SUMPOOL  2x2 sum (adjoint of UpSampling2D nearest)       training: same, through an upsample
enum { EPI_BN = 0, EPI_BN_POOL = 1, EPI_RELU = 2, EPI_PLAIN = 3, EPI_SUMPOOL = 4 };

// FOLD_ (upsampled layers only): nearest x2 upsampling makes the 3x3 taps of output pixel
// (2y+a, 2x+b) land on only 2x2 stored pixels, so the conv splits into 4 output phases (a,b), each
// a 2x2-tap conv over the STORED grid with the taps that share a stored pixel pre-summed
// (W_eff[a][b][ry][rx] = sum of W[dy][dx] over the taps mapping there).  Exact algebra, 4/9 of
// the multiply-adds; only the order of fp32 roundings changes.  Tiles are phase-pure (16 stored
// pixels, outputs at stride 2), the tap offsets are compile-time immediates again (no >>1).
template <int H_, int W_, int CIN_, int COUT_, int EPI_, bool UPS_, int SR_, int WPS_, int MODE_, bool FOLD_ = false>
struct ConvCfg {
    static constexpr bool FOLD = FOLD_;
    static constexpr int H = H_, W = W_, CIN = CIN_, COUT = COUT_, SR = SR_, WPS = WPS_, MODE = MODE_, EPI = EPI_;
    static constexpr bool POOL = (EPI_ == EPI_BN_POOL || EPI_ == EPI_SUMPOOL), UPS = UPS_;
    static constexpr int NSL = COUT / 16;            // 16-channel output slices (= waves along N)
    static constexpr int NMG = 4 / NSL;              // wave groups along M
    static constexpr int KQ = CIN / 16;              // 16-channel K blocks per tap (0 when CIN == 1)
    // FOLD: a wave holds the effective weights of PHW phases: both column phases b, and either its
    // own row phase a (NMG == 2: the M group IS the row phase) or both (NMG == 1)
    static constexpr int PHW = (4 / (COUT / 16) == 2) ? 2 : 4;
    static constexpr int NB = (CIN == 1) ? 3 : (FOLD_ ? PHW * 4 * (CIN / 16) * 4 : 9 * (CIN / 16) * 4);  // B fragment registers per lane
    static constexpr int HS = UPS ? H / 2 : H;       // stored input size
    static constexpr int WS = UPS ? W / 2 : W;
    static constexpr int R = UPS ? SR / 2 + 2 : SR + 2;     // staged rows (incl. halo)
    static constexpr int WP = WS + 2;                // staged cols (incl. halo)
    // floats per staged pixel: +8 makes the pixel stride 2*odd 16-B slots, so the 16 pixels x 2 channel
    // quads of each ds_read_b128 lane group land on 16 distinct slots (+4 left a 2-way conflict on
    // every read: PMC SQ_LDS_BANK_CONFLICT was 48 % of SQ_LDS_IDX_ACTIVE)
    static constexpr int PS = (CIN == 1) ? 1 : CIN + 8;
    static constexpr int STRIP_BYTES = (R * WP * PS * 4 + 15) / 16 * 16;
    static constexpr int LDS_BYTES = STRIP_BYTES * (MODE == STAGE_DB ? 2 : 1);
    static constexpr int NSTRIP = H / SR;
    static constexpr int TPR = (W >= 16) ? W / 16 : 1;      // tiles per conv row (W >= 16)
    static constexpr int TILES = SR * W / 16;
    static constexpr int NPAIR = TILES / 2;
    static constexpr int PPW = NPAIR / NMG;          // tile pairs per wave per item
    static constexpr int HO = POOL ? H / 2 : H, WO = POOL ? W / 2 : W;
    static constexpr int C4 = (CIN == 1) ? 1 : CIN / 4;     // staged elements (16 B; 4 B for conv1) per pixel
    static constexpr int TOT = R * WP * C4;          // staged elements per strip
    static constexpr int NLD = (TOT + 255) / 256;    // per-thread loads per strip
    static constexpr int LPP = (NLD + PPW - 1) / PPW;       // per-thread loads per pair iteration (STAGE_DB)
    static_assert(COUT % 16 == 0 && (NSL == 2 || NSL == 4), "cout must be 32 or 64");
    static_assert(CIN == 1 || CIN % 16 == 0, "cin must be 1 or a multiple of 16");
    static_assert(!POOL || W >= 16, "pooled layers need whole-row tiles");
    static_assert(!UPS || W >= 16, "upsampled layers need whole-row tiles");
    static_assert(TILES % 2 == 0 && NPAIR % NMG == 0, "strip must split into tile pairs");
    static_assert(H % SR == 0 && SR % 2 == 0, "strip rows");
    static_assert(MODE == STAGE_PF || MODE == STAGE_DB, "staging mode");
    static_assert(!FOLD_ || (UPS_ && (EPI_ == EPI_BN || EPI_ == EPI_RELU) && CIN_ >= 16), "FOLD needs an upsampled, unpooled layer");
    static_assert(LDS_BYTES <= 160 * 1024, "strip does not fit the 160 KB LDS");
};

// Strip-local conv-grid coordinates of pixel `i` (0..15) of tile `t`.
template <class C>
__device__ __forceinline__ void tile_pixel(int t, int i, int& py, int& px)
{
    if constexpr (C::W >= 16) {
        py = t / C::TPR;
        px = (t % C::TPR) * 16 + i;
    } else {  // W == 8: a tile is two rows of eight
        py = 2 * t + (i >> 3);
        px = i & 7;
    }
}

// Byte address in the staged strip of channel block `kq` of the input pixel that tap
// (dy,dx) of conv pixel (py,px) reads.
template <class C>
__device__ __forceinline__ int a_addr(int py, int px, int kq, int tap)
{
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    if constexpr (C::UPS) {
        const int r = ((py + dy) >> 1) + 1;   // arithmetic shift: -1 -> halo row 0
        const int c = ((px + dx) >> 1) + 1;
        return (r * C::WP + c) * (C::PS * 4) + kq * 16;
    } else {
        return ((py + dy + 1) * C::WP + (px + dx + 1)) * (C::PS * 4) + kq * 16;
    }
}

__device__ __forceinline__ float relu_bn(float v, float bias, float s, float t)
{
    v += bias;
    v = fmaxf(v, 0.0f);
    return fmaf(v, s, t);
}

template <class C>
struct Stager {
    using elem_t = typename std::conditional<C::CIN == 1, float, f32x4>::type;
    // element `idx` of the strip of (cell, strip y0): loaded from `in`, zero in the halo
    static __device__ __forceinline__ elem_t load(const float* __restrict__ in, long cell, int y0, int idx)
    {
        const int ybase = C::UPS ? (y0 / 2 - 1) : (y0 - 1);
        const float* src = in + (size_t)cell * C::HS * C::WS * C::CIN;
        const int pix = idx / C::C4, c4 = idx % C::C4;
        const int r = pix / C::WP, c = pix % C::WP;
        const int sy = ybase + r, sx = c - 1;
        const bool ok = idx < C::TOT && sy >= 0 && sy < C::HS && sx >= 0 && sx < C::WS;
        if constexpr (C::CIN == 1) {
            return ok ? src[sy * C::WS + sx] : 0.0f;
        } else {
            elem_t v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (ok) v = *(const f32x4*)(src + ((size_t)sy * C::WS + sx) * C::CIN + c4 * 4);
            return v;
        }
    }
    static __device__ __forceinline__ void store(char* strip, int idx, elem_t v)
    {
        if (idx < C::TOT) {
            if constexpr (C::CIN == 1) {
                *(float*)(strip + idx * 4) = v;
            } else {
                const int pix = idx / C::C4, c4 = idx % C::C4;
                *(f32x4*)(strip + (pix * C::PS + c4 * 4) * 4) = v;
            }
        }
    }
};

// stats_part (EPI_RELU only, may be NULL): the training forward also leaves, per workgroup, {count, mean, M2} of every channel of
// the relu outputs it wrote -- [gridDim.x][3][COUT] floats, the partials bn_stats_final_kernel merges (train.hip) -- so that
// BatchNormalization's statistics need no pass of their own over the tensor.  A lane adds the 8 values of a tile pair and their
// squares in fp32 and those into double accumulators; the 4 pixel-quad lanes of a channel are added by shuffles, the waves of a
// slice through LDS in wave order: deterministic.
template <class C>
__global__ __launch_bounds__(256, C::WPS) void conv_mfma_kernel(
    const float* __restrict__ in, const float* __restrict__ wfrag, const float* __restrict__ ep,
    float* __restrict__ out, long n_cells, float* __restrict__ stats_part)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using S = Stager<C>;
    using elem_t = typename S::elem_t;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nsl = wave % C::NSL;   // output-channel slice of this wave
    const int mg = wave / C::NSL;    // tile-pair group of this wave
    const int li = lane & 15;        // A: pixel in tile / B,D: channel in slice
    const int kq = lane >> 4;        // A,B: k within the MFMA / D: pixel quad

    // B fragments: this wave's [K x 16] weight slice, resident in registers.
    float B[C::NB];
#pragma unroll
    for (int s = 0; s < C::NB; ++s) {
        if constexpr (C::FOLD && C::NMG == 2) B[s] = wfrag[(((size_t)nsl * 2 + mg) * C::NB + s) * 64 + lane];   // [slice][a][b][tap]...
        else B[s] = wfrag[((size_t)nsl * C::NB + s) * 64 + lane];
    }

    const int co = nsl * 16 + li;
    float bias = 0.0f, bns = 1.0f, bnt = 0.0f;
    if constexpr (C::EPI == EPI_BN || C::EPI == EPI_BN_POOL) { bias = ep[co]; bns = ep[C::COUT + co]; bnt = ep[2 * C::COUT + co]; }
    if constexpr (C::EPI == EPI_RELU) bias = ep[co];
    // Touch the loop-invariant registers here.  Their first use is inside the persistent loop, and the
    // compiler's wait for these loads would sit there too -- a vmcnt(N) executed EVERY iteration that
    // also waits for the strip prefetch issued just before it.
#pragma unroll
    for (int s = 0; s < C::NB; ++s) asm volatile("" : "+v"(B[s]));
    asm volatile("" : "+v"(bias), "+v"(bns), "+v"(bnt));
    const bool all_up = __all(bns >= 0.0f) != 0;          // wave-uniform: pooled epilogue can take the max-only path

    // conv1: per-lane tap of each of the 3 K steps (k = 4 s + kq; k >= 9 is zero padding)
    int toff[3];
    bool tval[3];
    if constexpr (C::CIN == 1) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int k = 4 * s + kq;
            tval[s] = k < 9;
            const int kk = tval[s] ? k : 0;
            toff[s] = ((kk / 3) * C::WP + (kk % 3)) * 4;
        }
    }

    // EPI_RELU + stats_part: this lane's running {sum, sum of squares} live in LDS behind the strip(s) (two doubles per thread:
    // the register file of the cin = 64 layers is full), its count follows from the items the workgroup has walked
    double* const st_acc = (double*)(smem + C::LDS_BYTES) + 2 * threadIdx.x;
    int st_items = 0;
    // the sums run about a per-lane shift (the lane's first value): a channel whose variance is far below mean^2 would lose
    // its M2 to the rounding of the squares otherwise (v - shift is exact for values near the shift)
    float st_c = 0.0f;
    bool st_set = false;
    if constexpr (C::EPI == EPI_RELU) { if (stats_part) { st_acc[0] = 0.0; st_acc[1] = 0.0; } }
    const long total = n_cells * C::NSTRIP;
    const long first = blockIdx.x;
    if (first >= total) return;
    // ---- prologue: first strip ---------------------------------------------------------
    elem_t stg[C::MODE == STAGE_PF ? C::NLD : C::LPP];
    if constexpr (C::MODE == STAGE_PF) {
#pragma unroll
        for (int k = 0; k < C::NLD; ++k)
            stg[k] = S::load(in, first / C::NSTRIP, (int)(first % C::NSTRIP) * C::SR, tid + 256 * k);
    } else {
#pragma unroll 4
        for (int idx = tid; idx < C::TOT; idx += 256)
            S::store(smem, idx, S::load(in, first / C::NSTRIP, (int)(first % C::NSTRIP) * C::SR, idx));
        __syncthreads();
    }

    int buf = 0;
    for (long item = first; item < total; item += gridDim.x) {
        const long cell = item / C::NSTRIP;
        const int y0 = (int)(item % C::NSTRIP) * C::SR;
        const long nitem = item + gridDim.x;
        const bool has_next = nitem < total;
        const long ncell = nitem / C::NSTRIP;
        const int ny0 = (int)(nitem % C::NSTRIP) * C::SR;

        if constexpr (C::MODE == STAGE_PF) {
#pragma unroll
            for (int k = 0; k < C::NLD; ++k) S::store(smem, tid + 256 * k, stg[k]);   // waits for the loads
            __syncthreads();
            if (has_next) {   // in flight during the MFMA loop
#pragma unroll
                for (int k = 0; k < C::NLD; ++k) stg[k] = S::load(in, ncell, ny0, tid + 256 * k);
            }
        }
        const char* strip = smem + (C::MODE == STAGE_DB ? buf * C::STRIP_BYTES : 0);
        char* nstrip = smem + (C::MODE == STAGE_DB ? (buf ^ 1) * C::STRIP_BYTES : 0);

        // ---- epilogue pieces shared by the pair loops: D[row = 4*kq + r][col = li] -> per-mode transform
        auto post = [&](float v) -> float {
            if constexpr (C::EPI == EPI_BN || C::EPI == EPI_BN_POOL) return relu_bn(v, bias, bns, bnt);
            else if constexpr (C::EPI == EPI_RELU) return fmaxf(v + bias, 0.0f);
            else return v;
        };
        // pooled store of a tile pair (rows y, y+1; the lane holds pixels x..x+3 of both): two outputs at o, o + COUT
        // (the wave-uniform all_up test is a template argument so that an unrolled caller can hoist it out of its loop)
        auto pooled_store_t = [&](auto allup_c, const f32x4& acc0_in, const f32x4& acc1_in, float* o) {
            // The maxima below are inline-asm v_max / v_min: the compiler's hazard recogniser does not look inside inline asm and
            // the hardware does not interlock a VALU read of a register an MFMA has just written (8-pass XDL write -> VALU read:
            // 11 wait states).  conv12_fused.hip measured what that costs -- stale accumulators in the pooled maximum whenever the asm
            // lands right behind the last MFMA.  The fence takes the accumulators as read-write operands (so it sits behind the
            // producing MFMAs and ahead of every consumer) and spends the wait states.
            f32x4 acc0 = acc0_in, acc1 = acc1_in;
            if constexpr (C::EPI != EPI_SUMPOOL) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc0), "+v"(acc1));
            if constexpr (C::EPI == EPI_SUMPOOL) {
                o[0] = (acc0[0] + acc0[1]) + (acc1[0] + acc1[1]);
                o[C::COUT] = (acc0[2] + acc0[3]) + (acc1[2] + acc1[3]);
            } else {
                // MaxPooling2D after bias -> ReLU -> BN: that map is monotone (non-decreasing where the BN
                // scale is >= 0, non-increasing where it is negative), so the max over the window of the
                // mapped values is the map of the window's max (resp. min) of the raw sums -- the same bits
                // for a third of the epilogue's VALU work
                // raw v_max / v_min (fmaxf would first canonicalise each operand with a v_max x, x)
                auto vmax = [](float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; };
                auto vmin = [](float a, float b) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; };
                if constexpr (decltype(allup_c)::value) {   // every channel of this wave has a non-negative BN scale (wave-uniform, the usual case)
                    o[0] = post(vmax(vmax(acc0[0], acc0[1]), vmax(acc1[0], acc1[1])));
                    o[C::COUT] = post(vmax(vmax(acc0[2], acc0[3]), vmax(acc1[2], acc1[3])));
                } else {
                    const bool up = bns >= 0.0f;
                    auto ext = [&](float a, float b, float c, float d) {
                        const float mx = vmax(vmax(a, b), vmax(c, d)), mn = vmin(vmin(a, b), vmin(c, d));
                        return up ? mx : mn;
                    };
                    o[0] = post(ext(acc0[0], acc0[1], acc1[0], acc1[1]));
                    o[C::COUT] = post(ext(acc0[2], acc0[3], acc1[2], acc1[3]));
                }
            }
        };
        auto pooled_store = [&](const f32x4& acc0, const f32x4& acc1, float* o) {
            if (all_up) pooled_store_t(std::true_type{}, acc0, acc1, o);
            else pooled_store_t(std::false_type{}, acc0, acc1, o);
        };

        // ---- folded upsample conv: phase-pure tile pairs ((a,0),(a,1)) over stored pixels -------
        if constexpr (C::FOLD) {
            constexpr int SRS = C::SR / 2;                                   // stored rows in the strip
            constexpr int RPT = (C::WS >= 16) ? 1 : 16 / C::WS;              // stored rows per tile
            constexpr int TPRS = (C::WS >= 16) ? C::WS / 16 : 1;             // tiles per stored row
            constexpr int NT = (SRS / RPT) * TPRS;                           // tiles per strip and phase
            constexpr int NA = (C::NMG == 2) ? 1 : 2;                        // row phases this wave walks
            constexpr int GK = 4 * C::KQ;                                    // K groups per tile: 4 taps x 16-ch blocks
            static_assert(SRS % RPT == 0, "strip rows");
            int pi = 0;
            for (int ai = 0; ai < NA; ++ai) {
                const int a = (C::NMG == 2) ? mg : ai;                       // row phase (wave-uniform)
                for (int t = 0; t < NT; ++t, ++pi) {
                    if constexpr (C::MODE == STAGE_DB) {
                        if (has_next) {
#pragma unroll
                            for (int j = 0; j < C::LPP; ++j) stg[j] = S::load(in, ncell, ny0, tid + 256 * (pi * C::LPP + j));
                        }
                    }
                    // stored pixel of lane li in tile t
                    int ys, xs;
                    if constexpr (C::WS >= 16) { ys = t / TPRS; xs = (t % TPRS) * 16 + li; }
                    else { ys = t * RPT + li / C::WS; xs = li % C::WS; }
                    // LDS row 0 is stored row -1: stored (ys + a - 1 + ry, xs + b - 1 + rx) -> LDS (ys + a + ry, xs + b + rx)
                    const int base = ((ys + a) * C::WP + xs) * (C::PS * 4) + kq * 16;
                    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};   // column phases b = 0, 1
                    auto rd = [&](int b, int g) -> f32x4 {
                        const int tap = g / C::KQ, q = g % C::KQ, ry = tap >> 1, rx = tap & 1;
                        return *(const f32x4*)(strip + base + (ry * C::WP + b + rx) * (C::PS * 4) + q * 64);
                    };
                    f32x4 a0 = rd(0, 0), a1 = rd(1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                    for (int g = 0; g < GK; ++g) {
                        f32x4 n0 = a0, n1 = a1;
                        if (g + 1 < GK) { n0 = rd(0, g + 1); n1 = rd(1, g + 1); }
                        const int pw0 = (C::NMG == 2) ? 0 : ai * 2, pw1 = pw0 + 1;   // phase slots of b = 0, 1
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], B[(pw0 * GK + g) * 4 + j], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], B[(pw1 * GK + g) * 4 + j], acc1, 0, 0, 0);
                        }
                        a0 = n0;
                        a1 = n1;
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                    }
                    // epilogue: D row 4*kq + r -> stored pixel -> output (2*ys + a, 2*xs + b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i2 = 4 * kq + r;
                        int ys2, xs2;
                        if constexpr (C::WS >= 16) { ys2 = t / TPRS; xs2 = (t % TPRS) * 16 + i2; }
                        else { ys2 = t * RPT + i2 / C::WS; xs2 = i2 % C::WS; }
                        const int Y = y0 + 2 * ys2 + a, X = 2 * xs2;
                        float* o = out + (((size_t)cell * C::HO + Y) * C::WO + X) * C::COUT + co;
                        if constexpr (C::EPI == EPI_RELU) {
                            o[0] = fmaxf(acc0[r] + bias, 0.0f);
                            o[C::COUT] = fmaxf(acc1[r] + bias, 0.0f);
                        } else {
                            o[0] = relu_bn(acc0[r], bias, bns, bnt);
                            o[C::COUT] = relu_bn(acc1[r], bias, bns, bnt);
                        }
                    }
                    if constexpr (C::MODE == STAGE_DB) {
                        if (has_next) {
#pragma unroll
                            for (int j = 0; j < C::LPP; ++j) S::store(nstrip, tid + 256 * (pi * C::LPP + j), stg[j]);
                        }
                    }
                }
            }
        } else if constexpr (C::CIN == 1 && C::POOL && C::W >= 16 && C::TPR % C::NMG == 0 && C::MODE == STAGE_PF) {
        // ---- conv1 (and its transposed twin in training): the 16 tile pairs of a wave, fully unrolled.  Pair
        // p = mg + NMG it sits at tile row 2 ((NMG it) / TPR), tile column (NMG it) % TPR + mg: with the loop unrolled
        // everything but mg is a constant, so LDS reads and stores take one base register + immediate offsets.  The
        // rolled loop spent as many scalar and address instructions per pair as it has MFMAs' worth of issue time
        // (PMC: 5.2 k SALU + 6.1 k VALU against 1.5 k MFMA per cell) on div/mod of the pair index.
        {
            const char* lbase = strip + (16 * mg + li) * 4;
            float* obase = out + (((size_t)cell * C::HO + (y0 >> 1)) * C::WO + 8 * mg + 2 * kq) * C::COUT + co;
            auto pairs = [&](auto allup_c) {
#pragma unroll
                for (int it = 0; it < C::PPW; ++it) {
                    const int pq = C::NMG * it, ry = 2 * (pq / C::TPR), xq = pq % C::TPR;
                    const char* a = lbase + (ry * C::WP + 16 * xq) * 4;
                    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const float a0 = *(const float*)(a + toff[s]);
                        const float a1 = *(const float*)(a + toff[s] + C::WP * 4);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, B[s], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, B[s], acc1, 0, 0, 0);
                    }
                    pooled_store_t(allup_c, acc0, acc1, obase + ((size_t)(ry / 2) * C::WO + 8 * xq) * C::COUT);
                }
            };
            if (all_up) pairs(std::true_type{});
            else pairs(std::false_type{});
        }
        } else {
        // ---- tile pairs ----------------------------------------------------------
        int pi = 0;
        for (int p = mg; p < C::NPAIR; p += C::NMG, ++pi) {
            if constexpr (C::MODE == STAGE_DB) {
                if (has_next) {   // this iteration's slice of the next strip: loads now, LDS writes at the bottom
#pragma unroll
                    for (int j = 0; j < C::LPP; ++j) stg[j] = S::load(in, ncell, ny0, tid + 256 * (pi * C::LPP + j));
                }
            }
            int t0, t1;
            if constexpr (C::POOL) {  // vertical pool partners: same columns, rows 2r and 2r+1
                const int ry = 2 * (p / C::TPR), xb = p % C::TPR;
                t0 = ry * C::TPR + xb;
                t1 = (ry + 1) * C::TPR + xb;
            } else {
                t0 = 2 * p;
                t1 = 2 * p + 1;
            }
            int py0, px0, py1, px1;
            tile_pixel<C>(t0, li, py0, px0);
            tile_pixel<C>(t1, li, py1, px1);

            f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (C::CIN == 1) {
                // fp32 MFMA and VALU instructions do not overlap, and this layer has only 6 MFMAs
                // per tile pair, so every VALU instruction here is on the critical path:
                //  - the second tile of the pair sits a compile-time distance from the first (the row below when
                //    pooling, the next 16 pixels otherwise), so its reads share the address register (ds_read offset);
                //  - the lanes of the padded taps (k = 9..11) are not masked: their B values are zero, and what they read
                //    (tap 0 of the same pixel) is finite whenever the true taps are, so 0 * a contributes exactly 0.
                const int b0 = (py0 * C::WP + px0) * 4;
                constexpr int D01 = C::POOL ? C::WP * 4 : 16 * 4;
                static_assert(C::POOL || C::TPR % 2 == 0, "pair = two tiles of one row");
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const float a0 = *(const float*)(strip + b0 + toff[s]);
                    const float a1 = *(const float*)(strip + b0 + toff[s] + D01);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, B[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, B[s], acc1, 0, 0, 0);
                }
            } else {
                // K walk in groups g = (tap, 16-channel block): one 16-B LDS read per tile feeds 4 MFMAs.
                // The reads of group g+1 are issued before the MFMAs of group g (software prefetch),
                // so the matrix pipe never waits on an LDS round trip.
                constexpr int G = 9 * C::KQ;
                int ad0[9], ad1[9];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    ad0[tap] = a_addr<C>(py0, px0, kq, tap);
                    ad1[tap] = a_addr<C>(py1, px1, kq, tap);
                }
                f32x4 a0 = *(const f32x4*)(strip + ad0[0]);
                f32x4 a1 = *(const f32x4*)(strip + ad1[0]);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // group 0's reads
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    f32x4 n0 = a0, n1 = a1;
                    if (g + 1 < G) {
                        n0 = *(const f32x4*)(strip + ad0[(g + 1) / C::KQ] + ((g + 1) % C::KQ) * 64);
                        n1 = *(const f32x4*)(strip + ad1[(g + 1) / C::KQ] + ((g + 1) % C::KQ) * 64);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float b = B[g * 4 + j];
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b, acc1, 0, 0, 0);
                    }
                    a0 = n0;
                    a1 = n1;
                    // pin the schedule: this group's 2 prefetch reads first, then its 8 MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                }
            }

            // ---- epilogue: D[row = 4*kq + r][col = li] -> per-mode transform, store
            if constexpr (C::POOL) {
                // lane holds pixels x = xq..xq+3 of rows (y, y+1); t0/t1 share columns
                int qy, qx;
                tile_pixel<C>(t0, 4 * kq, qy, qx);
                const int yo = (y0 + qy) >> 1;
                const int xo = qx >> 1;
                float* o = out + (((size_t)cell * C::HO + yo) * C::WO + xo) * C::COUT + co;
                pooled_store(acc0, acc1, o);
            } else {
                float ps = 0.0f, pq = 0.0f;              // the pair's 8 values (minus the shift) in fp32, then into the double accumulators
                if constexpr (C::EPI == EPI_RELU) {
                    if (!st_set) { st_c = post(acc0[0]); st_set = true; }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int qy, qx;
                    tile_pixel<C>(t0, 4 * kq + r, qy, qx);
                    const float v0 = post(acc0[r]);
                    out[(((size_t)cell * C::HO + (y0 + qy)) * C::WO + qx) * C::COUT + co] = v0;
                    tile_pixel<C>(t1, 4 * kq + r, qy, qx);
                    const float v1 = post(acc1[r]);
                    out[(((size_t)cell * C::HO + (y0 + qy)) * C::WO + qx) * C::COUT + co] = v1;
                    if constexpr (C::EPI == EPI_RELU) {
                        const float d0 = v0 - st_c, d1 = v1 - st_c;
                        ps += d0 + d1;
                        pq = fmaf(d0, d0, pq);
                        pq = fmaf(d1, d1, pq);
                    }
                }
                if constexpr (C::EPI == EPI_RELU) {
                    if (stats_part) { st_acc[0] += (double)ps; st_acc[1] += (double)pq; }
                }
            }

            if constexpr (C::MODE == STAGE_DB) {
                if (has_next) {
#pragma unroll
                    for (int j = 0; j < C::LPP; ++j) S::store(nstrip, tid + 256 * (pi * C::LPP + j), stg[j]);
                }
            }
        }
        }   // !FOLD
        __syncthreads();  // all reads of this strip done; (STAGE_DB) next strip complete in the other buffer
        buf ^= 1;
        ++st_items;
    }
    if constexpr (C::EPI == EPI_RELU) {
        if (stats_part) {
            // per lane {mean, M2} from its shifted sums (double), then Chan's merge of equal counts over the lanes kq = 0..3 of a
            // channel, then over the NMG waves of the slice (wave = mg * NSL + nsl) in order
            double nl = (double)st_items * (8.0 * C::PPW);       // a lane writes 8 values per tile pair, PPW pairs per item
            if (nl == 0.0) nl = 1.0;                             // (never: the grid is at most the number of items)
            double st_m = (double)st_c + st_acc[0] / nl, st_v = st_acc[1] - st_acc[0] * (st_acc[0] / nl);
            st_v = st_v > 0.0 ? st_v : 0.0;
#pragma unroll
            for (int x = 16; x <= 32; x <<= 1) {
                const double om = __shfl_xor(st_m, x), ov = __shfl_xor(st_v, x), dm = om - st_m;
                st_v = (st_v + ov) + dm * dm * (0.5 * nl);
                st_m = 0.5 * (st_m + om);
                nl *= 2.0;
            }
            double* red = (double*)smem;                   // [wave 4][2][16]; the strip is dead (barrier above)
            if (kq == 0) { red[(wave * 2 + 0) * 16 + li] = st_m; red[(wave * 2 + 1) * 16 + li] = st_v; }
            __syncthreads();
            if (tid < C::COUT) {
                const int sl = tid >> 4, l16 = tid & 15;
                double n = 0.0, mu = 0.0, m2 = 0.0;
#pragma unroll
                for (int g = 0; g < C::NMG; ++g) {
                    const int w = g * C::NSL + sl;
                    const double om = red[(w * 2 + 0) * 16 + l16], ov = red[(w * 2 + 1) * 16 + l16], dm = om - mu, nn = n + nl;
                    m2 = (m2 + ov) + dm * dm * (n * nl / nn);
                    mu += dm * (nl / nn);
                    n = nn;
                }
                float* o = stats_part + (size_t)blockIdx.x * 3 * C::COUT;
                o[tid] = (float)n;
                o[C::COUT + tid] = (float)mu;
                o[2 * C::COUT + tid] = (float)m2;
            }
        }
    }
}

//                       H   W  CIN COUT EPI          UPS    SR WPS MODE
// inference (CAE_improved_modeltrain.py:191-214 with BatchNormalization in inference mode)
using CfgL1 = ConvCfg<64, 64,  1, 32, EPI_BN_POOL, false, 16, 4, STAGE_PF>;   // :191-193
using CfgL2 = ConvCfg<32, 32, 32, 64, EPI_BN_POOL, false,  8, 2, STAGE_PF>;   // :195-197
using CfgL3 = ConvCfg<16, 16, 64, 32, EPI_BN_POOL, false,  4, 2, STAGE_DB>;   // :199-201 -> encoded 8x8x32
using CfgL4 = ConvCfg< 8,  8, 32, 32, EPI_BN,      false,  8, 3, STAGE_PF>;   // :204-205
using CfgL5 = ConvCfg<16, 16, 32, 64, EPI_BN,      true,  16, 3, STAGE_PF>;   // :206-209 (reads up(a4))
using CfgL6 = ConvCfg<32, 32, 64, 32, EPI_BN,      true,   8, 2, STAGE_DB>;   // :210-213 (reads up(a5))
// the same two decoder layers with the upsample folded into 4 phase convs (4/9 of the MACs)
using CfgL5F = ConvCfg<16, 16, 32, 64, EPI_BN,     true,  16, 2, STAGE_PF, true>;
using CfgL6F = ConvCfg<32, 32, 64, 32, EPI_BN,     true,   8, 2, STAGE_DB, true>;
// training forward: conv + bias + ReLU at full conv-grid resolution (BN batch stats come next).  Strips are SHORT: a fit() batch is
// 32 cells, and 32 cells x H / SR strips is the whole grid -- at the inference kernels' strip heights half of these launches ran on
// 32 .. 128 of the 256 CUs
using CfgF1 = ConvCfg<64, 64,  1, 32, EPI_RELU,    false,  8, 4, STAGE_PF>;
using CfgF2 = ConvCfg<32, 32, 32, 64, EPI_RELU,    false,  4, 3, STAGE_PF>;
using CfgF3 = ConvCfg<16, 16, 64, 32, EPI_RELU,    false,  4, 2, STAGE_DB>;
using CfgF4 = ConvCfg< 8,  8, 32, 32, EPI_RELU,    false,  8, 3, STAGE_PF>;
using CfgF5 = ConvCfg<16, 16, 32, 64, EPI_RELU,    true,   4, 3, STAGE_PF>;
using CfgF6 = ConvCfg<32, 32, 64, 32, EPI_RELU,    true,   4, 2, STAGE_DB>;
// training backward-data: dX = conv(dZ, flipped/transposed kernel) [+ 2x2 sum through an upsample].
// D<l> is the gradient wrt the input of conv l (1-based): channels swap roles.
using CfgD7 = ConvCfg<64, 64,  1, 32, EPI_SUMPOOL, false,  8, 4, STAGE_PF>;   // dz7 (1 ch) -> d a6 (32x32x32)
using CfgD6 = ConvCfg<32, 32, 32, 64, EPI_SUMPOOL, false,  4, 3, STAGE_PF>;   // dz6 (32 ch) -> d a5 (16x16x64)
using CfgD5 = ConvCfg<16, 16, 64, 32, EPI_SUMPOOL, false,  4, 2, STAGE_DB>;   // dz5 (64 ch) -> d a4 (8x8x32)
using CfgD4 = ConvCfg< 8,  8, 32, 32, EPI_PLAIN,   false,  8, 3, STAGE_PF>;   // dz4 -> d p3
using CfgD3 = ConvCfg<16, 16, 32, 64, EPI_PLAIN,   false,  2, 3, STAGE_PF>;   // dz3 (32 ch) -> d p2 (16x16x64)
using CfgD2 = ConvCfg<32, 32, 64, 32, EPI_PLAIN,   false,  4, 2, STAGE_DB>;   // dz2 (64 ch) -> d p1 (32x32x32)

// the training-forward kernels keep two doubles per thread behind the strip(s) (the statistics accumulators)
template <class C> constexpr int kStatsLds = (C::EPI == EPI_RELU) ? 256 * 16 : 0;

template <class C>
static hipError_t launch_cfg(const float* in, const float* wfrag, const float* ep, float* out,
                             int64_t n_cells, hipStream_t stream, float* stats_part = nullptr, int* stats_parts = nullptr)
{
    // Persistent grid = exactly the number of workgroups the chip holds at once (CUs x resident
    // workgroups per CU for this kernel's registers and LDS): a larger grid would queue the
    // surplus behind the first wave of workgroups and run it at a fraction of the occupancy.
    static int resident = 0, cus = 0;
    if (!resident) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_mfma_kernel<C>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES + kStatsLds<C>);
        if (e != hipSuccess) return e;
        int dev = 0, per_cu = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv_mfma_kernel<C>, 256, C::LDS_BYTES + kStatsLds<C>);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        resident = cus * per_cu;
    }
    const long total = (long)n_cells * C::NSTRIP;
    if (total <= 0) return hipSuccess;
    unsigned grid = (unsigned)(total < resident ? total : resident);
    if (stats_part) {      // one {count, mean, M2} x COUT partial per workgroup in a BN_MAX_PARTS x 3 x 64-float buffer
        const unsigned cap = (unsigned)(BN_MAX_PARTS * 64 / C::COUT);
        if (grid > cap) grid = cap;
    }
    if (stats_parts) *stats_parts = (int)grid;
    hipLaunchKernelGGL(conv_mfma_kernel<C>, dim3(grid), dim3(256), C::LDS_BYTES + kStatsLds<C>, stream, in, wfrag, ep,
                       out, (long)n_cells, stats_part);
    return hipGetLastError();
}

hipError_t launch_conv_mfma(int layer, const float* in, const float* wfrag, const float* ep,
                            float* out, int64_t n_cells, hipStream_t stream, bool folded)
{
    if (folded && layer == 4) return launch_cfg<CfgL5F>(in, wfrag, ep, out, n_cells, stream);
    if (folded && layer == 5) return launch_cfg<CfgL6F>(in, wfrag, ep, out, n_cells, stream);
    switch (layer) {
        case 0: return launch_cfg<CfgL1>(in, wfrag, ep, out, n_cells, stream);
        case 1: return launch_cfg<CfgL2>(in, wfrag, ep, out, n_cells, stream);
        case 2: return launch_cfg<CfgL3>(in, wfrag, ep, out, n_cells, stream);
        case 3: return launch_cfg<CfgL4>(in, wfrag, ep, out, n_cells, stream);
        case 4: return launch_cfg<CfgL5>(in, wfrag, ep, out, n_cells, stream);
        case 5: return launch_cfg<CfgL6>(in, wfrag, ep, out, n_cells, stream);
        default: return hipErrorInvalidValue;
    }
}

// stats_part / stats_parts (optional): the kernel also leaves its workgroups' {count, mean, M2} per channel of what it wrote in
// stats_part [*stats_parts][3][cout] -- the input of launch_bn_stats_final, in place of launch_bn_stats' pass over the tensor
hipError_t launch_conv_train_fwd(int layer, const float* in, const float* wfrag, const float* bias,
                                 float* relu_out, int64_t n_cells, hipStream_t stream, float* stats_part, int* stats_parts)
{
    switch (layer) {
        case 0: return launch_cfg<CfgF1>(in, wfrag, bias, relu_out, n_cells, stream, stats_part, stats_parts);
        case 1: return launch_cfg<CfgF2>(in, wfrag, bias, relu_out, n_cells, stream, stats_part, stats_parts);
        case 2: return launch_cfg<CfgF3>(in, wfrag, bias, relu_out, n_cells, stream, stats_part, stats_parts);
        case 3: return launch_cfg<CfgF4>(in, wfrag, bias, relu_out, n_cells, stream, stats_part, stats_parts);
        case 4: return launch_cfg<CfgF5>(in, wfrag, bias, relu_out, n_cells, stream, stats_part, stats_parts);
        case 5: return launch_cfg<CfgF6>(in, wfrag, bias, relu_out, n_cells, stream, stats_part, stats_parts);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_conv_dgrad(int layer, const float* dz, const float* wfrag_t, float* dx,
                             int64_t n_cells, hipStream_t stream)
{
    switch (layer) {   // 0-based conv index whose INPUT gradient is produced (layer 0 has none)
        case 1: return launch_cfg<CfgD2>(dz, wfrag_t, nullptr, dx, n_cells, stream);
        case 2: return launch_cfg<CfgD3>(dz, wfrag_t, nullptr, dx, n_cells, stream);
        case 3: return launch_cfg<CfgD4>(dz, wfrag_t, nullptr, dx, n_cells, stream);
        case 4: return launch_cfg<CfgD5>(dz, wfrag_t, nullptr, dx, n_cells, stream);
        case 5: return launch_cfg<CfgD6>(dz, wfrag_t, nullptr, dx, n_cells, stream);
        case 6: return launch_cfg<CfgD7>(dz, wfrag_t, nullptr, dx, n_cells, stream);
        default: return hipErrorInvalidValue;
    }
}

// Folded-upsample fragments (ConvCfg FOLD): effective 2x2 kernels per output phase (a,b),
//   W_eff[a][b][ry][rx] = sum of W[dy][dx] over taps with ((a+dy)>>1)+1 == a+ry and ((b+dx)>>1)+1 == b+rx
// (dy,dx in -1..1), summed in fp32 in (dy,dx) order.  Layout: cout 32 (two waves per slice, one
// per row phase): [slice][a][(b*4 + tap)*KQ + q][j][lane]; cout 64: [slice][((a*2+b)*4 + tap)*KQ + q][j][lane].
size_t pack_conv_fragments_folded(int cin, int cout, const float* hwio, float* dst)
{
    const int nsl_n = cout / 16, nmg = 4 / nsl_n, kq_n = cin / 16, gk = 4 * kq_n;
    const int phw = (nmg == 2) ? 2 : 4, nb = phw * gk * 4;
    const size_t total = (size_t)nsl_n * (nmg == 2 ? 2 : 1) * nb * 64;
    if (!dst) return total;
    for (int nsl = 0; nsl < nsl_n; ++nsl)
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                for (int tap = 0; tap < 4; ++tap)
                    for (int q = 0; q < kq_n; ++q)
                        for (int j = 0; j < 4; ++j)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int li = lane & 15, kq = lane >> 4, ry = tap >> 1, rx = tap & 1;
                                const int ci = 16 * q + 4 * kq + j, co = nsl * 16 + li;
                                float sum = 0.0f;
                                for (int dy = -1; dy <= 1; ++dy)
                                    for (int dx = -1; dx <= 1; ++dx)
                                        if (((a + dy) >> 1) + 1 == a + ry && ((b + dx) >> 1) + 1 == b + rx)
                                            sum += hwio[((size_t)((dy + 1) * 3 + (dx + 1)) * cin + ci) * cout + co];
                                const int g = tap * kq_n + q;
                                size_t idx;
                                if (nmg == 2) idx = ((((size_t)nsl * 2 + a) * nb) + ((size_t)(b * gk + g) * 4 + j)) * 64 + lane;
                                else idx = (((size_t)nsl * nb) + ((size_t)((a * 2 + b) * gk + g) * 4 + j)) * 64 + lane;
                                dst[idx] = sum;
                            }
    return total;
}

// wfrag[nsl][s][lane]: the value lane (li = lane & 15, kq = lane >> 4) feeds as B[k = kq][n = li]
// of K step s.  cin >= 16: s = (tap*KQ + q)*4 + j reads input channel 16q + 4kq + j.
// cin == 1: s in 0..2 reads tap 4s + kq (zero for the padded taps 9..11).
size_t pack_conv_fragments(int cin, int cout, const float* hwio, float* dst)
{
    const int nsl_n = cout / 16;
    const int kq_n = cin / 16;
    const int nb = (cin == 1) ? 3 : 9 * kq_n * 4;
    const size_t total = (size_t)nsl_n * nb * 64;
    if (!dst) return total;
    for (int nsl = 0; nsl < nsl_n; ++nsl)
        for (int s = 0; s < nb; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int li = lane & 15, kq = lane >> 4;
                const int co = nsl * 16 + li;
                float v;
                if (cin == 1) {
                    const int k = 4 * s + kq;
                    v = (k < 9) ? hwio[(size_t)k * cout + co] : 0.0f;
                } else {
                    const int j = s & 3, q = (s >> 2) % kq_n, tap = (s >> 2) / kq_n;
                    const int ci = 16 * q + 4 * kq + j;
                    v = hwio[((size_t)tap * cin + ci) * cout + co];
                }
                dst[((size_t)nsl * nb + s) * 64 + lane] = v;
            }
    return total;
}

}  // namespace cs
