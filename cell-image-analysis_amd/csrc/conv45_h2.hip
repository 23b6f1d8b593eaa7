// conv45_h2.hip -- conv4 and conv5 of the reference autoencoder (CAE_improved_modeltrain.py:203-208) with the fp32 contraction as a
// TWO-term fp16 split on v_mfma_f32_16x16x32_f16 (CS_PRECISION_SPLIT16):
//   conv4_h2_kernel   3x3 'same' 32 -> 32 on the 8x8 bottleneck, bias -> ReLU -> BatchNormalization (stage taps, CS_DEBUG_NO_FUSE45)
//   conv5_h2_kernel   UpSampling2D -> 3x3 'same' 32 -> 64 -> ReLU -> BatchNormalization, the upsample folded into four 2x2-tap
//                     phase convs over the stored grid (4/9 of the multiply-adds)
//   conv45_h2_kernel  the two as one kernel, a4 in LDS: what cs_screen runs
//
// Mapping (D[16 pixels][16 couts] += A[16 pixels][32 ch] * B[32 ch][16 couts], one MFMA per tap and product):
//   lane l: A = 8 channels 8(l>>4)..+7 of pixel l&15 of the tile (two rows of eight), B = the same 8 channels of
//   cout l&15 of the wave's slice.  The weights stay in registers; the cell is split ONCE per element when it is staged
//   into two fp16 planes in LDS with a zero halo, pixel stride 64 B and row stride 672 B (the lane groups of a ds_read_b128
//   land on 16 distinct 16-byte slots), and every tap's A fragment is one ds_read_b128 per plane.
#include "common.hpp"

#include <cstring>

namespace cs {

namespace {

constexpr int G = 8;                      // the bottleneck grid
constexpr int CH = 32;                    // channels in = channels out
constexpr int WP = G + 2;                 // staged row length (halo)
constexpr int PXB = 64;                   // bytes per staged pixel and plane: 32 fp16
constexpr int ROWB = WP * PXB + 32;       // bytes per staged row.  With 4 and 42 16-byte slots per pixel and row every lane group of a
                                          // ds_read_b128 ({0-3,12-15,20-27}, ...: pixels x 0-3 of row 0 and 4-7 of row 1 at kq, the other
                                          // eight at kq + 1) lands on 16 distinct slots (enumerated; 80 B / 896 B was 2-way everywhere)
constexpr int PLANE = WP * ROWB;          // 6,720 B
constexpr int WG_PER_CU = 2;

constexpr int C5_OUT = 64;

// ---- conv4 and conv5 with the contraction as a TWO-term fp16 split (three products): conv_wino_up.hip (conv67_h2_kernel) has the
// algebra and the hardware facts.  x = hi + lo in fp16 after an exact power-of-two scale per CELL (the whole 8x8x32 cell is
// staged at once: S_a puts the cell's max|x| into [2^14, 2^15)), the weights likewise with one scale per layer (host), the
// product as hi hi | hi lo + lo hi on two accumulators, the scales undone in the epilogue's fma.  Half the matrix instructions
// and two thirds of the weight registers of the bf16 kernels above; the staged layout is theirs with two planes.  The maximum of
// the NEXT cell is taken while this cell's MFMAs run (its loads are in flight then anyway) and read behind the barrier that
// ends the cell: no extra barrier.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int H2_OFF_MAX = 2 * PLANE;              // two words: cell maxima, alternating
constexpr int H2_LDS_BYTES = 2 * PLANE + 16;

__device__ __forceinline__ unsigned int h2_rowmax(unsigned int m)
{
    unsigned int o;
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, true);  m = m > o ? m : o;     // quad_perm [1,0,3,2]
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, true);  m = m > o ? m : o;     // quad_perm [2,3,0,1]
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0x124, 0xF, 0xF, true); m = m > o ? m : o;     // row_ror:4
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xF, 0xF, true); m = m > o ? m : o;     // row_ror:8
    return m;
}
__device__ __forceinline__ void h2_scale(unsigned int mbits, float& S, float& invS)
{
    int E = (int)((mbits >> 23) & 0xffu);
    E = E < 40 ? 40 : (E > 254 ? 254 : E);
    S = __builtin_bit_cast(float, (unsigned int)(268 - E) << 23);          // 2^(14 - (E - 127))
    invS = __builtin_bit_cast(float, (unsigned int)(E - 14) << 23);
}
// max|.| of four values into a running maximum kept as float bits (non-negative floats order like their bit patterns).  The
// elements are copied to scalars first: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 whatever the
// index (clang 19 of ROCm 7.2 -- found as a scale taken from a quarter of the data).
__device__ __forceinline__ void h2_absmax4(const f32x4& v, unsigned int& mx)
{
    const float a = v[0], b = v[1], c = v[2], d = v[3];
    const float m = fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d)));
    const unsigned int u = __builtin_bit_cast(unsigned int, m);
    mx = mx > u ? mx : u;
}
__device__ __forceinline__ void h2_split_store(char* dst, const f32x4& x, float S)
{
    const f32x4 v = x * S;
    const f16x4 hi = __builtin_convertvector(v, f16x4);
    const f32x4 r = v - __builtin_convertvector(hi, f32x4);            // exact in fp32
    *(f16x4*)dst = hi;
    *(f16x4*)(dst + PLANE) = __builtin_convertvector(r, f16x4);
}

__global__ __launch_bounds__(256, WG_PER_CU) void conv4_h2_kernel(const float* __restrict__ in, const f16x8* __restrict__ wfrag,
                                                                 const float* __restrict__ ep, float* __restrict__ out,
                                                                 long n_cells, float inv_sw)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned int* const mxw = (unsigned int*)(smem + H2_OFF_MAX);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = wave & 1;           // couts 16 slice .. +15
    const int th = wave >> 1;             // tiles 2 th, 2 th + 1 (rows 4 th .. 4 th + 3)
    const int li = lane & 15, kq = lane >> 4;

    f16x8 B[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) B[t][p] = wfrag[((slice * 9 + t) * 2 + p) * 64 + lane];
    const int co = slice * 16 + li;
    const float bias = ep[co], bns = ep[CH + co], bnt = ep[2 * CH + co];

    // zero the planes (and the two max words) once: the interior is rewritten for every cell, the halo stays zero
    for (int i = tid; i < H2_LDS_BYTES / 16; i += 256) *(f32x4*)(smem + i * 16) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    long cell = blockIdx.x;
    if (cell >= n_cells) return;
    f32x4 stg[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) stg[k] = *(const f32x4*)(in + (size_t)cell * (G * G * CH) + (tid + 256 * k) * 4);
    __syncthreads();
    {
        unsigned int mx = 0;
        h2_absmax4(stg[0], mx);
        h2_absmax4(stg[1], mx);
        mx = h2_rowmax(mx);
        if (li == 0) atomicMax(&mxw[0], mx);
    }
    __syncthreads();
    float S, invS;
    h2_scale(mxw[0], S, invS);

    int woff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = tid + 256 * k, px = idx >> 3, c4 = idx & 7;
        woff[k] = ((px >> 3) + 1) * ROWB + ((px & 7) + 1) * PXB + c4 * 8;
    }
    const int abase = (4 * th + (li >> 3)) * ROWB + (li & 7) * PXB + kq * 16;

    for (int it = 0; cell < n_cells; cell += gridDim.x, ++it) {
        const float unscale = invS * inv_sw;
#pragma unroll
        for (int k = 0; k < 2; ++k) h2_split_store(smem + woff[k], stg[k], S);
        __syncthreads();
        if (tid == 0) mxw[it & 1] = 0;      // this cell's word: read by everyone before the barrier above, next used two cells on
        const long ncell = cell + gridDim.x;
        if (ncell < n_cells) {     // in flight during the MFMA phase
#pragma unroll
            for (int k = 0; k < 2; ++k) stg[k] = *(const f32x4*)(in + (size_t)ncell * (G * G * CH) + (tid + 256 * k) * 4);
        }

        f32x4 hi[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
        f32x4 lo[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * ROWB + (tap % 3) * PXB;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const char* a = smem + abase + toff + t * (2 * ROWB);
                const f16x8 a1 = *(const f16x8*)a;
                const f16x8 a2 = *(const f16x8*)(a + PLANE);
                lo[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, B[tap][1], lo[t], 0, 0, 0);
                hi[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, B[tap][0], hi[t], 0, 0, 0);
                lo[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, B[tap][0], lo[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float* o = out + ((size_t)cell * (G * G) + 16 * (2 * th + t) + 4 * kq) * CH + co;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = fmaxf(fmaf(hi[t][r] + lo[t][r], unscale, bias), 0.0f);
                o[r * CH] = fmaf(v, bns, bnt);
            }
        }
        if (ncell < n_cells) {
            unsigned int mx = 0;
            h2_absmax4(stg[0], mx);
            h2_absmax4(stg[1], mx);
            mx = h2_rowmax(mx);
            if (li == 0) atomicMax(&mxw[(it + 1) & 1], mx);
        }
        __syncthreads();           // every wave is done reading the planes; the next cell's maximum is complete
        h2_scale(mxw[(it + 1) & 1], S, invS);
    }
}

__global__ __launch_bounds__(512, 2) void conv5_h2_kernel(const float* __restrict__ in, const f16x8* __restrict__ wfrag,
                                                          const float* __restrict__ ep, float* __restrict__ out, long n_cells, float inv_sw)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned int* const mxw = (unsigned int*)(smem + H2_OFF_MAX);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = wave >> 1, pa = ph >> 1, pb = ph & 1;      // output phase (a,b)
    const int sp = wave & 1;                                  // filters 32 sp .. 32 sp + 31
    const int li = lane & 15, kq = lane >> 4;

    f16x8 B[4][2][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int p = 0; p < 2; ++p) B[t][k][p] = wfrag[(((wave * 4 + t) * 2 + k) * 2 + p) * 64 + lane];
    float bias[2], bns[2], bnt[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int co = (2 * sp + k) * 16 + li;
        bias[k] = ep[co]; bns[k] = ep[C5_OUT + co]; bnt[k] = ep[2 * C5_OUT + co];
    }

    for (int i = tid; i < H2_LDS_BYTES / 16; i += 512) *(f32x4*)(smem + i * 16) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    long cell = blockIdx.x;
    if (cell >= n_cells) return;
    f32x4 stg = *(const f32x4*)(in + (size_t)cell * (G * G * CH) + tid * 4);
    __syncthreads();
    {
        unsigned int mx = 0;
        h2_absmax4(stg, mx);
        mx = h2_rowmax(mx);
        if (li == 0) atomicMax(&mxw[0], mx);
    }
    __syncthreads();
    float S, invS;
    h2_scale(mxw[0], S, invS);
    const int woff = ((tid >> 6) + 1) * ROWB + (((tid >> 3) & 7) + 1) * PXB + (tid & 7) * 8;
    const int abase = ((li >> 3) + pa) * ROWB + ((li & 7) + pb) * PXB + kq * 16;

    for (int it = 0; cell < n_cells; cell += gridDim.x, ++it) {
        const float unscale = invS * inv_sw;
        h2_split_store(smem + woff, stg, S);
        __syncthreads();
        if (tid == 0) mxw[it & 1] = 0;
        const long ncell = cell + gridDim.x;
        if (ncell < n_cells) stg = *(const f32x4*)(in + (size_t)ncell * (G * G * CH) + tid * 4);     // in flight during the MFMA phase

        float* obase = out + (((size_t)cell * (2 * G) + pa) * (2 * G) + pb) * C5_OUT + (2 * sp) * 16 + li;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 ah[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
            f32x4 al[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                const char* a = smem + abase + t * (2 * ROWB) + (tap >> 1) * ROWB + (tap & 1) * PXB;
                const f16x8 a1 = *(const f16x8*)a;
                const f16x8 a2 = *(const f16x8*)(a + PLANE);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    al[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, B[tap][k][1], al[k], 0, 0, 0);
                    ah[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, B[tap][k][0], ah[k], 0, 0, 0);
                    al[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, B[tap][k][0], al[k], 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * kq + r, ys = 2 * t + (i >> 3), xs = i & 7;
                float* o = obase + ((size_t)(2 * ys) * (2 * G) + 2 * xs) * C5_OUT;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float v = fmaxf(fmaf(ah[k][r] + al[k][r], unscale, bias[k]), 0.0f);
                    o[16 * k] = fmaf(v, bns[k], bnt[k]);
                }
            }
        }
        if (ncell < n_cells) {
            unsigned int mx = 0;
            h2_absmax4(stg, mx);
            mx = h2_rowmax(mx);
            if (li == 0) atomicMax(&mxw[(it + 1) & 1], mx);
        }
        __syncthreads();           // every wave is done reading the planes; the next cell's maximum is complete
        h2_scale(mxw[(it + 1) & 1], S, invS);
    }
}

// ---- conv4 + conv5 in one kernel (fp16 split): a4 never leaves LDS ----------------------------------------------------------------
// conv5 writes a5 at what HBM takes (65.5 KB per cell at ~5 TB/s: profiles/r03_b_pmc_traffic.json), so conv4's 216 MFMAs per cell and
// its two extra barriers ride under that for free, and the a4 round trip (8 KB out, 8 KB in) and a launch per chunk go.  A workgroup is
// conv5_h2_kernel's (8 waves = 4 output phases x 2 filter halves for conv5; for conv4 wave = (16-pixel tile, 16-filter slice)); p3 is
// staged as conv4_h2_kernel stages it, conv4's weights sit in LDS (37 KB, one 16-byte read per MFMA operand: registers are conv5's), its
// output goes through the same per-cell maximum -> power-of-two scale -> [hi | lo] planes as a staged input would, into a second image.
constexpr int F45_OFF_A = 2 * PLANE;                       // a4 planes
constexpr int F45_OFF_W4 = 4 * PLANE;                      // conv4 weights: pack_conv4_f16x2's [slice 2][tap 9][plane 2][lane 64][8]
constexpr int F45_W4_BYTES = 2 * 9 * 2 * 64 * 16;
constexpr int F45_OFF_MAX = F45_OFF_W4 + F45_W4_BYTES;     // four words: p3 maxima [2], a4 maxima [2] (each alternating)
constexpr int F45_LDS_BYTES = F45_OFF_MAX + 16;
static_assert(2 * F45_LDS_BYTES <= 160 * 1024, "two workgroups per CU");

__global__ __launch_bounds__(512, 2) void conv45_h2_kernel(const float* __restrict__ in /* p3 */, const f16x8* __restrict__ w4frag,
                                                           const float* __restrict__ ep4, float inv_sw4, const f16x8* __restrict__ w5frag,
                                                           const float* __restrict__ ep5, float inv_sw5, float* __restrict__ out /* a5 */,
                                                           long n_cells)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned int* const mxw = (unsigned int*)(smem + F45_OFF_MAX);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    // conv5 roles
    const int ph = wave >> 1, pa = ph >> 1, pb = ph & 1;      // output phase (a,b)
    const int sp = wave & 1;                                  // filters 32 sp .. 32 sp + 31
    // conv4 roles
    const int slice4 = wave & 1, tile4 = wave >> 1;           // filters 16 slice4 .. +15; pixels 16 tile4 .. +15 (rows 2 tile4, 2 tile4 + 1)

    f16x8 B[4][2][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int p = 0; p < 2; ++p) B[t][k][p] = w5frag[(((wave * 4 + t) * 2 + k) * 2 + p) * 64 + lane];
    float bias[2], bns[2], bnt[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int co = (2 * sp + k) * 16 + li;
        bias[k] = ep5[co]; bns[k] = ep5[C5_OUT + co]; bnt[k] = ep5[2 * C5_OUT + co];
    }
    const int co4 = slice4 * 16 + li;
    const float bias4 = ep4[co4], bns4 = ep4[CH + co4], bnt4 = ep4[2 * CH + co4];

    for (int i = tid; i < F45_LDS_BYTES / 16; i += 512) *(f32x4*)(smem + i * 16) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    __syncthreads();
    for (int i = tid; i < F45_W4_BYTES / 16; i += 512) *(f16x8*)(smem + F45_OFF_W4 + i * 16) = w4frag[i];

    long cell = blockIdx.x;
    if (cell >= n_cells) return;
    f32x4 stg = *(const f32x4*)(in + (size_t)cell * (G * G * CH) + tid * 4);
    {
        unsigned int mx = 0;
        h2_absmax4(stg, mx);
        mx = h2_rowmax(mx);
        if (li == 0) atomicMax(&mxw[0], mx);
    }
    __syncthreads();
    float S3, invS3;
    h2_scale(mxw[0], S3, invS3);
    const int woff = ((tid >> 6) + 1) * ROWB + (((tid >> 3) & 7) + 1) * PXB + (tid & 7) * 8;      // this thread's p3 element
    const int abase4 = (2 * tile4 + (li >> 3)) * ROWB + (li & 7) * PXB + kq * 16;               // conv4 A operand (image P)
    const char* const w4l = smem + F45_OFF_W4 + (size_t)slice4 * 9 * 2 * 64 * 16 + lane * 16;    // + (tap * 2 + plane) * 1024
    const int abase5 = F45_OFF_A + ((li >> 3) + pa) * ROWB + ((li & 7) + pb) * PXB + kq * 16;  // conv5 A operand (image A)

    for (int it = 0; cell < n_cells; cell += gridDim.x, ++it) {
        h2_split_store(smem + woff, stg, S3);
        __syncthreads();                                   // image P complete (and, first cell, conv4's weights)
        if (tid == 0) mxw[it & 1] = 0;
        const long ncell = cell + gridDim.x;
        if (ncell < n_cells) stg = *(const f32x4*)(in + (size_t)ncell * (G * G * CH) + tid * 4);     // in flight during both MFMA phases

        // ---- conv4: one (tile, slice) per wave
        f32x4 a4v;
        {
            f32x4 hi = {0.0f, 0.0f, 0.0f, 0.0f}, lo = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const char* a = smem + abase4 + (tap / 3) * ROWB + (tap % 3) * PXB;
                const f16x8 a1 = *(const f16x8*)a;
                const f16x8 a2 = *(const f16x8*)(a + PLANE);
                const f16x8 bh = *(const f16x8*)(w4l + (tap * 2 + 0) * 1024), bl = *(const f16x8*)(w4l + (tap * 2 + 1) * 1024);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bl, lo, 0, 0, 0);
                hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bh, hi, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, bh, lo, 0, 0, 0);
            }
            const float us4 = invS3 * inv_sw4;
            float am = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = fmaxf(fmaf(hi[r] + lo[r], us4, bias4), 0.0f);
                a4v[r] = fmaf(v, bns4, bnt4);
                am = fmaxf(am, fabsf(a4v[r]));
            }
            const unsigned int m = h2_rowmax(__builtin_bit_cast(unsigned int, am));
            if (li == 0) atomicMax(&mxw[2 + (it & 1)], m);
        }
        __syncthreads();                                   // a4's maximum complete; image P fully read
        float S4, invS4;
        h2_scale(mxw[2 + (it & 1)], S4, invS4);
        {   // D row 4 kq + r = pixel 16 tile4 + 4 kq + r, column = channel co4: [hi | lo] into image A
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = 16 * tile4 + 4 * kq + r;
                char* d = smem + F45_OFF_A + ((p >> 3) + 1) * ROWB + ((p & 7) + 1) * PXB + co4 * 2;
                const float v = a4v[r] * S4;
                const _Float16 h = (_Float16)v;
                *(_Float16*)d = h;
                *(_Float16*)(d + PLANE) = (_Float16)(v - (float)h);
            }
        }
        __syncthreads();                                   // image A complete
        if (tid == 0) mxw[2 + (it & 1)] = 0;              // read by everyone before the barrier above; next used two cells on

        // ---- conv5 (conv5_h2_kernel's body on image A)
        const float unscale = invS4 * inv_sw5;
        float* obase = out + (((size_t)cell * (2 * G) + pa) * (2 * G) + pb) * C5_OUT + (2 * sp) * 16 + li;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 ah[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
            f32x4 al[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                const char* a = smem + abase5 + t * (2 * ROWB) + (tap >> 1) * ROWB + (tap & 1) * PXB;
                const f16x8 a1 = *(const f16x8*)a;
                const f16x8 a2 = *(const f16x8*)(a + PLANE);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    al[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, B[tap][k][1], al[k], 0, 0, 0);
                    ah[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, B[tap][k][0], ah[k], 0, 0, 0);
                    al[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, B[tap][k][0], al[k], 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * kq + r, ys = 2 * t + (i >> 3), xs = i & 7;
                float* o = obase + ((size_t)(2 * ys) * (2 * G) + 2 * xs) * C5_OUT;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float v = fmaxf(fmaf(ah[k][r] + al[k][r], unscale, bias[k]), 0.0f);
                    o[16 * k] = fmaf(v, bns[k], bnt[k]);
                }
            }
        }
        if (ncell < n_cells) {
            unsigned int mx = 0;
            h2_absmax4(stg, mx);
            mx = h2_rowmax(mx);
            if (li == 0) atomicMax(&mxw[(it + 1) & 1], mx);
        }
        __syncthreads();           // every wave is done reading image A; the next cell's p3 maximum is complete
        h2_scale(mxw[(it + 1) & 1], S3, invS3);
    }
}

}  // namespace

// conv4's weights as two fp16 planes of S_w W: [slice 2][tap 9][plane 2][lane 64][8]; *inv_sw = 1 / S_w
size_t pack_conv4_f16x2(const float* hwio, uint16_t* dst, float* inv_sw)
{
    const size_t n = (size_t)2 * 9 * 2 * 64 * 8;
    if (!dst) return n;
    const float S = f16x2_weight_scale(hwio, (size_t)9 * CH * CH);
    if (inv_sw) *inv_sw = 1.0f / S;
    for (int s = 0; s < 2; ++s)
        for (int t = 0; t < 9; ++t)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int li = l & 15, kq = l >> 4;
                    uint16_t pl[2];
                    f16x2_split(hwio[((size_t)t * CH + 8 * kq + j) * CH + 16 * s + li], S, pl[0], pl[1]);
                    for (int p = 0; p < 2; ++p) dst[((((size_t)s * 9 + t) * 2 + p) * 64 + l) * 8 + j] = pl[p];
                }
    return n;
}

template <class K>
static hipError_t resident_grid(K kernel, int threads, int lds, int& resident)
{
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t e;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, threads, lds)) != hipSuccess) return e;
    resident = cus * (per_cu < 1 ? 1 : per_cu);
    return hipSuccess;
}

hipError_t launch_conv4_h2(const float* in, const uint16_t* wfrag, float inv_sw, const float* ep, float* out, int64_t n_cells, hipStream_t stream)
{
    if (n_cells <= 0) return hipSuccess;
    static int resident = 0;
    if (!resident) {
        hipError_t e = resident_grid(conv4_h2_kernel, 256, H2_LDS_BYTES, resident);
        if (e != hipSuccess) return e;
    }
    const long grid = n_cells < resident ? n_cells : resident;
    hipLaunchKernelGGL(conv4_h2_kernel, dim3((unsigned)grid), dim3(256), H2_LDS_BYTES, stream, in, (const f16x8*)wfrag, ep, out,
                       (long)n_cells, inv_sw);
    return hipGetLastError();
}

// conv5's folded weights as two fp16 planes: [wave = phase * 2 + half][tap][slice-in-half 2][plane 2][lane 64][8]
size_t pack_conv5_f16x2(const float* weff, uint16_t* dst, float* inv_sw)
{
    const size_t n = (size_t)8 * 4 * 2 * 2 * 64 * 8;
    if (!dst) return n;
    const float S = f16x2_weight_scale(weff, (size_t)16 * CH * C5_OUT);
    if (inv_sw) *inv_sw = 1.0f / S;
    for (int w = 0; w < 8; ++w)
        for (int t = 0; t < 4; ++t)
            for (int k = 0; k < 2; ++k)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int ph = w >> 1, sp = w & 1, li = l & 15, kq = l >> 4;
                        uint16_t pl[2];
                        f16x2_split(weff[((size_t)(ph * 4 + t) * CH + 8 * kq + j) * C5_OUT + (2 * sp + k) * 16 + li], S, pl[0], pl[1]);
                        for (int p = 0; p < 2; ++p) dst[((((((size_t)w * 4 + t) * 2 + k) * 2 + p) * 64) + l) * 8 + j] = pl[p];
                    }
    return n;
}

hipError_t launch_conv5_h2(const float* in, const uint16_t* wfrag, float inv_sw, const float* ep, float* out, int64_t n_cells, hipStream_t stream)
{
    if (n_cells <= 0) return hipSuccess;
    static int resident = 0;
    if (!resident) {
        hipError_t e = resident_grid(conv5_h2_kernel, 512, H2_LDS_BYTES, resident);
        if (e != hipSuccess) return e;
    }
    const long grid = n_cells < resident ? n_cells : resident;
    hipLaunchKernelGGL(conv5_h2_kernel, dim3((unsigned)grid), dim3(512), H2_LDS_BYTES, stream, in, (const f16x8*)wfrag, ep, out,
                       (long)n_cells, inv_sw);
    return hipGetLastError();
}

// conv4 + conv5 as one kernel: w4 = pack_conv4_f16x2's planes, w5 = pack_conv5_f16x2's; in = p3 [n][8][8][32], out = a5 [n][16][16][64]
hipError_t launch_conv45_h2(const float* in, const uint16_t* w4, float inv_sw4, const float* ep4, const uint16_t* w5, float inv_sw5,
                            const float* ep5, float* out, int64_t n_cells, hipStream_t stream)
{
    if (n_cells <= 0) return hipSuccess;
    static int resident = 0;
    if (!resident) {
        hipError_t e = hipFuncSetAttribute((const void*)conv45_h2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F45_LDS_BYTES);
        if (e != hipSuccess) return e;
        e = resident_grid(conv45_h2_kernel, 512, F45_LDS_BYTES, resident);
        if (e != hipSuccess) return e;
    }
    const long grid = n_cells < resident ? n_cells : resident;
    hipLaunchKernelGGL(conv45_h2_kernel, dim3((unsigned)grid), dim3(512), F45_LDS_BYTES, stream, in, (const f16x8*)w4, ep4, inv_sw4,
                       (const f16x8*)w5, ep5, inv_sw5, out, (long)n_cells);
    return hipGetLastError();
}

}  // namespace cs
