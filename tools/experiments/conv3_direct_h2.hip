// EXPERIMENT, not built into libcellscreen.so (round 4; DESIGN.md 6b has the numbers): this kernel passed the -m gpu parity and
// determinism tests as cs_screen's conv3 and ran in 21.0 - 21.5 ms per 1 M cells against 24.4 - 25.0 for the Winograd kernel it
// would replace -- and the 1 M-cell step did not get shorter (179.6 / 178.5 / 186.7 ms against 180.7 / 177.7 / 182.8 in alternating
// runs on one box): the kernels that follow it in the chunk loop ran 3 - 6 % slower (conv4+5 +1.0 ms, conv6+7 +2.0, conv1+2 +1.1).
// To build it: add it to build.py's SOURCES, declare launch_conv3_h2 / pack_conv3_h2 in common.hpp, pack with the layer's BN scales
// and call it where api.hip calls launch_conv3_wino_h2.
//
// conv3 of the encoder, Conv2D 64 -> 32 on the 16x16 grid + ReLU + BatchNormalization + MaxPooling2D -> the 8x8x32
// `encoded` tensor (CAE_improved_modeltrain.py:199-201), with the fp32 contraction as a TWO-term fp16 split on
// v_mfma_f32_16x16x32_f16 (CS_PRECISION_SPLIT16; conv_wino_up.hip, conv67_h2_kernel, has the algebra).
//
// DIRECT form, whole cells.  The Winograd F(2x2,3x3) kernel this replaces (conv_wino_cs.hip, round 3) ran 768 matrix instructions
// per cell behind 8.1 k vector instructions -- every transformed value has to be split into its two fp16 terms again -- and
// staged 16-tile strips with their halo rows (98 KB read per cell for a 65.5 KB tensor).  Here p2 is split ONCE per element while
// the whole cell is staged (no halo re-read: 65.5 KB per cell), the nine taps are address offsets, and the price is 1,728 matrix
// instructions of 16 cycles per cell -- which is less than the vector work they replace, because on this chip the two do not
// overlap.
//
//   staged cell   [row 18][pixel 18][hi: 64 ch fp16 | lo | 32 B] with a zero halo; 288 B per pixel and 64 B behind every row put
//                 every lane group of a ds_read_b128 on 16 distinct 16-byte slots (tools/lds_bank_model.py) -- 94,464 B, one
//                 workgroup of 8 waves per CU;
//   tile          16 pixels = 2 rows x 8 columns = four 2x2 pool windows; A row i of the MFMA is pixel (row (i >> 1) & 1, column
//                 2 (i >> 2) + (i & 1)), so register r of lane (filter li, window kq) is pixel r of window kq: the max-pool is a
//                 max over the lane's four registers;
//   wave          (16-filter slice, four tiles): the slice's weights -- 9 taps x 2 channel blocks x 2 planes x 4 VGPRs = 144 --
//                 stay in registers for the life of the workgroup; two tiles at a time (four independent accumulation chains);
//   scale         one per cell: S puts max|p2| of the cell into [2^14, 2^15) (taken while the PREVIOUS cell's MFMAs run, from the
//                 prefetched registers), the weights carry the layer's S_w from the host, 1 / (S S_w) sits in the epilogue's fma;
//   pooling       filters with a negative BatchNormalization scale carry NEGATED weights (pack_conv3_h2), so the window's min is
//                 -max(-z) and one path serves both signs (as conv12_fused.hip's pool_post).
#include "common.hpp"

#include <cstring>
#include <vector>

namespace cs {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct C3 {
    static constexpr int G = 16, CIN = 64, COUT = 32;
    static constexpr int PLB = 2 * CIN;                        // bytes of one plane of a pixel
    static constexpr int PXB = 2 * PLB + 32;                   // 288
    static constexpr int ROWB = (G + 2) * PXB + 64;            // 5,248
    static constexpr int CELL = (G + 2) * ROWB;                // 94,464
    static constexpr int OFF_MAX = CELL;                       // two words: cell maxima, alternating
    static constexpr int LDS = CELL + 16;
    static constexpr int THREADS = 512, NLD = G * G * CIN / 4 / THREADS;      // 8 16-byte loads per thread and cell
    static_assert(LDS <= 160 * 1024 && NLD == 8, "layout");
};

__device__ __forceinline__ unsigned int c3_rowmax(unsigned int m)
{
    unsigned int o;
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xF, 0xF, true);  m = m > o ? m : o;     // quad_perm [1,0,3,2]
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xF, 0xF, true);  m = m > o ? m : o;     // quad_perm [2,3,0,1]
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0x124, 0xF, 0xF, true); m = m > o ? m : o;     // row_ror:4
    o = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xF, 0xF, true); m = m > o ? m : o;     // row_ror:8
    return m;
}
// S = the power of two that puts a maximum with float bits mbits into [2^14, 2^15), and 1 / S (exponents clamped: both stay normal)
__device__ __forceinline__ void c3_scale(unsigned int mbits, float& S, float& invS)
{
    int E = (int)((mbits >> 23) & 0xffu);
    E = E < 40 ? 40 : (E > 254 ? 254 : E);
    S = __builtin_bit_cast(float, (unsigned int)(268 - E) << 23);
    invS = __builtin_bit_cast(float, (unsigned int)(E - 14) << 23);
}
// (scalars first: __builtin_bit_cast of a vector ELEMENT expression reads element 0 whatever the index -- clang 19, ROCm 7.2)
__device__ __forceinline__ void c3_absmax4(const f32x4& v, unsigned int& mx)
{
    const float a = v[0], b = v[1], c = v[2], d = v[3];
    const unsigned int u = __builtin_bit_cast(unsigned int, fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d))));
    mx = mx > u ? mx : u;
}
__device__ __forceinline__ void c3_split_store(char* dst, const f32x4& x, float S)
{
    const f32x4 v = x * S;
    const f16x4 hi = __builtin_convertvector(v, f16x4);
    const f32x4 r = v - __builtin_convertvector(hi, f32x4);            // exact in fp32
    *(f16x4*)dst = hi;
    *(f16x4*)(dst + C3::PLB) = __builtin_convertvector(r, f16x4);
}

__global__ __launch_bounds__(C3::THREADS, 2) void conv3_h2_kernel(const float* __restrict__ in /* p2 [n][16][16][64] */,
                                                                  const f16x8* __restrict__ wfrag, const float* __restrict__ ep /* [3][32] */,
                                                                  float* __restrict__ out /* [n][8][8][32] */, long n_cells, float inv_sw)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned int* const mxw = (unsigned int*)(smem + C3::OFF_MAX);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = wave & 1;                                    // filters 16 slice .. +15
    const int tg = wave >> 1;                                      // tile rows 2 tg, 2 tg + 1 (pixel rows 4 tg .. 4 tg + 3), both column halves
    const int li = lane & 15, kq = lane >> 4;

    f16x8 B[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int p = 0; p < 2; ++p) B[t][k][p] = wfrag[((((size_t)slice * 9 + t) * 2 + k) * 2 + p) * 64 + lane];
    const int co = slice * 16 + li;
    const float bias = ep[co], bns = ep[C3::COUT + co], bnt = ep[2 * C3::COUT + co];
    const float sgn = bns >= 0.0f ? 1.0f : -1.0f;

    // zero everything once: the interior is rewritten for every cell, the halo stays zero
    for (int i = tid; i < C3::LDS / 16; i += C3::THREADS) *(f32x4*)(smem + i * 16) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    long cell = blockIdx.x;
    if (cell >= n_cells) return;
    f32x4 stg[C3::NLD];
#pragma unroll
    for (int k = 0; k < C3::NLD; ++k) stg[k] = *(const f32x4*)(in + (size_t)cell * (C3::G * C3::G * C3::CIN) + (size_t)(tid + C3::THREADS * k) * 4);
    __syncthreads();
    {
        unsigned int mx = 0;
#pragma unroll
        for (int k = 0; k < C3::NLD; ++k) c3_absmax4(stg[k], mx);
        mx = c3_rowmax(mx);
        if (li == 0) atomicMax(&mxw[0], mx);
    }
    __syncthreads();
    float S, invS;
    c3_scale(mxw[0], S, invS);

    // element tid + 512 k of the cell: pixel (2 k + (tid >> 8), (tid >> 4) & 15), channels 4 (tid & 15) .. +3
    const int woff = ((tid >> 8) + 1) * C3::ROWB + (((tid >> 4) & 15) + 1) * C3::PXB + (tid & 15) * 8;
    // A fragment of (tile row ty, column half tx, tap (dy, dx), channel block kb, plane p): this lane's pixel + 8 channels at 8 kq
    const int abase = ((li >> 1) & 1) * C3::ROWB + (2 * (li >> 2) + (li & 1)) * C3::PXB + kq * 16;

    for (int it = 0; cell < n_cells; cell += gridDim.x, ++it) {
        const float unscale = sgn * invS * inv_sw;
#pragma unroll
        for (int k = 0; k < C3::NLD; ++k) c3_split_store(smem + woff + 2 * k * C3::ROWB, stg[k], S);
        __syncthreads();
        if (tid == 0) mxw[it & 1] = 0;      // this cell's word: read by everyone before the barrier above, next used two cells on
        const long ncell = cell + gridDim.x;
        if (ncell < n_cells) {              // in flight during the MFMA phase
#pragma unroll
            for (int k = 0; k < C3::NLD; ++k) stg[k] = *(const f32x4*)(in + (size_t)ncell * (C3::G * C3::G * C3::CIN) + (size_t)(tid + C3::THREADS * k) * 4);
        }

#pragma unroll
        for (int half = 0; half < 2; ++half) {                     // tile row 2 tg + half, its two column halves as two chains each
            const int ty = 2 * tg + half;
            f32x4 hi[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
            f32x4 lo[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
            const char* const arow = smem + abase + (2 * ty) * C3::ROWB;
            // 36 fragments (tap, channel block, column half), each two ds_read_b128 (hi, lo plane) and three MFMAs; the reads run two
            // fragments ahead of the MFMAs that use them (pinned with sched_group_barrier: left alone, the compiler issues a fragment's
            // reads right in front of its MFMAs and waits for the LDS 79 times per cell)
            auto rd = [&](int f, f16x8 (&fr)[2]) {
                const int tap = f >> 2, kb = (f >> 1) & 1, tx = f & 1;
                const char* q = arow + (tap / 3) * C3::ROWB + (tap % 3) * C3::PXB + tx * (8 * C3::PXB) + kb * 64;
                fr[0] = *(const f16x8*)q;
                fr[1] = *(const f16x8*)(q + C3::PLB);
            };
            f16x8 a0[2], a1[2];
            rd(0, a0);
            rd(1, a1);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int f = 0; f < 36; ++f) {
                const int tap = f >> 2, kb = (f >> 1) & 1, tx = f & 1;
                f16x8 an[2] = {a1[0], a1[1]};
                if (f + 2 < 36) rd(f + 2, an);
                lo[tx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[0], B[tap][kb][1], lo[tx], 0, 0, 0);
                hi[tx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[0], B[tap][kb][0], hi[tx], 0, 0, 0);
                lo[tx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[1], B[tap][kb][0], lo[tx], 0, 0, 0);
                a0[0] = a1[0]; a0[1] = a1[1];
                a1[0] = an[0]; a1[1] = an[1];
                if (f + 2 < 36) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
            // register r = pixel r of pool window kq of the tile: max, then bias -> ReLU -> BN on the one value that survives
#pragma unroll
            for (int tx = 0; tx < 2; ++tx) {
                const f32x4 z = hi[tx] + lo[tx];
                const float m = fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3]));
                const float v = fmaxf(fmaf(m, unscale, bias), 0.0f);
                out[(((size_t)cell * (C3::G / 2) + ty) * (C3::G / 2) + 4 * tx + kq) * C3::COUT + co] = fmaf(v, bns, bnt);
            }
        }
        if (ncell < n_cells) {
            unsigned int mx = 0;
#pragma unroll
            for (int k = 0; k < C3::NLD; ++k) c3_absmax4(stg[k], mx);
            mx = c3_rowmax(mx);
            if (li == 0) atomicMax(&mxw[(it + 1) & 1], mx);
        }
        __syncthreads();           // every wave is done reading the cell; the next cell's maximum is complete
        c3_scale(mxw[(it + 1) & 1], S, invS);
    }
}

}  // namespace

// conv3's weights [3][3][64][32] as two fp16 planes of S_w w (S_w: max|w| into [2^14, 2^15)), negated for filters whose
// BatchNormalization scale is negative (the kernel pools -z for those).  dst[slice 2][tap 9][channel block 2][plane 2][lane 64][8]:
// lane (filter li, kq) holds channels 32 kb + 8 kq .. +7 of tap t for filter 16 slice + li.  Returns 16-bit words.
size_t pack_conv3_h2(const float* hwio, const float* bn_scale, uint16_t* dst, float* inv_sw)
{
    const size_t n = (size_t)2 * 9 * 2 * 2 * 64 * 8;
    if (!dst) return n;
    const float S = f16x2_weight_scale(hwio, (size_t)9 * C3::CIN * C3::COUT);
    if (inv_sw) *inv_sw = 1.0f / S;
    for (int s = 0; s < 2; ++s)
        for (int t = 0; t < 9; ++t)
            for (int kb = 0; kb < 2; ++kb)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int li = l & 15, kq = l >> 4, co = 16 * s + li;
                        float w = hwio[((size_t)t * C3::CIN + 32 * kb + 8 * kq + j) * C3::COUT + co];
                        if (bn_scale[co] < 0.0f) w = -w;
                        uint16_t pl[2];
                        f16x2_split(w, S, pl[0], pl[1]);
                        for (int p = 0; p < 2; ++p) dst[((((((size_t)s * 9 + t) * 2 + kb) * 2 + p) * 64) + l) * 8 + j] = pl[p];
                    }
    return n;
}

hipError_t launch_conv3_h2(const float* in, const uint16_t* wplanes, float inv_sw, const float* ep, float* out, int64_t n_cells,
                           hipStream_t stream)
{
    static int resident = 0;
    if (!resident) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3_h2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C3::LDS);
        if (e != hipSuccess) return e;
        int dev = 0, cus = 0, per_cu = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv3_h2_kernel, C3::THREADS, C3::LDS)) != hipSuccess) return e;
        resident = cus * (per_cu < 1 ? 1 : per_cu);
    }
    if (n_cells <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(n_cells < resident ? n_cells : resident);
    hipLaunchKernelGGL(conv3_h2_kernel, dim3(grid), dim3(C3::THREADS), C3::LDS, stream, in, (const f16x8*)wplanes, ep, out, (long)n_cells, inv_sw);
    return hipGetLastError();
}

}  // namespace cs
