// conv12_fused.hip -- conv1 + conv2 of the encoder as ONE kernel: crop -> p2, p1 never leaves the CU.
//   conv1: Conv2D 1->32 on 64x64 + ReLU + BN + MaxPool   (CAE_improved_modeltrain.py:191-193)
//   conv2: Conv2D 32->64 on 32x32 + ReLU + BN + MaxPool  (:195-197)
//
// conv2 is Winograd F(4x4, 3x3) (interpolation points 0, +-1, +-2, inf): a 4x4 output tile costs 36
// multiplies per channel pair where F(2x2,3x3) needed 64 -- 4,608 16x16x4 MFMAs per cell instead of
// 8,192 (direct: 18,432).  Measured in fp32 emulation before adopting (tools/wino_error_study.py): p2 error
// 1.6e-6 of its range, features 4.3e-7 of theirs (bar 1e-5).  conv1 (2.4 % of the path's MACs) is
// computed inside the staging of conv2's input rows, so the 131 KB/cell p1 tensor (written by one kernel,
// read by the next) and one launch disappear.
//
// One 512-thread workgroup per CU (8 waves = 2 per SIMD, 152 KB of LDS) walks whole cells; a cell is four
// GROUPS of 16 tiles (two tile rows = 8 conv2 rows).  Per group, four phases, one barrier after each:
//   P1 conv1   wave (x-tile, 16-channel slice) computes the 8 new p1 rows of the group from the crop in
//              LDS: K = 9 padded to 12 = three 16x16x4 MFMAs per 16 pixels, vertical tile pair = the pool
//              window; bias -> ReLU -> BN -> max; rows go to a 10-slot ring in LDS (slot = row mod 10;
//              two rows carry over to the next group).
//   P2 V=B^TdB thread (tile, channel) transforms its 6x6 patch (scalar LDS reads, conflict-free: a half
//              wave reads 32 consecutive channels) and writes V in the A-operand order of the MFMAs.
//   P3 MFMA    wave (column group g of 3 transform-domain columns, 16-filter slice): U = G g G^T of its
//              18 points x 32 channels stays in 144 VGPRs for the life of the workgroup; per column 6 rows x
//              8 MFMAs, row fold s = A^T M in registers.  Output rows (2g, 2g+1) of s stay; the other two
//              go to the partner wave (same slice, other column group) through LDS that is dead at this
//              point (the ring's 8 consumed slots + a 16 KB area).
//   P4 Y=sA    column fold of the wave's two output rows over all six columns, bias -> ReLU -> BN -> 2x2
//              max (those two rows are one pool row), store p2.
// LDS map: V 73,728 | ring 10 x 34 x 32 x 4 = 43,520 | exchange 16,384 | crop 66 x 72 x 4 = 19,008.
#include "common.hpp"

#include <cstdlib>

namespace cs {

namespace {

constexpr int NTHR = 512;
constexpr int V_BYTES = 36 * 2048;                   // [xi][q][kq][slot][4]: 2 KB per transform point
constexpr int RING_SLOTS = 10;
constexpr int RING_ROWF = 34 * 32;                   // floats per slot: 32 interior columns + 2 halo, 32 channels
constexpr int RING_BYTES = RING_SLOTS * RING_ROWF * 4;
constexpr int E2_BYTES = 8 * 2048;
constexpr int INP_STRIDE = 72;                       // floats per crop row: interior at 4..67 (16-byte aligned), halo at 3 and 68
constexpr int INP_BYTES = 66 * INP_STRIDE * 4;
constexpr int OFF_RING = V_BYTES;
constexpr int OFF_E2 = OFF_RING + RING_BYTES;
constexpr int OFF_INP = OFF_E2 + E2_BYTES;
constexpr int LDS_BYTES = OFF_INP + INP_BYTES;
static_assert(LDS_BYTES <= 160 * 1024 && OFF_RING % 16 == 0 && OFF_E2 % 16 == 0 && OFF_INP % 16 == 0, "LDS map");

__device__ __forceinline__ float vmaxf(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float vminf(float a, float b) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

// bias -> ReLU -> BN of the pooled raw sum: the map is monotone (direction = sign of the BN scale), so
// the max over the window of the mapped values is the map of the window's max (resp. min)
__device__ __forceinline__ float pool_post(float a, float b, float c, float d, float bias, float bns, float bnt)
{
    const float mx = vmaxf(vmaxf(a, b), vmaxf(c, d)), mn = vminf(vminf(a, b), vminf(c, d));
    float v = (bns >= 0.0f ? mx : mn) + bias;
    v = fmaxf(v, 0.0f);
    return fmaf(v, bns, bnt);
}

// 1-D input transform B^T of F(4,3), points (0, 1, -1, 2, -2, inf): 12 operations
__device__ __forceinline__ void bt6(const float d[6], float o[6])
{
    o[0] = fmaf(4.0f, d[0], fmaf(-5.0f, d[2], d[4]));
    o[5] = fmaf(4.0f, d[1], fmaf(-5.0f, d[3], d[5]));
    const float t1 = fmaf(-4.0f, d[2], d[4]), t2 = fmaf(-4.0f, d[1], d[3]);
    o[1] = t1 + t2;
    o[2] = t1 - t2;
    const float t3 = d[4] - d[2], t4 = d[3] - d[1];
    o[3] = fmaf(2.0f, t4, t3);
    o[4] = fmaf(-2.0f, t4, t3);
}

__global__ __launch_bounds__(NTHR, 2) void conv12_fused_kernel(const float* __restrict__ x, const float* __restrict__ w1frag,
                                                              const float* __restrict__ ep1, const float* __restrict__ ufrag,
                                                              const float* __restrict__ ep2, float* __restrict__ p2, long n_cells)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const ring = (float*)(smem + OFF_RING);
    float* const inp = (float*)(smem + OFF_INP);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    // ---- P3 operands: U of this wave's 18 transform points, all 32 input channels, its 16 filters
    const int gcol = w & 1, sl = w >> 1;
    float U[144];
#pragma unroll
    for (int s = 0; s < 144; ++s) U[s] = ufrag[((size_t)w * 144 + s) * 64 + lane];
    const int co = sl * 16 + li;
    float bias2 = ep2[co], bns2 = ep2[64 + co], bnt2 = ep2[128 + co];
    // ---- P1 operands: conv1 weights of slice s1 (K = 9 padded to 12), tile column xt
    const int xt = w & 3, s1 = w >> 2;
    float B1[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) B1[s] = w1frag[((size_t)s1 * 3 + s) * 64 + lane];
    const int c1 = s1 * 16 + li;
    float bias1 = ep1[c1], bns1 = ep1[32 + c1], bnt1 = ep1[64 + c1];
    // touch the loop invariants here so that the compiler's wait for these loads is not inside the loop
#pragma unroll
    for (int s = 0; s < 144; ++s) asm volatile("" : "+v"(U[s]));
    asm volatile("" : "+v"(B1[0]), "+v"(B1[1]), "+v"(B1[2]), "+v"(bias1), "+v"(bns1), "+v"(bnt1), "+v"(bias2), "+v"(bns2), "+v"(bnt2));

    const long my_cells = (n_cells - blockIdx.x + gridDim.x - 1) / gridDim.x;
    if (my_cells <= 0) return;

    // conv1 A operand: lane (pixel li, k = 4 s + kq) reads tap (k / 3, k % 3); the padded taps k >= 9 carry zero
    // weights and read tap 0 (finite whenever the true taps are)
    int toff[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int k = 4 * s + kq, kk = k < 9 ? k : 0;
        toff[s] = (kk / 3) * INP_STRIDE + (kk % 3) + 16 * xt + li + 3;
    }
    // conv1 pooled outputs of this lane: ring columns 8 xt + 2 kq + {0,1} (+1: halo), channel c1
    const int pwoff = (8 * xt + 2 * kq + 1) * 32 + c1;
    // crop staging: thread -> rows (tid >> 4) and +32, 16-byte column tid & 15
    const int srow = tid >> 4, sc16 = tid & 15;
    const int soff = (srow + 1) * INP_STRIDE + 4 + 4 * sc16;
    // P2: thread (tile, channel)
    const int tile = tid >> 5, ch = tid & 31, trow = w >> 2, tx = tile & 7;
    const int roff = (4 * tx) * 32 + ch;                              // first patch column of the tile, this channel
    const int vq = ch >> 4, vkq = (ch >> 2) & 3, vj = ch & 3;
    const int voff = (((vq * 4 + vkq) * 16 + ((tile + 2 * (2 * vq + (vkq >> 1))) & 15)) * 4 + vj) * 4;   // bytes
    // P3: A operand slots of lane (tile li, channel quad kq) for q = 0, 1
    const int aoff0 = ((0 * 4 + kq) * 16 + ((li + 2 * (0 + (kq >> 1))) & 15)) * 16;
    const int aoff1 = ((1 * 4 + kq) * 16 + ((li + 2 * (2 + (kq >> 1))) & 15)) * 16;

    // ---- LDS: zero everything once (halo columns / rows of the crop and the ring are never written again)
    for (int i = tid; i < LDS_BYTES / 16; i += NTHR) ((f32x4*)smem)[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    __syncthreads();
    {
        const float* src = x + (size_t)blockIdx.x * 4096 + srow * 64 + 4 * sc16;
        *(f32x4*)(inp + soff) = *(const f32x4*)src;
        *(f32x4*)(inp + soff + 32 * INP_STRIDE) = *(const f32x4*)(src + 32 * 64);
    }
    __syncthreads();

    for (long ci = 0; ci < my_cells; ++ci) {
        const long cell = blockIdx.x + ci * gridDim.x;
        const bool has_next = ci + 1 < my_cells;
        for (int g = 0; g < 4; ++g) {
            // ================= P1: the group's new p1 rows (ring positions q; p1 row y = q - 1; q = 0, 33: zero rows)
            f32x4 stg0 = {0.0f, 0.0f, 0.0f, 0.0f}, stg1 = stg0;
            if (g == 3 && has_next) {   // next cell's crop: in flight during this phase, written to LDS in P2
                const float* src = x + (size_t)(cell + gridDim.x) * 4096 + srow * 64 + 4 * sc16;
                stg0 = *(const f32x4*)src;
                stg1 = *(const f32x4*)(src + 32 * 64);
            }
            const int q0 = g == 0 ? 0 : 8 * g + 2, nq = g == 0 ? 10 : 8;
            for (int i = 0; i < nq; ++i) {
                const int q = q0 + i;
                const int slot = q % RING_SLOTS;
                float* const row = ring + slot * RING_ROWF;
                if (q == 0 || q == 33) {
                    if (tid < 256) *(f32x4*)(row + 32 + 4 * tid) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    continue;
                }
                const float* a = inp + (2 * (q - 1)) * INP_STRIDE;
                f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const float a0 = a[toff[s]];
                    const float a1 = a[toff[s] + INP_STRIDE];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, B1[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, B1[s], acc1, 0, 0, 0);
                }
                row[pwoff] = pool_post(acc0[0], acc0[1], acc1[0], acc1[1], bias1, bns1, bnt1);
                row[pwoff + 32] = pool_post(acc0[2], acc0[3], acc1[2], acc1[3], bias1, bns1, bnt1);
            }
            __syncthreads();

            // ================= P2: V = B^T d B of this thread's (tile, channel) patch
            {
                float d[6][6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int slot = (8 * g + 4 * trow + i) % RING_SLOTS;           // wave-uniform
                    const float* r = ring + slot * RING_ROWF + roff;
#pragma unroll
                    for (int j = 0; j < 6; ++j) d[i][j] = r[j * 32];
                }
                if (g == 3 && has_next) {   // the crop buffer is dead since the barrier above
                    *(f32x4*)(inp + soff) = stg0;
                    *(f32x4*)(inp + soff + 32 * INP_STRIDE) = stg1;
                }
                // rows first (B^T d), then columns ((B^T d) B): V[r][c]
                float t[6][6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    float col[6], o[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) col[i] = d[i][j];
                    bt6(col, o);
#pragma unroll
                    for (int i = 0; i < 6; ++i) t[i][j] = o[i];
                }
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    float o[6];
                    bt6(t[r], o);
#pragma unroll
                    for (int c = 0; c < 6; ++c) *(float*)(smem + (r * 6 + c) * 2048 + voff) = o[c];
                }
            }
            __syncthreads();

            // ================= P3: M = V U on the matrix pipe, row fold in registers
            f32x4 own[3][2];
            {
                const int e1slot = (8 * g + w) % RING_SLOTS;                        // a consumed ring slot: this wave's exchange area
                char* const e1 = (char*)(ring + e1slot * RING_ROWF + 32) + lane * 16;
                char* const e2 = smem + OFF_E2 + w * 2048 + lane * 16;
                auto mrow = [&](int cc, int r) -> f32x4 {
                    const int xi = r * 6 + 3 * gcol + cc;
                    const f32x4 a0 = *(const f32x4*)(smem + xi * 2048 + aoff0);
                    const f32x4 a1 = *(const f32x4*)(smem + xi * 2048 + aoff1);
                    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], U[(cc * 6 + r) * 8 + j], acc, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], U[(cc * 6 + r) * 8 + 4 + j], acc, 0, 0, 0);
                    return acc;
                };
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    // s = A^T m:  s0 = m0 + (m1+m2) + (m3+m4), s1 = (m1-m2) + 2 (m3-m4), s2 = (m1+m2) + 4 (m3+m4),
                    //             s3 = (m1-m2) + 8 (m3-m4) + m5
                    const f32x4 m1 = mrow(cc, 1), m2 = mrow(cc, 2);
                    const f32x4 p = m1 + m2, mq = m1 - m2;
                    const f32x4 m3 = mrow(cc, 3), m4 = mrow(cc, 4);
                    const f32x4 u = m3 + m4, v = m3 - m4;
                    f32x4 sa, sb, ta, tb;   // (sa, sb): rows this wave keeps; (ta, tb): rows of the partner
                    if (gcol == 0) {
                        const f32x4 m0 = mrow(cc, 0);
                        sa = m0 + p + u;
                        sb = mq + 2.0f * v;
                        ta = p + 4.0f * u;
                        const f32x4 m5 = mrow(cc, 5);
                        tb = mq + 8.0f * v + m5;
                    } else {
                        const f32x4 m5 = mrow(cc, 5);
                        sa = p + 4.0f * u;
                        sb = mq + 8.0f * v + m5;
                        const f32x4 m0 = mrow(cc, 0);
                        ta = m0 + p + u;
                        tb = mq + 2.0f * v;
                    }
                    own[cc][0] = sa;
                    own[cc][1] = sb;
                    if (cc < 2) {
                        *(f32x4*)(e1 + (2 * cc) * 1024) = ta;
                        *(f32x4*)(e1 + (2 * cc + 1) * 1024) = tb;
                    } else {
                        *(f32x4*)(e2) = ta;
                        *(f32x4*)(e2 + 1024) = tb;
                    }
                }
            }
            __syncthreads();

            // ================= P4: column fold of this wave's two output rows, epilogue, store
            {
                const int wp = w ^ 1;
                const int pslot = (8 * g + wp) % RING_SLOTS;
                const char* const e1 = (const char*)(ring + pslot * RING_ROWF + 32) + lane * 16;
                const char* const e2 = smem + OFF_E2 + wp * 2048 + lane * 16;
                f32x4 y[2][4];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x4 s[6];
                    const f32x4 pa = *(const f32x4*)(e1 + (0 + i) * 1024), pb = *(const f32x4*)(e1 + (2 + i) * 1024),
                                pc = *(const f32x4*)(e2 + i * 1024);
                    if (gcol == 0) { s[0] = own[0][i]; s[1] = own[1][i]; s[2] = own[2][i]; s[3] = pa; s[4] = pb; s[5] = pc; }
                    else           { s[0] = pa; s[1] = pb; s[2] = pc; s[3] = own[0][i]; s[4] = own[1][i]; s[5] = own[2][i]; }
                    const f32x4 p = s[1] + s[2], mq = s[1] - s[2], u = s[3] + s[4], v = s[3] - s[4];
                    y[i][0] = s[0] + p + u;
                    y[i][1] = mq + 2.0f * v;
                    y[i][2] = p + 4.0f * u;
                    y[i][3] = mq + 8.0f * v + s[5];
                }
                // register r <-> tile 4 kq + r of the group; rows (2 gcol, 2 gcol + 1) of the tile = pool row gcol
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = 4 * kq + r;
                    const int py = 2 * (2 * g + (t >> 3)) + gcol, px = 2 * (t & 7);
                    float* o = p2 + (((size_t)cell * 16 + py) * 16 + px) * 64 + co;
                    o[0] = pool_post(y[0][0][r], y[0][1][r], y[1][0][r], y[1][1][r], bias2, bns2, bnt2);
                    o[64] = pool_post(y[0][2][r], y[0][3][r], y[1][2][r], y[1][3][r], bias2, bns2, bnt2);
                }
            }
            __syncthreads();   // the exchange area inside the ring is consumed before the next P1 refills those slots
        }
    }
}

}  // namespace

// U = G g G^T of F(4x4,3x3) per (cin, cout), evaluated in double and rounded once.
// ufrag[wave w = 2 sl + gcol][(cc * 6 + r) * 8 + 4 q + j][lane] = U[row r][col 3 gcol + cc][ci = 16 q + 4 kq + j][co = 16 sl + li]
size_t pack_conv12_fragments(const float* hwio /* [3][3][32][64] */, float* dst)
{
    const size_t total = (size_t)8 * 144 * 64;
    if (!dst) return total;
    static const double G[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    for (int w = 0; w < 8; ++w) {
        const int gcol = w & 1, sl = w >> 1;
        for (int cc = 0; cc < 3; ++cc)
            for (int r = 0; r < 6; ++r)
                for (int kk = 0; kk < 8; ++kk)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int li = lane & 15, kq = lane >> 4, q = kk >> 2, j = kk & 3;
                        const int ci = 16 * q + 4 * kq + j, co = 16 * sl + li, c = 3 * gcol + cc;
                        double u = 0.0;
                        for (int a = 0; a < 3; ++a)
                            for (int b = 0; b < 3; ++b)
                                u += G[r][a] * (double)hwio[((size_t)(a * 3 + b) * 32 + ci) * 64 + co] * G[c][b];
                        dst[((size_t)w * 144 + (cc * 6 + r) * 8 + kk) * 64 + lane] = (float)u;
                    }
    }
    return total;
}

hipError_t launch_conv12_fused(const float* x, const float* w1frag, const float* ep1, const float* ufrag, const float* ep2, float* p2,
                               int64_t n_cells, hipStream_t stream)
{
    static int cus = 0;
    if (!cus) {
        hipError_t e = hipFuncSetAttribute((const void*)conv12_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        int dev = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        if (cus < 1) cus = 1;
    }
    if (n_cells <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(n_cells < cus ? n_cells : cus);      // one workgroup per CU (LDS-bound), persistent over cells
    hipLaunchKernelGGL(conv12_fused_kernel, dim3(grid), dim3(NTHR), LDS_BYTES, stream, x, w1frag, ep1, ufrag, ep2, p2, (long)n_cells);
    return hipGetLastError();
}

}  // namespace cs
