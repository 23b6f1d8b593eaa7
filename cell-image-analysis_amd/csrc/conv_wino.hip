// conv_wino.hip -- conv2 of the reference autoencoder (Conv2D 32->64, 3x3 same, relu; BatchNorm;
// MaxPooling2D 2x2 -- CAE_improved_modeltrain.py:195-197) as a Winograd F(2x2, 3x3) convolution
// on the exact-fp32 MFMA: 16 multiplies per 2x2 output tile and channel pair instead of 36, i.e.
// 4/9 of the multiply-adds of the direct form (conv2 is 45 % of the path's direct MACs).
//
//   V = B^T d B      input transform of each 4x4 patch d (stride 2), VALU, into LDS
//   M_xi = V_xi U_xi  16 independent [tiles x cin] x [cin x cout] products, xi = 0..15   (MFMA)
//   Y = A^T M A      output transform, 2x2 outputs per tile, VALU in the MFMA epilogue
// with U = G g G^T pre-computed on the host (in double, rounded once).  The 2x2 output tile IS
// the max-pool window, so bias -> ReLU -> BN -> max collapse to one value per tile and lane.
// F(2,3) has transform entries 0, +-1, +-1/2 only; measured fp32 error on this layer's data is
// 4.5e-7 of the output range against 2.8e-7 for the direct fp32 form (tolerance 1e-5).
//
// Work split: a workgroup = 4 waves = the 4 16-channel output slices; a work item = one cell x 4
// conv rows = 2 rows of 16 tiles.  Each wave keeps its slice of U (16 xi x 8 K steps = 128 VGPRs)
// resident and accumulates all 16 xi of a 16-tile row in registers (16 accumulators), so the
// output transform needs no cross-lane traffic.
#include "common.hpp"

namespace cs {

namespace {

constexpr int WN_H = 32, WN_W = 32, WN_CIN = 32, WN_COUT = 64;
constexpr int WN_SR = 4;                                   // conv rows per item (2 tile rows)
constexpr int WN_R = WN_SR + 2, WN_WP = WN_W + 2;          // staged rows / cols incl. halo
constexpr int WN_PS = WN_CIN + 8;                          // padded pixel stride (floats), conflict-free b128 reads
constexpr int WN_STRIP = WN_R * WN_WP * WN_PS * 4;         // 32,640 B
constexpr int WN_TILES = WN_W / 2;                         // 16 tiles per tile row
constexpr int WN_VS = WN_CIN + 8;                          // padded tile stride of V (floats)
constexpr int WN_V = 16 * WN_TILES * WN_VS * 4;            // 40,960 B
constexpr int WN_LDS = WN_STRIP + WN_V;
constexpr int WN_NSTRIP = WN_H / WN_SR;                    // 8 items per cell
constexpr int WN_NB = 16 * 8;                              // B registers: 16 xi x (cin / 4) K steps

__global__ __launch_bounds__(256, 2) void conv2_wino_kernel(
    const float* __restrict__ in /* p1 [n][32][32][32] */, const float* __restrict__ ufrag,
    const float* __restrict__ ep /* [3][64] bias, bn scale, bn shift */, float* __restrict__ out /* p2 [n][16][16][64] */,
    long n_cells)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* strip = (float*)smem;
    float* V = (float*)(smem + WN_STRIP);
    const int tid = threadIdx.x, lane = tid & 63;
    const int nsl = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = output-channel slice
    const int li = lane & 15, kq = lane >> 4;

    float B[WN_NB];
#pragma unroll
    for (int s = 0; s < WN_NB; ++s) B[s] = ufrag[((size_t)nsl * WN_NB + s) * 64 + lane];
    const int co = nsl * 16 + li;
    const float bias = ep[co], bns = ep[WN_COUT + co], bnt = ep[2 * WN_COUT + co];

    // input-transform role of this thread: tile, channel quad, half of the xi rows
    const int t_tile = tid & 15, t_cq = (tid >> 4) & 7, t_half = tid >> 7;

    // strip staging, software pipelined: item i+1's loads are issued before item i's second
    // tile row and written to LDS when item i+1 begins (see conv_mfma.hip STAGE_PF)
    constexpr int C4 = WN_CIN / 4, TOT = WN_R * WN_WP * C4, NLD = (TOT + 255) / 256;
    f32x4 stg[NLD];
    auto issue = [&](long item) {
        const long cell = item / WN_NSTRIP;
        const int y0 = (int)(item % WN_NSTRIP) * WN_SR;
        const float* src = in + (size_t)cell * WN_H * WN_W * WN_CIN;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tid + 256 * k;
            const int pix = idx / C4, c4 = idx % C4;
            const int r = pix / WN_WP, c = pix % WN_WP;
            const int sy = y0 - 1 + r, sx = c - 1;
            stg[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (idx < TOT && sy >= 0 && sy < WN_H && sx >= 0 && sx < WN_W)
                stg[k] = *(const f32x4*)(src + ((size_t)sy * WN_W + sx) * WN_CIN + c4 * 4);
        }
    };
    const long total = n_cells * WN_NSTRIP;
    if ((long)blockIdx.x < total) issue(blockIdx.x);
    for (long item = blockIdx.x; item < total; item += gridDim.x) {
        const long cell = item / WN_NSTRIP;
        const int y0 = (int)(item % WN_NSTRIP) * WN_SR;
        // ---- conv rows y0-1 .. y0+4 of p1 with a zero halo: registers -> LDS ------------------
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tid + 256 * k;
            if (idx < TOT) {
                const int pix = idx / C4, c4 = idx % C4;
                *(f32x4*)(strip + pix * WN_PS + c4 * 4) = stg[k];
            }
        }
        __syncthreads();

        for (int tr = 0; tr < WN_SR / 2; ++tr) {
            // ---- input transform V = B^T d B of the 16 tiles of tile row tr ------------------
            // tile (tr, t_tile): patch rows 2tr .. 2tr+3, cols 2t .. 2t+3 of the staged strip
            {
                const float* d0 = strip + ((2 * tr) * WN_WP + 2 * t_tile) * WN_PS + t_cq * 4;
                // this thread makes xi rows {0,1} (t_half = 0: needs patch rows 0,1,2) or {2,3} (rows 1,2,3)
                f32x4 tA[4], tB[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 r1 = *(const f32x4*)(d0 + (1 * WN_WP + c) * WN_PS);
                    const f32x4 r2 = *(const f32x4*)(d0 + (2 * WN_WP + c) * WN_PS);
                    if (t_half == 0) {
                        const f32x4 r0 = *(const f32x4*)(d0 + (0 * WN_WP + c) * WN_PS);
                        tA[c] = r0 - r2;      // B^T row 0
                        tB[c] = r1 + r2;      // B^T row 1
                    } else {
                        const f32x4 r3 = *(const f32x4*)(d0 + (3 * WN_WP + c) * WN_PS);
                        tA[c] = r2 - r1;      // B^T row 2
                        tB[c] = r1 - r3;      // B^T row 3
                    }
                }
                float* vo = V + (size_t)t_tile * WN_VS + t_cq * 4;
                const int xa = (2 * t_half) * 4, xb = (2 * t_half + 1) * 4;   // first xi of the two rows
                auto put = [&](int xi, const f32x4& v) { *(f32x4*)(vo + (size_t)xi * WN_TILES * WN_VS) = v; };
                put(xa + 0, tA[0] - tA[2]); put(xa + 1, tA[1] + tA[2]); put(xa + 2, tA[2] - tA[1]); put(xa + 3, tA[1] - tA[3]);
                put(xb + 0, tB[0] - tB[2]); put(xb + 1, tB[1] + tB[2]); put(xb + 2, tB[2] - tB[1]); put(xb + 3, tB[1] - tB[3]);
            }
            __syncthreads();
            if (tr == WN_SR / 2 - 1 && item + gridDim.x < total) issue(item + gridDim.x);   // in flight during the last GEMM

            // ---- 16 products M_xi[tile][cout] = V_xi[tile][cin] U_xi[cin][cout] -----------------
            f32x4 acc[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) acc[xi] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            const float* va = V + (size_t)li * WN_VS + kq * 4;
#pragma unroll
            for (int xp = 0; xp < 8; ++xp) {   // two xi at a time: two independent accumulation chains
                const int x0 = 2 * xp, x1 = 2 * xp + 1;
                const f32x4 a00 = *(const f32x4*)(va + (size_t)x0 * WN_TILES * WN_VS);
                const f32x4 a01 = *(const f32x4*)(va + (size_t)x0 * WN_TILES * WN_VS + 16);
                const f32x4 a10 = *(const f32x4*)(va + (size_t)x1 * WN_TILES * WN_VS);
                const f32x4 a11 = *(const f32x4*)(va + (size_t)x1 * WN_TILES * WN_VS + 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[x0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a00[j], B[x0 * 8 + j], acc[x0], 0, 0, 0);
                    acc[x1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a10[j], B[x1 * 8 + j], acc[x1], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[x0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a01[j], B[x0 * 8 + 4 + j], acc[x0], 0, 0, 0);
                    acc[x1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a11[j], B[x1 * 8 + 4 + j], acc[x1], 0, 0, 0);
                }
            }

            // ---- output transform Y = A^T M A, bias -> relu -> BN -> 2x2 max, per tile ---------
            // D layout: lane holds cout li, register r <-> tile 4*kq + r
            float res[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s0[4], s1[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    s0[c] = (acc[0 * 4 + c][r] + acc[1 * 4 + c][r]) + acc[2 * 4 + c][r];
                    s1[c] = (acc[1 * 4 + c][r] - acc[2 * 4 + c][r]) - acc[3 * 4 + c][r];
                }
                const float y00 = (s0[0] + s0[1]) + s0[2], y01 = (s0[1] - s0[2]) - s0[3];
                const float y10 = (s1[0] + s1[1]) + s1[2], y11 = (s1[1] - s1[2]) - s1[3];
                auto post = [&](float v) { v += bias; v = fmaxf(v, 0.0f); return fmaf(v, bns, bnt); };
                res[r] = fmaxf(fmaxf(post(y00), post(y01)), fmaxf(post(y10), post(y11)));
            }
            const int ty = (y0 >> 1) + tr;
            float* o = out + (((size_t)cell * (WN_H / 2) + ty) * (WN_W / 2) + 4 * kq) * WN_COUT + co;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(size_t)r * WN_COUT] = res[r];
            __syncthreads();   // V is rewritten by the next tile row / the strip by the next item
        }
    }
}

}  // namespace

// U = G g G^T per (cin, cout), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], evaluated in double and
// rounded once; B fragments [slice][xi][q*4 + j][lane] with ci = 16 q + 4 kq + j.
size_t pack_wino_fragments(const float* hwio /* [3][3][32][64] */, float* dst)
{
    const size_t total = (size_t)(WN_COUT / 16) * WN_NB * 64;
    if (!dst) return total;
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int nsl = 0; nsl < WN_COUT / 16; ++nsl)
        for (int xi = 0; xi < 16; ++xi)
            for (int s = 0; s < 8; ++s)
                for (int lane = 0; lane < 64; ++lane) {
                    const int li = lane & 15, kq = lane >> 4, q = s >> 2, j = s & 3;
                    const int ci = 16 * q + 4 * kq + j, co = nsl * 16 + li;
                    const int a = xi >> 2, b = xi & 3;   // U[a][b] = sum_{u,v} G[a][u] g[u][v] G[b][v]
                    double u = 0.0;
                    for (int uu = 0; uu < 3; ++uu)
                        for (int vv = 0; vv < 3; ++vv)
                            u += G[a][uu] * (double)hwio[((size_t)(uu * 3 + vv) * WN_CIN + ci) * WN_COUT + co] * G[b][vv];
                    dst[((size_t)nsl * WN_NB + xi * 8 + s) * 64 + lane] = (float)u;
                }
    return total;
}

hipError_t launch_conv2_wino(const float* in, const float* ufrag, const float* ep, float* out, int64_t n_cells,
                             hipStream_t stream)
{
    static int resident = 0;
    if (!resident) {
        hipError_t e = hipFuncSetAttribute((const void*)conv2_wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WN_LDS);
        if (e != hipSuccess) return e;
        int dev = 0, cus = 0, per_cu = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv2_wino_kernel, 256, WN_LDS);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        resident = cus * per_cu;
    }
    const long total = (long)n_cells * WN_NSTRIP;
    if (total <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(total < resident ? total : resident);
    hipLaunchKernelGGL(conv2_wino_kernel, dim3(grid), dim3(256), WN_LDS, stream, in, ufrag, ep, out, (long)n_cells);
    return hipGetLastError();
}

}  // namespace cs
