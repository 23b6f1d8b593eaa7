// conv_wino.hip -- conv2 of the reference autoencoder (Conv2D 32->64, 3x3 same, relu; BatchNorm;
// MaxPooling2D 2x2 -- CAE_improved_modeltrain.py:195-197) as a Winograd F(2x2, 3x3) convolution
// on the exact-fp32 MFMA: 16 multiplies per 2x2 output tile and channel pair instead of 36, i.e.
// 4/9 of the multiply-adds of the direct form (conv2 is 45 % of the path's direct MACs).
//
//   V = B^T d B      input transform of each 4x4 patch d (stride 2), VALU, in registers
//   M_xi = V_xi U_xi  16 independent [tiles x cin] x [cin x cout] products, xi = 0..15   (MFMA)
//   Y = A^T M A      output transform, 2x2 outputs per tile, VALU in the MFMA epilogue
// with U = G g G^T pre-computed on the host (in double, rounded once).  The 2x2 output tile IS
// the max-pool window, so bias -> ReLU -> BN -> max collapse to one value per tile and lane.
// F(2,3) has transform entries 0, +-1, +-1/2 only; measured fp32 error on this layer's data is
// 4.5e-7 of the output range against 2.8e-7 for the direct fp32 form (tolerance 1e-5).
//
// Work split: a workgroup = 4 waves = the 4 16-channel output slices; a work item = one cell x 4
// conv rows = 2 rows of 16 tiles.  Each wave keeps its slice of U (16 xi x 8 K steps = 128 VGPRs)
// resident and walks V one column at a time (4 accumulators), folding each column into the
// first half of the output transform, so neither transform needs cross-lane traffic.
#include "common.hpp"

#include <cstdlib>

namespace cs {

namespace {

constexpr int WN_H = 32, WN_W = 32, WN_CIN = 32, WN_COUT = 64;
constexpr int WN_SR = 4;                                   // conv rows per item (2 tile rows)
constexpr int WN_NTR = WN_SR / 2;                          // tile rows per item
constexpr int WN_R = WN_SR + 2, WN_WP = WN_W + 2;          // staged rows / cols incl. halo
// padded pixel stride (floats): tiles step 2 pixels, so an ODD number of 16-B slots per pixel makes
// the 16 tiles x 2 channel quads of a ds_read_b128 lane group land on 16 distinct slots
constexpr int WN_PS = WN_CIN + 4;
constexpr int WN_STRIP = WN_R * WN_WP * WN_PS * 4;         // 29,376 B, double buffered
constexpr int WN_LDS = 2 * WN_STRIP;
constexpr int WN_NSTRIP = WN_H / WN_SR;                    // 8 items per cell
constexpr int WN_NB = 16 * 8;                              // B registers: 16 xi x (cin / 4) K steps
constexpr int WN_C4 = WN_CIN / 4, WN_TOT = WN_R * WN_WP * WN_C4;
constexpr int WN_NLD = (WN_TOT + 255) / 256;               // per-thread 16-B loads per strip (7)
constexpr int WN_LPP = (WN_NLD + WN_NTR - 1) / WN_NTR;     // per tile-row iteration (4)

__device__ __forceinline__ f32x4 wn_load(const float* __restrict__ in, long cell, int y0, int idx)
{
    const float* src = in + (size_t)cell * WN_H * WN_W * WN_CIN;
    const int pix = idx / WN_C4, c4 = idx % WN_C4;
    const int r = pix / WN_WP, c = pix % WN_WP;
    const int sy = y0 - 1 + r, sx = c - 1;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (idx < WN_TOT && sy >= 0 && sy < WN_H && sx >= 0 && sx < WN_W)
        v = *(const f32x4*)(src + ((size_t)sy * WN_W + sx) * WN_CIN + c4 * 4);
    return v;
}
__device__ __forceinline__ void wn_store(float* strip, int idx, const f32x4& v)
{
    if (idx < WN_TOT) *(f32x4*)(strip + (idx / WN_C4) * WN_PS + (idx % WN_C4) * 4) = v;
}

// Every wave derives its own A operands: lane (tile, kq) reads the 4x4 patch of ITS tile for ITS
// four channels and applies B^T d B in registers, one output column of V at a time, feeding the
// MFMAs directly.  The four output-slice waves repeat the same transform (the VALU is otherwise
// idle) but no transformed tensor goes through LDS and no barrier separates transform and product:
// one barrier per item (strip double buffer) is all that is left.
// DIAG: a diagnostic build that stamps s_memtime around the phases of every tile row and adds the
// differences up per wave (never used for results or timing; `diag` = [blocks][4 waves][4] cycles:
// prepare (LDS reads + input transform), MFMA issue, fold/epilogue/stores, barrier wait).
__device__ __forceinline__ unsigned long long wn_stamp()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

template <bool DIAG>
__global__ __launch_bounds__(256, 2) void conv2_wino_kernel(
    const float* __restrict__ in /* p1 [n][32][32][32] */, const float* __restrict__ ufrag,
    const float* __restrict__ ep /* [3][64] bias, bn scale, bn shift */, float* __restrict__ out /* p2 [n][16][16][64] */,
    long n_cells, unsigned long long* __restrict__ diag)
{
    unsigned long long d_prep = 0, d_mfma = 0, d_epi = 0, d_bar = 0, d_t = 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int nsl = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = output-channel slice
    const int li = lane & 15, kq = lane >> 4;

    float B[WN_NB];
#pragma unroll
    for (int s = 0; s < WN_NB; ++s) B[s] = ufrag[((size_t)nsl * WN_NB + s) * 64 + lane];
    const int co = nsl * 16 + li;
    const float bias = ep[co], bns = ep[WN_COUT + co], bnt = ep[2 * WN_COUT + co];

    const long total = n_cells * WN_NSTRIP;
    const long first = blockIdx.x;
    if (first >= total) return;
#pragma unroll 4
    for (int idx = tid; idx < WN_TOT; idx += 256)
        wn_store((float*)smem, idx, wn_load(in, first / WN_NSTRIP, (int)(first % WN_NSTRIP) * WN_SR, idx));
    __syncthreads();

    int buf = 0;
    for (long item = first; item < total; item += gridDim.x) {
        const long cell = item / WN_NSTRIP;
        const int y0 = (int)(item % WN_NSTRIP) * WN_SR;
        const long nitem = item + gridDim.x;
        const bool has_next = nitem < total;
        const long ncell = nitem / WN_NSTRIP;
        const int ny0 = (int)(nitem % WN_NSTRIP) * WN_SR;
        const float* strip = (const float*)(smem + buf * WN_STRIP);
        float* nstrip = (float*)(smem + (buf ^ 1) * WN_STRIP);

        for (int tr = 0; tr < WN_NTR; ++tr) {
            f32x4 stg[WN_LPP];
            if (has_next) {   // this iteration's slice of the next strip: loads now, LDS writes at the bottom
#pragma unroll
                for (int j = 0; j < WN_LPP; ++j) stg[j] = wn_load(in, ncell, ny0, tid + 256 * (tr * WN_LPP + j));
            }
            // patch of tile li: strip rows 2tr .. 2tr+3, cols 2li .. 2li+3; this lane's channels 16q + 4kq ..
            const float* d0 = strip + ((2 * tr) * WN_WP + 2 * li) * WN_PS + 4 * kq;
            float s0[4][4], s1[4][4];   // [column c of M][tile register r]
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // V[:, c] = B^T (d B[:, c]):  d B[:, c] combines patch columns (ca, cb) with sign sg
                constexpr int CA[4] = {0, 1, 2, 1}, CB[4] = {2, 2, 1, 3};
                constexpr float SG[4] = {-1.0f, 1.0f, -1.0f, -1.0f};
                f32x4 acc[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    // keep the scheduler from hoisting every step's patch loads to the top (register spills)
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (DIAG) { d_t = wn_stamp(); __builtin_amdgcn_sched_barrier(0); }
                    f32x4 w[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 da = *(const f32x4*)(d0 + (i * WN_WP + CA[c]) * WN_PS + 16 * q);
                        const f32x4 db = *(const f32x4*)(d0 + (i * WN_WP + CB[c]) * WN_PS + 16 * q);
                        w[i] = da + SG[c] * db;
                    }
                    const f32x4 v[4] = {w[0] - w[2], w[1] + w[2], w[2] - w[1], w[1] - w[3]};   // xi = 4 r + c
                    if constexpr (DIAG) {
                        asm volatile("" ::"v"(v[0][0]), "v"(v[1][1]), "v"(v[2][2]), "v"(v[3][3]));
                        __builtin_amdgcn_sched_barrier(0);
                        const unsigned long long t = wn_stamp();
                        d_prep += t - d_t; d_t = t;
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[r][j], B[(4 * r + c) * 8 + 4 * q + j], acc[r], 0, 0, 0);
                    if constexpr (DIAG) {
                        __builtin_amdgcn_sched_barrier(0);
                        const unsigned long long t = wn_stamp();
                        d_mfma += t - d_t; d_t = t;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // first half of the output transform: s = A^T M  (rows of M live in acc[0..3])
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s0[c][r] = (acc[0][r] + acc[1][r]) + acc[2][r];
                    s1[c][r] = (acc[1][r] - acc[2][r]) - acc[3][r];
                }
            }
            // second half Y = s A, then bias -> relu -> BN -> 2x2 max per tile (register r <-> tile 4 kq + r)
            float res[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y00 = (s0[0][r] + s0[1][r]) + s0[2][r], y01 = (s0[1][r] - s0[2][r]) - s0[3][r];
                const float y10 = (s1[0][r] + s1[1][r]) + s1[2][r], y11 = (s1[1][r] - s1[2][r]) - s1[3][r];
                auto post = [&](float v) { v += bias; v = fmaxf(v, 0.0f); return fmaf(v, bns, bnt); };
                res[r] = fmaxf(fmaxf(post(y00), post(y01)), fmaxf(post(y10), post(y11)));
            }
            const int ty = (y0 >> 1) + tr;
            float* o = out + (((size_t)cell * (WN_H / 2) + ty) * (WN_W / 2) + 4 * kq) * WN_COUT + co;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(size_t)r * WN_COUT] = res[r];
            if (has_next) {
#pragma unroll
                for (int j = 0; j < WN_LPP; ++j) wn_store(nstrip, tid + 256 * (tr * WN_LPP + j), stg[j]);
            }
            if constexpr (DIAG) {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long t = wn_stamp();
                d_epi += t - d_t; d_t = t;
            }
        }
        __syncthreads();   // this strip fully read; the next strip complete in the other buffer
        if constexpr (DIAG) { const unsigned long long t = wn_stamp(); d_bar += t - d_t; d_t = t; }
        buf ^= 1;
    }
    if constexpr (DIAG) {
        if (lane == 0) {
            unsigned long long* o = diag + ((size_t)blockIdx.x * 4 + nsl) * 4;
            o[0] = d_prep; o[1] = d_mfma; o[2] = d_epi; o[3] = d_bar;
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

static unsigned long long* g_diag = nullptr;
static int g_diag_blocks = 0;

hipError_t launch_conv2_wino(const float* in, const float* ufrag, const float* ep, float* out, int64_t n_cells,
                             hipStream_t stream)
{
    static int resident = 0;
    static const bool diag = getenv("CS_WINO_DIAG") != nullptr;
    if (!resident) {
        hipError_t e = hipFuncSetAttribute((const void*)conv2_wino_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WN_LDS);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)conv2_wino_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WN_LDS);
        if (e != hipSuccess) return e;
        int dev = 0, cus = 0, per_cu = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv2_wino_kernel<false>, 256, WN_LDS);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        resident = cus * per_cu;
        if (diag) {
            if ((e = hipMalloc(&g_diag, (size_t)resident * 16 * sizeof(unsigned long long))) != hipSuccess) return e;
            g_diag_blocks = resident;
        }
    }
    const long total = (long)n_cells * WN_NSTRIP;
    if (total <= 0) return hipSuccess;
    const unsigned grid = (unsigned)(total < resident ? total : resident);
    if (diag)
        hipLaunchKernelGGL(conv2_wino_kernel<true>, dim3(grid), dim3(256), WN_LDS, stream, in, ufrag, ep, out, (long)n_cells, g_diag);
    else
        hipLaunchKernelGGL(conv2_wino_kernel<false>, dim3(grid), dim3(256), WN_LDS, stream, in, ufrag, ep, out, (long)n_cells,
                           (unsigned long long*)nullptr);
    return hipGetLastError();
}

}  // namespace cs

// Diagnostic only (CS_WINO_DIAG=1): per-wave phase cycles of the LAST conv2 launch, averaged over waves.
extern "C" int cs_debug_wino_diag(double out4[4])
{
    using namespace cs;
    if (!g_diag) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    const size_t n = (size_t)g_diag_blocks * 16;
    unsigned long long* h = new unsigned long long[n];
    if (hipMemcpy(h, g_diag, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) { delete[] h; return -3; }
    for (int k = 0; k < 4; ++k) out4[k] = 0.0;
    for (size_t i = 0; i < n; ++i) out4[i & 3] += (double)h[i];
    for (int k = 0; k < 4; ++k) out4[k] /= (double)(n / 4);
    delete[] h;
    return 0;
}
