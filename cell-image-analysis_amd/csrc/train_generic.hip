// train_generic.hip -- one `autoencoder.fit` step (CAE_improved_modeltrain.py:223-227, 286-293) for ANY instance of the
// reference's layer grammar (create_improved_autoencoder(input_shape) is generic, :184; BASELINE.json configs[4]: 128x128
// crops, filters 32-64-128 | 128-64-32-1), on run-time-shaped kernels:
//   forward        conv_generic.hip in GEN_EPI_RELU mode (bias -> ReLU at full resolution), the BatchNormalization /
//                  max-pool kernels of train.hip (they take N, H, W, C at run time), sigmoid conv + error sums
//   backward-data  the same generic conv with the kernel flipped and its channel roles swapped (GEN_EPI_PLAIN);
//                  through an UpSampling2D the adjoint is a 2x2 sum
//   weight grads   an MFMA GEMM per tap: dW[tap][ci][co] = sum over pixels of in[pixel + tap][ci] * dz[pixel][co]
//                  (M = 16 input channels, N = 16 filters, K = pixels), per-workgroup partials in a fixed order
//   reduction, Adam: train.hip (flat parameter vector, no float atomics: bit-reproducible run to run)
// The reference graph never takes this path (train_api.hip runs it on the tuned kernels); this one is shape-free and
// correct first -- the 128x128 variant's gradient all-reduce (same flat-gradient split as the reference graph:
// cs_train_forward_backward -> all-reduce -> cs_train_apply) is what BASELINE.json configs[4] needs from it.
#include "train_internal.hpp"

#include <cstdlib>

#include <vector>

using namespace cs;

namespace cs {
namespace {

__global__ void flip_transpose_kernel(const float* __restrict__ w, int cin, int cout, float* __restrict__ dst)
{
    const long total = 9L * cin * cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout);
        const int ci = (int)((i / cout) % cin);
        const int tap = (int)(i / ((long)cout * cin));
        dst[((long)(8 - tap) * cout + co) * cin + ci] = w[i];       // (2-dy, 2-dx) = tap 8 - tap; channel roles swapped
    }
}

__global__ void sumpool2x2_kernel(const float* __restrict__ in, float* __restrict__ out, long n, int H, int W, int C)
{
    const int Ho = H / 2, Wo = W / 2;
    const long total = n * Ho * Wo * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        const int x = (int)(p % Wo), y = (int)((p / Wo) % Ho);
        const long cell = p / ((long)Wo * Ho);
        const float* s = in + (((cell * H + 2 * y) * W + 2 * x) * C + c);
        out[i] = (s[0] + s[C]) + (s[(long)W * C] + s[(long)W * C + C]);
    }
}

// Weight gradient of one conv: part[p][tap][ci][co] for the pixels of part p.  A wave owns one 16 x 16 (ci, co) tile
// for all nine taps (nine accumulators): per 4 consecutive pixels of a conv row one B load (dz) feeds nine MFMAs.
struct WgArgs {
    const float* xin;    // stored input [n][Hs][Ws][cin]  (Hs = H/2 when ups)
    const float* dz;     // [n][H][W][cout]
    float* part;         // [nparts][9][cin][cout]
    long n;
    int H, W, cin, cout, ups, tm, tn, nparts;
};

// NCO = 16-filter tiles per wave.  A workgroup (4 waves = 4 (input-channel tile, filter-tile group) units of ONE part) walks the
// part's conv rows; per row it stages the three activation rows in conv-grid coordinates (halo and UpSampling2D resolved while
// staging: 16-byte global loads along the channels) and the dz row in LDS, then every wave runs W / 4 pixel groups x 9 taps x NCO
// MFMAs from LDS (lane = (channel, pixel): 4-byte reads, 16 consecutive channels per pixel; the pixel stride is padded to
// 16 (mod 32) floats so the two pixels of a half wave fall on different banks).
// History: the first version gave a wave one filter tile, loaded every operand with scalar global loads and ran three 64-bit
// divisions per 4-pixel group -- 61 % of the 128 x 128 variant's training step at 8 % of the matrix peak; walking whole rows took
// the step from 4.68 to 3.26 ms, the LDS staging below to 2.8 ms.
__device__ __forceinline__ int wg_pad(int c) { return c + ((48 - (c & 31)) & 31); }      // smallest c' >= c with c' = 16 (mod 32)

template <int NCO>
__global__ __launch_bounds__(256) void wgrad_generic_kernel(WgArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int tng = (g.tn + NCO - 1) / NCO;                 // filter-tile groups
    const int units = g.tm * tng, wg_per_part = (units + 3) / 4;
    const int p = blockIdx.x / wg_per_part, unit = (blockIdx.x % wg_per_part) * 4 + wave;
    const bool active = unit < units;                       // idle waves still stage and take the barriers
    const int cib = active ? unit / tng : 0, cog = active ? unit % tng : 0;
    const int ci = cib * 16 + li;
    const int H = g.H, W = g.W, cin = g.cin, cout = g.cout, Ws = g.ups ? W / 2 : W, Hs = g.ups ? H / 2 : H;
    const int xs_ = wg_pad(cin), zs_ = wg_pad(cout);        // staged pixel strides (floats)
    float* const Xs = wsm;                                  // [3][W + 2][xs_]
    float* const Zs = wsm + 3 * (W + 2) * xs_;              // [W][zs_]
    const long rows = g.n * H;
    const long r0 = (rows * p) / g.nparts, r1 = (rows * (p + 1)) / g.nparts;
    int co[NCO];
    bool cok[NCO];
#pragma unroll
    for (int j = 0; j < NCO; ++j) { co[j] = (cog * NCO + j) * 16 + li; cok[j] = active && co[j] < cout; }
    const bool ciok = active && ci < cin;
    f32x4 acc[NCO][9];
#pragma unroll
    for (int j = 0; j < NCO; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[j][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (long row = r0; row < r1; ++row) {
        const long cell = row / H;
        const int y = (int)(row - cell * H);
        __syncthreads();                                    // the previous row's MFMAs are done with the staged rows
        // activation rows y - 1 .. y + 1 at conv-grid columns -1 .. W live in a three-slot ring (row yy in slot (yy + 3) % 3):
        // inside a cell a conv row brings ONE new row (y + 1); the part's first row and a cell's first row stage all three
        const int d0 = (row == r0 || y == 0) ? 0 : 2, nd = 3 - d0;
        if ((cin & 3) == 0) {
            const int c4n = cin >> 2, per_row = (W + 2) * c4n;
            for (int e = tid; e < nd * per_row; e += 256) {
                const int dd = e / per_row, d = d0 + dd, rem = e - dd * per_row, xx = rem / c4n - 1, c4 = rem - (xx + 1) * c4n;
                const int yy = y + d - 1;
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (yy >= 0 && yy < H && xx >= 0 && xx < W)
                    v = *(const f32x4*)(g.xin + ((cell * Hs + (g.ups ? yy >> 1 : yy)) * Ws + (g.ups ? xx >> 1 : xx)) * cin + 4 * c4);
                *(f32x4*)(Xs + (((yy + 3) % 3) * (W + 2) + xx + 1) * xs_ + 4 * c4) = v;
            }
        } else {
            const int per_row = (W + 2) * cin;
            for (int e = tid; e < nd * per_row; e += 256) {
                const int dd = e / per_row, d = d0 + dd, rem = e - dd * per_row, xx = rem / cin - 1, c = rem - (xx + 1) * cin;
                const int yy = y + d - 1;
                float v = 0.0f;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = g.xin[((cell * Hs + (g.ups ? yy >> 1 : yy)) * Ws + (g.ups ? xx >> 1 : xx)) * cin + c];
                Xs[(((yy + 3) % 3) * (W + 2) + xx + 1) * xs_ + c] = v;
            }
        }
        const float* dzr = g.dz + ((cell * H + y) * W) * cout;
        if ((cout & 3) == 0) {
            const int c4n = cout >> 2;
            for (int e = tid; e < W * c4n; e += 256) {
                const int x = e / c4n, c4 = e - x * c4n;
                *(f32x4*)(Zs + x * zs_ + 4 * c4) = *(const f32x4*)(dzr + (size_t)x * cout + 4 * c4);
            }
        } else {
            for (int e = tid; e < W * cout; e += 256) {
                const int x = e / cout, c = e - x * cout;
                Zs[x * zs_ + c] = dzr[e];
            }
        }
        __syncthreads();
        if (active) {
            const float* xa = Xs + kq * xs_ + (ciok ? ci : 0);       // tap (d, dx) of pixel x: row d, staged column x + dx
            const float* zb = Zs + kq * zs_;
            const int rslot[3] = {(y + 2) % 3, y % 3, (y + 1) % 3};       // ring slots of rows y - 1, y, y + 1
            for (int x0 = 0; x0 < W; x0 += 4) {
                float b[NCO], a[9];
#pragma unroll
                for (int j = 0; j < NCO; ++j) b[j] = cok[j] ? zb[x0 * zs_ + co[j]] : 0.0f;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float v = xa[(rslot[t / 3] * (W + 2) + x0 + t % 3) * xs_];
                    a[t] = ciok ? v : 0.0f;
                }
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int j = 0; j < NCO; ++j) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[j], acc[j][t], 0, 0, 0);
            }
        }
    }
    // D: lane = filter li of the tile, registers = input channels 4 kq .. 4 kq + 3
    float* o = g.part + (size_t)p * 9 * cin * cout;
#pragma unroll
    for (int j = 0; j < NCO; ++j) {
        if (!cok[j]) continue;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c2 = cib * 16 + 4 * kq + r;
                if (c2 < cin) o[((size_t)t * cin + c2) * cout + co[j]] = acc[j][t][r];
            }
    }
}

}  // namespace

hipError_t launch_flip_transpose(const float* hwio, int cin, int cout, float* dst, hipStream_t s)
{
    const long total = 9L * cin * cout;
    hipLaunchKernelGGL(flip_transpose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, hwio, cin, cout, dst);
    return hipGetLastError();
}

hipError_t launch_sumpool2x2(const float* in, float* out, int64_t n, int H, int W, int C, hipStream_t s)
{
    const long total = (long)n * (H / 2) * (W / 2) * C;
    long grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(sumpool2x2_kernel, dim3((unsigned)grid), dim3(256), 0, s, in, out, (long)n, H, W, C);
    return hipGetLastError();
}

// Layers with a small kernel (the first and the last conv: 9 x 1 x 32 and 9 x 32 x 1 weights) have one or two (channel tile, filter
// tile) units, i.e. one or two busy waves per part: they get four times the parts (their partial sums are 1 KB each)
static int wgrad_generic_part_cap(int cin, int cout, int max_parts) { return 9L * cin * cout <= 16384 ? 4 * max_parts : max_parts; }

// filter tiles per wave and workgroups per part for a layer (shared by the launcher and the partial-sum buffer sizing)
static void wgrad_generic_shape(int cin, int cout, int* nco, int* wg_per_part)
{
    const int tm = (cin + 15) / 16, tn = (cout + 15) / 16;
    // two filter tiles per wave: 160 VGPRs (three waves per SIMD); four (300 VGPRs, one wave per SIMD) measured 3 % slower, one equal
    *nco = tn >= 2 ? 2 : 1;
    *wg_per_part = (tm * ((tn + *nco - 1) / *nco) + 3) / 4;
}

// bytes of LDS the weight-gradient kernel stages for a layer (three input rows + one dz row, padded pixel strides)
size_t wgrad_generic_lds_bytes(int W, int cin, int cout)
{
    auto pad = [](int c) { return c + ((48 - (c & 31)) & 31); };
    return ((size_t)3 * (W + 2) * pad(cin) + (size_t)W * pad(cout)) * sizeof(float);
}

hipError_t launch_wgrad_generic(const float* xin, const float* dz, float* part, int64_t n, int H, int W, int cin, int cout, int ups,
                                int max_parts, int* nparts, hipStream_t s)
{
    WgArgs g;
    g.xin = xin; g.dz = dz; g.part = part; g.n = n; g.H = H; g.W = W; g.cin = cin; g.cout = cout; g.ups = ups;
    g.tm = (cin + 15) / 16; g.tn = (cout + 15) / 16;
    int nco, wg_per_part;
    wgrad_generic_shape(cin, cout, &nco, &wg_per_part);
    const long rows = (long)n * H;
    long np = 2048 / wg_per_part;
    max_parts = wgrad_generic_part_cap(cin, cout, max_parts);
    if (np > max_parts) np = max_parts;
    if (np > rows) np = rows;
    if (np < 1) np = 1;
    g.nparts = (int)np;
    *nparts = g.nparts;
    const dim3 grid((unsigned)(g.nparts * wg_per_part));
    auto pad = [](int c) { return c + ((48 - (c & 31)) & 31); };
    const size_t lds = ((size_t)3 * (W + 2) * pad(cin) + (size_t)W * pad(cout)) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e;
#define WG_GO(N)                                                                                                                   \
    do {                                                                                                                           \
        e = hipFuncSetAttribute((const void*)wgrad_generic_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
        if (e == hipSuccess) hipLaunchKernelGGL(wgrad_generic_kernel<N>, grid, dim3(256), lds, s, g);                              \
    } while (0)
    if (nco == 4) WG_GO(4);
    else if (nco == 2) WG_GO(2);
    else WG_GO(1);
#undef WG_GO
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace cs

// ------------------------------------------------------------------------------------------------ orchestration
static int gen_parts(const cs_trainer* t, int l)
{
    int nco, wg_per_part;
    cs::wgrad_generic_shape(t->cin(l), t->ch[l], &nco, &wg_per_part);
    long np = 2048 / wg_per_part;
    const int cap = cs::wgrad_generic_part_cap(t->cin(l), t->ch[l], TRAIN_MAX_PARTS);
    if (np > cap) np = cap;
    return np < 1 ? 1 : (int)np;
}

int gen_train_setup(cs_trainer* t)
{
    int rc;
    int cmax = 1;
    for (int l = 0; l < t->n_conv; ++l) cmax = t->ch[l] > cmax ? t->ch[l] : cmax;
    for (int l = 0; l < t->n_conv - 1; ++l) {
        if (256 % t->ch[l]) return fail(CS_ERR_UNSUPPORTED, "training: conv %d has %d filters; the BatchNormalization kernels need a divisor of 256", l, t->ch[l]);
        if ((rc = t->ep_inf[l].ensure(3 * (size_t)t->ch[l] * 4)) || (rc = t->stats[l].ensure(2 * (size_t)t->ch[l] * 4))) return rc;
    }
    for (int l = 1; l < t->n_conv; ++l)
        if ((rc = t->wft[l].ensure(9 * (size_t)t->cin(l) * t->ch[l] * 4))) return rc;
    if ((rc = t->part_stats.ensure((size_t)BN_MAX_PARTS * 3 * cmax * 4)) || (rc = t->part_bwd.ensure((size_t)BN_MAX_PARTS * 2 * cmax * 4)) ||
        (rc = t->bwd_sums.ensure(2 * (size_t)cmax * 4)))
        return rc;
    for (int l = 0; l < t->n_conv; ++l) {
        if ((rc = t->dzsum_part[l].ensure((size_t)BN_MAX_PARTS * cmax * 4))) return rc;
        if ((rc = t->wpart[l].ensure((size_t)gen_parts(t, l) * 9 * t->cin(l) * t->ch[l] * 4))) return rc;
    }
    if ((rc = t->descs.ensure(2 * TR_MAXL * sizeof(ReduceDesc))) || (rc = t->scal.ensure(16))) return rc;
    // convs that run on bf16 MFMAs (conv_generic_x3.hip: the shape has a plan, cin 32 / 64 / 128, not upsample-fed): forward with
    // the kernel as it is, backward-data with the flipped kernel (channel roles swapped, never upsample-fed: the 2x2 sum follows)
    for (int l = 0; l < t->n_conv; ++l) {
        const int cin = t->cin(l), C = t->ch[l];
        t->x3f[l] = l < t->n_conv - 1 && l <= t->n_enc && conv_generic_x3_takes(t->gh[l], t->gw[l], cin, C, 0);
        t->x3t[l] = l > 0 && conv_generic_x3_takes(t->gh[l], t->gw[l], C, cin, 0);
        if (t->x3f[l] && (rc = t->wx3f[l].ensure(pack_generic_bf16x3(9, cin, C, nullptr, nullptr) * 2))) return rc;
        if (t->x3t[l] && (rc = t->wx3t[l].ensure(pack_generic_bf16x3(9, C, cin, nullptr, nullptr) * 2))) return rc;
    }
    return gen_train_repack(t);
}

int gen_train_repack(cs_trainer* t)
{
    float* P = t->P.as<float>();
    for (int l = 1; l < t->n_conv; ++l)
        LCHK(launch_flip_transpose(P + t->off_k[l], t->cin(l), t->ch[l], t->wft[l].as<float>(), t->stream));
    for (int l = 0; l < t->n_conv; ++l) {
        if (t->x3f[l]) LCHK(launch_pack_generic_bf16x3(P + t->off_k[l], 9, t->cin(l), t->ch[l], t->wx3f[l].as<uint16_t>(), t->stream));
        if (t->x3t[l]) LCHK(launch_pack_generic_bf16x3(t->wft[l].as<float>(), 9, t->ch[l], t->cin(l), t->wx3t[l].as<uint16_t>(), t->stream));
    }
    return CS_OK;
}

int gen_train_ensure_batch(cs_trainer* t, int64_t b)
{
    if (b <= t->maxb) return CS_OK;
    for (int l = 0; l < t->n_conv - 1; ++l)      // the BatchNormalization / pooling kernels index with 32-bit shifts and masks
        if ((double)b * (double)t->rfl[l] >= 2147483648.0)
            return fail(CS_ERR_UNSUPPORTED, "a batch of %lld cells makes conv %d's tensor %g elements: the BatchNormalization kernels index below 2^31",
                        (long long)b, l, (double)b * (double)t->rfl[l]);
    int rc;
    const size_t npix = (size_t)t->H * t->W;
    if ((rc = t->x.ensure(b * npix * 4)) || (rc = t->y.ensure(b * npix * 4))) return rc;
    size_t dup = 0;
    const int last = t->n_conv - 1;
    for (int l = 0; l < last; ++l) {
        if ((rc = t->r[l].ensure(b * t->rfl[l] * 4)) || (rc = t->a[l].ensure(b * t->afl[l] * 4)) || (rc = t->da[l].ensure(b * t->afl[l] * 4)) ||
            (rc = t->dz[l].ensure(b * t->rfl[l] * 4)))
            return rc;
    }
    for (int l = t->n_enc + 1; l < t->n_conv; ++l) {          // upsample-fed convs: gradient wrt the upsampled input, before the 2x2 sum
        const size_t f = (size_t)t->gh[l] * t->gw[l] * t->cin(l);
        dup = f > dup ? f : dup;
    }
    if ((rc = t->dup.ensure(b * dup * 4))) return rc;
    if ((rc = t->dz[last].ensure(b * npix * 4)) || (rc = t->out.ensure(b * npix * 4)) || (rc = t->errpart.ensure((size_t)b * 8 * 4))) return rc;
    t->maxb = b;
    return CS_OK;
}

static int gen_copy_in(cs_trainer* t, DevBuf& dst, const float* src, int kind, size_t floats)
{
    HIPCHK(hipMemcpyAsync(dst.p, src, floats * 4, kind == CS_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, t->stream));
    return CS_OK;
}

// Forward + backward of one batch as stream work only (no host synchronisation once the reduction descriptors of this batch
// size are uploaded).  The weight gradient of a layer needs only that layer's dz and input: it runs on a second stream beside the
// backward-data conv and the BatchNormalization-backward kernels of the layers below (a third of the step's kernel time at
// batch 32 of the 128 x 128 variant: profiles/r04_*_train_variant_trace.txt); both streams join before the partial sums are reduced.
int gen_train_fb_enqueue(cs_trainer* t, const float* x, const float* y, int64_t batch, int kind)
{
    int rc = gen_train_ensure_batch(t, batch);
    if (rc) return rc;
    const int64_t B = batch;
    hipStream_t s = t->stream;
    float* P = t->P.as<float>();
    float* G = t->G;
    float* MOV = t->MOV.as<float>();
    const int last = t->n_conv - 1;
    const size_t npix = (size_t)t->H * t->W;
    if ((rc = gen_copy_in(t, t->x, x, kind, B * npix)) || (rc = gen_copy_in(t, t->y, y, kind, B * npix))) return rc;

    // ---- forward, BatchNormalization in training mode
    for (int l = 0; l < last; ++l) {
        const int C = t->ch[l], pool = l < t->n_enc;
        const float* in = l == 0 ? t->x.as<float>() : t->a[l - 1].as<float>();
        if (t->x3f[l])
            LCHK(launch_conv_generic_x3(in, t->wx3f[l].as<uint16_t>(), P + t->off_b[l], t->r[l].as<float>(), B, t->gh[l], t->gw[l], t->cin(l), C,
                                        0, GEN_EPI_RELU, s));
        else
            LCHK(launch_conv_generic(in, P + t->off_k[l], P + t->off_b[l], t->r[l].as<float>(), B, t->gh[l], t->gw[l], t->cin(l), C,
                                     l > t->n_enc, GEN_EPI_RELU, s));
        int G1 = 0;
        LCHK(launch_bn_stats(t->r[l].as<float>(), (long)B * t->gh[l] * t->gw[l], C, t->part_stats.as<float>(), &G1, s));
        LCHK(launch_bn_stats_final(t->part_stats.as<float>(), G1, C, t->cfg.bn_eps, t->cfg.bn_momentum, MOV + t->off_mm[l],
                                   MOV + t->off_mv[l], t->stats[l].as<float>(), s));
        LCHK(launch_bn_apply(t->r[l].as<float>(), C, P + t->off_g[l], P + t->off_be[l], t->stats[l].as<float>(), t->a[l].as<float>(), B,
                             t->gh[l], t->gw[l], pool, s));
    }
    LCHK(launch_conv_generic(t->a[last - 1].as<float>(), P + t->off_k[last], P + t->off_b[last], t->out.as<float>(), B, t->gh[last],
                             t->gw[last], t->cin(last), 1, last > t->n_enc, GEN_EPI_SIGMOID, s));
    LCHK(launch_recon_err(t->out.as<float>(), t->y.as<float>(), B, (int)npix, t->errpart.as<float>(), s));
    // ---- backward
    LCHK(launch_loss_dz(t->out.as<float>(), t->y.as<float>(), (long)B * npix, t->dz[last].as<float>(), t->dzsum_part[last].as<float>(),
                        &t->np_b[last], s, t->errpart.as<float>(), B * 4, t->scal.as<float>()));
    if (!t->stream2) {
        HIPCHK(hipStreamCreateWithFlags(&t->stream2, hipStreamNonBlocking));
        for (int l = 0; l < TR_MAXL; ++l) HIPCHK(hipEventCreateWithFlags(&t->ev_dz[l], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&t->ev_wg, hipEventDisableTiming));
    }
    hipStream_t s2 = t->stream2;
    for (int l = last; l >= 0; --l) {
        const int C = t->ch[l], pool = l < t->n_enc, ups = l > t->n_enc;
        if (l < last) {
            int G2 = 0;
            LCHK(launch_bn_bwd_reduce(t->da[l].as<float>(), t->r[l].as<float>(), t->stats[l].as<float>(), P + t->off_g[l], P + t->off_be[l], B,
                                      t->gh[l], t->gw[l], C, pool, t->part_bwd.as<float>(), &G2, s));
            LCHK(launch_bn_bwd_final(t->part_bwd.as<float>(), G2, C, (double)B * t->gh[l] * t->gw[l], t->bwd_sums.as<float>(), G + t->off_g[l],
                                     G + t->off_be[l], s));
            LCHK(launch_bn_bwd_dz(t->da[l].as<float>(), t->r[l].as<float>(), t->stats[l].as<float>(), P + t->off_g[l], P + t->off_be[l],
                                  t->bwd_sums.as<float>(), B, t->gh[l], t->gw[l], C, pool, t->dz[l].as<float>(), t->dzsum_part[l].as<float>(),
                                  &t->np_b[l], s));
        }
        const float* in = l == 0 ? t->x.as<float>() : t->a[l - 1].as<float>();
        HIPCHK(hipEventRecord(t->ev_dz[l], s));
        HIPCHK(hipStreamWaitEvent(s2, t->ev_dz[l], 0));
        LCHK(launch_wgrad_generic(in, t->dz[l].as<float>(), t->wpart[l].as<float>(), B, t->gh[l], t->gw[l], t->cin(l), C, ups, TRAIN_MAX_PARTS,
                                  &t->np_w[l], s2));
        if (l > 0) {   // dL/d(input of conv l): conv of dz with the flipped kernel, channel roles swapped
            float* dst = ups ? t->dup.as<float>() : t->da[l - 1].as<float>();
            if (t->x3t[l])
                LCHK(launch_conv_generic_x3(t->dz[l].as<float>(), t->wx3t[l].as<uint16_t>(), nullptr, dst, B, t->gh[l], t->gw[l], C, t->cin(l), 0,
                                            GEN_EPI_PLAIN, s));
            else
                LCHK(launch_conv_generic(t->dz[l].as<float>(), t->wft[l].as<float>(), nullptr, dst, B, t->gh[l], t->gw[l], C, t->cin(l), 0,
                                         GEN_EPI_PLAIN, s));
            if (ups) LCHK(launch_sumpool2x2(t->dup.as<float>(), t->da[l - 1].as<float>(), B, t->gh[l], t->gw[l], t->cin(l), s));
        }
    }
    HIPCHK(hipEventRecord(t->ev_wg, s2));
    HIPCHK(hipStreamWaitEvent(s, t->ev_wg, 0));
    // ---- all partial sums -> flat gradient, in workgroup order
    long total = 0;
    for (int l = 0; l < t->n_conv; ++l) total += 9L * t->cin(l) * t->ch[l] + t->ch[l];
    if (t->descs_batch != B) {          // the partial counts depend on the batch size only; hdescs outlives the copy
        for (int l = 0; l < t->n_conv; ++l) {
            const long klen = 9L * t->cin(l) * t->ch[l];
            t->hdescs[2 * l] = ReduceDesc{t->off_k[l], klen, t->wpart[l].as<float>(), t->np_w[l], klen};
            t->hdescs[2 * l + 1] = ReduceDesc{t->off_b[l], (long)t->ch[l], t->dzsum_part[l].as<float>(), t->np_b[l], (long)t->ch[l]};
        }
        HIPCHK(hipMemcpyAsync(t->descs.p, t->hdescs, 2 * t->n_conv * sizeof(ReduceDesc), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
        t->descs_batch = B;
    }
    LCHK(launch_reduce_all(t->descs.as<ReduceDesc>(), 2 * t->n_conv, total, G, s));
    return CS_OK;
}

int gen_train_forward_backward(cs_trainer* t, const float* x, const float* y, int64_t batch, int kind, float* loss, float* mae)
{
    int rc = gen_train_fb_enqueue(t, x, y, batch, kind);
    if (rc) return rc;
    float h[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(h, t->scal.p, 8, hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    if (loss) *loss = h[0];
    if (mae) *mae = h[1];
    return CS_OK;
}

// evaluation needs the forward tensors only: x, y, the stored activations, the output and the error partials -- not the
// relu / gradient buffers of a training batch (~15 GB at 1,024 cells of the 128 x 128 variant)
static int gen_eval_ensure(cs_trainer* t, int64_t b)
{
    if (b <= t->maxb || b <= t->eval_maxb) return CS_OK;
    int rc;
    const size_t npix = (size_t)t->H * t->W;
    if ((rc = t->x.ensure(b * npix * 4)) || (rc = t->y.ensure(b * npix * 4))) return rc;
    for (int l = 0; l < t->n_conv - 1; ++l)
        if ((rc = t->a[l].ensure(b * t->afl[l] * 4))) return rc;
    if ((rc = t->out.ensure(b * npix * 4)) || (rc = t->errpart.ensure((size_t)b * 8 * 4))) return rc;
    t->eval_maxb = b;
    return CS_OK;
}

int gen_train_eval(cs_trainer* t, const float* x, const float* y, int64_t n, int kind, float* loss, float* mae)
{
    const int64_t ch = n < 1024 ? n : 1024;
    int rc = gen_eval_ensure(t, ch);
    if (rc) return rc;
    hipStream_t s = t->stream;
    float* P = t->P.as<float>();
    float* MOV = t->MOV.as<float>();
    const int last = t->n_conv - 1;
    const size_t npix = (size_t)t->H * t->W;
    for (int l = 0; l < last; ++l)
        LCHK(launch_pack_ep(P + t->off_b[l], P + t->off_g[l], P + t->off_be[l], MOV + t->off_mm[l], MOV + t->off_mv[l], t->cfg.bn_eps, t->ch[l],
                            t->ep_inf[l].as<float>(), s));
    double s2 = 0.0, s1 = 0.0;
    std::vector<float> part;
    for (int64_t off = 0; off < n; off += ch) {
        const int64_t nc = (n - off) < ch ? (n - off) : ch;
        if ((rc = gen_copy_in(t, t->x, x + (size_t)off * npix, kind, (size_t)nc * npix)) ||
            (rc = gen_copy_in(t, t->y, y + (size_t)off * npix, kind, (size_t)nc * npix)))
            return rc;
        for (int l = 0; l < last; ++l) {
            const float* in = l == 0 ? t->x.as<float>() : t->a[l - 1].as<float>();
            LCHK(launch_conv_generic(in, P + t->off_k[l], t->ep_inf[l].as<float>(), t->a[l].as<float>(), nc, t->gh[l], t->gw[l], t->cin(l), t->ch[l],
                                     l > t->n_enc, l < t->n_enc ? GEN_EPI_BN_POOL : GEN_EPI_BN, s));
        }
        LCHK(launch_conv_generic(t->a[last - 1].as<float>(), P + t->off_k[last], P + t->off_b[last], t->out.as<float>(), nc, t->gh[last],
                                 t->gw[last], t->cin(last), 1, last > t->n_enc, GEN_EPI_SIGMOID, s));
        LCHK(launch_recon_err(t->out.as<float>(), t->y.as<float>(), nc, (int)npix, t->errpart.as<float>(), s));
        part.resize((size_t)nc * 8);
        HIPCHK(hipMemcpyAsync(part.data(), t->errpart.p, part.size() * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (size_t i = 0; i < part.size(); i += 2) { s2 += part[i]; s1 += part[i + 1]; }
    }
    if (loss) *loss = (float)(s2 / ((double)n * npix));
    if (mae) *mae = (float)(s1 / ((double)n * npix));
    return CS_OK;
}
