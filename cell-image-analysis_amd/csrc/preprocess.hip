// preprocess.hip -- per-crop preprocess of the screening path on gfx950:
//
//     cell_image_eq      = exposure.equalize_adapthist(cell_image, clip_limit=0.02)
//     cell_image_resized = resize(cell_image_eq, (64, 64), anti_aliasing=True)
//
// (improved_detection.py:98-99, CAE_improved_modeltrain.py:92-93) plus the float32 cast of
// improved_detection.py:122.  The arithmetic is scikit-image 0.18.3 / SciPy 1.7.1 (see
// oracle/preprocess_oracle.py for the restatement and how it is pinned).
//
// One workgroup (4 waves) owns one bounding-box crop from its raw integer pixels to the 64x64 fp32
// tile the autoencoder reads.  CLAHE is integer work and is reproduced bit-exactly, including
// skimage's float32 accumulation order of the four blended look-ups; every floating-point step
// that feeds a rounding or truncation decision is done in fp64 with explicitly rounded operations
// (no FMA contraction).  The resize (Gaussian anti-alias + bilinear warp) is fp64 as in SciPy.
//
// LDS: the contrast maps of all tiles of the crop (uint16 [tiles][256], <= 225 tiles = 115 KB; a
// typical 8x8..9x9 tiling is 32-41 KB) + one 256-bin histogram per wave.
// HBM per crop: H*W raw pixels in, 16 KB out; the fp64 blur planes are a per-chunk scratch that
// stays in L2 (a crop is a few thousand pixels).
#include "api_internal.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>

namespace cs {

struct CropDesc {
    long long off;      // element offset of the crop's first pixel in the ragged pixel buffer
    int H, W;
};

static constexpr int PP_THREADS = 512;
static constexpr int PP_WAVES = PP_THREADS / 64;
static constexpr int PP_NBINS = 256;
static constexpr int PP_GRAY = 1 << 14;                 // NR_OF_GRAY, _adapthist.py:23
static constexpr int PP_BIN_SIZE = 1 + PP_GRAY / PP_NBINS;
static constexpr int PP_OUT = 64;
static constexpr int PP_MAX_TAPS = 32;                  // Gaussian radius limit: side <= 1024

__device__ inline int wave_sum(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ inline int wave_min(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m));
    return v;
}
__device__ inline int wave_max(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m));
    return v;
}
__device__ inline double wave_min(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m));
    return v;
}
__device__ inline double wave_max(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
    return v;
}

// orders this wave's LDS traffic (a wave's ds operations execute in issue order; the fences keep the
// compiler from moving them across)
__device__ inline void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// workgroup min/max over per-thread values; every thread gets the result
template <typename T>
__device__ inline void block_minmax(T& lo, T& hi, T* red /* [2*PP_WAVES] */)
{
    lo = wave_min(lo);
    hi = wave_max(hi);
    const int wave = threadIdx.x >> 6;
    __syncthreads();                                    // red[] may still be read from a previous use
    if ((threadIdx.x & 63) == 0) { red[wave] = lo; red[PP_WAVES + wave] = hi; }
    __syncthreads();
    lo = red[0];
    hi = red[PP_WAVES];
#pragma unroll
    for (int w = 1; w < PP_WAVES; ++w) {
        lo = red[w] < lo ? red[w] : lo;
        hi = red[PP_WAVES + w] > hi ? red[PP_WAVES + w] : hi;
    }
}

template <typename PIX> __device__ inline int to_u16(PIX v);
template <> __device__ inline int to_u16<unsigned char>(unsigned char v) { return (int)v * 257; }   // img_as_uint 8 -> 16
template <> __device__ inline int to_u16<unsigned short>(unsigned short v) { return (int)v; }

// rescale_intensity(out_range=(0, 2^14-1)) + np.round + // bin_size  (_adapthist.py:78-81,139-143).
// skimage evaluates rint(fl(fl((v - imin) / range) * 16383)) in float64.  The exact rational
// a*16383/range (a = v - imin < 2^16) is either a tie (2*rem == range) or at least 1/(2*range) >= 7.6e-6
// away from one, while the two float64 roundings move it by < 4e-12: away from ties the integer
// round-to-nearest below is the same number; exact ties take the float64 path itself.
__device__ inline int gray_bin(int v16, int vlo, int irange, double rcp_range)
{
    int g;
    if (irange != 0) {
        const int a = v16 - vlo;
        const unsigned num = (unsigned)a * (unsigned)(PP_GRAY - 1);                 // < 2^30
        int q = (int)((double)num * rcp_range);
        int rem = (int)num - q * irange;
        if (rem < 0) { --q; rem += irange; } else if (rem >= irange) { ++q; rem -= irange; }
        if (2 * rem > irange) {
            g = q + 1;
        } else if (2 * rem < irange) {
            g = q;
        } else {
            const double x = __dmul_rn(__ddiv_rn((double)a, (double)irange), (double)(PP_GRAY - 1));
            g = (int)rint(x);                           // rint: half-to-even, as np.round
        }
    } else {
        g = min(max(v16, 0), PP_GRAY - 1);              // constant crop: np.clip to the output range
    }
    return g / PP_BIN_SIZE;
}

// exact p / d for 0 <= p < 2^22, 1 <= d: float reciprocal estimate + one correction
__device__ inline int fast_div(int p, int d, float rcp, int& rem)
{
    int q = (int)((float)p * rcp);
    rem = p - q * d;
    if (rem < 0) { --q; rem += d; } else if (rem >= d) { ++q; rem -= d; }
    return q;
}

__device__ inline int reflect_once(int i, int n) { return i >= n ? 2 * (n - 1) - i : i; }   // np.pad 'reflect'

// scipy.ndimage mode='mirror' and skimage's coord_map mode 'R' are the same map (reflect about the
// centre of the edge pixel).  One reflection is enough here: the Gaussian radius int(4*sigma + 0.5)
// with sigma = (n/64 - 1)/2 is below n/32, and the warp samples rows -1 .. n at most.
__device__ inline int mirror_once(int i, int n)
{
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// clip_histogram (_adapthist.py:241-289) + map_histogram (:292-330) for one tile, one wave.
// Lane l owns bins 4l..4l+3.  Returns the four map entries.
__device__ inline void clip_and_map(int h[4], int clim, double scale, int out[4])
{
    const int lane = threadIdx.x & 63;
    int exc = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (h[k] > clim) { exc += h[k] - clim; h[k] = clim; }
    }
    int n_excess = wave_sum(exc);
    const int bin_incr = n_excess / PP_NBINS;           // n_excess >= 0 here
    const int upper = clim - bin_incr;
    int cnt_low = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (h[k] < upper) { ++cnt_low; h[k] += bin_incr; }
    }
    n_excess -= wave_sum(cnt_low) * bin_incr;
    int mid = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (h[k] >= upper && h[k] < clim) { mid += h[k] - clim; h[k] = clim; }
    }
    n_excess += wave_sum(mid);

    bool stuck = false;
    while (n_excess > 0 && !stuck) {
        const int prev = n_excess;
        for (int index = 0; index < PP_NBINS; ++index) {
            int under = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) under += __popcll(__ballot(h[k] < clim));
            if (under == 0) { stuck = true; break; }    // nothing can change any more
            const int step = max(1, under / n_excess);
            int sel = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int b = 4 * lane + k;
                const bool s = h[k] < clim && b >= index && (b - index) % step == 0;
                if (s) ++h[k];
                sel += __popcll(__ballot(s));
            }
            n_excess -= sel;
            if (n_excess <= 0) break;
        }
        if (prev == n_excess) break;
    }

    // cumulative sum over the 256 bins
    int c0 = h[0], c1 = c0 + h[1], c2 = c1 + h[2], c3 = c2 + h[3];
    int incl = c3;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    const int base = incl - c3;
    const int cs4[4] = {base + c0, base + c1, base + c2, base + c3};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double m = __dmul_rn((double)cs4[k], scale);
        m = fmin(m, (double)(PP_GRAY - 1));
        out[k] = (int)m;                                // astype(int): truncation
    }
}

template <typename PIX>
__global__ __launch_bounds__(PP_THREADS) void preprocess_kernel(const PIX* __restrict__ pix,
                                                                const CropDesc* __restrict__ desc, double clip_limit,
                                                                unsigned short* __restrict__ clahe, float* __restrict__ out)
{
    // [tiles][256] uint16 contrast maps during CLAHE; afterwards the same bytes hold one fp64 line of
    // W values per wave for the resize
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    unsigned short* maps = (unsigned short*)dyn_lds;
    __shared__ unsigned int hist[PP_WAVES][PP_NBINS];
    __shared__ int red_i[2 * PP_WAVES];
    __shared__ double wts[2][PP_MAX_TAPS + 1];
    __shared__ double crow[128], ccol[128];             // in-tile offset / tile size (tile side <= 1024/8)
    __shared__ unsigned short rowtab[1024], coltab[1024];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const CropDesc d = desc[blockIdx.x];
    const int H = d.H, W = d.W, npx = H * W;
    const PIX* src = pix + d.off;
    unsigned short* cl = clahe + d.off;
    float* dst = out + (size_t)blockIdx.x * PP_OUT * PP_OUT;

    // ---- A: intensity range of the crop, then the grey-level bin of every pixel (parked in cl[]) ----
    const float rcpW = 1.0f / (float)W;
    int vlo = 0x7fffffff, vhi = -1;
    for (int p = tid; p < npx; p += PP_THREADS) {
        const int v = to_u16<PIX>(src[p]);
        vlo = min(vlo, v);
        vhi = max(vhi, v);
    }
    block_minmax(vlo, vhi, red_i);
    const int irange = vhi - vlo;
    const double rcp_range = irange ? 1.0 / (double)irange : 0.0;
    for (int p = tid; p < npx; p += PP_THREADS) cl[p] = (unsigned short)gray_bin(to_u16<PIX>(src[p]), vlo, irange, rcp_range);

    // per-row / per-column block index and in-block offset of the padded image, the blend weights,
    // and the two anti-aliasing kernels (scipy _gaussian_kernel1d: sigma = (n/64 - 1)/2, truncate 4)
    const int kh = H / 8, kw = W / 8;
    const int ph0 = kh / 2, pw0 = kw / 2;
    for (int r = tid; r < H; r += PP_THREADS) {
        const int pr = r + ph0, br = pr / kh;
        rowtab[r] = (unsigned short)((br << 8) | (pr - br * kh));
    }
    for (int c = tid; c < W; c += PP_THREADS) {
        const int pc = c + pw0, bc = pc / kw;
        coltab[c] = (unsigned short)((bc << 8) | (pc - bc * kw));
    }
    if (tid < kh) crow[tid] = __ddiv_rn((double)tid, (double)kh);
    if (tid < kw) ccol[tid] = __ddiv_rn((double)tid, (double)kw);
    const double sig_r = fmax(0.0, ((double)H / (double)PP_OUT - 1.0) / 2.0);
    const double sig_c = fmax(0.0, ((double)W / (double)PP_OUT - 1.0) / 2.0);
    const int lw_r = sig_r > 1e-15 ? (int)(4.0 * sig_r + 0.5) : -1;     // -1: no filter along this axis
    const int lw_c = sig_c > 1e-15 ? (int)(4.0 * sig_c + 0.5) : -1;
    if (tid >= PP_THREADS - 128 && (tid & 63) == 0) {                    // lane 0 of the last two waves
        const int ax = (tid >> 6) & 1;
        const double sg = ax ? sig_c : sig_r;
        const int lw = ax ? lw_c : lw_r;
        double sum = 0.0;
        for (int j = -lw; j <= lw; ++j) sum += exp(-0.5 / (sg * sg) * (double)(j * j));
        for (int j = 0; j <= lw; ++j) wts[ax][j] = exp(-0.5 / (sg * sg) * (double)(j * j)) / sum;
    }
    __syncthreads();                                     // publishes cl[] and the tables

    // ---- B: per-tile histogram -> clip -> cumulative map; a wave owns a tile, no workgroup barrier ----
    const int nty = (H + kh - 1) / kh, ntx = (W + kw - 1) / kw;      // histogram tiles (= ns_hist)
    const int tiles = nty * ntx, tpx = kh * kw;
    const float rcpkw = 1.0f / (float)kw;
    int clim;
    if (clip_limit > 0.0) {
        const double c = __dmul_rn(clip_limit, (double)tpx);
        clim = (int)fmax(c, 1.0);
    } else {
        clim = tpx;
    }
    const double scale = __ddiv_rn((double)(PP_GRAY - 1), (double)tpx);
    for (int t = wave; t < tiles; t += PP_WAVES) {
#pragma unroll
        for (int k = 0; k < 4; ++k) hist[wave][4 * lane + k] = 0;
        wave_sync();
        const int ty = t / ntx, tx = t - ty * ntx;
        for (int p = lane; p < tpx; p += 64) {
            int j;
            const int i = fast_div(p, kw, rcpkw, j);
            const int r = reflect_once(ty * kh + i, H), c = reflect_once(tx * kw + j, W);
            atomicAdd(&hist[wave][cl[r * W + c]], 1u);
        }
        wave_sync();
        int h[4], m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k] = (int)hist[wave][4 * lane + k];
        wave_sync();
        clip_and_map(h, clim, scale, m);
#pragma unroll
        for (int k = 0; k < 4; ++k) maps[(size_t)t * PP_NBINS + 4 * lane + k] = (unsigned short)m[k];
    }
    __syncthreads();

    // ---- C: blend the four neighbouring maps (float32 accumulation, skimage's edge order) --------
    int ulo = 0x7fffffff, uhi = -1;
    for (int p = tid; p < npx; p += PP_THREADS) {
        int c;
        const int r = fast_div(p, W, rcpW, c);
        const int b = cl[p];
        const int rt = rowtab[r], ct = coltab[c];
        const int br = rt >> 8, bc = ct >> 8;
        const double cr = crow[rt & 255], cc = ccol[ct & 255];
        const double wr[2] = {__dsub_rn(1.0, cr), cr}, wc[2] = {__dsub_rn(1.0, cc), cc};
        float acc = 0.0f;
#pragma unroll
        for (int e0 = 0; e0 < 2; ++e0) {
#pragma unroll
            for (int e1 = 0; e1 < 2; ++e1) {
                const int ty = min(max(br + e0 - 1, 0), nty - 1);
                const int tx = min(max(bc + e1 - 1, 0), ntx - 1);
                const double mv = (double)maps[(size_t)(ty * ntx + tx) * PP_NBINS + b];
                const double coef = __dmul_rn(wc[e1], wr[e0]);
                acc = __fadd_rn(acc, (float)__dmul_rn(mv, coef));
            }
        }
        const int u = (int)acc;                          // astype(uint16): truncation
        cl[p] = (unsigned short)u;
        ulo = min(ulo, u);
        uhi = max(uhi, u);
    }
    block_minmax(ulo, uhi, red_i);                       // its barriers also retire the maps and publish cl[]

    // ---- D: img_as_float + rescale_intensity + resize, fused -------------------------------------------
    // skimage: eq = (u/65535 - min)/(max - min) (_adapthist.py:93-94), Gaussian along rows then columns,
    // bilinear warp at r = H/64*(i+0.5)-0.5 (mode 'reflect'), clip to the filtered image's range.  Every
    // step after the uint16 image is linear with weights that sum to 1, so the same numbers come from
    // filtering + sampling (u - ulo) in fp64 and dividing by (uhi - ulo) at the end (differences ~1e-16
    // relative; the result is rounded to float32 anyway), and the clip can only act on rounding noise:
    // it becomes a clamp to [0,1].  A wave produces one output row: the vertically filtered and
    // row-interpolated line Y[0..W) goes to LDS, then each lane filters + interpolates along it.
    double* yline = (double*)dyn_lds + (size_t)wave * W;
    const double fr = (double)H / (double)PP_OUT, fc = (double)W / (double)PP_OUT;
    const double or_ = fr * 0.5 - 0.5, oc_ = fc * 0.5 - 0.5;
    const bool eflat = ulo == uhi;
    const double inv = eflat ? 0.0 : 1.0 / (double)(uhi - ulo);
    const double cflat = fmin(fmax((double)ulo * (1.0 / 65535.0), 0.0), 1.0);
    // column sample points of this lane
    const double cpos = fc * (double)lane + oc_;
    const double c0f = floor(cpos), dc = cpos - c0f;
    const int c0 = (int)c0f, c1 = (int)ceil(cpos);
    for (int i = wave; i < PP_OUT; i += PP_WAVES) {
        const double rpos = fr * (double)i + or_;
        const double r0f = floor(rpos), dr = rpos - r0f;
        const int r0 = (int)r0f, r1 = (int)ceil(rpos);
        for (int c = lane; c < W; c += 64) {
            double y0, y1;
            if (lw_r < 0) {
                y0 = (double)((int)cl[mirror_once(r0, H) * W + c] - ulo);
                y1 = (double)((int)cl[mirror_once(r1, H) * W + c] - ulo);
            } else {
                // the filtered image is sampled at mirror(r0), mirror(r1); taps mirror again around them
                const int q0 = mirror_once(r0, H), q1 = mirror_once(r1, H);
                y0 = (double)((int)cl[q0 * W + c] - ulo) * wts[0][0];
                y1 = (double)((int)cl[q1 * W + c] - ulo) * wts[0][0];
                for (int j = lw_r; j >= 1; --j) {
                    const double w = wts[0][j];
                    y0 += (double)((int)cl[mirror_once(q0 - j, H) * W + c] + (int)cl[mirror_once(q0 + j, H) * W + c] - 2 * ulo) * w;
                    y1 += (double)((int)cl[mirror_once(q1 - j, H) * W + c] + (int)cl[mirror_once(q1 + j, H) * W + c] - 2 * ulo) * w;
                }
            }
            yline[c] = (1.0 - dr) * y0 + dr * y1;
        }
        wave_sync();
        double z0, z1;
        const int q0 = mirror_once(c0, W), q1 = mirror_once(c1, W);
        if (lw_c < 0) {
            z0 = yline[q0];
            z1 = yline[q1];
        } else {
            z0 = yline[q0] * wts[1][0];
            z1 = yline[q1] * wts[1][0];
            for (int j = lw_c; j >= 1; --j) {
                const double w = wts[1][j];
                z0 += (yline[mirror_once(q0 - j, W)] + yline[mirror_once(q0 + j, W)]) * w;
                z1 += (yline[mirror_once(q1 - j, W)] + yline[mirror_once(q1 + j, W)]) * w;
            }
        }
        const double z = (1.0 - dc) * z0 + dc * z1;
        const double v = eflat ? cflat : fmin(fmax(z * inv, 0.0), 1.0);
        dst[i * PP_OUT + lane] = (float)v;              // .astype('float32'), improved_detection.py:122
        wave_sync();                                     // yline is rewritten in the next round
    }
}

}  // namespace cs

// ---- C ABI ----------------------------------------------------------------------------------
using namespace cs;

struct cs_preproc {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DevBuf pix, clahe, out;
    // crop descriptors: pinned host memory the kernel reads directly (24 B per crop).  A host-to-device COPY of them would
    // queue on the DMA engine behind whatever the caller has in flight there -- e.g. the raw pixels of the NEXT chunk it is
    // uploading while this one computes -- and the kernel would start only when that upload has finished.
    void* hdesc = nullptr;
    const void* ddesc = nullptr;        // the same memory as the device sees it
    double last_kernel_ms = 0.0;
    int64_t last_pixels = 0;
    ~cs_preproc() { if (hdesc) (void)hipHostFree(hdesc); }
};

static const int64_t kChunkPixels = 256ll << 20;        // pixel span of one launch (staging + uint16 plane)
static const int64_t kChunkCrops = 1 << 16;

int cs_preproc_create(int device_id, cs_preproc** out)
{
    if (!out) return fail(CS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int rc = require_gfx950(device_id);
    if (rc) return rc;
    cs_preproc* p = new cs_preproc();
    p->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&p->ev0);
    if (e == hipSuccess) e = hipEventCreate(&p->ev1);
    if (e == hipSuccess) e = hipHostMalloc(&p->hdesc, (size_t)kChunkCrops * sizeof(CropDesc), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&p->ddesc, p->hdesc, 0);
    if (e != hipSuccess) {
        delete p;
        return fail(CS_ERR_HIP, "stream / event / pinned descriptor buffer creation failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return CS_OK;
}

int cs_preproc_wait_stream(cs_preproc* p, void* hip_stream)
{
    if (!p) return fail(CS_ERR_INVALID, "preprocess handle is NULL");
    HIPCHK(hipSetDevice(p->device));
    return wait_on_stream(p->stream, hip_stream);
}

void cs_preproc_free(cs_preproc* p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

int cs_preproc_last_timing(const cs_preproc* p, double* kernel_ms, int64_t* pixels)
{
    if (!p) return fail(CS_ERR_INVALID, "handle is NULL");
    if (kernel_ms) *kernel_ms = p->last_kernel_ms;
    if (pixels) *pixels = p->last_pixels;
    return CS_OK;
}

int cs_preprocess(cs_preproc* p, const void* pixels, int pixel_type, int64_t n_pixels, int pixels_kind,
                  const int64_t* offsets, const int32_t* heights, const int32_t* widths, int64_t n,
                  double clip_limit, float* out, uint16_t* clahe_out, int out_kind)
{
    if (!p) return fail(CS_ERR_INVALID, "handle is NULL");
    HIPCHK(hipSetDevice(p->device));
    if (n < 0) return fail(CS_ERR_INVALID, "n is negative");
    if (n == 0) return CS_OK;
    if (!pixels || !offsets || !heights || !widths || !out) return fail(CS_ERR_INVALID, "NULL argument");
    if (pixel_type != CS_PIX_U8 && pixel_type != CS_PIX_U16) return fail(CS_ERR_INVALID, "pixel_type must be CS_PIX_U8 or CS_PIX_U16");
    if ((pixels_kind != CS_MEM_HOST && pixels_kind != CS_MEM_DEVICE) || (out_kind != CS_MEM_HOST && out_kind != CS_MEM_DEVICE))
        return fail(CS_ERR_INVALID, "memory kind must be CS_MEM_HOST or CS_MEM_DEVICE");
    if (!(clip_limit == clip_limit)) return fail(CS_ERR_INVALID, "clip_limit is NaN");
    const size_t esz = pixel_type == CS_PIX_U8 ? 1 : 2;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t H = heights[i], W = widths[i];
        if (H < 8 || W < 8)
            return fail(CS_ERR_INVALID, "crop %lld is %lldx%lld: below 8 px kernel_size = shape//8 is 0 (skimage raises ZeroDivisionError)",
                        (long long)i, (long long)H, (long long)W);
        if (H > 1024 || W > 1024) return fail(CS_ERR_UNSUPPORTED, "crop %lld is %lldx%lld: side above 1024", (long long)i, (long long)H, (long long)W);
        if (offsets[i] < 0 || offsets[i] + H * W > n_pixels)
            return fail(CS_ERR_INVALID, "crop %lld [%lld, +%lld) lies outside the pixel buffer of %lld", (long long)i,
                        (long long)offsets[i], (long long)(H * W), (long long)n_pixels);
        if (i > 0 && offsets[i] < offsets[i - 1] + (int64_t)heights[i - 1] * widths[i - 1])
            return fail(CS_ERR_INVALID, "crop %lld overlaps crop %lld: offsets must ascend without overlap", (long long)i, (long long)(i - 1));
    }

    p->last_kernel_ms = 0.0;
    p->last_pixels = 0;
    CropDesc* const hdesc = (CropDesc*)p->hdesc;     // free: every launch below is followed by a stream synchronisation
    int64_t i0 = 0;
    while (i0 < n) {
        // a chunk: consecutive crops, bounded in count and in the pixel span the fp64 planes must cover
        int64_t i1 = i0, lo = offsets[i0], hi = offsets[i0];
        size_t lds = 0;
        while (i1 < n && i1 - i0 < kChunkCrops) {
            const int64_t H = heights[i1], W = widths[i1];
            const int64_t nlo = std::min(lo, offsets[i1]), nhi = std::max(hi, offsets[i1] + H * W);
            if (i1 > i0 && nhi - nlo > kChunkPixels) break;
            lo = nlo; hi = nhi;
            const int kh = (int)H / 8, kw = (int)W / 8;
            const size_t tiles = (size_t)((H + kh - 1) / kh) * (size_t)((W + kw - 1) / kw);
            lds = std::max(lds, std::max(tiles * PP_NBINS * sizeof(unsigned short), (size_t)PP_WAVES * (size_t)W * sizeof(double)));
            ++i1;
        }
        const int64_t nc = i1 - i0, span = hi - lo;
        if (span > (1ll << 31)) return fail(CS_ERR_UNSUPPORTED, "crops of one chunk are spread over more than 2^31 pixels");
        for (int64_t i = 0; i < nc; ++i) hdesc[(size_t)i] = CropDesc{(long long)(offsets[i0 + i] - lo), heights[i0 + i], widths[i0 + i]};

        int rc;
        const char* d_pix;
        if (pixels_kind == CS_MEM_DEVICE) {
            d_pix = (const char*)pixels + (size_t)lo * esz;
        } else {
            if ((rc = p->pix.ensure((size_t)span * esz))) return rc;
            HIPCHK(hipMemcpyAsync(p->pix.p, (const char*)pixels + (size_t)lo * esz, (size_t)span * esz, hipMemcpyHostToDevice, p->stream));
            d_pix = (const char*)p->pix.p;
        }
        unsigned short* d_clahe;
        if (clahe_out && out_kind == CS_MEM_DEVICE) {
            d_clahe = clahe_out + lo;
        } else {
            if ((rc = p->clahe.ensure((size_t)span * sizeof(unsigned short)))) return rc;
            d_clahe = p->clahe.as<unsigned short>();
            // gaps between crops read 0 -- only when the plane goes back to the caller: a crop reads nothing but its own region,
            // and a fill is DMA work that would queue behind whatever upload the caller has in flight (see hdesc above)
            if (clahe_out) HIPCHK(hipMemsetAsync(d_clahe, 0, (size_t)span * sizeof(unsigned short), p->stream));
        }
        float* d_out;
        if (out_kind == CS_MEM_DEVICE) {
            d_out = out + (size_t)i0 * PP_OUT * PP_OUT;
        } else {
            if ((rc = p->out.ensure((size_t)nc * PP_OUT * PP_OUT * sizeof(float)))) return rc;
            d_out = p->out.as<float>();
        }

        HIPCHK(hipEventRecord(p->ev0, p->stream));
        if (pixel_type == CS_PIX_U8) {
            HIPCHK(hipFuncSetAttribute((const void*)preprocess_kernel<unsigned char>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(preprocess_kernel<unsigned char>, dim3((unsigned)nc), dim3(PP_THREADS), lds, p->stream,
                               (const unsigned char*)d_pix, (const CropDesc*)p->ddesc, clip_limit, d_clahe, d_out);
        } else {
            HIPCHK(hipFuncSetAttribute((const void*)preprocess_kernel<unsigned short>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(preprocess_kernel<unsigned short>, dim3((unsigned)nc), dim3(PP_THREADS), lds, p->stream,
                               (const unsigned short*)d_pix, (const CropDesc*)p->ddesc, clip_limit, d_clahe, d_out);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(p->ev1, p->stream));
        if (out_kind == CS_MEM_HOST) {
            HIPCHK(hipMemcpyAsync(out + (size_t)i0 * PP_OUT * PP_OUT, d_out, (size_t)nc * PP_OUT * PP_OUT * sizeof(float),
                                  hipMemcpyDeviceToHost, p->stream));
            if (clahe_out)
                HIPCHK(hipMemcpyAsync(clahe_out + lo, d_clahe, (size_t)span * sizeof(unsigned short), hipMemcpyDeviceToHost, p->stream));
        }
        HIPCHK(hipStreamSynchronize(p->stream));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, p->ev0, p->ev1));
        p->last_kernel_ms += ms;
        p->last_pixels += span;
        i0 = i1;
    }
    return CS_OK;
}
