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

static constexpr int PP_THREADS = 256;
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

// rescale_intensity(out_range=(0, 2^14-1)) + np.round + // bin_size  (_adapthist.py:78-81,139-143)
__device__ inline int gray_bin(int v16, double imin, double range, bool flat)
{
    double x;
    if (!flat) {
        x = __ddiv_rn(__dsub_rn((double)v16, imin), range);
        x = __dmul_rn(x, (double)(PP_GRAY - 1));
    } else {
        x = fmin(fmax((double)v16, 0.0), (double)(PP_GRAY - 1));
    }
    return (int)rint(x) / PP_BIN_SIZE;                  // rint: half-to-even, as np.round
}

__device__ inline int reflect_once(int i, int n) { return i >= n ? 2 * (n - 1) - i : i; }   // np.pad 'reflect'

__device__ inline int mirror_any(int i, int n)         // scipy.ndimage mode='mirror'
{
    const int p = 2 * (n - 1);
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - i : i;
}

__device__ inline int warp_reflect(int i, int n)       // skimage coord_map mode 'R'
{
    const int cmax = n - 1;
    const int a = i < 0 ? -i : i;
    const int q = a / cmax, r = a % cmax;
    return (q & 1) ? cmax - r : r;
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
                                                                unsigned short* __restrict__ clahe,
                                                                double* __restrict__ buf0, double* __restrict__ buf1,
                                                                float* __restrict__ out)
{
    extern __shared__ unsigned short maps[];            // [tiles][256]
    __shared__ unsigned int hist[PP_WAVES][PP_NBINS];
    __shared__ int red_i[2 * PP_WAVES];
    __shared__ double red_d[2 * PP_WAVES];
    __shared__ double wts[PP_MAX_TAPS + 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const CropDesc d = desc[blockIdx.x];
    const int H = d.H, W = d.W, npx = H * W;
    const PIX* src = pix + d.off;
    unsigned short* cl = clahe + d.off;
    double* b0 = buf0 + d.off;
    double* b1 = buf1 + d.off;
    float* dst = out + (size_t)blockIdx.x * PP_OUT * PP_OUT;

    // ---- A: intensity range of the crop --------------------------------------------------
    int vlo = 0x7fffffff, vhi = -1;
    for (int p = tid; p < npx; p += PP_THREADS) {
        const int v = to_u16<PIX>(src[p]);
        vlo = min(vlo, v);
        vhi = max(vhi, v);
    }
    block_minmax(vlo, vhi, red_i);
    const double imin = (double)vlo, range = (double)(vhi - vlo);
    const bool flat = vlo == vhi;

    // ---- B: per-tile histogram -> clip -> cumulative map ------------------------------------
    const int kh = H / 8, kw = W / 8;
    const int nty = (H + kh - 1) / kh, ntx = (W + kw - 1) / kw;      // histogram tiles (= ns_hist)
    const int tiles = nty * ntx, tpx = kh * kw;
    int clim;
    if (clip_limit > 0.0) {
        const double c = __dmul_rn(clip_limit, (double)tpx);
        clim = (int)fmax(c, 1.0);
    } else {
        clim = tpx;
    }
    const double scale = __ddiv_rn((double)(PP_GRAY - 1), (double)tpx);
    for (int t0 = 0; t0 < tiles; t0 += PP_WAVES) {
        const int t = t0 + wave;
        const bool live = t < tiles;
#pragma unroll
        for (int k = 0; k < 4; ++k) hist[wave][4 * lane + k] = 0;
        __syncthreads();
        if (live) {
            const int ty = t / ntx, tx = t - ty * ntx;
            for (int p = lane; p < tpx; p += 64) {
                const int i = p / kw, j = p - i * kw;
                const int r = reflect_once(ty * kh + i, H), c = reflect_once(tx * kw + j, W);
                const int b = gray_bin(to_u16<PIX>(src[r * W + c]), imin, range, flat);
                atomicAdd(&hist[wave][b], 1u);
            }
        }
        __syncthreads();
        if (live) {
            int h[4], m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) h[k] = (int)hist[wave][4 * lane + k];
            clip_and_map(h, clim, scale, m);
#pragma unroll
            for (int k = 0; k < 4; ++k) maps[(size_t)t * PP_NBINS + 4 * lane + k] = (unsigned short)m[k];
        }
        __syncthreads();
    }

    // ---- C: blend the four neighbouring maps (float32 accumulation, skimage's edge order) --------
    const int ph0 = kh / 2, pw0 = kw / 2;
    int ulo = 0x7fffffff, uhi = -1;
    for (int p = tid; p < npx; p += PP_THREADS) {
        const int r = p / W, c = p - r * W;
        const int b = gray_bin(to_u16<PIX>(src[p]), imin, range, flat);
        const int pr = r + ph0, pc = c + pw0;
        const int br = pr / kh, ir = pr - br * kh;
        const int bc = pc / kw, ic = pc - bc * kw;
        const double cr = __ddiv_rn((double)ir, (double)kh), cc = __ddiv_rn((double)ic, (double)kw);
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
    block_minmax(ulo, uhi, red_i);                       // also makes cl[] visible to the workgroup

    // ---- D: img_as_float + rescale_intensity (_adapthist.py:93-94) -> fp64 plane -------------------
    const double rcp = 1.0 / 65535.0;
    const double emin = __dmul_rn((double)ulo, rcp), erange = __dsub_rn(__dmul_rn((double)uhi, rcp), emin);
    const bool eflat = ulo == uhi;
    for (int p = tid; p < npx; p += PP_THREADS) {
        const double x = __dmul_rn((double)cl[p], rcp);
        b0[p] = eflat ? fmin(fmax(x, 0.0), 1.0) : __ddiv_rn(__dsub_rn(x, emin), erange);
    }
    __syncthreads();

    // ---- E: anti-aliasing Gaussian, axis 0 then axis 1 (scipy gaussian_filter, mode='mirror') -----
    double* cur = b0;
    double* oth = b1;
#pragma unroll 1
    for (int axis = 0; axis < 2; ++axis) {
        const int n_ax = axis == 0 ? H : W;
        const double f = (double)n_ax / (double)PP_OUT;
        const double sigma = fmax(0.0, (f - 1.0) / 2.0);
        if (!(sigma > 1e-15)) continue;
        const int lw = (int)(4.0 * sigma + 0.5);
        if (tid == 0) {
            double s = 0.0;
            for (int j = -lw; j <= lw; ++j) s += exp(-0.5 / (sigma * sigma) * (double)(j * j));
            for (int j = 0; j <= lw; ++j) wts[j] = exp(-0.5 / (sigma * sigma) * (double)(j * j)) / s;
        }
        __syncthreads();
        for (int p = tid; p < npx; p += PP_THREADS) {
            const int r = p / W, c = p - r * W;
            double acc = cur[p] * wts[0];
            if (axis == 0) {
                for (int j = lw; j >= 1; --j)
                    acc += (cur[mirror_any(r - j, H) * W + c] + cur[mirror_any(r + j, H) * W + c]) * wts[j];
            } else {
                for (int j = lw; j >= 1; --j)
                    acc += (cur[r * W + mirror_any(c - j, W)] + cur[r * W + mirror_any(c + j, W)]) * wts[j];
            }
            oth[p] = acc;
        }
        __syncthreads();
        double* t = cur; cur = oth; oth = t;
    }

    // ---- F: range of the filtered image (warp's clip=True) ----------------------------------------------
    double lo = 1e300, hi = -1e300;
    for (int p = tid; p < npx; p += PP_THREADS) {
        const double v = cur[p];
        lo = fmin(lo, v);
        hi = fmax(hi, v);
    }
    block_minmax(lo, hi, red_d);

    // ---- G: bilinear warp to 64x64, half-pixel centres, mode='reflect' (_warps.py:153-178) --------------
    const double fr = (double)H / (double)PP_OUT, fc = (double)W / (double)PP_OUT;
    const double or_ = fr * 0.5 - 0.5, oc_ = fc * 0.5 - 0.5;
    for (int p = tid; p < PP_OUT * PP_OUT; p += PP_THREADS) {
        const int i = p >> 6, j = p & 63;
        const double r = fr * (double)i + or_, c = fc * (double)j + oc_;
        const double r0 = floor(r), c0 = floor(c);
        const double dr = r - r0, dc = c - c0;
        const int r0i = warp_reflect((int)r0, H), r1i = warp_reflect((int)ceil(r), H);
        const int c0i = warp_reflect((int)c0, W), c1i = warp_reflect((int)ceil(c), W);
        const double top = (1.0 - dc) * cur[r0i * W + c0i] + dc * cur[r0i * W + c1i];
        const double bot = (1.0 - dc) * cur[r1i * W + c0i] + dc * cur[r1i * W + c1i];
        double v = (1.0 - dr) * top + dr * bot;
        v = fmin(fmax(v, lo), hi);
        dst[p] = (float)v;                               // .astype('float32'), improved_detection.py:122
    }
}

}  // namespace cs

// ---- C ABI ----------------------------------------------------------------------------------
using namespace cs;

struct cs_preproc {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DevBuf pix, desc, clahe, buf0, buf1, out;
    double last_kernel_ms = 0.0;
    int64_t last_pixels = 0;
};

static const int64_t kChunkPixels = 32ll << 20;         // fp64 planes: 2 x 256 MB per chunk
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
    if (e != hipSuccess) {
        delete p;
        return fail(CS_ERR_HIP, "stream/event creation failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return CS_OK;
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
    std::vector<CropDesc> hdesc;
    int64_t i0 = 0;
    while (i0 < n) {
        // a chunk: consecutive crops, bounded in count and in the pixel span the fp64 planes must cover
        int64_t i1 = i0, lo = offsets[i0], hi = offsets[i0];
        int max_tiles = 0;
        while (i1 < n && i1 - i0 < kChunkCrops) {
            const int64_t H = heights[i1], W = widths[i1];
            const int64_t nlo = std::min(lo, offsets[i1]), nhi = std::max(hi, offsets[i1] + H * W);
            if (i1 > i0 && nhi - nlo > kChunkPixels) break;
            lo = nlo; hi = nhi;
            const int kh = (int)H / 8, kw = (int)W / 8;
            max_tiles = std::max(max_tiles, (int)((H + kh - 1) / kh) * (int)((W + kw - 1) / kw));
            ++i1;
        }
        const int64_t nc = i1 - i0, span = hi - lo;
        if (span > (1ll << 31)) return fail(CS_ERR_UNSUPPORTED, "crops of one chunk are spread over more than 2^31 pixels");
        hdesc.resize((size_t)nc);
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
        if ((rc = p->desc.ensure((size_t)nc * sizeof(CropDesc)))) return rc;
        HIPCHK(hipMemcpyAsync(p->desc.p, hdesc.data(), (size_t)nc * sizeof(CropDesc), hipMemcpyHostToDevice, p->stream));
        if ((rc = p->buf0.ensure((size_t)span * sizeof(double))) || (rc = p->buf1.ensure((size_t)span * sizeof(double)))) return rc;
        unsigned short* d_clahe;
        if (clahe_out && out_kind == CS_MEM_DEVICE) {
            d_clahe = clahe_out + lo;
        } else {
            if ((rc = p->clahe.ensure((size_t)span * sizeof(unsigned short)))) return rc;
            d_clahe = p->clahe.as<unsigned short>();
            HIPCHK(hipMemsetAsync(d_clahe, 0, (size_t)span * sizeof(unsigned short), p->stream));   // gaps between crops read 0
        }
        float* d_out;
        if (out_kind == CS_MEM_DEVICE) {
            d_out = out + (size_t)i0 * PP_OUT * PP_OUT;
        } else {
            if ((rc = p->out.ensure((size_t)nc * PP_OUT * PP_OUT * sizeof(float)))) return rc;
            d_out = p->out.as<float>();
        }

        const size_t lds = (size_t)max_tiles * PP_NBINS * sizeof(unsigned short);
        HIPCHK(hipEventRecord(p->ev0, p->stream));
        if (pixel_type == CS_PIX_U8) {
            HIPCHK(hipFuncSetAttribute((const void*)preprocess_kernel<unsigned char>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(preprocess_kernel<unsigned char>, dim3((unsigned)nc), dim3(PP_THREADS), lds, p->stream,
                               (const unsigned char*)d_pix, p->desc.as<CropDesc>(), clip_limit, d_clahe, p->buf0.as<double>(),
                               p->buf1.as<double>(), d_out);
        } else {
            HIPCHK(hipFuncSetAttribute((const void*)preprocess_kernel<unsigned short>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(preprocess_kernel<unsigned short>, dim3((unsigned)nc), dim3(PP_THREADS), lds, p->stream,
                               (const unsigned short*)d_pix, p->desc.as<CropDesc>(), clip_limit, d_clahe, p->buf0.as<double>(),
                               p->buf1.as<double>(), d_out);
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
