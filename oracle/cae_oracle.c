/*
 * cae_oracle.c -- CPU restatement of the cell-crop anomaly-screening hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call it, and there only as the
 * checker / the timed CPU baseline -- never as the thing shipped.  The product path is
 * the HIP library (cell-image-analysis_amd/csrc) and it fails loudly without a GPU.
 *
 * What it restates (reference = /root/reference, read as text only):
 *   - the conv autoencoder graph      CAE_improved_modeltrain.py:184-229
 *   - per-cell scoring                improved_detection.py:117-153
 * The arithmetic of that path lives in third-party wheels the reference does not pin
 * (README.md:20-28): TensorFlow/Keras for conv/BN/pool/upsample/sigmoid, scikit-learn
 * (+ bundled libsvm) for RobustScaler / PCA / OneClassSVM.  Published algorithms restated:
 *   Conv2D(3x3, 'same', relu), BatchNormalization(eps=1e-3, inference),
 *   MaxPooling2D(2x2), UpSampling2D(2x2, nearest), sigmoid           (Keras layer defaults)
 *   RobustScaler.transform   sklearn/preprocessing/_data.py:1715-1718 (sklearn 1.7.2)
 *   PCA.transform            sklearn/decomposition/_base.py:147-155
 *   OneClassSVM RBF decision sklearn/svm/src/libsvm/svm.cpp:461-476, 2818-2838
 *
 * PINNING STATUS.  The reference has no tests, no fixtures and no golden vectors, and
 * TensorFlow/Keras is not installed here, so the Keras half is "PARITY UNPINNED" by the
 * reference itself; it is cross-checked against an independent implementation (torch-CPU
 * functional ops) in tests/test_oracle_cpu.py and tests/golden/.  The scikit-learn half
 * IS pinned: tests compare against the real sklearn 1.7.2 objects the reference calls.
 *
 * Layouts: activations NHWC fp32, conv kernels HWIO (kh,kw,cin,cout) as Keras stores
 * them, features flattened (h,w,c) as improved_detection.py:131 does.
 *
 * `acc64` selects double accumulation for every dot product (the "fp64-evaluated
 * oracle" the stated tolerances are measured against); acc64=0 is plain fp32 fmaf
 * accumulation in (tap, cin) order, which is what the timed CPU baseline runs.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_LAYERS 16

/* ---------------------------------------------------------------- synthetic crops */
/* Counter-based generator keyed (seed, cell, pixel): the HIP library implements the
 * identical integer hash (csrc/synth.hip) so CPU and GPU see bit-identical crops at
 * any size without shipping data.  Values are k/2^24, k in [0, 2^24): U[0,1) fp32,
 * the post-CLAHE range of improved_detection.py:98-99. */
static inline uint32_t orc_hash24(uint64_t seed, uint64_t cell, uint32_t pix)
{
    uint64_t z = seed + cell * 0x9E3779B97F4A7C15ULL + (uint64_t)pix * 0xD1B54A32D192ED03ULL;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (uint32_t)(z >> 40);
}

void orc_synth_crops(uint64_t seed, int64_t first_cell, int64_t n, int npix, float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < n; ++c)
        for (int p = 0; p < npix; ++p)
            out[c * npix + p] = (float)orc_hash24(seed, (uint64_t)(first_cell + c), (uint32_t)p)
                                * (1.0f / 16777216.0f);
}

/* ---------------------------------------------------------------- model description */
typedef struct {
    int H, W;               /* input spatial size (64,64) */
    int n_conv;             /* 7 */
    int n_enc;              /* encoder convs (3): each followed by BN + maxpool */
    int cin[ORC_MAX_LAYERS], cout[ORC_MAX_LAYERS];
    const float *kernel[ORC_MAX_LAYERS];   /* HWIO [3][3][cin][cout] */
    const float *bias[ORC_MAX_LAYERS];     /* [cout] */
    const float *bn_scale[ORC_MAX_LAYERS]; /* gamma/sqrt(var+eps)  (NULL on the last conv) */
    const float *bn_shift[ORC_MAX_LAYERS]; /* beta - mean*scale */
} orc_cae;

/* conv 3x3 'same' (cross-correlation, zero pad 1) + bias, NHWC, optional virtual
 * nearest x2 upsample of the input (UpSampling2D then Conv2D, CAE...:206-208). */
static void conv3x3(const float *in, int H, int W, int cin, int ups,
                    const float *k, const float *b, int cout, float *out, int acc64)
{
    /* H, W are the conv grid (= output) size; the stored input is (H>>ups, W>>ups). */
    const int Ws = W >> ups;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float *o = out + ((size_t)y * W + x) * cout;
            if (acc64) {
                double acc[256];
                for (int co = 0; co < cout; ++co) acc[co] = (double)b[co];
                for (int dy = -1; dy <= 1; ++dy) {
                    int yy = y + dy; if (yy < 0 || yy >= H) continue;
                    for (int dx = -1; dx <= 1; ++dx) {
                        int xx = x + dx; if (xx < 0 || xx >= W) continue;
                        const float *ip = in + ((size_t)(yy >> ups) * Ws + (xx >> ups)) * cin;
                        const float *kp = k + (size_t)((dy + 1) * 3 + (dx + 1)) * cin * cout;
                        for (int ci = 0; ci < cin; ++ci) {
                            double v = (double)ip[ci];
                            const float *kr = kp + (size_t)ci * cout;
                            for (int co = 0; co < cout; ++co) acc[co] += v * (double)kr[co];
                        }
                    }
                }
                for (int co = 0; co < cout; ++co) o[co] = (float)acc[co];
            } else {
                float acc[256];
                for (int co = 0; co < cout; ++co) acc[co] = 0.0f;
                for (int dy = -1; dy <= 1; ++dy) {
                    int yy = y + dy; if (yy < 0 || yy >= H) continue;
                    for (int dx = -1; dx <= 1; ++dx) {
                        int xx = x + dx; if (xx < 0 || xx >= W) continue;
                        const float *ip = in + ((size_t)(yy >> ups) * Ws + (xx >> ups)) * cin;
                        const float *kp = k + (size_t)((dy + 1) * 3 + (dx + 1)) * cin * cout;
                        for (int ci = 0; ci < cin; ++ci) {
                            float v = ip[ci];
                            const float *kr = kp + (size_t)ci * cout;
                            for (int co = 0; co < cout; ++co) acc[co] = fmaf(v, kr[co], acc[co]);
                        }
                    }
                }
                for (int co = 0; co < cout; ++co) o[co] = acc[co] + b[co];
            }
        }
}

static void relu_bn(float *x, size_t npix, int c, const float *s, const float *t)
{
    for (size_t p = 0; p < npix; ++p)
        for (int ch = 0; ch < c; ++ch) {
            float v = x[p * c + ch];
            v = v > 0.0f ? v : 0.0f;                 /* activation='relu' inside Conv2D */
            x[p * c + ch] = v * s[ch] + t[ch];       /* BatchNormalization, inference */
        }
}

static void maxpool2(const float *in, int H, int W, int c, float *out)
{
    int Ho = H / 2, Wo = W / 2;
    for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Wo; ++x)
            for (int ch = 0; ch < c; ++ch) {
                const float *p = in + ((size_t)(2 * y) * W + 2 * x) * c + ch;
                float a = p[0], b = p[c], d = p[(size_t)W * c], e = p[(size_t)W * c + c];
                float m = a > b ? a : b, n = d > e ? d : e;
                out[((size_t)y * Wo + x) * c + ch] = m > n ? m : n;
            }
}

/* Per-cell scratch big enough for any layer: H*W*maxC floats, two buffers. */
static size_t scratch_floats(const orc_cae *m)
{
    int maxc = 1;
    for (int l = 0; l < m->n_conv; ++l) if (m->cout[l] > maxc) maxc = m->cout[l];
    return (size_t)m->H * m->W * maxc;
}

/* One cell through the graph.  Outputs any of: features (encoded, h*w*c), recon (H*W),
 * layer_out[l] (the tensor the NEXT conv reads: post relu/BN and, in the encoder, pool;
 * for the last conv the sigmoid output).  buf[0..2]: three scratch buffers. */
static void cae_forward_cell(const orc_cae *m, const float *x, int acc64,
                             float *features, float *recon, float **layer_out,
                             float *buf0, float *buf1, float *buf2)
{
    float *bufs[3] = { buf0, buf1, buf2 };
    int H = m->H, W = m->W;           /* stored spatial size of `cur` */
    const float *cur = x;
    int cur_idx = -1;                 /* which scratch buffer `cur` lives in (-1: caller's x) */
    for (int l = 0; l < m->n_conv; ++l) {
        int cin = m->cin[l], cout = m->cout[l];
        int is_enc = l < m->n_enc;
        int is_last = l == m->n_conv - 1;
        /* CAE...:204-216: every decoder conv but the first reads an UpSampling2D output */
        int ups = (!is_enc && l > m->n_enc) ? 1 : 0;
        int Hc = H << ups, Wc = W << ups;
        int ci_idx = (cur_idx + 1) % 3; if (ci_idx < 0) ci_idx = 0;
        float *conv_out = bufs[ci_idx];
        conv3x3(cur, Hc, Wc, cin, ups, m->kernel[l], m->bias[l], cout, conv_out, acc64);
        size_t npix = (size_t)Hc * Wc;
        if (is_last) {
            for (size_t p = 0; p < npix * cout; ++p)  /* activation='sigmoid' */
                conv_out[p] = acc64 ? (float)(1.0 / (1.0 + exp(-(double)conv_out[p])))
                                    : 1.0f / (1.0f + expf(-conv_out[p]));
            if (recon) memcpy(recon, conv_out, npix * cout * sizeof(float));
            if (layer_out && layer_out[l]) memcpy(layer_out[l], conv_out, npix * cout * sizeof(float));
            break;
        }
        relu_bn(conv_out, npix, cout, m->bn_scale[l], m->bn_shift[l]);
        if (is_enc) {
            int po_idx = (ci_idx + 1) % 3;
            maxpool2(conv_out, Hc, Wc, cout, bufs[po_idx]);
            H = Hc / 2; W = Wc / 2;
            cur = bufs[po_idx]; cur_idx = po_idx;
        } else {
            H = Hc; W = Wc;
            cur = conv_out; cur_idx = ci_idx;
        }
        if (layer_out && layer_out[l]) memcpy(layer_out[l], cur, (size_t)H * W * cout * sizeof(float));
        if (l == m->n_enc - 1 && features) memcpy(features, cur, (size_t)H * W * cout * sizeof(float));
    }
}

/* ---------------------------------------------------------------- public: CAE forward */
/* autoencoder.predict + encoder.predict + per-cell MSE/MAE (improved_detection.py:125-131).
 * Any output pointer may be NULL.  layer_out: n_conv pointers (each n * layer size) or NULL.
 * mse/mae: np.mean(np.square(X-R)), np.mean(np.abs(X-R)) over the H*W*1 elements of a cell
 * (:126-127); the fp32 differences are summed in double and rounded once. */
int orc_cae_forward(int H, int W, int n_conv, int n_enc, const int *cin, const int *cout,
                    const float *const *kernels, const float *const *biases,
                    const float *const *bn_scale, const float *const *bn_shift,
                    const float *x, int64_t n, int acc64,
                    float *features, float *recon, float *mse, float *mae,
                    float *const *layer_out)
{
    if (n_conv > ORC_MAX_LAYERS || n_enc >= n_conv) return -1;
    orc_cae m; memset(&m, 0, sizeof m);
    m.H = H; m.W = W; m.n_conv = n_conv; m.n_enc = n_enc;
    for (int l = 0; l < n_conv; ++l) {
        if (cout[l] > 256) return -2;
        m.cin[l] = cin[l]; m.cout[l] = cout[l];
        m.kernel[l] = kernels[l]; m.bias[l] = biases[l];
        m.bn_scale[l] = bn_scale ? bn_scale[l] : NULL;
        m.bn_shift[l] = bn_shift ? bn_shift[l] : NULL;
    }
    /* per-layer stored output sizes (floats per cell) */
    size_t lsz[ORC_MAX_LAYERS]; size_t fsz = 0;
    {
        int h = H, w = W;
        for (int l = 0; l < n_conv; ++l) {
            int ups = (l > n_enc) ? 1 : 0;
            int hc = h << ups, wc = w << ups;
            if (l < n_enc) { h = hc / 2; w = wc / 2; } else { h = hc; w = wc; }
            lsz[l] = (size_t)h * w * cout[l];
            if (l == n_enc - 1) fsz = lsz[l];
        }
    }
    const size_t sf = scratch_floats(&m);
    const size_t npix = (size_t)H * W;
    int fail = 0;
#pragma omp parallel
    {
        float *b0 = (float *)malloc(sf * sizeof(float));
        float *b1 = (float *)malloc(sf * sizeof(float));
        float *b2 = (float *)malloc(sf * sizeof(float));
        float *rec = (float *)malloc(npix * sizeof(float));
        float *lo[ORC_MAX_LAYERS];
        if (!b0 || !b1 || !b2 || !rec) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 4)
            for (int64_t c = 0; c < n; ++c) {
                const float *xc = x + c * npix;
                float **lop = NULL;
                if (layer_out) {
                    for (int l = 0; l < n_conv; ++l)
                        lo[l] = layer_out[l] ? layer_out[l] + (size_t)c * lsz[l] : NULL;
                    lop = lo;
                }
                cae_forward_cell(&m, xc, acc64, features ? features + (size_t)c * fsz : NULL,
                                 rec, lop, b0, b1, b2);
                if (recon) memcpy(recon + c * npix, rec, npix * sizeof(float));
                if (mse || mae) {
                    double s2 = 0.0, s1 = 0.0;
                    for (size_t p = 0; p < npix; ++p) {
                        float d = xc[p] - rec[p];
                        s2 += (double)(d * d);
                        s1 += (double)fabsf(d);
                    }
                    if (mse) mse[c] = (float)(s2 / (double)npix);
                    if (mae) mae[c] = (float)(s1 / (double)npix);
                }
            }
        }
        free(b0); free(b1); free(b2); free(rec);
    }
    return fail ? -3 : 0;
}

/* ---------------------------------------------------------------- public: detector half */
/* scaler.transform (improved_detection.py:134): in-place `X -= center_; X /= scale_` on a
 * float32 array with float32 center_ and float64 scale_ (sklearn _data.py:1715-1718): numpy
 * evaluates the division in double and rounds the quotient to float32.
 * pca.transform (:135): `X @ components_.T - mean_ @ components_.T` in float32
 * (sklearn _base.py:147-155); mean_proj = mean_ @ components_.T is precomputed by the
 * model exporter with the same numpy expression.  BLAS's summation order is unspecified:
 * acc64=0 sums in feature order with fmaf, acc64=1 in double. */
void orc_scaler_pca(const float *feat, int64_t n, int F,
                    const float *center, const double *scale,
                    const float *comps, const float *mean_proj, int C, int acc64,
                    float *scaled_out, float *pca_out)
{
#pragma omp parallel
    {
        float *s = (float *)malloc((size_t)F * sizeof(float));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            const float *f = feat + (size_t)i * F;
            for (int k = 0; k < F; ++k) {
                float t = f[k] - center[k];
                s[k] = (float)((double)t / scale[k]);
            }
            if (scaled_out) memcpy(scaled_out + (size_t)i * F, s, (size_t)F * sizeof(float));
            if (pca_out)
                for (int c = 0; c < C; ++c) {
                    const float *w = comps + (size_t)c * F;
                    float r;
                    if (acc64) {
                        double a = 0.0;
                        for (int k = 0; k < F; ++k) a += (double)s[k] * (double)w[k];
                        r = (float)a;
                    } else {
                        float a = 0.0f;
                        for (int k = 0; k < F; ++k) a = fmaf(s[k], w[k], a);
                        r = a;
                    }
                    pca_out[(size_t)i * C + c] = r - mean_proj[c];
                }
        }
        free(s);
    }
}

/* OneClassSVM.decision_function / predict (improved_detection.py:138-142).
 * libsvm svm_predict_values, ONE_CLASS branch (svm.cpp:2818-2838):
 *   sum_i sv_coef[i] * exp(-gamma * ||x - SV_i||^2) - rho ;  predict = (sum > 0) ? +1 : -1
 * with the RBF distance accumulated over the difference vector (svm.cpp:461-476), inputs
 * cast to double (sklearn svm/_base.py:552,618).  The reference returns score = -dec
 * (:149-150).  Outputs: dec (decision_function), pred (+1/-1). */
void orc_ocsvm_decision(const float *x, int64_t n, int D,
                        const double *sv, const double *coef, int nsv,
                        double gamma, double rho, double *dec, int8_t *pred)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float *xi = x + (size_t)i * D;
        double sum = 0.0;
        for (int j = 0; j < nsv; ++j) {
            const double *s = sv + (size_t)j * D;
            double d2 = 0.0;
            for (int k = 0; k < D; ++k) { double d = (double)xi[k] - s[k]; d2 += d * d; }
            sum += coef[j] * exp(-gamma * d2);
        }
        sum -= rho;
        if (dec) dec[i] = sum;
        if (pred) pred[i] = (sum > 0) ? 1 : -1;
    }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int t)
{
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}
