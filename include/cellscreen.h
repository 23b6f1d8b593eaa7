/*
 * cellscreen.h -- C ABI of libcellscreen.so, the MI355X (gfx950) cell-crop
 * anomaly-screening path.
 *
 * The reference (Kmatsuo57/cell-image-analysis) is two Python classes with no
 * FFI; what this library replaces, entry point by entry point:
 *
 *   cs_model_load / cs_model_from_arrays
 *        ProductionMutantScreening.load_trained_models    improved_detection.py:23-46
 *        (the StarDist fetch at :44 is cell extraction, out of scope)
 *   cs_screen
 *        ProductionMutantScreening.compute_anomaly_scores improved_detection.py:117-153
 *        = autoencoder.predict (:125) + per-cell MSE/MAE (:126-127) + encoder.predict
 *          (:130-131) + scaler.transform (:134) + pca.transform (:135) + 2x OneClassSVM
 *          predict / decision_function (:138-142) + score negation (:149-150)
 *   cs_reconstruct
 *        evaluate_reconstruction_quality numerics   CAE_improved_modeltrain.py:335-339
 *   cs_encode
 *        encoder.predict + flatten      improved_detection.py:130-131, CAE...:401-402
 *   cs_layer_output, cs_scaler_pca, cs_svm_decision
 *        stage-level taps used by the parity tests (oracle inputs to each stage)
 *   cs_preprocess
 *        equalize_adapthist + resize of each crop   improved_detection.py:98-99, CAE...:92-93
 *   cs_fit_scaler / cs_fit_pca_moments / cs_fit_project / cs_fit_ocsvm
 *        RobustScaler / PCA / OneClassSVM fits of create_anomaly_detector
 *                                                 CAE_improved_modeltrain.py:408-427
 *   cs_synth_crops
 *        synthetic U[0,1) crops (no reference counterpart; benchmark/test input)
 *   cs_train_create / cs_train_step / cs_train_eval / cs_train_export
 *        the per-batch work of autoencoder.fit       CAE_improved_modeltrain.py:286-293
 *        on the model compiled at :223-227 (Adam 1e-3, loss 'mse', metric 'mae'); the
 *        callbacks of :263-283 are host-side scalars (cellscreen/training.py)
 *
 * Conventions
 *   - Every function returns CS_OK (0) or a negative cs_status.  cs_last_error()
 *     returns a thread-local message for the most recent failure on this thread.
 *   - Handles are opaque, created and destroyed by the library.  One handle = one
 *     device + one HIP stream + one workspace; a handle is not thread-safe, distinct
 *     handles are independent.  No exceptions or C++ types cross the boundary.
 *   - Every buffer passed in is caller-owned and borrowed for the duration of the call.
 *     `*_kind` says where it lives: CS_MEM_HOST (pageable or pinned host memory) or
 *     CS_MEM_DEVICE (memory of the handle's device, e.g. a torch tensor's data_ptr()).
 *   - Calls are synchronous: outputs are complete when the call returns.
 *   - Device INPUTS must be complete before the call: a handle works on its own non-blocking HIP
 *     stream, which is not ordered with any stream of the caller.  If a CS_MEM_DEVICE buffer (or a
 *     gradient buffer given to cs_train_set_grad_buffer) was produced by work still in flight on
 *     another stream -- a torch kernel, an RCCL collective -- either synchronise that stream, or
 *     call cs_model_wait_stream / cs_train_wait_stream / cs_fit_wait_stream /
 *     cs_preproc_wait_stream with it first (the Python wrappers do, with torch's current stream):
 *     the handle's next work then starts after everything enqueued on that stream so far.
 *   - Layouts: crops [n][H][W] fp32 (the trailing channel of 1 is implicit, as
 *     np.expand_dims at improved_detection.py:122 adds it); conv kernels HWIO
 *     [3][3][cin][cout] as Keras stores them; features [n][h*w*c] in (h,w,c) order
 *     as the reshape at improved_detection.py:131 produces.
 *   - The library links no communication library.  Multi-GPU is one process and one handle per GPU; every
 *     exchange between processes (the gradient all-reduce, the all-gather of BatchNormalization partials,
 *     the gather of results) happens ABOVE this ABI, in the host's own communicator (RCCL through
 *     torch.distributed in cellscreen/dist.py), on buffers the caller owns: cs_train_set_grad_buffer and
 *     cs_train_set_sync_bn are the two hooks.  Screening has no exchange on its data path.
 *   - There is NO CPU fallback: without a gfx950 device every compute entry point
 *     returns CS_ERR_NO_DEVICE.
 */
#ifndef CELLSCREEN_H
#define CELLSCREEN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CS_ABI_VERSION 2       /* 2: cs_model_options (precision) on cs_model_load / cs_model_from_arrays */
#define CS_MAX_CONV 16

typedef enum cs_status {
    CS_OK = 0,
    CS_ERR_INVALID = -1,      /* bad argument (NULL, negative size, wrong shape) */
    CS_ERR_IO = -2,           /* file missing / unreadable */
    CS_ERR_FORMAT = -3,       /* malformed model file */
    CS_ERR_NO_DEVICE = -4,    /* no usable gfx950 device */
    CS_ERR_HIP = -5,          /* HIP runtime error (message in cs_last_error) */
    CS_ERR_UNSUPPORTED = -6,  /* architecture / size this build has no kernel for */
    CS_ERR_NOMEM = -7,
    CS_ERR_NO_DETECTOR = -8   /* model was created without detector parameters */
} cs_status;

typedef enum cs_mem_kind { CS_MEM_HOST = 0, CS_MEM_DEVICE = 1 } cs_mem_kind;

typedef struct cs_model cs_model;

/* One conv autoencoder weight set (CAE_improved_modeltrain.py:184-229).
 * n_conv convs; the first n_enc are each followed by BatchNormalization + MaxPooling2D,
 * the next n_conv-n_enc-1 by BatchNormalization + UpSampling2D, the last has sigmoid.
 * BN arrays are NULL for the last conv. */
typedef struct cs_cae_weights {
    int32_t height, width;            /* input crop size: 64, 64 */
    int32_t n_conv, n_enc;            /* 7, 3 */
    int32_t channels[CS_MAX_CONV];    /* filters of each conv: 32,64,32,32,64,32,1 */
    const float *kernel[CS_MAX_CONV]; /* HWIO [3][3][cin][cout] */
    const float *bias[CS_MAX_CONV];   /* [cout] */
    const float *bn_gamma[CS_MAX_CONV];
    const float *bn_beta[CS_MAX_CONV];
    const float *bn_mean[CS_MAX_CONV];
    const float *bn_var[CS_MAX_CONV];
    float bn_eps;                     /* Keras default 1e-3 */
} cs_cae_weights;

/* One fitted OneClassSVM(kernel='rbf') (CAE_improved_modeltrain.py:420-427). */
typedef struct cs_ocsvm_params {
    int32_t n_sv;
    const double *support_vectors;    /* [n_sv][n_components]  (sklearn support_vectors_) */
    const double *dual_coef;          /* [n_sv]                (dual_coef_[0]) */
    double gamma;                     /* _gamma */
    double rho;                       /* -intercept_[0] == offset_[0] */
} cs_ocsvm_params;

/* RobustScaler + PCA + the two detectors (CAE_improved_modeltrain.py:408-427). */
typedef struct cs_detector_params {
    int32_t n_features;               /* 2048 */
    int32_t n_components;             /* <= 100 */
    const float *scaler_center;       /* [n_features]  center_  (float32) */
    const double *scaler_scale;       /* [n_features]  scale_   (float64) */
    const float *pca_components;      /* [n_components][n_features]  components_ */
    const float *pca_mean_proj;       /* [n_components] = mean_ @ components_.T (float32) */
    cs_ocsvm_params conservative;     /* nu = 0.05 */
    cs_ocsvm_params moderate;         /* nu = 0.10 */
} cs_detector_params;

/* How the fp32 contractions of the convs and of the PCA projection are evaluated.  Either way every tensor, bias,
 * BatchNormalization constant, accumulator and result is fp32 (the SVMs fp64), as in the reference
 * (improved_detection.py:122,125,130: Keras's float32 predict).
 *   CS_PRECISION_SPLIT16     (default) each fp32 operand as a two-term fp16 split (22 of 24 mantissa bits; exact
 *                            power-of-two scales per cell / strip) contracted by three products on the 16-bit
 *                            matrix instructions with fp32 accumulation; the PCA GEMM as a three-term bf16 split
 *                            (24 bits).  Inside every fp32 tolerance of tests/helpers.py; DESIGN.md section 3h.
 *   CS_PRECISION_FP32_EXACT  every contraction on v_mfma_f32_16x16x4_f32: fp32 operands, fp32 products, fp32
 *                            accumulation -- the reference's own arithmetic up to summation order.  ~0.58x the rate. */
typedef enum cs_precision { CS_PRECISION_SPLIT16 = 0, CS_PRECISION_FP32_EXACT = 1 } cs_precision;

/* Debug-only switches (A/B runs and the bit-identity tests): each keeps an UNFUSED form of the same arithmetic. */
#define CS_DEBUG_NO_FUSE12      0x1u   /* conv1 and conv2 as two kernels, p1 through HBM (conv2 as F(2x2,3x3): other bits) */
#define CS_DEBUG_NO_FUSE45      0x2u   /* conv4 and conv5 as two kernels, a4 through HBM (bit-identical) */
#define CS_DEBUG_NO_FUSE67      0x4u   /* conv6, conv7 + error as two kernels, a6 through HBM */
#define CS_DEBUG_NO_SMALL_SPLIT 0x8u   /* small calls keep the one-workgroup detector tail (bit-identical) */

/* Options of cs_model_load / cs_model_from_arrays; NULL = all defaults.  Set struct_size = sizeof(cs_model_options)
 * and zero the rest before filling in what is wanted. */
typedef struct cs_model_options {
    uint32_t struct_size;
    int32_t precision;                /* cs_precision */
    uint32_t debug_flags;             /* CS_DEBUG_* ; 0 in production */
    uint32_t reserved[5];             /* must be 0 */
} cs_model_options;

typedef struct cs_model_info {
    int32_t height, width;
    int32_t n_conv, n_enc;
    int32_t feature_dim;              /* h*w*c of the encoded tensor */
    int32_t n_components;
    int32_t n_sv_conservative, n_sv_moderate;
    int32_t shared_encoder;           /* 1: encoder weights are bit-identical to the
                                         autoencoder's encoder half, so one pass serves both */
    int32_t has_detector;
    int32_t device_id;
    int64_t chunk_cells;              /* cells processed per internal pass */
    int32_t channels[CS_MAX_CONV];    /* filters of each conv */
    int32_t reference_arch;           /* 1: the reference graph (tuned kernels); 0: generic-shape kernels */
    int32_t precision;                /* cs_precision the handle was created with */
    uint32_t debug_flags;
} cs_model_info;

/* ---- library / device ---------------------------------------------------------- */
int cs_abi_version(void);
const char *cs_status_string(int status);
const char *cs_last_error(void);
/* Number of visible HIP devices (0 if none).  Never fails. */
int cs_device_count(void);

/* ---- model --------------------------------------------------------------------- */
/* Reads <model_dir>/cae.bin (+ detector.bin if present) in the native tensor-archive
 * format written by cellscreen.model_io (see DESIGN.md "model_dir").
 * Replaces load_trained_models, improved_detection.py:23-46. */
int cs_model_load(const char *model_dir, int device_id, const cs_model_options *options, cs_model **out);

/* autoencoder: weights of best_autoencoder.keras (improved_detection.py:28).
 * encoder:     weights of encoder.keras (:29), n_conv = n_enc convs; NULL = same as the
 *              autoencoder's encoder half.  The two files may differ
 *              (CAE_improved_modeltrain.py:270-275 vs :300).
 * detector:    may be NULL; then cs_screen returns CS_ERR_NO_DETECTOR.
 * Architectures: the reference graph (64x64, filters 32-64-32 | 32-64-32-1) runs on kernels tuned for
 * it.  Any other instance of the same layer grammar -- create_improved_autoencoder(input_shape) is
 * generic in its input size (CAE_improved_modeltrain.py:184), e.g. BASELINE.json configs[4]: 128x128,
 * filters 32-64-128 | 128-64-32-1 -- runs on run-time-shaped MFMA kernels (csrc/conv_generic.hip):
 * n_conv = 2 n_enc + 1, last conv 1 filter, every conv grid's width a multiple of 16 and <= 128,
 * channel counts multiples of 4.  Anything else: CS_ERR_UNSUPPORTED.  Training handles (cs_train_*) take
 * the same architectures (cs_train_create). */
int cs_model_from_arrays(const cs_cae_weights *autoencoder, const cs_cae_weights *encoder,
                         const cs_detector_params *detector, int device_id, const cs_model_options *options,
                         cs_model **out);
void cs_model_free(cs_model *m);
/* Orders the handle's stream after all work enqueued so far on `hip_stream` (a hipStream_t; NULL = the legacy
 * default stream).  See "Device INPUTS" above.  Same for the three other handle types. */
int cs_model_wait_stream(cs_model *m, void *hip_stream);
int cs_model_get_info(const cs_model *m, cs_model_info *info);
/* Cells per internal pass (workspace ~0.4 MB per cell for the reference graph).  Default: automatic -- 16,384 for
 * host input (pipelined staging), up to 65,536 for device-resident input (a ~19 GB workspace for the reference graph);
 * this call fixes it, 0 returns to automatic.  The workspace is sized for min(n, chunk) cells of the largest call so far. */
int cs_model_set_chunk(cs_model *m, int64_t chunk_cells);

/* ---- the hot path -------------------------------------------------------------- */
/* compute_anomaly_scores, improved_detection.py:117-153, for n crops.
 * Outputs (each may be NULL to skip), all length n:
 *   mse, mae            reconstruction_mse / reconstruction_mae (float32)
 *   cons_score, mod_score   -decision_function (float64; "higher = more anomalous", :149-150)
 *   cons_pred, mod_pred     predict: +1 inlier / -1 anomaly (:138-139; libsvm dec > 0 ? 1 : -1)
 * n == 0 is valid and touches nothing (the reference returns {} at :119-120). */
int cs_screen(cs_model *m, const float *crops, int64_t n, int crops_kind,
              float *mse, float *mae, double *cons_score, double *mod_score,
              int8_t *cons_pred, int8_t *mod_pred, int out_kind);

/* autoencoder.predict + MSE/MAE (CAE_improved_modeltrain.py:335-339).
 * recon [n][H][W] may be NULL. */
int cs_reconstruct(cs_model *m, const float *crops, int64_t n, int crops_kind,
                   float *recon, float *mse, float *mae, int out_kind);

/* encoder.predict + reshape (improved_detection.py:130-131).
 * which = 0: the autoencoder's encoder half; 1: the encoder.keras weight set. */
int cs_encode(cs_model *m, const float *crops, int64_t n, int crops_kind, int which,
              float *features, int out_kind);

/* ---- stage taps for parity tests ------------------------------------------------ */
/* Output of conv `layer` (0-based) of the autoencoder after its relu/BN/pool (the tensor
 * the next conv reads), NHWC; for the last conv the sigmoid output.  out: n * layer size. */
int cs_layer_output(cs_model *m, const float *crops, int64_t n, int crops_kind, int layer,
                    float *out, int out_kind);
/* scaler.transform + pca.transform on caller-supplied features [n][n_features]. */
int cs_scaler_pca(cs_model *m, const float *features, int64_t n, int in_kind,
                  float *pca_out /* [n][n_components] */, int out_kind);
/* decision_function of both detectors on caller-supplied PCA vectors [n][n_components]. */
int cs_svm_decision(cs_model *m, const float *pca, int64_t n, int in_kind,
                    double *cons_dec, double *mod_dec, int out_kind);

/* ---- synthetic input ------------------------------------------------------------ */
/* Fills out_device[n][npix] with U[0,1) fp32 from the counter-based generator keyed
 * (seed, first_cell + i, pixel); bit-identical to oracle/cae_oracle.c:orc_synth_crops. */
int cs_synth_crops(cs_model *m, uint64_t seed, int64_t first_cell, int64_t n, int32_t npix,
                   float *out_device);

/* ---- crop preprocess (the caller side of the hot path) ----------------------------- */
/* What the reference does to every bounding-box crop before compute_anomaly_scores sees it
 * (improved_detection.py:98-99, CAE_improved_modeltrain.py:92-93):
 *     exposure.equalize_adapthist(crop, clip_limit=0.02)  ->  resize(., (64, 64), anti_aliasing=True)
 * and the float32 cast of improved_detection.py:122.  Arithmetic of scikit-image 0.18.3 /
 * SciPy 1.7.1 (CLAHE bit-exact, resize in fp64).  Independent of cs_model: own handle, own stream. */
typedef struct cs_preproc cs_preproc;
typedef enum cs_pixel_type { CS_PIX_U8 = 0, CS_PIX_U16 = 1 } cs_pixel_type;   /* TIFF channel dtypes */

int cs_preproc_create(int device_id, cs_preproc **out);
void cs_preproc_free(cs_preproc *p);
int cs_preproc_wait_stream(cs_preproc *p, void *hip_stream);
/* pixels:  ragged buffer of n_pixels elements (pixels_kind: host or device); crop i is the
 *          row-major heights[i] x widths[i] block at element offsets[i].  offsets must ascend
 *          and crops must not overlap.  offsets/heights/widths are host arrays of length n.
 * Sides below 8 return CS_ERR_INVALID (kernel_size = shape // 8 would be 0: skimage raises),
 * above 1024 CS_ERR_UNSUPPORTED.
 * out:       [n][64][64] float32 (out_kind: host or device).
 * clahe_out: optional stage tap, same layout/offsets as pixels (uint16, out_kind): the image
 *            skimage's _clahe returns before the final rescale.  Elements between crops: 0 / untouched.
 * n == 0 is valid and touches nothing. */
int cs_preprocess(cs_preproc *p, const void *pixels, int pixel_type, int64_t n_pixels, int pixels_kind,
                  const int64_t *offsets, const int32_t *heights, const int32_t *widths, int64_t n,
                  double clip_limit, float *out, uint16_t *clahe_out, int out_kind);
/* Device time (HIP events on the handle's stream around the kernel launches) and the pixel count
 * of the last cs_preprocess call. */
int cs_preproc_last_timing(const cs_preproc *p, double *kernel_ms, int64_t *pixels);

/* ---- detector fitting (create_anomaly_detector, CAE_improved_modeltrain.py:394-446) -------- */
/* The fit of what cs_screen's tail evaluates, for the training set's encoder features
 * (cs_encode output, [n][n_features] fp32, host or device).  Own handle, own stream.
 *   cs_fit_scaler       RobustScaler().fit (:408-409): center_ = per-feature median (float32), scale_ = 75th - 25th
 *                       percentile (float64, numpy's linear interpolation; spreads below 10 eps become 1).  Exact
 *                       (radix select of the order statistics + numpy's arithmetic on them).  NaNs: CS_ERR_UNSUPPORTED.
 *   cs_fit_pca_moments  what PCA(...).fit (:412-414) needs from the data: mean_ (float32, numpy's row-after-row sum)
 *                       of the scaled features and the fp64 scatter matrix Xc^T Xc of the centred scaled features
 *                       ([F][F], host).  The principal axes are the leading eigenvectors of scatter / (n - 1); the
 *                       F x F eigenproblem is the host's (cellscreen/detector_fit.py uses LAPACK through numpy).
 *                       n_features must be a multiple of 128, at most 8192.
 *   cs_fit_project      scaler.transform + pca.transform of the training features with freshly fitted parameters
 *                       (the kernel cs_screen uses), out [n][n_components] fp32 on the host.
 *   cs_fit_ocsvm        OneClassSVM(kernel='rbf', nu).fit (:420-427): libsvm's SMO (second-order working-set selection,
 *                       Qfloat kernel rows, eps stopping rule) with every iteration on the device.  x: [n][n_components]
 *                       float64 host (what sklearn hands libsvm), gamma as resolved by sklearn ('scale':
 *                       1 / (n_components * x.var())), eps = tol (sklearn default 1e-3), max_iter < 0 = unbounded.
 *                       alpha: [n] dual variables (support vectors are the points with alpha > 0; dual_coef_ = alpha),
 *                       rho = -intercept_ = offset_.  status: 0 converged, 1 stopped at max_iter. */
typedef struct cs_fit cs_fit;
int cs_fit_create(int device_id, cs_fit **out);
void cs_fit_free(cs_fit *f);
int cs_fit_wait_stream(cs_fit *f, void *hip_stream);
int cs_fit_scaler(cs_fit *f, const float *features, int64_t n, int32_t n_features, int kind, float *center, double *scale);
int cs_fit_pca_moments(cs_fit *f, const float *features, int64_t n, int32_t n_features, int kind, const float *center,
                       const double *scale, float *mean, double *scatter);
int cs_fit_project(cs_fit *f, const float *features, int64_t n, int32_t n_features, int kind, const float *center,
                   const double *scale, const float *components, const float *mean_proj, int32_t n_components, float *out);
int cs_fit_ocsvm(cs_fit *f, const double *x, int64_t n, int32_t n_components, double gamma, double nu, double eps,
                 int64_t max_iter, double *alpha, double *rho, double *obj, int64_t *n_iter, int32_t *status);
/* Device time (HIP events on the handle's stream) of the last cs_fit_* call, transfers of results included. */
int cs_fit_last_ms(const cs_fit *f, double *device_ms);

/* ---- measurement ---------------------------------------------------------------- */
/* When enabled, every kernel launch of this handle is bracketed by HIP events on the
 * handle's stream; totals are per kernel family.  Adds a stream sync per call. */
int cs_profile_enable(cs_model *m, int on);
int cs_profile_reset(cs_model *m);
int cs_profile_kernel_count(void);
const char *cs_profile_kernel_name(int kernel_id);
/* total_ms: summed device time; launches: number of launches; cells: cells processed;
 * flops: algorithmic FLOPs of those launches (2 x MACs of the reference graph). */
int cs_profile_get(cs_model *m, int kernel_id, double *total_ms, int64_t *launches,
                   int64_t *cells, double *flops);
/* Matrix-pipe instructions (v_mfma_f32_16x16x4_f32, 2,048 FLOP each) one cell costs in that kernel family with
 * the kernels this handle runs: the EXECUTED work a roofline fraction is priced with (Winograd / folded-upsample
 * kernels execute fewer multiply-adds than the layer's algorithmic count).  Equals SQ_INSTS_MFMA per cell. */
int cs_profile_mfma_per_cell(cs_model *m, int kernel_id, double *mfma);
/* The same for kernels that carry the fp32 contraction on the bf16 matrix pipe (three-way operand split, six
 * products: conv4 of the reference graph, the MFMA convs of other architectures): v_mfma_f32_16x16x32_bf16
 * instructions (16,384 FLOP each) per cell; such a kernel reports 0 from cs_profile_mfma_per_cell.
 * CS_NO_BF16X3=1 in the environment keeps every conv on the fp32 matrix instructions (A/B timing). */
int cs_profile_bf16_mfma_per_cell(cs_model *m, int kernel_id, double *mfma);

/* ---- training ---------------------------------------------------------------------- */
typedef struct cs_trainer cs_trainer;

/* Keras defaults of the reference: Adam(beta_1 0.9, beta_2 0.999, epsilon 1e-7)
 * (CAE_improved_modeltrain.py:224), BatchNormalization(momentum 0.99, epsilon 1e-3) (:192). */
typedef struct cs_train_cfg {
    float beta1, beta2, adam_eps;
    float bn_momentum, bn_eps;
} cs_train_cfg;

/* Number of trainable parameters (84,289) and of BatchNormalization moving statistics (512).
 * Flat layouts -- trainable: per conv l in order {kernel HWIO, bias, [gamma, beta]};
 * moving: per BN l in order {moving_mean, moving_variance}. */
int cs_train_param_count(int64_t *n_trainable, int64_t *n_moving);
/* The same two counts for the architecture of a handle (any instance of the layer grammar). */
int cs_train_param_count_of(const cs_trainer *t, int64_t *n_trainable, int64_t *n_moving);
/* init: starting weights + moving statistics (create_improved_autoencoder, :184-229).  The reference graph trains on
 * kernels tuned for it; any other instance of the layer grammar (see cs_model_from_arrays; additionally every filter
 * count but the last must divide 256) trains on run-time-shaped kernels (csrc/train_generic.hip) -- BASELINE.json
 * configs[4]'s 128x128 / 128-channel variant with the same forward_backward -> all-reduce -> apply split. */
int cs_train_create(const cs_cae_weights *init, const cs_train_cfg *cfg, int device_id, cs_trainer **out);
void cs_train_free(cs_trainer *t);
int cs_train_wait_stream(cs_trainer *t, void *hip_stream);
/* One fit() batch: forward (BN batch statistics, moving-average update), loss = mean((out-y)^2),
 * mae = mean|out-y|, backward, Adam update with learning rate lr.  x = network input (the
 * augmented image of datagen.flow(X_train, X_train), :287), y = target; [batch][H][W] fp32. */
int cs_train_step(cs_trainer *t, const float *x, const float *y, int64_t batch, int kind, float lr,
                  float *loss, float *mae);
/* The same batch with NO host synchronisation: copies, forward, backward, Adam and the operand re-pack are enqueued on the
 * handle's stream and the call returns.  The batch's loss / mae are added to running sums on the device -- Keras's epoch
 * metrics are the means over the epoch's batches (fit(), CAE_improved_modeltrain.py:286-293) -- and cs_train_read_metrics
 * fetches them: one host round trip per epoch instead of one per step.  x / y must stay valid until the step's input copies
 * have run; cs_train_inputs_consumed(t, stream) makes `stream` (the caller's, e.g. torch's current stream) wait for exactly
 * that point, and likewise for the device output of the last cs_train_augment.  Host batches and run-time-shaped
 * architectures fall back to a synchronous step whose scalars are added on the host. */
int cs_train_step_async(cs_trainer *t, const float *x, const float *y, int64_t batch, int kind, float lr);
int cs_train_inputs_consumed(cs_trainer *t, void *hip_stream);
/* Mean loss / mae over the cs_train_step_async calls since the last reset, their number; reset != 0 clears the sums.
 * Synchronises the handle's stream. */
int cs_train_read_metrics(cs_trainer *t, double *loss_mean, double *mae_mean, int64_t *steps, int reset);
/* The two halves of cs_train_step, for data-parallel training: gradients are left in the
 * gradient buffer (all-reduce it between the two calls). */
int cs_train_forward_backward(cs_trainer *t, const float *x, const float *y, int64_t batch, int kind,
                              float *loss, float *mae);
int cs_train_apply(cs_trainer *t, float lr);
/* BatchNormalization over the WHOLE batch when the batch is split over `world` processes (the reference normalises over its
 * single batch of 32, CAE_improved_modeltrain.py:192-213 with batch_size=32 at :287).  cs_train_forward_backward then stops at
 * two points per BN layer -- after the layer's local {count, mean, M2} per channel (forward) and after its local
 * {sum dy, sum dy xhat} (backward) -- writes this rank's floats_per_rank values at device_buf + rank * floats_per_rank,
 * synchronises its stream and calls fn(ctx, floats_per_rank): the caller all-gathers in place (RCCL / torch.distributed; a
 * fake communicator in tests) and returns 0 once device_buf[0 .. world * floats_per_rank) is complete and visible to the
 * device.  The library merges the `world` triples in rank order (Chan's formula, double), so every rank gets identical
 * statistics, identical moving averages, and -- with the usual mean of the per-rank weight gradients -- the gradient of the
 * single-process step.  capacity_floats >= world * 3 * (largest filter count).  fn == NULL switches it off.  Reference graph
 * only (CS_ERR_UNSUPPORTED otherwise). */
typedef int (*cs_allgather_fn)(void *ctx, int64_t floats_per_rank);
int cs_train_set_sync_bn(cs_trainer *t, cs_allgather_fn fn, void *ctx, float *device_buf, int64_t capacity_floats, int rank, int world);
/* Use caller-owned device memory (n_trainable floats, e.g. a torch tensor) as the gradient
 * buffer; NULL restores the internal one. */
int cs_train_set_grad_buffer(cs_trainer *t, float *device_buffer);
/* Inference-mode loss / mae over n cells with the moving statistics (fit()'s validation pass). */
int cs_train_eval(cs_trainer *t, const float *x, const float *y, int64_t n, int kind, float *loss, float *mae);
/* Training augmentation: ImageDataGenerator(rotation_range=2, width/height_shift_range=0.02,
 * zoom_range=0.02, horizontal/vertical_flip, fill_mode='nearest') applied to the INPUT batch only
 * (CAE_improved_modeltrain.py:246-254, datagen.flow(X_train, X_train) at :287).
 * One transform per image, already reduced on the host to what
 * scipy.ndimage.affine_transform(order=1, mode='nearest') takes: input coordinate =
 * m . (row, col) + off, built in double exactly as Keras's apply_affine_transform does
 * (cellscreen/augment.py draws the parameters in Keras's order).  identity != 0 skips the
 * resampling (Keras does when no rotation/shift/zoom was drawn); flips come last. */
typedef struct cs_aug_affine {
    double m[4];      /* row-major 2x2 */
    double off[2];
    int32_t identity, flip_h, flip_v, reserved;
} cs_aug_affine;
/* x, out: [n][64][64] fp32, both `kind`; tf: host array of n transforms (copied before the call returns).  out may not
 * alias x.  A CS_MEM_DEVICE output is left on the handle's stream (the next cs_train_step* consumes it there); any other
 * reader orders itself with cs_train_inputs_consumed or a synchronising call.  CS_MEM_HOST returns when out is written. */
int cs_train_augment(cs_trainer *t, const float *x, int64_t n, const cs_aug_affine *tf, float *out, int kind);

/* The generator's settings (Keras names; CAE_improved_modeltrain.py:246-254 is {2, 0.02, 0.02, 0.98, 1.02, 1, 1, -0.5}).
 * Shift ranges below 1 are fractions of the image side, as in Keras; center: the transform's centre is size/2 + center. */
typedef struct cs_aug_config {
    double rotation_range;                        /* degrees: theta ~ U(-r, r) */
    double width_shift_range, height_shift_range;
    double zoom_lo, zoom_hi;                      /* zx, zy ~ U(lo, hi) each */
    int32_t horizontal_flip, vertical_flip;
    double center;
} cs_aug_config;
/* The n transforms of fit() step `step`: image b's seven draws (theta, tx = height shift, ty = width shift, zx, zy, flip_h,
 * flip_v -- Keras's get_random_transform order) are u(seed, step, b, j), j = 0..6, of a counter-based generator (three rounds
 * of splitmix64 over the key; cellscreen/augment.py has the same function), so a step's augmentation does not depend on how many
 * were drawn before it or on which rank draws it.  Host only, no device needed. */
int cs_train_draw_transforms(const cs_aug_config *aug, uint64_t seed, uint64_t step, int64_t n, int32_t height, int32_t width,
                             cs_aug_affine *out);
/* One fit() batch straight from a training set resident on the handle's device (CAE_improved_modeltrain.py:286-293:
 * datagen.flow(X_train, X_train, batch_size=32) -> one step of model.fit): gathers train[idx[b]], b < batch, draws the
 * batch's transforms (cs_train_draw_transforms; aug == NULL: none), resamples the INPUT only -- the target stays the original
 * crop -- and enqueues forward + backward + Adam like cs_train_step_async: no host synchronisation, no other call, no
 * intermediate tensor of the caller's.  idx: host array (copied before the call returns).  Metrics: cs_train_read_metrics. */
int cs_train_fit_step(cs_trainer *t, const float *train_device, int64_t n_train, const int32_t *idx, int64_t batch,
                      const cs_aug_config *aug, uint64_t seed, uint64_t step, float lr);

/* Copies to host (each pointer may be NULL): trainable parameters, moving statistics, last gradients. */
int cs_train_export(cs_trainer *t, float *params_host, float *moving_host, float *grads_host);
/* Stage tap for parity tests: copies one tensor of the last forward_backward to host.
 * which: 0 relu(conv) output, 1 BN(+pool) output, 2 dL/dz, 3 dL/d(BN output), 4 sigmoid output. */
int cs_train_tensor(cs_trainer *t, int which, int layer, int64_t batch, float *host);
/* Overwrites trainable parameters and/or moving statistics from host arrays (restore best weights). */
int cs_train_import(cs_trainer *t, const float *params_host, const float *moving_host);

#ifdef __cplusplus
}
#endif
#endif /* CELLSCREEN_H */
