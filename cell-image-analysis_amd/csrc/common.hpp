// Shared declarations for libcellscreen's HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/cellscreen.h"

namespace cs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- the two-term fp16 split x S = hi + lo on the mixed-precision fma ------------------------------------------
// hi = fp16(x S) and lo = fp16(x S - hi) are ONE v_fma_mix*_f16 each (the product with a power of two and the residual are
// exact in fp32, so the instruction's single rounding is the conversion's): 2 VALU per value where scale / convert / convert
// back / subtract / convert took 3 to 4.  tools/microbench/fp16_split_probe.hip: bit-identical on 16.7 M values x 3 scales,
// except that fma(-0, S, +0) is +0 where the conversion kept -0.
// The instructions sit in inline asm (there is no builtin), and the hazard recogniser does not look inside: ordinary VALU
// consumers are interlocked by the hardware, but a DPP / MFMA read of a result needs two wait states.  Used where the split
// is a phase of its own (conv12_fused.hip: P2 3.6 k -> 3.0 k cycles per group, the kernel 83.2 -> 79.5 ms per 1 M cells).  In
// the kernels that split while staging under MFMAs (conv3, conv4, conv5, conv6) the same helpers measured SLOWER on the same
// box (conv3 24.2 -> 25.2 ms, conv5 15.6 -> 16.4): opaque asm blocks cost the compiler more scheduling freedom than the saved
// instructions return; those keep the convert / subtract form.
// One value -> the dword [fp16(x S) | fp16(x S - hi)] (conv1's crop records).
__device__ __forceinline__ unsigned int f16x2_split_word_scaled(float x, float S)
{
    unsigned int pk;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0\n\t"
        "v_fma_mixhi_f16 %0, %1, %2, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]"
        : "=&v"(pk) : "v"(x), "v"(S));
    return pk;
}
// Six values, already scaled -> six dwords for conv2's V (conv12_fused.hip, P2): [hi | lo] of each value, exchanged with the
// neighbouring lane (quad_perm [1,0,3,2]) and merged by `sel` (even lane [hi c | hi c+1], odd lane [lo c-1 | lo c]).  The DPP
// read of a VALU result needs two wait states: the six independent chains are interleaved so that five instructions separate them.
__device__ __forceinline__ void f16x2_split6_exchange(const float (&v)[6], unsigned int sel, unsigned int (&out)[6])
{
    unsigned int p0, p1, p2, p3, p4, p5;
    asm("v_cvt_f16_f32 %6, %12\n\t"
        "v_cvt_f16_f32 %7, %13\n\t"
        "v_cvt_f16_f32 %8, %14\n\t"
        "v_cvt_f16_f32 %9, %15\n\t"
        "v_cvt_f16_f32 %10, %16\n\t"
        "v_cvt_f16_f32 %11, %17\n\t"
        "v_fma_mixhi_f16 %6, %6, -1.0, %12 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %7, %7, -1.0, %13 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %8, %8, -1.0, %14 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %9, %9, -1.0, %15 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %10, %10, -1.0, %16 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %11, %11, -1.0, %17 op_sel_hi:[1,0,0]\n\t"
        "v_mov_b32_dpp %0, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %1, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %3, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %4, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %5, %11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_perm_b32 %0, %0, %6, %18\n\t"
        "v_perm_b32 %1, %1, %7, %18\n\t"
        "v_perm_b32 %2, %2, %8, %18\n\t"
        "v_perm_b32 %3, %3, %9, %18\n\t"
        "v_perm_b32 %4, %4, %10, %18\n\t"
        "v_perm_b32 %5, %5, %11, %18"
        : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]),
          "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(sel));
}

// Geometry of one conv layer of the reference graph (CAE_improved_modeltrain.py:191-216)
// as the kernels see it.  H, W: conv grid (= conv output, pre-pool) size.
struct LayerGeom {
    int H, W, cin, cout;
    bool pool;   // MaxPooling2D follows (encoder)
    bool ups;    // the stored input is (H/2, W/2): UpSampling2D precedes this conv
    bool last;   // sigmoid output conv
};

// Kernel families, for cs_profile_*.
enum KernelId {
    K_CONV1 = 0, K_CONV2, K_CONV3, K_CONV4, K_CONV5, K_CONV6, K_CONV7_ERR,
    K_SCALER_PCA, K_SVM, K_FINALIZE, K_SYNTH, K_CONV67_FUSED, K_CONV12_FUSED, K_COUNT
};

// ---- launchers (each enqueues on `stream`, returns hipGetLastError()) -------------
// conv layers 1..6 of the 64x64 reference graph; `layer` is 0-based.
// in/out NHWC fp32; wfrag = MFMA B-operand fragments built by pack_conv_fragments();
// ep = [3][cout] {bias, bn_scale, bn_shift}.
// folded = true (layers 4 and 5 only): wfrag comes from pack_conv_fragments_folded and the
// upsample is folded into four 2x2-tap phase convs (4/9 of the MACs).
hipError_t launch_conv_mfma(int layer, const float* in, const float* wfrag, const float* ep,
                            float* out, int64_t n_cells, hipStream_t stream, bool folded = false);
size_t pack_conv_fragments_folded(int cin, int cout, const float* hwio, float* dst);
// conv2 (layer 1) / conv3 (layer 2) as Winograd with the transform domain split by column over the waves: conv_wino_cs.hip
hipError_t launch_conv_wino_cs(int layer, const float* in, const float* ufrag, const float* ep, float* out, int64_t n_cells,
                               hipStream_t stream);
size_t pack_wino_cs_fragments(int layer, const float* hwio, float* dst);
// conv3 (layer 2) in the same Winograd form with the contraction as a two-term fp16 split (three products, two workgroups per CU):
// planes + 1 / (their power-of-two scale)
hipError_t launch_conv3_wino_h2(const float* in, const uint16_t* uplanes, float inv_sw, const float* ep, float* out, int64_t n_cells,
                                hipStream_t stream);
size_t pack_wino3_h2(const float* hwio, uint16_t* dst, float* inv_sw);
// conv1 + conv2 in one kernel (crop -> p2), conv2 as Winograd F(4x4,3x3): conv12_fused.hip.  w1frag comes from
// pack_conv12_conv1_fragments, ep1 / ep2 are the layers' [3][cout] epilogue arrays, ufrag comes from pack_conv12_fragments (conv2's HWIO kernel).
// CS_PRECISION_SPLIT16: ufrag_h2 (pack_conv12_fragments_h2) and w1h2 (pack_conv12_conv1_h2) both non-null -- conv2's and conv1's
// contractions as two-term fp16 splits; p1a / p1b = pack_conv12_p1_bound, inv_sw / inv_sw1 = 1 / the scales of ufrag_h2 / w1h2.
// Both null: everything on the fp32 matrix instructions.
hipError_t launch_conv12_fused(const float* x, const float* w1frag, const float* ep1, const float* ufrag, const float* ep2, float* p2,
                               int64_t n_cells, hipStream_t stream, const unsigned int* ufrag_h2 = nullptr,
                               float p1a = 0.0f, float p1b = 0.0f, float inv_sw = 1.0f, const unsigned int* w1h2 = nullptr, float inv_sw1 = 1.0f);
size_t pack_conv12_conv1_h2(const float* hwio, const float* bn_scale, unsigned int* dst, float* inv_sw1);     // returns 32-bit words
size_t pack_conv12_fragments_h2(const float* hwio, const float* bn_scale, unsigned int* dst, float* inv_sw);     // returns 32-bit words
void pack_conv12_p1_bound(const float* hwio1, const float* ep1, float* a1, float* b1);
size_t pack_conv12_fragments(const float* hwio, const float* bn_scale, float* dst);
size_t pack_conv12_conv1_fragments(const float* hwio, const float* bn_scale, float* dst);
// conv4 / conv5 as a two-term fp16 split (three products; conv45_h2.hip): planes + 1 / (their power-of-two scale)
hipError_t launch_conv4_h2(const float* in, const uint16_t* wfrag, float inv_sw, const float* ep, float* out, int64_t n_cells, hipStream_t stream);
size_t pack_conv4_f16x2(const float* hwio, uint16_t* dst, float* inv_sw);
hipError_t launch_conv5_h2(const float* in, const uint16_t* wfrag, float inv_sw, const float* ep, float* out, int64_t n_cells, hipStream_t stream);
// conv4 + conv5 as one kernel (a4 stays in LDS): in = p3, out = a5; the two layers' own packed planes and epilogue tables
hipError_t launch_conv45_h2(const float* in, const uint16_t* w4, float inv_sw4, const float* ep4, const uint16_t* w5, float inv_sw5,
                            const float* ep5, float* out, int64_t n_cells, hipStream_t stream);
size_t pack_conv5_f16x2(const float* weff, uint16_t* dst, float* inv_sw);
// conv5 (layer 4) / conv6 (layer 5), the upsample-fed decoder convs, as four Winograd F(2x2,2x2) phase convs: conv_wino_up.hip
hipError_t launch_conv_wino_up(int layer, const float* in, const float* ufrag, const float* ep, float* out, int64_t n_cells,
                               hipStream_t stream);
size_t pack_wino_up_fragments(int layer, const float* hwio, float* dst);
// Host-side packing of HWIO weights into the per-lane B fragments of launch_conv_mfma.
// Returns the number of floats written (or required if dst == nullptr).
size_t pack_conv_fragments(int cin, int cout, const float* hwio, float* dst);

// conv7 + sigmoid + per-cell squared/absolute error partial sums.
// a6: [n][32][32][32]; x: [n][64][64] (the array the reconstruction is compared with);
// conv6 + conv7 + reconstruction error fused (the screening path): a5 -> per-cell error partials
// errpart[n][conv67_fused_nparts()][2]; neither a6 nor the reconstruction is written.
hipError_t launch_conv67_fused(const float* a5, const float* ufrag, const float* ep, const float* x, const float* weff_dev,
                               const float* b7_dev, float* errpart, int64_t n_cells, hipStream_t stream);
int conv67_fused_nparts();
// the same kernel with conv6 (folded direct form) as a TWO-term fp16 split (three products, power-of-two operand scales; conv_wino_up.hip has the algebra):
// wplanes = pack_conv6_f16x2(pack_generic_folded(64, 32, hwio, .), ., &inv_sw)
hipError_t launch_conv67_h2(const float* a5, const uint16_t* wplanes, float inv_sw, const float* ep, const float* x, const float* weff_dev,
                            const float* b7_dev, float* errpart, int64_t n_cells, hipStream_t stream);
size_t pack_conv6_f16x2(const float* weff, uint16_t* dst, float* inv_sw);
// helpers of the fp16-split packers: the power of two that puts max|w| into [2^14, 2^15), and one value's two fp16 terms
float f16x2_weight_scale(const float* w, size_t n);
void f16x2_split(float w, float S, uint16_t& hi, uint16_t& lo);

// weff_dev: device, [16][32] effective weights (conv7_effective_weights / launch_pack_w7eff);
// b7_dev: device, the conv's bias; errpart: [n][4][2]; recon (may be null): [n][64][64].
hipError_t launch_conv7_err(const float* a6, const float* x, const float* weff_dev, const float* b7_dev,
                            float* errpart, float* recon, int64_t n_cells, hipStream_t stream);
// the training step's form: also dz = dL/dz of the sigmoid conv for L = mean((out - x)^2) over n*64*64 elements, and its sum per
// (cell, strip) in dzsum_part [n][4] (the bias gradient's partial sums) -- what loss_dz_kernel computed from a stored reconstruction
hipError_t launch_conv7_err_train(const float* a6, const float* x, const float* weff_dev, const float* b7_dev, float* errpart,
                                  float* recon, float* dz, float* dzsum_part, int64_t n_cells, hipStream_t stream);
void conv7_effective_weights(const float* w7_hwio, float* weff);

// ---- training (train.hip, conv_mfma.hip) -----------------------------------------------
constexpr int TRAIN_MAX_PARTS = 256;   // workgroups (= partial sums) per weight-gradient reduction
constexpr int BN_MAX_PARTS = 1024;     // workgroups per BatchNorm statistics / backward pass: these kernels stream whole
                                       // tensors with ~1 element in flight per thread, so they need every SIMD several waves deep
struct ReduceDesc { long dst; long len; const float* src; int nparts; long stride; };
hipError_t launch_conv_train_fwd(int layer, const float* in, const float* wfrag, const float* bias,
                                 float* relu_out, int64_t n_cells, hipStream_t stream, float* stats_part = nullptr,
                                 int* stats_parts = nullptr);
hipError_t launch_conv_dgrad(int layer, const float* dz, const float* wfrag_t, float* dx, int64_t n_cells,
                             hipStream_t stream);
hipError_t launch_bn_stats(const float* r, long P, int C, float* part, int* G, hipStream_t s);
hipError_t launch_bn_stats_final(const float* part, int G, int C, float eps, float momentum, float* mov_mean,
                                 float* mov_var, float* stats, hipStream_t s);
hipError_t launch_bn_stats_merge(const float* part, int G, int C, float* out /*[3][C]*/, hipStream_t s);
hipError_t launch_bn_apply(const float* r, int C, const float* gamma, const float* beta, const float* stats, float* a,
                           long N, int H, int W, int pool, hipStream_t s);
// errpart / nparts / out2 (optional): also reduce the error partial sums of the forward pass to {loss, mae} (saves a launch)
hipError_t launch_loss_dz(const float* out, const float* y, long total, float* dz, float* dzsum_part, int* G, hipStream_t s,
                          const float* errpart = nullptr, long nparts = 0, float* out2 = nullptr);
hipError_t launch_loss_scalar(const float* errpart, long nparts, long nelem, float* out2, hipStream_t s);
hipError_t launch_bn_bwd_reduce(const float* da, const float* r, const float* stats, const float* gamma, const float* beta,
                                long N, int H, int W, int C, int pool, float* part, int* G, hipStream_t s);
hipError_t launch_bn_bwd_final(const float* part, int G, int C, double nred, float* sums, float* dgamma, float* dbeta,
                               hipStream_t s);
hipError_t launch_bn_bwd_dz(const float* da, const float* r, const float* stats, const float* gamma, const float* beta,
                            const float* sums, long N, int H, int W, int C, int pool, float* dz, float* dzsum_part,
                            int* Gz, hipStream_t s);
hipError_t launch_wgrad(int layer, const float* xin, const float* dz, float* part, int64_t n_cells, int* nparts,
                        hipStream_t s);
// errpart / nparts / nelem / out2 (optional): thread 0 also reduces the forward pass's error partial sums to {loss, mae}
hipError_t launch_reduce_all(const ReduceDesc* descs_dev, int ndesc, long total_len, float* flat_grad, hipStream_t s,
                             const float* errpart = nullptr, long nparts = 0, long nelem = 0, float* out2 = nullptr);
// alpha_dev == NULL: the step size is alpha_val; macc (optional): {sum loss, sum mae, batches} += batch_scal's {loss, mae}, 1
hipError_t launch_adam(float* p, const float* g, float* m, float* v, long n, const float* alpha_dev, float b1, float b2, float eps,
                       hipStream_t s, float alpha_val = 0.0f, const float* batch_scal = nullptr, double* macc = nullptr);
// run-time-shaped conv for non-reference architectures (conv_generic.hip)
// GEN_EPI_RELU (bias -> ReLU, full resolution) and GEN_EPI_PLAIN (the raw sums; ep may be NULL) serve training:
// forward with BatchNormalization in batch mode, and the backward-data convs (train_generic.hip)
enum { GEN_EPI_BN = 0, GEN_EPI_BN_POOL = 1, GEN_EPI_SIGMOID = 2, GEN_EPI_RELU = 3, GEN_EPI_PLAIN = 4 };
int conv_generic_supported(int H, int W, int cin, int cout, char* why, size_t why_len);
// w_folded (optional, upsample-fed convs only): pack_generic_folded's effective 2x2 kernels; when the layer's shape has a
// folded plan (conv_generic_folds) the conv runs as four phase convs on the stored grid, 4/9 of the multiply-adds
hipError_t launch_conv_generic(const float* in, const float* w_hwio, const float* ep, float* out, int64_t n, int H, int W, int cin,
                               int cout, int ups, int epi, hipStream_t stream, const float* w_folded = nullptr);
size_t pack_generic_folded(int cin, int cout, const float* hwio, float* dst);
// the same convs with the fp32 contraction on the bf16 matrix pipe (three-way operand split, six products): conv_generic_x3.hip.
// Inference only (the weights are split on the host); conv_generic_x3_takes says whether a layer's shape has a plan (ups = the
// folded-upsample form, fed pack_generic_folded's kernels with ntaps = 16; otherwise the HWIO kernel with ntaps = 9)
int conv_generic_x3_takes(int H, int W, int cin, int cout, int ups);
size_t pack_generic_bf16x3(int ntaps, int cin, int cout, const float* w, uint16_t* dst);     // returns the number of bf16 values
hipError_t launch_pack_generic_bf16x3(const float* w, int ntaps, int cin, int cout, uint16_t* dst, hipStream_t stream);   // the same on the device
// inv_sw != 0: the two-term fp16 split instead (wplanes = pack_generic_f16x2's; per-strip activation scale taken in the kernel)
hipError_t launch_conv_generic_x3(const float* in, const uint16_t* wplanes, const float* ep, float* out, int64_t n, int H, int W, int cin,
                                  int cout, int ups, int epi, hipStream_t stream, float inv_sw = 0.0f);
size_t pack_generic_f16x2(int ntaps, int cin, int cout, const float* w, uint16_t* dst, float* inv_sw);
// the 1-filter sigmoid conv behind an UpSampling2D as a cin -> 16 GEMM on bf16 MFMAs + a gather (cin 32 or 64):
// wplanes = pack_last_bf16x3(cin, pack_generic_folded(cin, 1, hwio, .), .); out = the reconstruction [n][H][W]
int conv_last_x3_takes(int H, int W, int cin);
size_t pack_last_bf16x3(int cin, const float* weff, uint16_t* dst);
hipError_t launch_conv_last_x3(const float* in, const uint16_t* wplanes, const float* ep, float* out, int64_t n, int H, int W, int cin,
                               hipStream_t stream);
int conv_generic_folds(int H, int W, int cin, int cout);
// run-time-shaped training kernels (train_generic.hip)
hipError_t launch_flip_transpose(const float* hwio, int cin, int cout, float* dst, hipStream_t s);
hipError_t launch_sumpool2x2(const float* in, float* out, int64_t n, int H, int W, int C, hipStream_t s);
size_t wgrad_generic_lds_bytes(int W, int cin, int cout);
hipError_t launch_wgrad_generic(const float* xin, const float* dz, float* part, int64_t n, int H, int W, int cin, int cout, int ups,
                                int max_parts, int* nparts, hipStream_t s);
hipError_t launch_recon_err(const float* recon, const float* x, int64_t n, int npix, float* errpart, hipStream_t stream);
// training augmentation: affine bilinear resample (nearest fill) + flips, one image per workgroup
hipError_t launch_augment(const float* in, const cs_aug_affine* tf_dev, float* out, int64_t n, int H, int W, hipStream_t s);
// cs_train_fit_step: gather train[idx[b]] -> y_out[b] (unchanged) and x_out[b] (resampled by tf[b] when has_tf); tf / idx may be pinned host memory
hipError_t launch_fit_gather(const float* train, const cs_aug_affine* tf, const int* idx, float* x_out, float* y_out, int64_t n, int H,
                             int W, bool has_tf, hipStream_t s);
// one launch for every operand pack of a training step: transposed = 0 forward fragments, 1 backward-data fragments,
// 2 conv7's effective weights (cin/cout ignored)
struct PackJob { const float* src; float* dst; int cin, cout, transposed, blocks; };
struct PackTable { PackJob job[16]; int n; };
hipError_t launch_pack_all(PackTable& tab, hipStream_t s);
hipError_t launch_pack_frag(const float* hwio, int cin, int cout, int transposed, float* dst, hipStream_t s);
hipError_t launch_pack_w7eff(const float* w7, float* weff, hipStream_t s);
hipError_t launch_pack_ep(const float* bias, const float* gamma, const float* beta, const float* mov_mean,
                          const float* mov_var, float eps, int C, float* ep, hipStream_t s);

// scaler.transform + pca.transform.  comps_pad: [cpad][fpad], zero beyond [C][F];
// cpad % 16 == 0, fpad % 512 == 0.
hipError_t launch_scaler_pca(const float* feat, const float* center, const double* scale,
                             const float* comps_pad, const float* mean_proj, int F, int fpad, int C,
                             int cpad, float* pca_out, int64_t n_cells, hipStream_t stream);

// the same on the bf16 matrix pipe (six split products per multiply): comps_planes = pack_pca_bf16x3(comps_pad, cpad, fpad, .)
hipError_t launch_scaler_pca_x3(const float* feat, const float* center, const double* scale, const uint16_t* comps_planes,
                                const float* mean_proj, int F, int fpad, int C, int cpad, float* pca_out, int64_t n_cells,
                                hipStream_t stream, void* split_ws = nullptr);
// Small calls (n <= DET_SPLIT_MAX_CELLS, split_ws != NULL: det_split_ws_bytes(C) bytes of device memory): the PCA GEMM's feature
// ranges / the SVM's support-vector ranges run side by side in separate workgroups and are added in the order the one-workgroup
// form adds them -- bit-identical results, a 128-cell call 0.65 -> 0.3 ms (the reference screens one sample per call).
constexpr int64_t DET_SPLIT_MAX_CELLS = 16384;
size_t det_split_ws_bytes(int C);
size_t pack_pca_bf16x3(const float* comps_pad, int cpad, int fpad, uint16_t* dst);     // returns the number of bf16 values

// One-class SVM decision for one detector.  sv: [nsv_pad][D] row-major, svT: [D][nsv_pad]
// (transposed), coef: [nsv_pad]; all zero padded.  dec[n] = sum - rho.
hipError_t launch_ocsvm(const float* pca, int D, const double* svT /* [D][nsv_pad] */, const double* svn /* ||sv||^2 */,
                        const double* coef, int nsv_pad, double gamma, double rho, double* dec, int64_t n_cells, hipStream_t stream,
                        void* split_ws = nullptr);

hipError_t launch_ocsvm_pair_split(const float* pca, int D, const double* const svT[2], const double* const svn[2],
                                   const double* const coef[2], const int nsv_pad[2], const double gamma[2], const double rho[2],
                                   double* dec0, double* dec1, int64_t n_cells, hipStream_t stream, void* split_ws);

// errpart -> mse/mae ; dec -> score (= -dec) and pred.
hipError_t launch_finalize(const float* errpart, int nparts, int npix, const double* dec_c,
                           const double* dec_m, float* mse, float* mae, double* score_c,
                           double* score_m, int8_t* pred_c, int8_t* pred_m, int64_t n_cells,
                           hipStream_t stream);

hipError_t launch_synth(uint64_t seed, int64_t first_cell, int64_t n, int npix, float* out,
                        hipStream_t stream);

}  // namespace cs
