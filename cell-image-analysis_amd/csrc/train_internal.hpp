// train_internal.hpp -- the trainer handle shared by train_api.hip (reference graph, tuned kernels) and
// train_generic.hip (any instance of the layer grammar, run-time-shaped kernels).
#pragma once
#include "api_internal.hpp"

constexpr int TR_MAXL = CS_MAX_CONV;

struct cs_trainer {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;      // weight gradients run here, beside the BatchNormalization-backward chain of the next layer
    hipEvent_t ev_dz[TR_MAXL] = {nullptr}, ev_wg = nullptr;
    cs_train_cfg cfg;
    int64_t maxb = 0;
    int64_t eval_maxb = 0;              // generic trainer: cells the forward-only buffers of cs_train_eval hold
    // architecture: ref = the reference graph (64x64, 32-64-32 | 32-64-32-1) on the tuned kernels
    bool ref = true;
    int H = 64, W = 64, n_conv = 7, n_enc = 3;
    int ch[TR_MAXL] = {0}, gh[TR_MAXL] = {0}, gw[TR_MAXL] = {0};       // filters, conv grid of layer l
    size_t rfl[TR_MAXL] = {0}, afl[TR_MAXL] = {0};                     // per-cell floats: conv-grid tensor, stored (BN/pool) output
    int cin(int l) const { return l == 0 ? 1 : ch[l - 1]; }
    // flat parameter layout, Keras order: conv l kernel (HWIO), bias, [gamma, beta]
    long off_k[TR_MAXL], off_b[TR_MAXL], off_g[TR_MAXL], off_be[TR_MAXL], nparam = 0;
    long off_mm[TR_MAXL], off_mv[TR_MAXL], nmov = 0;
    cs::DevBuf P, Gown, M, V, MOV;
    float* G = nullptr;                 // gradient buffer in use (own or caller's)
    long step = 0;
    // packed operands, rebuilt after every update (reference graph: MFMA fragments; generic: flipped/transposed HWIO kernels)
    cs::DevBuf wf[TR_MAXL], wft[TR_MAXL], w7eff, ep_inf[TR_MAXL];
    // generic trainer: split-bf16 planes of the forward kernels / of the flipped kernels (conv_generic_x3.hip), re-packed with the weights
    cs::DevBuf wx3f[TR_MAXL], wx3t[TR_MAXL];
    bool x3f[TR_MAXL] = {false}, x3t[TR_MAXL] = {false};
    // batch tensors
    cs::DevBuf x, y, r[TR_MAXL], a[TR_MAXL], out, errpart, dz[TR_MAXL], da[TR_MAXL], stats[TR_MAXL], dup;
    cs::DevBuf aug_tf, aug_in, aug_out;
    cs::DevBuf part_stats, part_bwd, bwd_sums, dzsum_part[TR_MAXL], wpart[TR_MAXL], descs, scal, zeros;
    int np_w[TR_MAXL], np_b[TR_MAXL];
    // the reduction descriptors depend on the batch size only: uploaded when it changes, from memory that outlives the copy
    cs::ReduceDesc hdescs[2 * TR_MAXL];
    int64_t descs_batch = -1;
    float* hloss = nullptr;             // pinned {loss, mae, alpha staging, -}: read after the step's single synchronisation
    bool defer_sync = false;            // cs_train_step: forward_backward leaves its results to the sync at the end of apply
    // cs_train_step_async: epoch metrics on the device {sum loss, sum mae, batches} (double), what synchronous generic steps add
    // on the host, the event behind the step's input copies, and a pinned ring for the augmentation parameters
    cs::DevBuf macc;
    double hacc[3] = {0.0, 0.0, 0.0};
    // synchronised BatchNormalization under data parallelism (cs_train_set_sync_bn): the caller's all-gather, its exchange buffer
    cs_allgather_fn sync_fn = nullptr;
    void* sync_ctx = nullptr;
    float* sync_buf = nullptr;
    int64_t sync_cap = 0;
    int sync_rank = 0, sync_world = 1;
    cs::DevBuf sync_scratch;
    hipEvent_t ev_in = nullptr;
    static constexpr int AUG_SLOTS = 8;
    void* aug_pin = nullptr;
    size_t aug_pin_slot = 0;            // bytes per slot
    hipEvent_t ev_aug[AUG_SLOTS] = {nullptr};
    bool aug_used[AUG_SLOTS] = {false};
    int aug_next = 0;
    // cs_train_fit_step: per-step {transforms, indices} in a pinned ring the gather kernel reads directly
    static constexpr int FIT_SLOTS = 16;
    void* fit_pin = nullptr;
    size_t fit_pin_slot = 0;
    hipEvent_t ev_fit[FIT_SLOTS] = {nullptr};
    bool fit_used[FIT_SLOTS] = {false};
    int fit_next = 0;
    ~cs_trainer()
    {
        if (hloss) (void)hipHostFree(hloss);
        if (aug_pin) (void)hipHostFree(aug_pin);
        if (fit_pin) (void)hipHostFree(fit_pin);
        for (auto& e : ev_fit) if (e) (void)hipEventDestroy(e);
        if (ev_in) (void)hipEventDestroy(ev_in);
        for (auto& e : ev_aug) if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_dz) if (e) (void)hipEventDestroy(e);
        if (ev_wg) (void)hipEventDestroy(ev_wg);
        if (stream2) (void)hipStreamDestroy(stream2);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

#define LCHK(call)                                                                             \
    do {                                                                                       \
        hipError_t le__ = (call);                                                              \
        if (le__ != hipSuccess) return cs::fail(CS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(le__)); \
    } while (0)

// train_generic.hip
int gen_train_setup(cs_trainer* t);                      // buffers that depend on the architecture only
int gen_train_repack(cs_trainer* t);                     // flipped / transposed kernels for the backward-data convs
int gen_train_ensure_batch(cs_trainer* t, int64_t b);
int gen_train_fb_enqueue(cs_trainer* t, const float* x, const float* y, int64_t batch, int kind);
int gen_train_forward_backward(cs_trainer* t, const float* x, const float* y, int64_t batch, int kind, float* loss, float* mae);
int gen_train_eval(cs_trainer* t, const float* x, const float* y, int64_t n, int kind, float* loss, float* mae);
