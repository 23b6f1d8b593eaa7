// train_api.hip -- C ABI of the CAE trainer (include/cellscreen.h, cs_train_*): one
// `autoencoder.fit` step of CAE_improved_modeltrain.py:286-293 on the graph compiled at
// :223-227 -- forward with BatchNormalization in training mode, MSE loss / MAE metric,
// backward, Keras Adam -- plus the inference-mode evaluation used for val_loss.
// Callbacks (EarlyStopping, ModelCheckpoint, ReduceLROnPlateau, :263-283) are host-side
// scalars and live in cellscreen/training.py.
#include "train_internal.hpp"

#include <cmath>

using namespace cs;

static int cin_of(int l) { return l == 0 ? 1 : kRefChannels[l - 1]; }
static size_t a_floats(int l) { return kLayerFloats[l]; }                                   // stored BN output per cell
static size_t r_floats(int l) { return (size_t)kConvGrid[l] * kConvGrid[l] * kRefChannels[l]; }  // conv-grid tensor per cell

static int repack(cs_trainer* t)
{
    float* P = t->P.as<float>();
    PackTable tab;
    tab.n = 0;
    for (int l = 0; l < 6; ++l) tab.job[tab.n++] = PackJob{P + t->off_k[l], t->wf[l].as<float>(), cin_of(l), kRefChannels[l], 0, 0};
    for (int l = 1; l < 7; ++l) tab.job[tab.n++] = PackJob{P + t->off_k[l], t->wft[l].as<float>(), cin_of(l), kRefChannels[l], 1, 0};
    tab.job[tab.n++] = PackJob{P + t->off_k[6], t->w7eff.as<float>(), 32, 1, 2, 0};
    LCHK(launch_pack_all(tab, t->stream));
    return CS_OK;
}

constexpr int kTrainMaxBatch = 8192;
static_assert((long)kTrainMaxBatch * 4 <= (long)BN_MAX_PARTS * 64, "conv7's loss epilogue leaves 4 partials per cell in dzsum_part[6]");
static_assert((long)kTrainMaxBatch * 64 * 64 * 32 < (1L << 31), "the BatchNormalization kernels index below 2^31");

static int ensure_batch(cs_trainer* t, int64_t b)
{
    if (b <= t->maxb) return CS_OK;
    // conv7's loss epilogue leaves 4 bias-gradient partials per cell in dzsum_part[6] (BN_MAX_PARTS * 64 floats), and the
    // BatchNormalization kernels index the 32x64x64 tensors below 2^31: both hold up to this batch
    if (b > kTrainMaxBatch)
        return fail(CS_ERR_UNSUPPORTED, "a batch of %lld cells exceeds the trainer's limit of %d (the reference trains on 32)", (long long)b, kTrainMaxBatch);
    int rc;
    if ((rc = t->x.ensure((size_t)b * kH * kW * 4)) || (rc = t->y.ensure((size_t)b * kH * kW * 4))) return rc;
    for (int l = 0; l < 6; ++l) {
        if ((rc = t->r[l].ensure(b * r_floats(l) * 4)) || (rc = t->a[l].ensure(b * a_floats(l) * 4)) ||
            (rc = t->da[l].ensure(b * a_floats(l) * 4)) || (rc = t->dz[l].ensure(b * r_floats(l) * 4)))
            return rc;
    }
    if ((rc = t->dz[6].ensure((size_t)b * kH * kW * 4)) || (rc = t->out.ensure((size_t)b * kH * kW * 4)) ||
        (rc = t->errpart.ensure((size_t)b * 8 * 4)))
        return rc;
    t->maxb = b;
    return CS_OK;
}

extern "C" {

int cs_train_param_count(int64_t* n_trainable, int64_t* n_moving)
{
    long n = 0, m = 0;
    for (int l = 0; l < 7; ++l) {
        n += 9L * cin_of(l) * kRefChannels[l] + kRefChannels[l];
        if (l < 6) { n += 2L * kRefChannels[l]; m += 2L * kRefChannels[l]; }
    }
    if (n_trainable) *n_trainable = n;
    if (n_moving) *n_moving = m;
    return CS_OK;
}

// Describes the architecture of `init`; non-reference instances must follow the layer grammar and fit the run-time-shaped
// kernels in both directions (the backward-data conv of layer l has cin' = filters(l), cout' = cin(l)).
static int describe_trainer(cs_trainer* t, const cs_cae_weights* w)
{
    if (!w) return fail(CS_ERR_INVALID, "initial weights are NULL");
    t->H = w->height; t->W = w->width; t->n_conv = w->n_conv; t->n_enc = w->n_enc;
    if (t->n_conv < 3 || t->n_conv > TR_MAXL || t->n_enc < 1 || t->n_conv != 2 * t->n_enc + 1)
        return fail(CS_ERR_UNSUPPORTED, "initial weights: n_conv=%d n_enc=%d is not the reference grammar (n_conv = 2 n_enc + 1)", t->n_conv, t->n_enc);
    if (t->H <= 0 || t->W <= 0 || t->H % (1 << t->n_enc) || t->W % (1 << t->n_enc))
        return fail(CS_ERR_UNSUPPORTED, "initial weights: input %dx%d is not divisible by 2^n_enc", t->H, t->W);
    t->ref = t->H == kH && t->W == kW && t->n_conv == kNConv && t->n_enc == kNEnc;
    int h = t->H, wd = t->W;
    for (int l = 0; l < t->n_conv; ++l) {
        t->ch[l] = w->channels[l];
        if (t->ch[l] <= 0) return fail(CS_ERR_INVALID, "initial weights: conv %d has %d filters", l, t->ch[l]);
        if (t->ref && t->ch[l] != kRefChannels[l]) t->ref = false;
        if (!w->kernel[l] || !w->bias[l]) return fail(CS_ERR_INVALID, "initial weights: conv %d kernel/bias is NULL", l);
        if (l < t->n_conv - 1 && (!w->bn_gamma[l] || !w->bn_beta[l] || !w->bn_mean[l] || !w->bn_var[l]))
            return fail(CS_ERR_INVALID, "initial weights: conv %d BatchNormalization arrays are NULL", l);
        if (l > t->n_enc) { h *= 2; wd *= 2; }
        t->gh[l] = h; t->gw[l] = wd;
        t->rfl[l] = (size_t)h * wd * t->ch[l];
        if (l < t->n_enc) { h /= 2; wd /= 2; }
        t->afl[l] = (size_t)h * wd * t->ch[l];
    }
    if (t->ch[t->n_conv - 1] != 1) return fail(CS_ERR_UNSUPPORTED, "initial weights: the last conv must have 1 filter");
    if (!t->ref) {
        char why[160];
        for (int l = 0; l < t->n_conv; ++l) {
            if (!conv_generic_supported(t->gh[l], t->gw[l], t->cin(l), t->ch[l], why, sizeof why))
                return fail(CS_ERR_UNSUPPORTED, "training, conv %d: %s", l, why);
            if (l > 0 && !conv_generic_supported(t->gh[l], t->gw[l], t->ch[l], t->cin(l), why, sizeof why))
                return fail(CS_ERR_UNSUPPORTED, "training, backward-data conv %d: %s", l, why);
            // the BatchNormalization / max-pool kernels shared with the reference graph index with shifts and masks
            // (train.hip: "H, W, C are powers of two"): anything else would train on silently wrong statistics
            auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
            if (l < t->n_conv - 1 && (!pow2(t->gh[l]) || !pow2(t->gw[l]) || !pow2(t->ch[l]) || t->ch[l] < 4 || t->ch[l] > 256))
                return fail(CS_ERR_UNSUPPORTED, "training, conv %d: a %dx%d grid with %d filters -- the BatchNormalization kernels need powers of two (4 .. 256 filters)",
                            l, t->gh[l], t->gw[l], t->ch[l]);
            // the weight-gradient kernel stages three input rows and one dz row in LDS
            if (wgrad_generic_lds_bytes(t->gw[l], t->cin(l), t->ch[l]) > 160 * 1024)
                return fail(CS_ERR_UNSUPPORTED, "training, conv %d: its weight gradient needs %zu bytes of LDS (160 KiB per workgroup)", l,
                            wgrad_generic_lds_bytes(t->gw[l], t->cin(l), t->ch[l]));
        }
    }
    return CS_OK;
}

int cs_train_param_count_of(const cs_trainer* t, int64_t* n_trainable, int64_t* n_moving)
{
    if (!t) return fail(CS_ERR_INVALID, "trainer is NULL");
    if (n_trainable) *n_trainable = t->nparam;
    if (n_moving) *n_moving = t->nmov;
    return CS_OK;
}

int cs_train_create(const cs_cae_weights* init, const cs_train_cfg* cfg, int device_id, cs_trainer** out)
{
    if (!out) return fail(CS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!cfg) return fail(CS_ERR_INVALID, "cfg is NULL");
    int rc;
    if ((rc = require_gfx950(device_id))) return rc;
    cs_trainer* t = new (std::nothrow) cs_trainer();
    if (!t) return fail(CS_ERR_NOMEM, "host allocation failed");
    t->device = device_id;
    t->cfg = *cfg;
#define TFAIL(x) do { int r__ = (x); if (r__) { delete t; return r__; } } while (0)
    TFAIL(describe_trainer(t, init));
    {
        hipError_t e = hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete t; return fail(CS_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    }
    const int NL = t->n_conv;
    long o = 0, mo = 0;
    for (int l = 0; l < NL; ++l) {
        t->off_k[l] = o; o += 9L * t->cin(l) * t->ch[l];
        t->off_b[l] = o; o += t->ch[l];
        if (l < NL - 1) {
            t->off_g[l] = o; o += t->ch[l];
            t->off_be[l] = o; o += t->ch[l];
            t->off_mm[l] = mo; mo += t->ch[l];
            t->off_mv[l] = mo; mo += t->ch[l];
        }
    }
    t->nparam = o; t->nmov = mo;
    std::vector<float> hp(o), hm(mo);
    for (int l = 0; l < NL; ++l) {
        const int c = t->ch[l];
        memcpy(&hp[t->off_k[l]], init->kernel[l], sizeof(float) * 9 * t->cin(l) * c);
        memcpy(&hp[t->off_b[l]], init->bias[l], sizeof(float) * c);
        if (l < NL - 1) {
            memcpy(&hp[t->off_g[l]], init->bn_gamma[l], sizeof(float) * c);
            memcpy(&hp[t->off_be[l]], init->bn_beta[l], sizeof(float) * c);
            memcpy(&hm[t->off_mm[l]], init->bn_mean[l], sizeof(float) * c);
            memcpy(&hm[t->off_mv[l]], init->bn_var[l], sizeof(float) * c);
        }
    }
    {
        hipError_t e = hipHostMalloc((void**)&t->hloss, 4 * sizeof(float), hipHostMallocDefault);
        if (e != hipSuccess) { delete t; return fail(CS_ERR_HIP, "hipHostMalloc: %s", hipGetErrorString(e)); }
        t->hloss[0] = t->hloss[1] = t->hloss[2] = t->hloss[3] = 0.0f;
    }
    TFAIL(upload(t->P, hp.data(), o * 4));
    TFAIL(upload(t->MOV, hm.data(), mo * 4));
    {
        const double z3[4] = {0.0, 0.0, 0.0, 0.0};
        TFAIL(upload(t->macc, z3, sizeof z3));
        hipError_t e = hipEventCreateWithFlags(&t->ev_in, hipEventDisableTiming);
        if (e != hipSuccess) { delete t; return fail(CS_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e)); }
    }
    TFAIL(t->Gown.ensure(o * 4)); TFAIL(t->M.ensure(o * 4)); TFAIL(t->V.ensure(o * 4));
    t->G = t->Gown.as<float>();
    {
        hipError_t e1 = hipMemset(t->M.p, 0, o * 4), e2 = hipMemset(t->V.p, 0, o * 4), e3 = hipMemset(t->Gown.p, 0, o * 4);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { delete t; return fail(CS_ERR_HIP, "hipMemset failed"); }
    }
    if (!t->ref) {
        TFAIL(gen_train_setup(t));
        hipError_t e = hipStreamSynchronize(t->stream);
        if (e != hipSuccess) { delete t; return fail(CS_ERR_HIP, "initial repack: %s", hipGetErrorString(e)); }
        *out = t;
        return CS_OK;
    }
    for (int l = 0; l < 6; ++l) {
        TFAIL(t->wf[l].ensure(pack_conv_fragments(cin_of(l), kRefChannels[l], nullptr, nullptr) * 4));
        TFAIL(t->ep_inf[l].ensure(3 * kRefChannels[l] * 4));
        TFAIL(t->stats[l].ensure(2 * kRefChannels[l] * 4));
    }
    for (int l = 1; l < 7; ++l)   // backward-data fragments: effective conv (cin' = cout, cout' = cin)
        TFAIL(t->wft[l].ensure(pack_conv_fragments(kRefChannels[l], cin_of(l), nullptr, nullptr) * 4));
    TFAIL(t->w7eff.ensure(16 * 32 * 4));
    TFAIL(t->part_stats.ensure((size_t)BN_MAX_PARTS * 3 * 64 * 4));
    TFAIL(t->part_bwd.ensure((size_t)BN_MAX_PARTS * 2 * 64 * 4));
    TFAIL(t->bwd_sums.ensure(2 * 64 * 4));
    for (int l = 0; l < 7; ++l) {
        TFAIL(t->dzsum_part[l].ensure((size_t)BN_MAX_PARTS * 64 * 4));
        TFAIL(t->wpart[l].ensure((size_t)TRAIN_MAX_PARTS * 9 * cin_of(l) * kRefChannels[l] * 4));
    }
    TFAIL(t->descs.ensure(14 * sizeof(ReduceDesc)));
    TFAIL(t->scal.ensure(16));
    TFAIL(repack(t));
    {
        hipError_t e = hipStreamSynchronize(t->stream);
        if (e != hipSuccess) { delete t; return fail(CS_ERR_HIP, "initial repack: %s", hipGetErrorString(e)); }
    }
#undef TFAIL
    *out = t;
    return CS_OK;
}

int cs_train_wait_stream(cs_trainer* t, void* hip_stream)
{
    if (!t) return fail(CS_ERR_INVALID, "trainer handle is NULL");
    HIPCHK(hipSetDevice(t->device));
    return wait_on_stream(t->stream, hip_stream);
}

void cs_train_free(cs_trainer* t)
{
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->stream2) (void)hipStreamSynchronize(t->stream2);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    delete t;
}

int cs_train_set_grad_buffer(cs_trainer* t, float* device_buffer)
{
    if (!t) return fail(CS_ERR_INVALID, "trainer is NULL");
    t->G = device_buffer ? device_buffer : t->Gown.as<float>();
    return CS_OK;
}

int cs_train_set_sync_bn(cs_trainer* t, cs_allgather_fn fn, void* ctx, float* device_buf, int64_t capacity_floats, int rank, int world)
{
    if (!t) return fail(CS_ERR_INVALID, "trainer is NULL");
    if (!fn) { t->sync_fn = nullptr; t->sync_world = 1; t->sync_rank = 0; return CS_OK; }
    if (!t->ref) return fail(CS_ERR_UNSUPPORTED, "synchronised BatchNormalization is built for the reference graph's trainer");
    if (!device_buf || world < 1 || rank < 0 || rank >= world || capacity_floats < (int64_t)world * 3 * 64)
        return fail(CS_ERR_INVALID, "sync buffer NULL / too small (needs world x 192 floats) or bad rank/world");
    int rc = t->sync_scratch.ensure(2 * 64 * sizeof(float));
    if (rc) return rc;
    t->sync_fn = fn; t->sync_ctx = ctx; t->sync_buf = device_buf; t->sync_cap = capacity_floats; t->sync_rank = rank; t->sync_world = world;
    return CS_OK;
}

// this rank's values are in its slot of the exchange buffer (enqueued): drain the stream, let the caller all-gather
static int sync_gather(cs_trainer* t, int64_t floats_per_rank)
{
    HIPCHK(hipStreamSynchronize(t->stream));
    const int rc = t->sync_fn(t->sync_ctx, floats_per_rank);
    if (rc) return fail(CS_ERR_INVALID, "the all-gather hook of cs_train_set_sync_bn returned %d", rc);
    return CS_OK;
}

static int copy_in(cs_trainer* t, DevBuf& dst, const float* src, int kind, size_t floats)
{
    HIPCHK(hipMemcpyAsync(dst.p, src, floats * 4, kind == CS_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, t->stream));
    return CS_OK;
}

static int fb_enqueue(cs_trainer* t, int64_t B);

// loss / mae of the batch -> the caller: a pinned read-back enqueued behind the step's kernels.  Within cs_train_step
// the wait is left to the single synchronisation at the end of cs_train_apply (one host round trip per step, not three).
static int finish_forward_backward(cs_trainer* t, float* loss, float* mae)
{
    HIPCHK(hipMemcpyAsync(t->hloss, t->scal.p, 8, hipMemcpyDeviceToHost, t->stream));
    if (t->defer_sync) return CS_OK;
    HIPCHK(hipStreamSynchronize(t->stream));
    if (loss) *loss = t->hloss[0];
    if (mae) *mae = t->hloss[1];
    return CS_OK;
}

int cs_train_forward_backward(cs_trainer* t, const float* x, const float* y, int64_t batch, int kind, float* loss, float* mae)
{
    if (!t) return fail(CS_ERR_INVALID, "trainer is NULL");
    if (!x || !y || batch <= 0) return fail(CS_ERR_INVALID, "x/y NULL or batch <= 0");
    if (kind != CS_MEM_HOST && kind != CS_MEM_DEVICE) return fail(CS_ERR_INVALID, "bad mem kind");
    HIPCHK(hipSetDevice(t->device));
    if (!t->ref) return gen_train_forward_backward(t, x, y, batch, kind, loss, mae);
    int rc = ensure_batch(t, batch);
    if (rc) return rc;
    if ((rc = copy_in(t, t->x, x, kind, (size_t)batch * kH * kW)) || (rc = copy_in(t, t->y, y, kind, (size_t)batch * kH * kW))) return rc;
    if ((rc = fb_enqueue(t, batch))) return rc;
    return finish_forward_backward(t, loss, mae);
}

// Everything of forward + backward between the input copies and the loss read-back, as stream work only (no host
// synchronisation once the reduction descriptors of this batch size are uploaded): what a captured step graph replays.
static int fb_enqueue(cs_trainer* t, int64_t B)
{
    hipStream_t s = t->stream;
    float* P = t->P.as<float>();
    float* G = t->G;
    float* MOV = t->MOV.as<float>();

    // ---- forward, BatchNormalization in training mode ---------------------------------
    for (int l = 0; l < 6; ++l) {
        const int C = kRefChannels[l], Hc = kConvGrid[l], pool = l < kNEnc;
        const float* in = l == 0 ? t->x.as<float>() : t->a[l - 1].as<float>();
        // the conv leaves its workgroups' {count, mean, M2} per channel beside the tensor: no statistics pass over r
        int G1 = 0;
        LCHK(launch_conv_train_fwd(l, in, t->wf[l].as<float>(), P + t->off_b[l], t->r[l].as<float>(), B, s, t->part_stats.as<float>(), &G1));
        if (t->sync_fn) {
            // the local partials -> ONE {count, mean, M2} triple per channel in this rank's slot, all-gather, and the final merge
            // runs over the ranks' triples: the statistics (and the moving averages) of the whole batch, identical on every rank
            LCHK(launch_bn_stats_merge(t->part_stats.as<float>(), G1, C, t->sync_buf + (size_t)t->sync_rank * 3 * C, s));
            int rc = sync_gather(t, 3 * C);
            if (rc) return rc;
            LCHK(launch_bn_stats_final(t->sync_buf, t->sync_world, C, t->cfg.bn_eps, t->cfg.bn_momentum, MOV + t->off_mm[l],
                                       MOV + t->off_mv[l], t->stats[l].as<float>(), s));
        } else
        LCHK(launch_bn_stats_final(t->part_stats.as<float>(), G1, C, t->cfg.bn_eps, t->cfg.bn_momentum, MOV + t->off_mm[l],
                                   MOV + t->off_mv[l], t->stats[l].as<float>(), s));
        LCHK(launch_bn_apply(t->r[l].as<float>(), C, P + t->off_g[l], P + t->off_be[l], t->stats[l].as<float>(),
                             t->a[l].as<float>(), B, Hc, Hc, pool, s));
    }
    // the output conv with the loss gradient in its epilogue: dz7 and its per-(cell, strip) sums (the bias gradient's partials)
    LCHK(launch_conv7_err_train(t->a[5].as<float>(), t->y.as<float>(), t->w7eff.as<float>(), P + t->off_b[6], t->errpart.as<float>(),
                                t->out.as<float>(), t->dz[6].as<float>(), t->dzsum_part[6].as<float>(), B, s));
    t->np_b[6] = (int)(B * 4);

    // ---- backward -------------------------------------------------------------------------
    // The weight gradient of a layer needs only that layer's dz and input; it runs on a second stream beside the
    // backward-data conv and the BatchNormalization-backward kernels of the layers below, none of which fills the chip at
    // batch 32.  Both streams join again before the partial sums are reduced.
    if (!t->stream2) {
        HIPCHK(hipStreamCreateWithFlags(&t->stream2, hipStreamNonBlocking));
        for (int l = 0; l < 7; ++l) HIPCHK(hipEventCreateWithFlags(&t->ev_dz[l], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&t->ev_wg, hipEventDisableTiming));
    }
    hipStream_t s2 = t->stream2;
    HIPCHK(hipEventRecord(t->ev_dz[6], s));
    HIPCHK(hipStreamWaitEvent(s2, t->ev_dz[6], 0));
    LCHK(launch_wgrad(6, t->a[5].as<float>(), t->dz[6].as<float>(), t->wpart[6].as<float>(), B, &t->np_w[6], s2));
    LCHK(launch_conv_dgrad(6, t->dz[6].as<float>(), t->wft[6].as<float>(), t->da[5].as<float>(), B, s));
    for (int l = 5; l >= 0; --l) {
        const int C = kRefChannels[l], Hc = kConvGrid[l], pool = l < kNEnc;
        int G2 = 0;
        LCHK(launch_bn_bwd_reduce(t->da[l].as<float>(), t->r[l].as<float>(), t->stats[l].as<float>(), P + t->off_g[l],
                                  P + t->off_be[l], B, Hc, Hc, C, pool, t->part_bwd.as<float>(), &G2, s));
        if (t->sync_fn) {
            // dgamma / dbeta stay this rank's share (the gradient all-reduce averages them like every other gradient); the two
            // MEANS that dz needs are those of the whole batch: raw local sums -> slot, all-gather, summed in rank order
            LCHK(launch_bn_bwd_final(t->part_bwd.as<float>(), G2, C, 1.0, t->sync_buf + (size_t)t->sync_rank * 2 * C,
                                     G + t->off_g[l], G + t->off_be[l], s));
            int rc = sync_gather(t, 2 * C);
            if (rc) return rc;
            LCHK(launch_bn_bwd_final(t->sync_buf, t->sync_world, C, (double)B * t->sync_world * Hc * Hc, t->bwd_sums.as<float>(),
                                     t->sync_scratch.as<float>(), t->sync_scratch.as<float>() + 64, s));
        } else
        LCHK(launch_bn_bwd_final(t->part_bwd.as<float>(), G2, C, (double)B * Hc * Hc, t->bwd_sums.as<float>(),
                                 G + t->off_g[l], G + t->off_be[l], s));
        LCHK(launch_bn_bwd_dz(t->da[l].as<float>(), t->r[l].as<float>(), t->stats[l].as<float>(), P + t->off_g[l],
                              P + t->off_be[l], t->bwd_sums.as<float>(), B, Hc, Hc, C, pool, t->dz[l].as<float>(),
                              t->dzsum_part[l].as<float>(), &t->np_b[l], s));
        const float* in = l == 0 ? t->x.as<float>() : t->a[l - 1].as<float>();
        HIPCHK(hipEventRecord(t->ev_dz[l], s));
        HIPCHK(hipStreamWaitEvent(s2, t->ev_dz[l], 0));
        LCHK(launch_wgrad(l, in, t->dz[l].as<float>(), t->wpart[l].as<float>(), B, &t->np_w[l], s2));
        if (l > 0) LCHK(launch_conv_dgrad(l, t->dz[l].as<float>(), t->wft[l].as<float>(), t->da[l - 1].as<float>(), B, s));
    }
    HIPCHK(hipEventRecord(t->ev_wg, s2));
    HIPCHK(hipStreamWaitEvent(s, t->ev_wg, 0));
    // ---- all partial sums -> flat gradient, in workgroup order -----------------------------
    long total = 0;
    for (int l = 0; l < 7; ++l) total += 9L * cin_of(l) * kRefChannels[l] + kRefChannels[l];
    if (t->descs_batch != B) {          // the partial counts depend on the batch size only
        for (int l = 0; l < 7; ++l) {
            const long klen = 9L * cin_of(l) * kRefChannels[l];
            t->hdescs[2 * l] = ReduceDesc{t->off_k[l], klen, t->wpart[l].as<float>(), t->np_w[l], klen};
            t->hdescs[2 * l + 1] = ReduceDesc{t->off_b[l], (long)kRefChannels[l], t->dzsum_part[l].as<float>(), t->np_b[l], (long)kRefChannels[l]};
        }
        HIPCHK(hipMemcpyAsync(t->descs.p, t->hdescs, 14 * sizeof(ReduceDesc), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
        t->descs_batch = B;
    }
    // ... and the batch's {loss, mae} from the forward pass's error partial sums (thread 0 of the same launch)
    LCHK(launch_reduce_all(t->descs.as<ReduceDesc>(), 14, total, G, s, t->errpart.as<float>(), B * 4, B * (long)kH * kW, t->scal.as<float>()));
    return CS_OK;
}

// Adam's step size alpha = lr sqrt(1 - b2^t) / (1 - b1^t) changes every step: it reaches the kernel through device memory
// (scal[2]), staged from pinned host memory, so that the update is the same stream work every step (graph-replayable).
static int stage_alpha(cs_trainer* t, float lr)
{
    t->step += 1;
    const double b1 = t->cfg.beta1, b2 = t->cfg.beta2;
    t->hloss[2] = (float)((double)lr * std::sqrt(1.0 - std::pow(b2, (double)t->step)) / (1.0 - std::pow(b1, (double)t->step)));
    HIPCHK(hipMemcpyAsync(t->scal.as<float>() + 2, t->hloss + 2, sizeof(float), hipMemcpyHostToDevice, t->stream));
    return CS_OK;
}

static int apply_enqueue(cs_trainer* t)
{
    LCHK(launch_adam(t->P.as<float>(), t->G, t->M.as<float>(), t->V.as<float>(), t->nparam, t->scal.as<float>() + 2, t->cfg.beta1,
                     t->cfg.beta2, t->cfg.adam_eps, t->stream));
    return t->ref ? repack(t) : gen_train_repack(t);
}

int cs_train_apply(cs_trainer* t, float lr)
{
    if (!t) return fail(CS_ERR_INVALID, "trainer is NULL");
    HIPCHK(hipSetDevice(t->device));
    int rc = stage_alpha(t, lr);
    if (rc || (rc = apply_enqueue(t))) return rc;
    HIPCHK(hipStreamSynchronize(t->stream));
    return CS_OK;
}

// cs_train_step = forward_backward + apply with ONE host synchronisation (at the end of apply).  Replaying the step as a
// captured hipGraph (~65 launches on two streams) was measured SLOWER on ROCm 7.2 -- 0.74 ms against 0.59 ms for the plain
// two-stream launches at batch 32 -- and aborted inside the runtime in the test suite; it was taken out again.
int cs_train_step(cs_trainer* t, const float* x, const float* y, int64_t batch, int kind, float lr, float* loss, float* mae)
{
    if (!t) return fail(CS_ERR_INVALID, "trainer is NULL");
    t->defer_sync = t->ref;             // the generic path keeps its own synchronisation
    int rc = cs_train_forward_backward(t, x, y, batch, kind, loss, mae);
    t->defer_sync = false;
    if (rc) return rc;
    rc = cs_train_apply(t, lr);         // ends with the step's one synchronisation
    if (rc) return rc;
    if (t->ref) {
        if (loss) *loss = t->hloss[0];
        if (mae) *mae = t->hloss[1];
    }
    return CS_OK;
}

// One fit() batch with NO host synchronisation: input copies, forward, backward, Adam and the re-pack are enqueued on the
// handle's stream and the call returns.  The batch's loss / MAE are added to running sums on the device (what Keras shows as
// an epoch's loss: the mean over its batches); cs_train_read_metrics fetches them with one round trip per epoch.  x and y must
// stay valid until the step's input copies have run: cs_train_inputs_consumed makes a stream of the caller's wait for exactly that
// (the Python wrapper does it with torch's current stream, whose allocator may otherwise hand the batch's memory out again).
int cs_train_step_async(cs_trainer* t, const float* x, const float* y, int64_t batch, int kind, float lr)
{
    if (!t) return fail(CS_ERR_INVALID, "trainer is NULL");
    if (!x || !y || batch <= 0) return fail(CS_ERR_INVALID, "x/y NULL or batch <= 0");
    if (kind != CS_MEM_HOST && kind != CS_MEM_DEVICE) return fail(CS_ERR_INVALID, "bad mem kind");
    if (!t->ref && kind == CS_MEM_DEVICE) {
        // run-time-shaped architectures: the same asynchronous step on train_generic.hip's kernels
        HIPCHK(hipSetDevice(t->device));
        int rc = gen_train_fb_enqueue(t, x, y, batch, kind);
        if (rc) return rc;
        HIPCHK(hipEventRecord(t->ev_in, t->stream));
        t->step += 1;
        const double b1 = t->cfg.beta1, b2 = t->cfg.beta2;
        const float alpha = (float)((double)lr * std::sqrt(1.0 - std::pow(b2, (double)t->step)) / (1.0 - std::pow(b1, (double)t->step)));
        LCHK(launch_adam(t->P.as<float>(), t->G, t->M.as<float>(), t->V.as<float>(), t->nparam, nullptr, t->cfg.beta1, t->cfg.beta2,
                         t->cfg.adam_eps, t->stream, alpha, t->scal.as<float>(), t->macc.as<double>()));
        return gen_train_repack(t);
    }
    if (kind == CS_MEM_HOST) {
        // pageable host batches: a synchronous step, its scalars added on the host
        float l = 0.0f, m = 0.0f;
        int rc = cs_train_step(t, x, y, batch, kind, lr, &l, &m);
        if (rc) return rc;
        t->hacc[0] += l; t->hacc[1] += m; t->hacc[2] += 1.0;
        HIPCHK(hipEventRecord(t->ev_in, t->stream));
        return CS_OK;
    }
    HIPCHK(hipSetDevice(t->device));
    int rc = ensure_batch(t, batch);
    if (rc) return rc;
    if ((rc = copy_in(t, t->x, x, kind, (size_t)batch * kH * kW)) || (rc = copy_in(t, t->y, y, kind, (size_t)batch * kH * kW))) return rc;
    HIPCHK(hipEventRecord(t->ev_in, t->stream));
    if ((rc = fb_enqueue(t, batch))) return rc;
    t->step += 1;
    const double b1 = t->cfg.beta1, b2 = t->cfg.beta2;
    const float alpha = (float)((double)lr * std::sqrt(1.0 - std::pow(b2, (double)t->step)) / (1.0 - std::pow(b1, (double)t->step)));
    LCHK(launch_adam(t->P.as<float>(), t->G, t->M.as<float>(), t->V.as<float>(), t->nparam, nullptr, t->cfg.beta1, t->cfg.beta2,
                     t->cfg.adam_eps, t->stream, alpha, t->scal.as<float>(), t->macc.as<double>()));
    return repack(t);
}

// ---- cs_train_fit_step: the generator's draws on the host, counter-based -------------------------------------------------
// u(seed, step, b, j) in [0, 1): three rounds of splitmix64's finaliser over the key, top 53 bits.  cellscreen/augment.py
// (counter_uniforms) is the same function; tests/test_augment_cpu.py holds the two together.
static inline uint64_t fit_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline double fit_u01(uint64_t seed, uint64_t step, uint64_t b, uint64_t j)
{
    const uint64_t h = fit_mix64(fit_mix64(fit_mix64(seed) ^ step) ^ (b * 8 + j));
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

// One image's transform from its seven uniforms: Keras's get_random_transform (theta, tx = height shift, ty = width shift, zx, zy,
// flip_h, flip_v) and apply_affine_transform's matrix (rotation . shift . zoom about the image centre) in the closed form of
// cellscreen/augment.py: ImageDataGenerator.random_transforms.
static void fit_transform(const cs_aug_config& c, const double u[7], int H, int W, cs_aug_affine* o)
{
    const double h = (double)H, w = (double)W;
    const double theta_deg = c.rotation_range != 0.0 ? -c.rotation_range + (c.rotation_range - -c.rotation_range) * u[0] : 0.0;
    const double theta = theta_deg * (3.14159265358979323846 / 180.0);
    double tx = 0.0, ty = 0.0;
    if (c.height_shift_range != 0.0) tx = (-c.height_shift_range + (c.height_shift_range - -c.height_shift_range) * u[1]) * (c.height_shift_range < 1.0 ? h : 1.0);
    if (c.width_shift_range != 0.0) ty = (-c.width_shift_range + (c.width_shift_range - -c.width_shift_range) * u[2]) * (c.width_shift_range < 1.0 ? w : 1.0);
    double zx = 1.0, zy = 1.0;
    if (!(c.zoom_lo == 1.0 && c.zoom_hi == 1.0)) {
        zx = c.zoom_lo + (c.zoom_hi - c.zoom_lo) * u[3];
        zy = c.zoom_lo + (c.zoom_hi - c.zoom_lo) * u[4];
    }
    const double cs_ = std::cos(theta), sn = std::sin(theta);
    const double m00 = cs_ * zx, m01 = -sn * zy, m10 = sn * zx, m11 = cs_ * zy;
    const double ox = h / 2 + c.center, oy = w / 2 + c.center;
    o->m[0] = m00; o->m[1] = m01; o->m[2] = m10; o->m[3] = m11;
    o->off[0] = ox + (cs_ * tx - sn * ty) - (m00 * ox + m01 * oy);
    o->off[1] = oy + (sn * tx + cs_ * ty) - (m10 * ox + m11 * oy);
    o->identity = (theta == 0.0 && tx == 0.0 && ty == 0.0 && zx == 1.0 && zy == 1.0) ? 1 : 0;
    o->flip_h = (u[5] < 0.5 && c.horizontal_flip) ? 1 : 0;
    o->flip_v = (u[6] < 0.5 && c.vertical_flip) ? 1 : 0;
    o->reserved = 0;
}

int cs_train_draw_transforms(const cs_aug_config* aug, uint64_t seed, uint64_t step, int64_t n, int32_t height, int32_t width,
                             cs_aug_affine* out)
{
    if (!aug || !out || n < 0 || height <= 0 || width <= 0) return fail(CS_ERR_INVALID, "cs_train_draw_transforms: NULL argument or bad size");
    if (!(aug->zoom_lo > 0.0) || aug->zoom_hi < aug->zoom_lo) return fail(CS_ERR_INVALID, "cs_aug_config: zoom range [%g, %g]", aug->zoom_lo, aug->zoom_hi);
    for (int64_t b = 0; b < n; ++b) {
        double u[7];
        for (int j = 0; j < 7; ++j) u[j] = fit_u01(seed, step, (uint64_t)b, (uint64_t)j);
        fit_transform(*aug, u, height, width, out + b);
    }
    return CS_OK;
}

int cs_train_fit_step(cs_trainer* t, const float* train_device, int64_t n_train, const int32_t* idx, int64_t batch,
                      const cs_aug_config* aug, uint64_t seed, uint64_t step, float lr)
{
    if (!t) return fail(CS_ERR_INVALID, "trainer is NULL");
    if (!train_device || !idx || batch <= 0 || n_train <= 0) return fail(CS_ERR_INVALID, "train/idx NULL or batch, n_train <= 0");
    for (int64_t b = 0; b < batch; ++b)
        if (idx[b] < 0 || idx[b] >= n_train) return fail(CS_ERR_INVALID, "idx[%lld] = %d is outside the training set of %lld crops", (long long)b, idx[b], (long long)n_train);
    HIPCHK(hipSetDevice(t->device));
    int rc = t->ref ? ensure_batch(t, batch) : gen_train_ensure_batch(t, batch);
    if (rc) return rc;
    // {transforms, indices} of this step in a pinned slot the gather kernel reads directly; a slot is reused once its kernel ran
    const size_t need = (size_t)batch * (sizeof(cs_aug_affine) + sizeof(int32_t));
    if (need > t->fit_pin_slot) {
        HIPCHK(hipStreamSynchronize(t->stream));
        if (t->fit_pin) { (void)hipHostFree(t->fit_pin); t->fit_pin = nullptr; }
        HIPCHK(hipHostMalloc(&t->fit_pin, need * cs_trainer::FIT_SLOTS, hipHostMallocDefault));
        t->fit_pin_slot = need;
        for (int k = 0; k < cs_trainer::FIT_SLOTS; ++k) {
            t->fit_used[k] = false;
            if (!t->ev_fit[k]) HIPCHK(hipEventCreateWithFlags(&t->ev_fit[k], hipEventDisableTiming));
        }
    }
    const int slot = t->fit_next;
    t->fit_next = (slot + 1) % cs_trainer::FIT_SLOTS;
    if (t->fit_used[slot]) HIPCHK(hipEventSynchronize(t->ev_fit[slot]));
    char* pin = (char*)t->fit_pin + (size_t)slot * t->fit_pin_slot;
    cs_aug_affine* tf = (cs_aug_affine*)pin;
    int32_t* ix = (int32_t*)(pin + (size_t)batch * sizeof(cs_aug_affine));
    if (aug && (rc = cs_train_draw_transforms(aug, seed, step, batch, t->H, t->W, tf))) return rc;
    memcpy(ix, idx, (size_t)batch * sizeof(int32_t));
    if (!t->ref) {
        // run-time-shaped architectures: their step synchronises by itself (cs_train_step_async's rule)
        const size_t bytes = (size_t)batch * t->H * t->W * sizeof(float);
        if ((rc = t->aug_in.ensure(bytes)) || (rc = t->aug_out.ensure(bytes))) return rc;
        LCHK(launch_fit_gather(train_device, tf, ix, t->aug_out.as<float>(), t->aug_in.as<float>(), batch, t->H, t->W, aug != nullptr, t->stream));
        HIPCHK(hipEventRecord(t->ev_fit[slot], t->stream));
        t->fit_used[slot] = true;
        return cs_train_step_async(t, t->aug_out.as<float>(), t->aug_in.as<float>(), batch, CS_MEM_DEVICE, lr);
    }
    LCHK(launch_fit_gather(train_device, tf, ix, t->x.as<float>(), t->y.as<float>(), batch, t->H, t->W, aug != nullptr, t->stream));
    HIPCHK(hipEventRecord(t->ev_fit[slot], t->stream));
    t->fit_used[slot] = true;
    HIPCHK(hipEventRecord(t->ev_in, t->stream));          // cs_train_inputs_consumed: the training set has been read
    if ((rc = fb_enqueue(t, batch))) return rc;
    t->step += 1;
    const double b1 = t->cfg.beta1, b2 = t->cfg.beta2;
    const float alpha = (float)((double)lr * std::sqrt(1.0 - std::pow(b2, (double)t->step)) / (1.0 - std::pow(b1, (double)t->step)));
    LCHK(launch_adam(t->P.as<float>(), t->G, t->M.as<float>(), t->V.as<float>(), t->nparam, nullptr, t->cfg.beta1, t->cfg.beta2,
                     t->cfg.adam_eps, t->stream, alpha, t->scal.as<float>(), t->macc.as<double>()));
    return repack(t);
}

int cs_train_inputs_consumed(cs_trainer* t, void* hip_stream)
{
    if (!t) return fail(CS_ERR_INVALID, "trainer handle is NULL");
    HIPCHK(hipSetDevice(t->device));
    HIPCHK(hipStreamWaitEvent((hipStream_t)hip_stream, t->ev_in, 0));
    return CS_OK;
}

// Mean loss / MAE over the steps since the last reset (Keras's epoch metrics), the number of steps; reset != 0 clears the sums.
// Synchronises the handle's stream: the one host round trip of an epoch of cs_train_step_async calls.
int cs_train_read_metrics(cs_trainer* t, double* loss_mean, double* mae_mean, int64_t* steps, int reset)
{
    if (!t) return fail(CS_ERR_INVALID, "trainer is NULL");
    HIPCHK(hipSetDevice(t->device));
    double d[3] = {0.0, 0.0, 0.0};
    HIPCHK(hipMemcpyAsync(d, t->macc.p, sizeof d, hipMemcpyDeviceToHost, t->stream));
    HIPCHK(hipStreamSynchronize(t->stream));
    const double n = d[2] + t->hacc[2];
    if (loss_mean) *loss_mean = n > 0 ? (d[0] + t->hacc[0]) / n : 0.0;
    if (mae_mean) *mae_mean = n > 0 ? (d[1] + t->hacc[1]) / n : 0.0;
    if (steps) *steps = (int64_t)n;
    if (reset) {
        HIPCHK(hipMemsetAsync(t->macc.p, 0, sizeof d, t->stream));
        t->hacc[0] = t->hacc[1] = t->hacc[2] = 0.0;
    }
    return CS_OK;
}

int cs_train_eval(cs_trainer* t, const float* x, const float* y, int64_t n, int kind, float* loss, float* mae)
{
    if (!t) return fail(CS_ERR_INVALID, "trainer is NULL");
    if (!x || !y || n <= 0) return fail(CS_ERR_INVALID, "x/y NULL or n <= 0");
    if (kind != CS_MEM_HOST && kind != CS_MEM_DEVICE) return fail(CS_ERR_INVALID, "bad mem kind");
    HIPCHK(hipSetDevice(t->device));
    if (!t->ref) return gen_train_eval(t, x, y, n, kind, loss, mae);
    const int64_t ch = n < 4096 ? n : 4096;
    int rc = ensure_batch(t, ch);
    if (rc) return rc;
    hipStream_t s = t->stream;
    float* P = t->P.as<float>();
    float* MOV = t->MOV.as<float>();
    for (int l = 0; l < 6; ++l)
        LCHK(launch_pack_ep(P + t->off_b[l], P + t->off_g[l], P + t->off_be[l], MOV + t->off_mm[l], MOV + t->off_mv[l],
                            t->cfg.bn_eps, kRefChannels[l], t->ep_inf[l].as<float>(), s));
    double s2 = 0.0, s1 = 0.0;
    std::vector<float> part;
    for (int64_t off = 0; off < n; off += ch) {
        const int64_t nc = (n - off) < ch ? (n - off) : ch;
        if ((rc = copy_in(t, t->x, x + (size_t)off * kH * kW, kind, (size_t)nc * kH * kW)) ||
            (rc = copy_in(t, t->y, y + (size_t)off * kH * kW, kind, (size_t)nc * kH * kW)))
            return rc;
        for (int l = 0; l < 6; ++l) {
            const float* in = l == 0 ? t->x.as<float>() : t->a[l - 1].as<float>();
            LCHK(launch_conv_mfma(l, in, t->wf[l].as<float>(), t->ep_inf[l].as<float>(), t->a[l].as<float>(), nc, s));
        }
        LCHK(launch_conv7_err(t->a[5].as<float>(), t->y.as<float>(), t->w7eff.as<float>(), P + t->off_b[6],
                              t->errpart.as<float>(), nullptr, nc, s));
        part.resize((size_t)nc * 8);
        HIPCHK(hipMemcpyAsync(part.data(), t->errpart.p, part.size() * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (size_t i = 0; i < part.size(); i += 2) { s2 += part[i]; s1 += part[i + 1]; }
    }
    if (loss) *loss = (float)(s2 / ((double)n * kH * kW));
    if (mae) *mae = (float)(s1 / ((double)n * kH * kW));
    return CS_OK;
}

int cs_train_augment(cs_trainer* t, const float* x, int64_t n, const cs_aug_affine* tf, float* out, int kind)
{
    if (!t) return fail(CS_ERR_INVALID, "trainer is NULL");
    if (n < 0) return fail(CS_ERR_INVALID, "n is negative");
    if (n == 0) return CS_OK;
    if (!x || !tf || !out) return fail(CS_ERR_INVALID, "NULL argument");
    if (x == out) return fail(CS_ERR_INVALID, "out aliases x");
    if (kind != CS_MEM_HOST && kind != CS_MEM_DEVICE) return fail(CS_ERR_INVALID, "bad mem kind");
    HIPCHK(hipSetDevice(t->device));
    const size_t bytes = (size_t)n * t->H * t->W * sizeof(float);
    int rc;
    if ((rc = t->aug_tf.ensure((size_t)n * sizeof(cs_aug_affine)))) return rc;
    // the parameters travel through a pinned ring (the caller's array may be gone before an asynchronous copy reads it); a slot
    // is reused only after the copy out of it has run
    const size_t need = (size_t)n * sizeof(cs_aug_affine);
    if (need > t->aug_pin_slot) {
        HIPCHK(hipStreamSynchronize(t->stream));
        if (t->aug_pin) { (void)hipHostFree(t->aug_pin); t->aug_pin = nullptr; }
        HIPCHK(hipHostMalloc(&t->aug_pin, need * cs_trainer::AUG_SLOTS, hipHostMallocDefault));
        t->aug_pin_slot = need;
        for (int k = 0; k < cs_trainer::AUG_SLOTS; ++k) {
            t->aug_used[k] = false;
            if (!t->ev_aug[k]) HIPCHK(hipEventCreateWithFlags(&t->ev_aug[k], hipEventDisableTiming));
        }
    }
    const int slot = t->aug_next;
    t->aug_next = (slot + 1) % cs_trainer::AUG_SLOTS;
    if (t->aug_used[slot]) HIPCHK(hipEventSynchronize(t->ev_aug[slot]));
    char* pin = (char*)t->aug_pin + (size_t)slot * t->aug_pin_slot;
    memcpy(pin, tf, need);
    HIPCHK(hipMemcpyAsync(t->aug_tf.p, pin, need, hipMemcpyHostToDevice, t->stream));
    HIPCHK(hipEventRecord(t->ev_aug[slot], t->stream));
    t->aug_used[slot] = true;
    const float* d_in = x;
    float* d_out = out;
    if (kind == CS_MEM_HOST) {
        if ((rc = t->aug_in.ensure(bytes)) || (rc = t->aug_out.ensure(bytes))) return rc;
        HIPCHK(hipMemcpyAsync(t->aug_in.p, x, bytes, hipMemcpyHostToDevice, t->stream));
        d_in = t->aug_in.as<float>();
        d_out = t->aug_out.as<float>();
    }
    LCHK(launch_augment(d_in, t->aug_tf.as<cs_aug_affine>(), d_out, n, t->H, t->W, t->stream));
    if (kind == CS_MEM_HOST) {
        HIPCHK(hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, t->stream));
        HIPCHK(hipStreamSynchronize(t->stream));
    }
    // device output: left on the handle's stream (what consumes it next -- cs_train_step / cs_train_step_async -- runs there;
    // any other reader orders itself with cs_train_inputs_consumed / a synchronising call)
    HIPCHK(hipEventRecord(t->ev_in, t->stream));
    return CS_OK;
}

int cs_train_export(cs_trainer* t, float* params_host, float* moving_host, float* grads_host)
{
    if (!t) return fail(CS_ERR_INVALID, "trainer is NULL");
    HIPCHK(hipSetDevice(t->device));
    HIPCHK(hipStreamSynchronize(t->stream));
    if (params_host) HIPCHK(hipMemcpy(params_host, t->P.p, t->nparam * 4, hipMemcpyDeviceToHost));
    if (moving_host) HIPCHK(hipMemcpy(moving_host, t->MOV.p, t->nmov * 4, hipMemcpyDeviceToHost));
    if (grads_host) HIPCHK(hipMemcpy(grads_host, t->G, t->nparam * 4, hipMemcpyDeviceToHost));
    return CS_OK;
}

int cs_train_tensor(cs_trainer* t, int which, int layer, int64_t batch, float* host)
{
    if (!t || !host) return fail(CS_ERR_INVALID, "NULL argument");
    const int last = t->n_conv - 1;
    if (layer < 0 || layer > last || batch <= 0 || batch > t->maxb) return fail(CS_ERR_INVALID, "bad layer/batch");
    HIPCHK(hipSetDevice(t->device));
    const DevBuf* b = nullptr;
    size_t per = 0;
    const size_t npix = (size_t)t->H * t->W;
    switch (which) {
        case 0: if (layer < last) { b = &t->r[layer]; per = t->rfl[layer]; } break;    // relu(conv) output
        case 1: if (layer < last) { b = &t->a[layer]; per = t->afl[layer]; } break;    // BN (+pool) output
        case 2: b = &t->dz[layer]; per = layer < last ? t->rfl[layer] : npix; break;   // dL/dz
        case 3: if (layer < last) { b = &t->da[layer]; per = t->afl[layer]; } break;   // dL/d(BN output)
        case 4: b = &t->out; per = npix; break;                                         // sigmoid output
        default: break;
    }
    if (!b) return fail(CS_ERR_INVALID, "no such tensor");
    HIPCHK(hipStreamSynchronize(t->stream));
    HIPCHK(hipMemcpy(host, b->p, (size_t)batch * per * 4, hipMemcpyDeviceToHost));
    return CS_OK;
}

int cs_train_import(cs_trainer* t, const float* params_host, const float* moving_host)
{
    if (!t) return fail(CS_ERR_INVALID, "trainer is NULL");
    HIPCHK(hipSetDevice(t->device));
    if (params_host) HIPCHK(hipMemcpy(t->P.p, params_host, t->nparam * 4, hipMemcpyHostToDevice));
    if (moving_host) HIPCHK(hipMemcpy(t->MOV.p, moving_host, t->nmov * 4, hipMemcpyHostToDevice));
    int rc = t->ref ? repack(t) : gen_train_repack(t);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(t->stream));
    return CS_OK;
}

}  // extern "C"
