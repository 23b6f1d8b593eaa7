// api.hip -- the C ABI of libcellscreen.so (include/cellscreen.h): model construction,
// workspace, chunked orchestration of the kernels, measurement hooks.
// There is deliberately no CPU path here: without a HIP device every compute entry
// point fails with CS_ERR_NO_DEVICE.
#include "api_internal.hpp"
#include "tensor_archive.hpp"

using namespace cs;

// ---------------------------------------------------------------- errors
static thread_local std::string g_err;

namespace cs {
int fail(int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
const char* last_error_cstr() { return g_err.c_str(); }
}  // namespace cs

static const char* kKernelNames[K_COUNT] = {"conv1_relu_bn_pool", "conv2_relu_bn_pool", "conv3_relu_bn_pool",
                                            "conv4_relu_bn",      "conv5_up_relu_bn",   "conv6_up_relu_bn",
                                            "conv7_up_sigmoid_err", "scaler_pca",       "ocsvm_decision",
                                            "finalize",           "synth_crops",        "conv6_conv7_fused_err",
                                            "conv1_conv2_fused"};

struct ConvSet {           // one weight set on device, packed for the kernels
    DevBuf winocs[3];      // conv2 / conv3 as Winograd F(2x2,3x3): transformed-kernel fragments (index = layer)
    DevBuf c12w1;          // conv1's fragments for the fused kernel (negated for filters with a negative BN scale)
    DevBuf c3h2, c4h2, c5h2, c6h2;   // conv3's Winograd U / conv4's / conv5's / conv6's (folded) weights as two fp16 planes (the *_h2 kernels) ...
    float c3h2_inv = 1.0f, c4h2_inv = 1.0f, c5h2_inv = 1.0f, c6h2_inv = 1.0f;   // ... and 1 / their power-of-two scales
    DevBuf c12w1h2;        // conv1's weights for the fp16 form of the fused kernel's P1, and 1 / their scale
    float c12w1h2_inv = 1.0f;
    DevBuf c12h2;          // the same U as two fp16 planes (C2H form of the fused kernel) ...
    float c12h2_inv = 1.0f, p1a = 0.0f, p1b = 0.0f;   // ... 1 / their scale, and the bound max|p1| <= p1a max|x| + p1b
    DevBuf c12;            // conv1 + conv2 fused, conv2 as Winograd F(4x4,3x3): transformed-kernel fragments (conv12_fused.hip)
    DevBuf winoup[6];      // conv5 / conv6 as four Winograd F(2x2,2x2) phase convs (index = layer)
    DevBuf wfrag[6];       // MFMA B fragments of convs 1..6
    DevBuf ep[6];          // [3][cout] bias, bn_scale, bn_shift
    int n = 0;             // number of convs packed (6 for the autoencoder, 3 for encoder.keras)
};

// weights of a non-reference architecture, as conv_generic.hip takes them (HWIO kernels + [3][cout] epilogue)
struct GenSet {
    DevBuf w[CS_MAX_CONV], ep[CS_MAX_CONV], wf[CS_MAX_CONV], wx3[CS_MAX_CONV];      // wx3: split-bf16 planes (conv_generic_x3.hip)
    bool folded[CS_MAX_CONV] = {false}, x3[CS_MAX_CONV] = {false};
    DevBuf wh2[CS_MAX_CONV];            // two fp16 planes (pack_generic_f16x2) of the layers conv_generic_x3_kernel<.., H2> runs
    float h2_inv[CS_MAX_CONV] = {0};    // 1 / their weight scale; 0 = no fp16 form
};

// The autoencoder's shape.  ref = the reference graph (64x64, 32-64-32 | 32-64-32-1): tuned kernels;
// otherwise the same layer grammar with other sizes: conv_generic.hip.
struct Arch {
    bool ref = true;
    int H = kH, W = kW, n_conv = kNConv, n_enc = kNEnc;
    int ch[CS_MAX_CONV] = {0};
    int gh[CS_MAX_CONV] = {0}, gw[CS_MAX_CONV] = {0};     // conv grid of layer l
    size_t floats[CS_MAX_CONV] = {0};                     // stored output of layer l per cell (after pool)
    size_t npix = (size_t)kH * kW;
    size_t feat() const { return floats[n_enc - 1]; }
    int cin(int l) const { return l == 0 ? 1 : ch[l - 1]; }
};

struct ProfEvent { int kid; hipEvent_t a, b; int64_t cells; };

constexpr int64_t kPackedResultCells = 65536;   // cs_screen: host results of calls up to this size come back in one copy

struct cs_model {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;                     // host-input path: H2D of chunk i+1 while chunk i computes
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_used[2] = {nullptr, nullptr};
    Arch arch;
    ConvSet ae, enc;
    GenSet gae, genc;
    bool shared_encoder = true;
    // How the fp32 contractions run (cs_model_options.precision): split16 = two-term fp16 split on the 16-bit matrix instructions
    // (PCA: three-term bf16), otherwise everything on v_mfma_f32_16x16x4_f32.  The fuse* / small_split switches are
    // cs_model_options.debug_flags: unfused forms of the same arithmetic for A/B runs and the bit-identity tests.
    int precision = CS_PRECISION_SPLIT16;
    unsigned debug_flags = 0;
    bool split16 = true;
    bool fuse12 = true, fuse45 = true, fuse67 = true, small_split = true;
    int errparts = 4;                                      // error partial sums per cell left by the last run_convs
    DevBuf w7eff, b7;      // conv7: effective weights [16][32] and bias, on device
    // detector
    bool has_det = false;
    int F = 0, fpad = 0, C = 0, cpad = 0;
    DevBuf center, scale, comps, comps_x3, mean_proj;     // comps_x3: components_ as three bf16 planes (scaler_pca_x3_kernel)
    struct Svm { DevBuf svT, svn, coef; int nsv = 0, nsv_pad = 0; double gamma = 0, rho = 0; } svm[2];
    // workspace (per chunk)
    int64_t chunk = 0;     // cells per internal pass; 0 = automatic (eff_chunk), otherwise what cs_model_set_chunk asked for
    int64_t ws_cells = 0;
    DevBuf xin, xin2, act[CS_MAX_CONV], featE, pca, errpart, dec[2], o_mse, o_mae, o_sc[2], o_pr[2], recon;
    DevBuf det_ws;      // range sums of the detector tail's split form (small calls)
    DevBuf o_pack;      // cs_screen's host results, collected on the device: six arrays back to back
    void* h_pack = nullptr;             // pinned landing buffer of a small call's results
    size_t h_pack_bytes = 0;
    // profiling
    bool prof = false;
    std::vector<ProfEvent> pending;
    double prof_ms[K_COUNT] = {0};
    int64_t prof_launches[K_COUNT] = {0};
    int64_t prof_cells[K_COUNT] = {0};
    ~cs_model()
    {
        if (h_pack) (void)hipHostFree(h_pack);
        for (auto& e : pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        for (int i = 0; i < 2; ++i) {
            if (ev_in[i]) (void)hipEventDestroy(ev_in[i]);
            if (ev_used[i]) (void)hipEventDestroy(ev_used[i]);
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

// ---------------------------------------------------------------- small helpers
namespace cs {
int upload(DevBuf& d, const void* src, size_t bytes)
{
    int rc = d.ensure(bytes ? bytes : 16);
    if (rc) return rc;
    if (bytes) HIPCHK(hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice));
    return CS_OK;
}

int check_arch(const cs_cae_weights* w, int expect_convs, const char* what)
{
    if (!w) return fail(CS_ERR_INVALID, "%s weights are NULL", what);
    if (w->height != kH || w->width != kW)
        return fail(CS_ERR_UNSUPPORTED, "%s: input %dx%d unsupported (this build has kernels for 64x64)", what, w->height, w->width);
    if (w->n_conv != expect_convs || (expect_convs == kNConv && w->n_enc != kNEnc))
        return fail(CS_ERR_UNSUPPORTED, "%s: n_conv=%d n_enc=%d unsupported (expected %d/%d)", what, w->n_conv, w->n_enc, expect_convs, kNEnc);
    for (int l = 0; l < expect_convs; ++l) {
        if (w->channels[l] != kRefChannels[l])
            return fail(CS_ERR_UNSUPPORTED, "%s: conv %d has %d filters, this build expects %d", what, l, w->channels[l], kRefChannels[l]);
        if (!w->kernel[l] || !w->bias[l]) return fail(CS_ERR_INVALID, "%s: conv %d kernel/bias is NULL", what, l);
        const bool has_bn = l < kNConv - 1;
        if (has_bn && (!w->bn_gamma[l] || !w->bn_beta[l] || !w->bn_mean[l] || !w->bn_var[l]))
            return fail(CS_ERR_INVALID, "%s: conv %d BatchNormalization arrays are NULL", what, l);
    }
    return CS_OK;
}

// Fills `a` from a weight set and decides which kernels serve it.  Generic architectures must follow the
// reference's grammar: n_enc x (conv, BN, pool), conv + BN, (n_enc - 1) x (upsample-fed conv + BN), upsample-fed
// conv with 1 filter + sigmoid -- i.e. n_conv = 2 n_enc + 1 (CAE_improved_modeltrain.py:188-216).
int describe_arch(const cs_cae_weights* w, Arch& a)
{
    if (!w) return fail(CS_ERR_INVALID, "autoencoder weights are NULL");
    a.H = w->height; a.W = w->width; a.n_conv = w->n_conv; a.n_enc = w->n_enc;
    if (a.n_conv < 3 || a.n_conv > CS_MAX_CONV || a.n_enc < 1 || a.n_conv != 2 * a.n_enc + 1)
        return fail(CS_ERR_UNSUPPORTED, "autoencoder: n_conv=%d n_enc=%d is not the reference grammar (n_conv = 2 n_enc + 1)", a.n_conv, a.n_enc);
    if (a.H <= 0 || a.W <= 0 || a.H % (1 << a.n_enc) || a.W % (1 << a.n_enc))
        return fail(CS_ERR_UNSUPPORTED, "autoencoder: input %dx%d is not divisible by 2^n_enc", a.H, a.W);
    a.ref = a.H == kH && a.W == kW && a.n_conv == kNConv && a.n_enc == kNEnc;
    for (int l = 0; l < a.n_conv; ++l) {
        a.ch[l] = w->channels[l];
        if (a.ch[l] <= 0) return fail(CS_ERR_INVALID, "autoencoder: conv %d has %d filters", l, a.ch[l]);
        if (a.ref && a.ch[l] != kRefChannels[l]) a.ref = false;
    }
    if (a.ch[a.n_conv - 1] != 1) return fail(CS_ERR_UNSUPPORTED, "autoencoder: the last conv must have 1 filter (has %d)", a.ch[a.n_conv - 1]);
    int h = a.H, wd = a.W;                                 // stored size of the tensor the next conv reads
    for (int l = 0; l < a.n_conv; ++l) {
        if (l > a.n_enc) { h *= 2; wd *= 2; }             // behind an UpSampling2D
        a.gh[l] = h; a.gw[l] = wd;
        if (l < a.n_enc) { h /= 2; wd /= 2; }             // MaxPooling2D
        a.floats[l] = (size_t)h * wd * a.ch[l];
    }
    a.npix = (size_t)a.H * a.W;
    if (!a.ref) {
        char why[160];
        for (int l = 0; l < a.n_conv; ++l)
            if (!conv_generic_supported(a.gh[l], a.gw[l], a.cin(l), a.ch[l], why, sizeof why))
                return fail(CS_ERR_UNSUPPORTED, "autoencoder conv %d: %s", l, why);
    }
    for (int l = 0; l < a.n_conv; ++l) {
        if (!w->kernel[l] || !w->bias[l]) return fail(CS_ERR_INVALID, "autoencoder: conv %d kernel/bias is NULL", l);
        if (l < a.n_conv - 1 && (!w->bn_gamma[l] || !w->bn_beta[l] || !w->bn_mean[l] || !w->bn_var[l]))
            return fail(CS_ERR_INVALID, "autoencoder: conv %d BatchNormalization arrays are NULL", l);
    }
    return CS_OK;
}

// encoder.keras weight set: must be the autoencoder's encoder half in shape
int check_encoder(const cs_cae_weights* e, const Arch& a)
{
    if (e->height != a.H || e->width != a.W || e->n_conv != a.n_enc)
        return fail(CS_ERR_UNSUPPORTED, "encoder: %dx%d with %d convs does not match the autoencoder's encoder half (%dx%d, %d)",
                    e->height, e->width, e->n_conv, a.H, a.W, a.n_enc);
    for (int l = 0; l < a.n_enc; ++l) {
        if (e->channels[l] != a.ch[l]) return fail(CS_ERR_UNSUPPORTED, "encoder: conv %d has %d filters, the autoencoder %d", l, e->channels[l], a.ch[l]);
        if (!e->kernel[l] || !e->bias[l] || !e->bn_gamma[l] || !e->bn_beta[l] || !e->bn_mean[l] || !e->bn_var[l])
            return fail(CS_ERR_INVALID, "encoder: conv %d arrays are NULL", l);
    }
    return CS_OK;
}

int require_gfx950(int device_id)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(CS_ERR_NO_DEVICE, "no HIP device visible; libcellscreen has no CPU path");
    if (device_id < 0 || device_id >= ndev) return fail(CS_ERR_INVALID, "device_id %d out of range [0,%d)", device_id, ndev);
    HIPCHK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(CS_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
    return CS_OK;
}
}  // namespace cs

// Pack convs [0, count) of a weight set.  BatchNormalization (inference) is reduced to
// y = x*s + t with s = gamma / sqrt(var + eps), t = beta - mean*s, all in fp32.
static int pack_set(ConvSet& set, const cs_cae_weights* w, int count)
{
    std::vector<float> tmp;
    for (int l = 0; l < count; ++l) {
        const int cin = l == 0 ? 1 : kRefChannels[l - 1], cout = kRefChannels[l];
        const bool folded = (l == 4 || l == 5);   // decoder convs behind an UpSampling2D (conv7 folds in conv_out.hip)
        const size_t nf = folded ? pack_conv_fragments_folded(cin, cout, nullptr, nullptr) : pack_conv_fragments(cin, cout, nullptr, nullptr);
        tmp.resize(nf);
        if (folded) pack_conv_fragments_folded(cin, cout, w->kernel[l], tmp.data());
        else pack_conv_fragments(cin, cout, w->kernel[l], tmp.data());
        int rc = upload(set.wfrag[l], tmp.data(), nf * sizeof(float));
        if (rc) return rc;
        std::vector<float> ep(3 * cout);
        for (int c = 0; c < cout; ++c) {
            const float s = w->bn_gamma[l][c] / sqrtf(w->bn_var[l][c] + w->bn_eps);
            ep[c] = w->bias[l][c];
            ep[cout + c] = s;
            ep[2 * cout + c] = w->bn_beta[l][c] - w->bn_mean[l][c] * s;
        }
        rc = upload(set.ep[l], ep.data(), ep.size() * sizeof(float));
        if (rc) return rc;
        if (l == 4 || l == 5) {
            tmp.resize(pack_wino_up_fragments(l, nullptr, nullptr));
            pack_wino_up_fragments(l, w->kernel[l], tmp.data());
            rc = upload(set.winoup[l], tmp.data(), tmp.size() * sizeof(float));
            if (rc) return rc;
        }
        if (l == 1 || l == 2) {
            tmp.resize(pack_wino_cs_fragments(l, nullptr, nullptr));
            pack_wino_cs_fragments(l, w->kernel[l], tmp.data());
            rc = upload(set.winocs[l], tmp.data(), tmp.size() * sizeof(float));
            if (rc) return rc;
        }
        if (l == 2) {
            std::vector<uint16_t> h2(pack_wino3_h2(nullptr, nullptr, nullptr));
            pack_wino3_h2(w->kernel[l], h2.data(), &set.c3h2_inv);
            rc = upload(set.c3h2, h2.data(), h2.size() * sizeof(uint16_t));
            if (rc) return rc;
        }
        if (l == 3) {
            std::vector<uint16_t> h2(pack_conv4_f16x2(nullptr, nullptr, nullptr));
            pack_conv4_f16x2(w->kernel[l], h2.data(), &set.c4h2_inv);
            rc = upload(set.c4h2, h2.data(), h2.size() * sizeof(uint16_t));
            if (rc) return rc;
        }
        if (l == 4) {
            std::vector<float> weff(pack_generic_folded(cin, cout, nullptr, nullptr));
            pack_generic_folded(cin, cout, w->kernel[l], weff.data());
            std::vector<uint16_t> h2(pack_conv5_f16x2(nullptr, nullptr, nullptr));
            pack_conv5_f16x2(weff.data(), h2.data(), &set.c5h2_inv);
            rc = upload(set.c5h2, h2.data(), h2.size() * sizeof(uint16_t));
            if (rc) return rc;
        }
        if (l == 5) {
            std::vector<float> weff(pack_generic_folded(cin, cout, nullptr, nullptr));
            pack_generic_folded(cin, cout, w->kernel[l], weff.data());
            std::vector<uint16_t> h2(pack_conv6_f16x2(nullptr, nullptr, nullptr));
            pack_conv6_f16x2(weff.data(), h2.data(), &set.c6h2_inv);
            rc = upload(set.c6h2, h2.data(), h2.size() * sizeof(uint16_t));
            if (rc) return rc;
        }
        if (l == 0) {
            tmp.resize(pack_conv12_conv1_fragments(nullptr, nullptr, nullptr));
            pack_conv12_conv1_fragments(w->kernel[l], ep.data() + cout, tmp.data());
            rc = upload(set.c12w1, tmp.data(), tmp.size() * sizeof(float));
            if (rc) return rc;
            pack_conv12_p1_bound(w->kernel[l], ep.data(), &set.p1a, &set.p1b);
            std::vector<unsigned int> wh(pack_conv12_conv1_h2(nullptr, nullptr, nullptr, nullptr));
            pack_conv12_conv1_h2(w->kernel[l], ep.data() + cout, wh.data(), &set.c12w1h2_inv);
            rc = upload(set.c12w1h2, wh.data(), wh.size() * sizeof(unsigned int));
            if (rc) return rc;
        }
        if (l == 1) {
            tmp.resize(pack_conv12_fragments(nullptr, nullptr, nullptr));
            pack_conv12_fragments(w->kernel[l], ep.data() + cout, tmp.data());
            rc = upload(set.c12, tmp.data(), tmp.size() * sizeof(float));
            if (rc) return rc;
            std::vector<unsigned int> uh(pack_conv12_fragments_h2(nullptr, nullptr, nullptr, nullptr));
            pack_conv12_fragments_h2(w->kernel[l], ep.data() + cout, uh.data(), &set.c12h2_inv);
            rc = upload(set.c12h2, uh.data(), uh.size() * sizeof(unsigned int));
            if (rc) return rc;
        }
    }
    set.n = count;
    return CS_OK;
}

// Generic architectures: HWIO kernels as they are + the [3][cout] epilogue (bias only for the sigmoid conv).
static int pack_generic(GenSet& set, const cs_cae_weights* w, const Arch& a, int count, bool split16)
{
    for (int l = 0; l < count; ++l) {
        const int cin = a.cin(l), cout = a.ch[l];
        int rc = upload(set.w[l], w->kernel[l], sizeof(float) * 9 * cin * cout);
        if (rc) return rc;
        // upsample-fed convs with a folded form: the MFMA phase convs, and the 1-filter last conv on the vector ALU
        set.folded[l] = l > a.n_enc && ((l < a.n_conv - 1 && conv_generic_folds(a.gh[l], a.gw[l], cin, cout)) ||
                                        (l == a.n_conv - 1 && cout == 1 && cin % 4 == 0));
        if (set.folded[l]) {   // upsample-fed conv: effective 2x2 kernels per output phase
            std::vector<float> wf(pack_generic_folded(cin, cout, nullptr, nullptr));
            pack_generic_folded(cin, cout, w->kernel[l], wf.data());
            if ((rc = upload(set.wf[l], wf.data(), wf.size() * sizeof(float)))) return rc;
            if (split16 && l == a.n_conv - 1 && cout == 1 && conv_last_x3_takes(a.gh[l], a.gw[l], cin)) {
                std::vector<uint16_t> pl(pack_last_bf16x3(cin, nullptr, nullptr));
                pack_last_bf16x3(cin, wf.data(), pl.data());
                if ((rc = upload(set.wx3[l], pl.data(), pl.size() * sizeof(uint16_t)))) return rc;
                set.x3[l] = true;
            }
            if (split16 && l < a.n_conv - 1 && conv_generic_x3_takes(a.gh[l], a.gw[l], cin, cout, 1)) {
                std::vector<uint16_t> pl(pack_generic_bf16x3(16, cin, cout, nullptr, nullptr));
                pack_generic_bf16x3(16, cin, cout, wf.data(), pl.data());
                if ((rc = upload(set.wx3[l], pl.data(), pl.size() * sizeof(uint16_t)))) return rc;
                set.x3[l] = true;
                std::vector<uint16_t> ph(pack_generic_f16x2(16, cin, cout, nullptr, nullptr, nullptr));
                pack_generic_f16x2(16, cin, cout, wf.data(), ph.data(), &set.h2_inv[l]);
                if ((rc = upload(set.wh2[l], ph.data(), ph.size() * sizeof(uint16_t)))) return rc;
            }
        } else if (split16 && l <= a.n_enc && l < a.n_conv - 1 && conv_generic_x3_takes(a.gh[l], a.gw[l], cin, cout, 0)) {
            std::vector<uint16_t> pl(pack_generic_bf16x3(9, cin, cout, nullptr, nullptr));
            pack_generic_bf16x3(9, cin, cout, w->kernel[l], pl.data());
            if ((rc = upload(set.wx3[l], pl.data(), pl.size() * sizeof(uint16_t)))) return rc;
            set.x3[l] = true;
            std::vector<uint16_t> ph(pack_generic_f16x2(9, cin, cout, nullptr, nullptr, nullptr));
            pack_generic_f16x2(9, cin, cout, w->kernel[l], ph.data(), &set.h2_inv[l]);
            if ((rc = upload(set.wh2[l], ph.data(), ph.size() * sizeof(uint16_t)))) return rc;
        }
        std::vector<float> ep(3 * cout, 0.0f);
        const bool has_bn = w->bn_gamma[l] != nullptr;
        for (int c = 0; c < cout; ++c) {
            ep[c] = w->bias[l][c];
            if (has_bn) {
                const float s_ = w->bn_gamma[l][c] / sqrtf(w->bn_var[l][c] + w->bn_eps);
                ep[cout + c] = s_;
                ep[2 * cout + c] = w->bn_beta[l][c] - w->bn_mean[l][c] * s_;
            }
        }
        if ((rc = upload(set.ep[l], ep.data(), ep.size() * sizeof(float)))) return rc;
    }
    return CS_OK;
}

static bool same_encoder(const cs_cae_weights* a, const cs_cae_weights* e, const Arch& ar)
{
    if (a->bn_eps != e->bn_eps) return false;
    for (int l = 0; l < ar.n_enc; ++l) {
        const int cin = ar.cin(l), cout = ar.ch[l];
        if (memcmp(a->kernel[l], e->kernel[l], sizeof(float) * 9 * cin * cout)) return false;
        if (memcmp(a->bias[l], e->bias[l], sizeof(float) * cout)) return false;
        if (memcmp(a->bn_gamma[l], e->bn_gamma[l], sizeof(float) * cout)) return false;
        if (memcmp(a->bn_beta[l], e->bn_beta[l], sizeof(float) * cout)) return false;
        if (memcmp(a->bn_mean[l], e->bn_mean[l], sizeof(float) * cout)) return false;
        if (memcmp(a->bn_var[l], e->bn_var[l], sizeof(float) * cout)) return false;
    }
    return true;
}

static int pack_svm(cs_model::Svm& s, const cs_ocsvm_params& p, int D, const char* what)
{
    if (p.n_sv <= 0 || !p.support_vectors || !p.dual_coef)
        return fail(CS_ERR_INVALID, "%s detector: n_sv=%d or NULL arrays", what, p.n_sv);
    s.nsv = p.n_sv;
    s.nsv_pad = (p.n_sv + 15) / 16 * 16;      // blocks of 16 support vectors (padding rows carry a zero coefficient: exact zeros, but they cost time)
    s.gamma = p.gamma;
    s.rho = p.rho;
    std::vector<double> svT((size_t)D * s.nsv_pad, 0.0), coef(s.nsv_pad, 0.0), svn(s.nsv_pad, 0.0);
    for (int i = 0; i < p.n_sv; ++i) {
        coef[i] = p.dual_coef[i];
        double nn = 0.0;
        for (int d = 0; d < D; ++d) {
            const double v = p.support_vectors[(size_t)i * D + d];
            svT[(size_t)d * s.nsv_pad + i] = v;
            nn = fma(v, v, nn);
        }
        svn[i] = nn;
    }
    {
        int rcn = upload(s.svn, svn.data(), svn.size() * sizeof(double));
        if (rcn) return rcn;
    }
    int rc = upload(s.svT, svT.data(), svT.size() * sizeof(double));
    if (rc) return rc;
    return upload(s.coef, coef.data(), coef.size() * sizeof(double));
}

// Cells per internal pass.  Unless the caller fixed it (cs_model_set_chunk): host input streams through two staging
// buffers and is fastest in 16,384-cell chunks (the first copy is the only exposed one); device-resident input takes
// as many cells per pass as a ~28 GB workspace holds, up to 65,536 -- measured on the reference graph: 2.46 / 2.57 /
// 2.58 / 2.60 M cells/s at 16,384 / 32,768 / 65,536 / 131,072 (the detector-tail kernels want >> 256 workgroups).
static int64_t eff_chunk(const cs_model* m, int in_kind)
{
    if (m->chunk > 0) return m->chunk;
    if (in_kind == CS_MEM_HOST) return 16384;
    size_t per_cell = m->arch.npix * sizeof(float);
    for (int l = 0; l < m->arch.n_conv - 1; ++l) per_cell += m->arch.floats[l] * sizeof(float);
    int64_t c = (int64_t)((size_t)28e9 / (per_cell ? per_cell : 1));
    c = c / 1024 * 1024;
    return c < 1024 ? 1024 : (c > 65536 ? 65536 : c);
}

static int ensure_workspace(cs_model* m, int64_t cells, bool need_recon)
{
    if (cells > m->ws_cells) {
        int rc;
        if ((rc = m->xin.ensure((size_t)cells * m->arch.npix * sizeof(float)))) return rc;
        // p1 (131 KB per cell, the largest activation) is not materialised when conv1 + conv2 run fused: allocated on demand
        const bool skip_p1 = m->arch.ref && m->fuse12;
        for (int l = skip_p1 ? 1 : 0; l < m->arch.n_conv - 1; ++l)
            if ((rc = m->act[l].ensure((size_t)cells * m->arch.floats[l] * sizeof(float)))) return rc;
        if ((rc = m->featE.ensure((size_t)cells * m->arch.feat() * sizeof(float)))) return rc;
        if ((rc = m->pca.ensure((size_t)cells * 256 * sizeof(float)))) return rc;
        if (m->small_split && (rc = m->det_ws.ensure(det_split_ws_bytes(m->C)))) return rc;
        if ((rc = m->errpart.ensure((size_t)cells * 16 * sizeof(float)))) return rc;      // up to 8 partial (sq, abs) pairs per cell
        for (int d = 0; d < 2; ++d) {
            if ((rc = m->dec[d].ensure((size_t)cells * sizeof(double)))) return rc;
            if ((rc = m->o_sc[d].ensure((size_t)cells * sizeof(double)))) return rc;
            if ((rc = m->o_pr[d].ensure((size_t)cells))) return rc;
        }
        if ((rc = m->o_mse.ensure((size_t)cells * sizeof(float)))) return rc;
        if ((rc = m->o_mae.ensure((size_t)cells * sizeof(float)))) return rc;
        m->ws_cells = cells;
    }
    if (need_recon || !m->arch.ref) {   // the generic path always materialises the reconstruction
        int rc = m->recon.ensure((size_t)m->ws_cells * m->arch.npix * sizeof(float));
        if (rc) return rc;
    }
    return CS_OK;
}

// ---------------------------------------------------------------- profiled launches
struct Launch {
    cs_model* m; int kid; int64_t cells; ProfEvent ev; bool on;
    Launch(cs_model* m_, int kid_, int64_t cells_) : m(m_), kid(kid_), cells(cells_), on(m_->prof)
    {
        if (on) {
            ev.kid = kid; ev.cells = cells;
            if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) { on = false; return; }
            (void)hipEventRecord(ev.a, m->stream);
        }
    }
    ~Launch()
    {
        if (on) { (void)hipEventRecord(ev.b, m->stream); m->pending.push_back(ev); }
    }
};

static int drain_profile(cs_model* m)
{
    for (auto& e : m->pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            m->prof_ms[e.kid] += ms;
            m->prof_launches[e.kid] += 1;
            m->prof_cells[e.kid] += e.cells;
        }
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    m->pending.clear();
    return CS_OK;
}

#define LAUNCH(kid, cells, call)                                                              \
    do {                                                                                      \
        hipError_t le__;                                                                      \
        { Launch l__(m, kid, cells); le__ = (call); }                                         \
        if (le__ != hipSuccess)                                                               \
            return fail(CS_ERR_HIP, "launch %s failed: %s", kKernelNames[kid], hipGetErrorString(le__)); \
    } while (0)

// Runs convs [first, last] (0-based, inclusive) of weight set `set` on `nc` cells.
// Layer l reads act[l-1] (or x for l == 0) and writes act[l]; conv 7 (index 6) writes
// errpart (and recon when asked).
static int run_convs_generic(cs_model* m, const GenSet& set, const float* x, int64_t nc, int first, int last, float* recon)
{
    const Arch& a = m->arch;
    for (int l = first; l <= last; ++l) {
        const float* in = l == 0 ? x : m->act[l - 1].as<float>();
        const bool is_last = l == a.n_conv - 1;
        float* out = is_last ? (recon ? recon : m->recon.as<float>()) : m->act[l].as<float>();
        const int epi = is_last ? GEN_EPI_SIGMOID : (l < a.n_enc ? GEN_EPI_BN_POOL : GEN_EPI_BN);
        const int kid = l < 6 ? K_CONV1 + l : K_CONV7_ERR;       // profile bucket: by position
        if (set.x3[l] && m->split16 && is_last) {
            LAUNCH(kid, nc,
                   launch_conv_last_x3(in, set.wx3[l].as<uint16_t>(), set.ep[l].as<float>(), out, nc, a.gh[l], a.gw[l], a.cin(l), m->stream));
            LAUNCH(K_CONV7_ERR, nc, launch_recon_err(out, x, nc, (int)a.npix, m->errpart.as<float>(), m->stream));
            continue;
        }
        if (set.x3[l] && m->split16) {
            const bool h2 = set.h2_inv[l] != 0.0f;
            LAUNCH(kid, nc,
                   launch_conv_generic_x3(in, h2 ? set.wh2[l].as<uint16_t>() : set.wx3[l].as<uint16_t>(), set.ep[l].as<float>(), out, nc,
                                          a.gh[l], a.gw[l], a.cin(l), a.ch[l], l > a.n_enc, epi, m->stream, h2 ? set.h2_inv[l] : 0.0f));
            continue;
        }
        LAUNCH(kid, nc,
               launch_conv_generic(in, set.w[l].as<float>(), set.ep[l].as<float>(), out, nc, a.gh[l], a.gw[l], a.cin(l), a.ch[l],
                                   l > a.n_enc, epi, m->stream, set.folded[l] ? set.wf[l].as<float>() : nullptr));
        if (is_last)
            LAUNCH(K_CONV7_ERR, nc, launch_recon_err(out, x, nc, (int)a.npix, m->errpart.as<float>(), m->stream));
    }
    m->errparts = 4;
    return CS_OK;
}

static int run_convs(cs_model* m, const ConvSet& set, const float* x, int64_t nc, int first, int last,
                     float* recon)
{
    if (!m->arch.ref) return run_convs_generic(m, &set == &m->enc ? m->genc : m->gae, x, nc, first, last, recon);
    // screening needs neither a6 nor the reconstruction: conv6, conv7 and the error sums run as one kernel
    const bool fused = m->fuse67 && first <= 5 && last >= 6 && !recon;
    // conv1 + conv2 as one kernel whenever p1 itself is not asked for (it is never written then)
    const bool fused12 = m->fuse12 && first == 0 && last >= 1;
    const bool h2 = m->split16;
    if (fused12) {
        LAUNCH(K_CONV12_FUSED, nc,
               launch_conv12_fused(x, set.c12w1.as<float>(), set.ep[0].as<float>(), set.c12.as<float>(), set.ep[1].as<float>(),
                                   m->act[1].as<float>(), nc, m->stream, h2 ? set.c12h2.as<unsigned int>() : nullptr, set.p1a, set.p1b,
                                   set.c12h2_inv, h2 ? set.c12w1h2.as<unsigned int>() : nullptr, set.c12w1h2_inv));
    }
    if (!fused12 && first == 0) {   // the stand-alone conv1 (stage tap / debug flag) needs p1 in HBM
        int rc = m->act[0].ensure((size_t)m->ws_cells * m->arch.floats[0] * sizeof(float));
        if (rc) return rc;
    }
    // conv4 + conv5 as one kernel when a4 itself is not asked for (conv5 is bound by its HBM writes: conv4 rides under them)
    const bool fused45 = m->fuse45 && h2 && first <= 3 && last >= 4;
    for (int l = fused12 ? 2 : first; l <= last && l < (fused ? 5 : 6); ++l) {
        const float* in = l == 0 ? x : m->act[l - 1].as<float>();
        if (l == 3 && fused45) {
            LAUNCH(K_CONV5, nc,
                   launch_conv45_h2(in, set.c4h2.as<uint16_t>(), set.c4h2_inv, set.ep[3].as<float>(), set.c5h2.as<uint16_t>(), set.c5h2_inv,
                                    set.ep[4].as<float>(), m->act[4].as<float>(), nc, m->stream));
            ++l;            // conv5 is done too
            continue;
        }
        if (l == 4 && h2) {
            LAUNCH(K_CONV5, nc,
                   launch_conv5_h2(in, set.c5h2.as<uint16_t>(), set.c5h2_inv, set.ep[l].as<float>(), m->act[l].as<float>(), nc, m->stream));
            continue;
        }
        if (l == 4 || l == 5) {     // fp32: four Winograd F(2x2,2x2) phase convs over the stored grid
            LAUNCH(K_CONV1 + l, nc,
                   launch_conv_wino_up(l, in, set.winoup[l].as<float>(), set.ep[l].as<float>(), m->act[l].as<float>(), nc, m->stream));
            continue;
        }
        if (l == 2 && h2) {
            LAUNCH(K_CONV3, nc,
                   launch_conv3_wino_h2(in, set.c3h2.as<uint16_t>(), set.c3h2_inv, set.ep[l].as<float>(), m->act[l].as<float>(), nc, m->stream));
            continue;
        }
        if (l == 1 || l == 2) {     // fp32 Winograd F(2x2,3x3)
            LAUNCH(K_CONV1 + l, nc,
                   launch_conv_wino_cs(l, in, set.winocs[l].as<float>(), set.ep[l].as<float>(), m->act[l].as<float>(), nc, m->stream));
            continue;
        }
        if (l == 3 && h2) {
            LAUNCH(K_CONV4, nc,
                   launch_conv4_h2(in, set.c4h2.as<uint16_t>(), set.c4h2_inv, set.ep[l].as<float>(), m->act[l].as<float>(), nc, m->stream));
            continue;
        }
        LAUNCH(K_CONV1 + l, nc,      // conv1 alone; conv4 in fp32
               launch_conv_mfma(l, in, set.wfrag[l].as<float>(), set.ep[l].as<float>(), m->act[l].as<float>(), nc, m->stream, false));
    }
    if (fused && h2) {
        LAUNCH(K_CONV67_FUSED, nc,
               launch_conv67_h2(m->act[4].as<float>(), set.c6h2.as<uint16_t>(), set.c6h2_inv, set.ep[5].as<float>(), x, m->w7eff.as<float>(),
                                m->b7.as<float>(), m->errpart.as<float>(), nc, m->stream));
        m->errparts = conv67_fused_nparts();
    } else if (fused) {
        LAUNCH(K_CONV67_FUSED, nc,
               launch_conv67_fused(m->act[4].as<float>(), set.winoup[5].as<float>(), set.ep[5].as<float>(), x, m->w7eff.as<float>(),
                                   m->b7.as<float>(), m->errpart.as<float>(), nc, m->stream));
        m->errparts = conv67_fused_nparts();
    } else if (last >= 6) {
        LAUNCH(K_CONV7_ERR, nc,
               launch_conv7_err(m->act[5].as<float>(), x, m->w7eff.as<float>(), m->b7.as<float>(), m->errpart.as<float>(), recon, nc, m->stream));
        m->errparts = 4;
    }
    return CS_OK;
}

// An error return from inside a call must not leave a copy from / to the caller's buffers in flight: every compute
// entry point holds one of these; its destructor drains both streams unless the call reached its normal end
// (end_call synchronises there itself).
struct DrainGuard {
    cs_model* m;
    bool armed = true;
    explicit DrainGuard(cs_model* mm) : m(mm) {}
    ~DrainGuard()
    {
        if (armed && m) {
            if (m->copy_stream) (void)hipStreamSynchronize(m->copy_stream);
            if (m->stream) (void)hipStreamSynchronize(m->stream);
        }
    }
    int done(int rc) { armed = false; return rc; }
};

static int begin_call(cs_model* m)
{
    if (!m) return fail(CS_ERR_INVALID, "model handle is NULL");
    HIPCHK(hipSetDevice(m->device));
    return CS_OK;
}

static int end_call(cs_model* m)
{
    HIPCHK(hipStreamSynchronize(m->stream));
    return drain_profile(m);
}

// Makes chunk [off, off+nc) of a caller buffer available on the device.
static int stage_in(cs_model* m, const float* base, int kind, int64_t off, int64_t nc, size_t per_cell,
                    DevBuf& staging, const float** dev)
{
    if (kind == CS_MEM_DEVICE) { *dev = base + (size_t)off * per_cell; return CS_OK; }
    int rc = staging.ensure((size_t)nc * per_cell * sizeof(float));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(staging.p, base + (size_t)off * per_cell, (size_t)nc * per_cell * sizeof(float),
                          hipMemcpyHostToDevice, m->stream));
    *dev = staging.as<float>();
    return CS_OK;
}

template <class T>
static int stage_out(cs_model* m, T* user, int kind, int64_t off, int64_t count, const T* dev)
{
    if (!user || kind == CS_MEM_DEVICE) return CS_OK;  // device outputs are written in place
    HIPCHK(hipMemcpyAsync(user + off, dev, (size_t)count * sizeof(T), hipMemcpyDeviceToHost, m->stream));
    return CS_OK;
}

template <class T>
static T* out_ptr(T* user, int kind, int64_t off, DevBuf& tmp)
{
    if (!user) return nullptr;
    return kind == CS_MEM_DEVICE ? user + off : tmp.as<T>();
}

static int check_kind(int k, const char* what)
{
    if (k != CS_MEM_HOST && k != CS_MEM_DEVICE) return fail(CS_ERR_INVALID, "%s must be CS_MEM_HOST or CS_MEM_DEVICE", what);
    return CS_OK;
}

// ================================================================ C ABI
extern "C" {

int cs_abi_version(void) { return CS_ABI_VERSION; }

const char* cs_status_string(int s)
{
    switch (s) {
        case CS_OK: return "ok";
        case CS_ERR_INVALID: return "invalid argument";
        case CS_ERR_IO: return "i/o error";
        case CS_ERR_FORMAT: return "malformed model file";
        case CS_ERR_NO_DEVICE: return "no usable gfx950 device";
        case CS_ERR_HIP: return "HIP runtime error";
        case CS_ERR_UNSUPPORTED: return "unsupported architecture or size";
        case CS_ERR_NOMEM: return "out of memory";
        case CS_ERR_NO_DETECTOR: return "model has no detector parameters";
        default: return "unknown status";
    }
}

const char* cs_last_error(void) { return last_error_cstr(); }

int cs_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// cs_model_options -> (precision, debug flags).  The environment may override both for a debugging session without a rebuild
// of the caller (CS_DEBUG_PRECISION=exact|split16, CS_DEBUG_FLAGS=<mask>); nothing else in the library reads the environment
// to choose arithmetic.
static int resolve_options(const cs_model_options* o, int* precision, unsigned* flags)
{
    *precision = CS_PRECISION_SPLIT16;
    *flags = 0;
    if (o) {
        if (o->struct_size < 12 || o->struct_size > 4096) return fail(CS_ERR_INVALID, "cs_model_options.struct_size = %u (set it to sizeof(cs_model_options))", o->struct_size);
        if (o->precision != CS_PRECISION_SPLIT16 && o->precision != CS_PRECISION_FP32_EXACT)
            return fail(CS_ERR_INVALID, "cs_model_options.precision = %d (CS_PRECISION_SPLIT16 or CS_PRECISION_FP32_EXACT)", o->precision);
        const unsigned known = CS_DEBUG_NO_FUSE12 | CS_DEBUG_NO_FUSE45 | CS_DEBUG_NO_FUSE67 | CS_DEBUG_NO_SMALL_SPLIT;
        if (o->debug_flags & ~known) return fail(CS_ERR_INVALID, "cs_model_options.debug_flags = 0x%x has unknown bits", o->debug_flags);
        if (o->struct_size >= sizeof(cs_model_options))
            for (unsigned r : o->reserved) if (r) return fail(CS_ERR_INVALID, "cs_model_options.reserved must be 0");
        *precision = o->precision;
        *flags = o->debug_flags;
    }
    if (const char* e = getenv("CS_DEBUG_PRECISION")) {
        if (!strcmp(e, "exact")) *precision = CS_PRECISION_FP32_EXACT;
        else if (!strcmp(e, "split16")) *precision = CS_PRECISION_SPLIT16;
        else return fail(CS_ERR_INVALID, "CS_DEBUG_PRECISION=%s (exact or split16)", e);
    }
    if (const char* e = getenv("CS_DEBUG_FLAGS")) *flags |= (unsigned)strtoul(e, nullptr, 0) & 0xfu;
    return CS_OK;
}

int cs_model_from_arrays(const cs_cae_weights* autoencoder, const cs_cae_weights* encoder,
                         const cs_detector_params* det, int device_id, const cs_model_options* options, cs_model** out)
{
    if (!out) return fail(CS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int precision;
    unsigned dflags;
    {
        int rco = resolve_options(options, &precision, &dflags);
        if (rco) return rco;
    }
    Arch arch;
    int rc = describe_arch(autoencoder, arch);
    if (rc) return rc;
    if (arch.ref && (rc = check_arch(autoencoder, kNConv, "autoencoder"))) return rc;
    if (encoder && (rc = check_encoder(encoder, arch))) return rc;
    if ((rc = require_gfx950(device_id))) return rc;

    cs_model* m = new (std::nothrow) cs_model();
    if (!m) return fail(CS_ERR_NOMEM, "host allocation failed");
    m->device = device_id;
    m->arch = arch;
    m->precision = precision;
    m->debug_flags = dflags;
    m->split16 = precision == CS_PRECISION_SPLIT16;
    m->fuse12 = !(dflags & CS_DEBUG_NO_FUSE12);
    m->fuse45 = !(dflags & CS_DEBUG_NO_FUSE45);
    m->fuse67 = !(dflags & CS_DEBUG_NO_FUSE67);
    m->small_split = !(dflags & CS_DEBUG_NO_SMALL_SPLIT);
#define FAIL_IF(x) do { int r__ = (x); if (r__) { delete m; return r__; } } while (0)
    {
        hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking);
        for (int i = 0; i < 2 && e == hipSuccess; ++i) {
            e = hipEventCreateWithFlags(&m->ev_in[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_used[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) { delete m; return fail(CS_ERR_HIP, "stream/event creation: %s", hipGetErrorString(e)); }
    }
    if (arch.ref) {
        FAIL_IF(pack_set(m->ae, autoencoder, 6));
        float weff[16 * 32];
        conv7_effective_weights(autoencoder->kernel[6], weff);
        FAIL_IF(upload(m->w7eff, weff, sizeof weff));
        FAIL_IF(upload(m->b7, autoencoder->bias[6], sizeof(float)));
    } else {
        FAIL_IF(pack_generic(m->gae, autoencoder, arch, arch.n_conv, m->split16));
    }
    m->shared_encoder = !encoder || same_encoder(autoencoder, encoder, arch);
    if (!m->shared_encoder) {
        if (arch.ref) FAIL_IF(pack_set(m->enc, encoder, kNEnc));
        else FAIL_IF(pack_generic(m->genc, encoder, arch, arch.n_enc, m->split16));
    }

    if (det) {
        if (det->n_features != (int)arch.feat()) {
            delete m;
            return fail(CS_ERR_INVALID, "detector n_features=%d but the encoder emits %zu", det->n_features, arch.feat());
        }
        if (det->n_components <= 0 || det->n_components > 128 || !det->scaler_center || !det->scaler_scale ||
            !det->pca_components || !det->pca_mean_proj) {
            delete m;
            return fail(CS_ERR_INVALID, "detector: n_components=%d (1..128) or NULL arrays", det->n_components);
        }
        m->F = det->n_features;
        m->fpad = (m->F + 511) / 512 * 512;
        m->C = det->n_components;
        m->cpad = (m->C + 15) / 16 * 16;
        FAIL_IF(upload(m->center, det->scaler_center, sizeof(float) * m->F));
        FAIL_IF(upload(m->scale, det->scaler_scale, sizeof(double) * m->F));
        std::vector<float> cp((size_t)m->cpad * m->fpad, 0.0f);
        for (int c = 0; c < m->C; ++c)
            memcpy(&cp[(size_t)c * m->fpad], det->pca_components + (size_t)c * m->F, sizeof(float) * m->F);
        FAIL_IF(upload(m->comps, cp.data(), cp.size() * sizeof(float)));
        {
            std::vector<uint16_t> pl(pack_pca_bf16x3(nullptr, m->cpad, m->fpad, nullptr));
            pack_pca_bf16x3(cp.data(), m->cpad, m->fpad, pl.data());
            FAIL_IF(upload(m->comps_x3, pl.data(), pl.size() * sizeof(uint16_t)));
        }
        FAIL_IF(upload(m->mean_proj, det->pca_mean_proj, sizeof(float) * m->C));
        FAIL_IF(pack_svm(m->svm[0], det->conservative, m->C, "conservative"));
        FAIL_IF(pack_svm(m->svm[1], det->moderate, m->C, "moderate"));
        m->has_det = true;
    }
#undef FAIL_IF
    *out = m;
    return CS_OK;
}

// ---- native model_dir ------------------------------------------------------------
static int fill_weights(const TensorArchive& ar, const std::string& prefix, int n_conv, cs_cae_weights* w,
                        const std::string& path)
{
    for (int l = 0; l < n_conv; ++l) {
        const int cin = l == 0 ? 1 : w->channels[l - 1], cout = w->channels[l];
        const std::string cl = prefix + ".conv" + std::to_string(l), bl = prefix + ".bn" + std::to_string(l);
        const Tensor* k = ar.get(cl + ".kernel");
        const Tensor* b = ar.get(cl + ".bias");
        if (!k || !b || !k->f32() || !b->f32() || k->numel() != (size_t)9 * cin * cout || b->numel() != (size_t)cout)
            return fail(CS_ERR_FORMAT, "%s: missing or mis-shaped %s.kernel/.bias", path.c_str(), cl.c_str());
        w->kernel[l] = k->f32();
        w->bias[l] = b->f32();
        if (l < w->n_conv - 1 || prefix == "enc") {
            const Tensor* g = ar.get(bl + ".gamma"); const Tensor* be = ar.get(bl + ".beta");
            const Tensor* mu = ar.get(bl + ".mean"); const Tensor* va = ar.get(bl + ".var");
            if (!g || !be || !mu || !va || !g->f32() || !be->f32() || !mu->f32() || !va->f32() ||
                g->numel() != (size_t)cout || be->numel() != (size_t)cout || mu->numel() != (size_t)cout || va->numel() != (size_t)cout)
                return fail(CS_ERR_FORMAT, "%s: missing or mis-shaped %s.*", path.c_str(), bl.c_str());
            w->bn_gamma[l] = g->f32(); w->bn_beta[l] = be->f32(); w->bn_mean[l] = mu->f32(); w->bn_var[l] = va->f32();
        }
    }
    return CS_OK;
}

static int fill_svm(const TensorArchive& ar, const std::string& prefix, int C, cs_ocsvm_params* p, const std::string& path)
{
    const Tensor* sv = ar.get(prefix + ".sv"); const Tensor* dc = ar.get(prefix + ".dual_coef");
    const Tensor* g = ar.get(prefix + ".gamma"); const Tensor* r = ar.get(prefix + ".rho");
    if (!sv || !dc || !g || !r || !sv->f64() || !dc->f64() || !g->f64() || !r->f64() || sv->dims.size() != 2 ||
        sv->dims[1] != (uint64_t)C || dc->numel() != sv->dims[0] || g->numel() != 1 || r->numel() != 1)
        return fail(CS_ERR_FORMAT, "%s: missing or mis-shaped %s.*", path.c_str(), prefix.c_str());
    p->n_sv = (int32_t)sv->dims[0];
    p->support_vectors = sv->f64();
    p->dual_coef = dc->f64();
    p->gamma = g->f64()[0];
    p->rho = r->f64()[0];
    return CS_OK;
}

int cs_model_load(const char* model_dir, int device_id, const cs_model_options* options, cs_model** out)
{
    if (!out) return fail(CS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!model_dir) return fail(CS_ERR_INVALID, "model_dir is NULL");
    const std::string dir(model_dir);
    TensorArchive cae, det;
    std::string err = cae.load(dir + "/cae.bin");
    if (!err.empty()) return fail(err.rfind("cannot open", 0) == 0 ? CS_ERR_IO : CS_ERR_FORMAT, "%s", err.c_str());

    const Tensor* meta = cae.get("meta");
    const Tensor* eps = cae.get("bn_eps");
    if (!meta || !meta->i32() || meta->numel() < 5 || !eps || !eps->f32() || eps->numel() != 1)
        return fail(CS_ERR_FORMAT, "%s/cae.bin: missing meta / bn_eps", model_dir);
    const int32_t* mi = meta->i32();
    cs_cae_weights ae;
    memset(&ae, 0, sizeof ae);
    ae.height = mi[0]; ae.width = mi[1]; ae.n_conv = mi[2]; ae.n_enc = mi[3];
    if (ae.n_conv < 2 || ae.n_conv > CS_MAX_CONV || meta->numel() != (size_t)(4 + ae.n_conv))
        return fail(CS_ERR_FORMAT, "%s/cae.bin: bad meta", model_dir);
    for (int l = 0; l < ae.n_conv; ++l) ae.channels[l] = mi[4 + l];
    ae.bn_eps = eps->f32()[0];
    int rc = fill_weights(cae, "ae", ae.n_conv, &ae, dir + "/cae.bin");
    if (rc) return rc;

    cs_cae_weights en;
    const cs_cae_weights* enp = nullptr;
    if (cae.get("enc.conv0.kernel")) {
        en = ae;
        en.n_conv = ae.n_enc;
        for (int l = 0; l < CS_MAX_CONV; ++l) en.kernel[l] = en.bias[l] = en.bn_gamma[l] = en.bn_beta[l] = en.bn_mean[l] = en.bn_var[l] = nullptr;
        rc = fill_weights(cae, "enc", en.n_conv, &en, dir + "/cae.bin");
        if (rc) return rc;
        enp = &en;
    }

    cs_detector_params dp;
    const cs_detector_params* dpp = nullptr;
    FILE* probe = fopen((dir + "/detector.bin").c_str(), "rb");
    if (probe) {
        fclose(probe);
        err = det.load(dir + "/detector.bin");
        if (!err.empty()) return fail(CS_ERR_FORMAT, "%s", err.c_str());
        memset(&dp, 0, sizeof dp);
        const Tensor* ce = det.get("scaler.center"); const Tensor* sc = det.get("scaler.scale");
        const Tensor* co = det.get("pca.components"); const Tensor* mp = det.get("pca.mean_proj");
        if (!ce || !sc || !co || !mp || !ce->f32() || !sc->f64() || !co->f32() || !mp->f32() || co->dims.size() != 2 ||
            ce->numel() != co->dims[1] || sc->numel() != co->dims[1] || mp->numel() != co->dims[0])
            return fail(CS_ERR_FORMAT, "%s/detector.bin: missing or mis-shaped scaler/pca tensors", model_dir);
        dp.n_features = (int32_t)co->dims[1];
        dp.n_components = (int32_t)co->dims[0];
        dp.scaler_center = ce->f32(); dp.scaler_scale = sc->f64();
        dp.pca_components = co->f32(); dp.pca_mean_proj = mp->f32();
        if ((rc = fill_svm(det, "svm_conservative", dp.n_components, &dp.conservative, dir + "/detector.bin"))) return rc;
        if ((rc = fill_svm(det, "svm_moderate", dp.n_components, &dp.moderate, dir + "/detector.bin"))) return rc;
        dpp = &dp;
    }
    return cs_model_from_arrays(&ae, enp, dpp, device_id, options, out);
}

void cs_model_free(cs_model* m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    delete m;
}

int cs_model_get_info(const cs_model* m, cs_model_info* info)
{
    if (!m || !info) return fail(CS_ERR_INVALID, "NULL argument");
    memset(info, 0, sizeof *info);
    info->height = m->arch.H; info->width = m->arch.W; info->n_conv = m->arch.n_conv; info->n_enc = m->arch.n_enc;
    info->feature_dim = (int32_t)m->arch.feat();
    for (int l = 0; l < m->arch.n_conv; ++l) info->channels[l] = m->arch.ch[l];
    info->reference_arch = m->arch.ref ? 1 : 0;
    info->precision = m->precision;
    info->debug_flags = m->debug_flags;
    info->n_components = m->C;
    info->n_sv_conservative = m->svm[0].nsv;
    info->n_sv_moderate = m->svm[1].nsv;
    info->shared_encoder = m->shared_encoder ? 1 : 0;
    info->has_detector = m->has_det ? 1 : 0;
    info->device_id = m->device;
    info->chunk_cells = eff_chunk(m, CS_MEM_DEVICE);
    return CS_OK;
}

int cs_model_set_chunk(cs_model* m, int64_t chunk_cells)
{
    if (!m || chunk_cells < 0 || chunk_cells > (1 << 20)) return fail(CS_ERR_INVALID, "chunk_cells must be in [0, 2^20] (0 = automatic)");
    m->chunk = chunk_cells;
    return CS_OK;
}

// ---- detector tail on a chunk ------------------------------------------------------
static int run_tail(cs_model* m, const float* feat, int64_t nc, float* mse, float* mae, double* sc, double* sm,
                    int8_t* pc, int8_t* pm, bool with_err)
{
    if (m->split16)
        LAUNCH(K_SCALER_PCA, nc,
               launch_scaler_pca_x3(feat, m->center.as<float>(), m->scale.as<double>(), m->comps_x3.as<uint16_t>(),
                                    m->mean_proj.as<float>(), m->F, m->fpad, m->C, m->cpad, m->pca.as<float>(), nc, m->stream, m->det_ws.p));
    else
        LAUNCH(K_SCALER_PCA, nc,
               launch_scaler_pca(feat, m->center.as<float>(), m->scale.as<double>(), m->comps.as<float>(),
                                 m->mean_proj.as<float>(), m->F, m->fpad, m->C, m->cpad, m->pca.as<float>(), nc, m->stream));
    if (m->det_ws.p && nc <= DET_SPLIT_MAX_CELLS) {       // a small call: both detectors' ranges in one launch
        const double* svT[2] = {m->svm[0].svT.as<double>(), m->svm[1].svT.as<double>()};
        const double* svn[2] = {m->svm[0].svn.as<double>(), m->svm[1].svn.as<double>()};
        const double* coef[2] = {m->svm[0].coef.as<double>(), m->svm[1].coef.as<double>()};
        const int nsv[2] = {m->svm[0].nsv_pad, m->svm[1].nsv_pad};
        const double gam[2] = {m->svm[0].gamma, m->svm[1].gamma}, rho[2] = {m->svm[0].rho, m->svm[1].rho};
        LAUNCH(K_SVM, nc,
               launch_ocsvm_pair_split(m->pca.as<float>(), m->C, svT, svn, coef, nsv, gam, rho, m->dec[0].as<double>(), m->dec[1].as<double>(), nc,
                                       m->stream, m->det_ws.p));
    } else
    for (int d = 0; d < 2; ++d)
        LAUNCH(K_SVM, nc,
               launch_ocsvm(m->pca.as<float>(), m->C, m->svm[d].svT.as<double>(), m->svm[d].svn.as<double>(), m->svm[d].coef.as<double>(),
                            m->svm[d].nsv_pad, m->svm[d].gamma, m->svm[d].rho, m->dec[d].as<double>(), nc, m->stream, m->det_ws.p));
    LAUNCH(K_FINALIZE, nc,
           launch_finalize(with_err ? m->errpart.as<float>() : nullptr, m->errparts, (int)m->arch.npix, m->dec[0].as<double>(),
                           m->dec[1].as<double>(), mse, mae, sc, sm, pc, pm, nc, m->stream));
    return CS_OK;
}

int cs_screen(cs_model* m, const float* crops, int64_t n, int crops_kind, float* mse, float* mae,
              double* cons_score, double* mod_score, int8_t* cons_pred, int8_t* mod_pred, int out_kind)
{
    int rc = begin_call(m);
    if (rc) return rc;
    DrainGuard guard(m);
    if (n < 0) return fail(CS_ERR_INVALID, "n is negative");
    if (n == 0) return guard.done(CS_OK);  // improved_detection.py:119-120
    if (!crops) return fail(CS_ERR_INVALID, "crops is NULL");
    if ((rc = check_kind(crops_kind, "crops_kind")) || (rc = check_kind(out_kind, "out_kind"))) return rc;
    if (!m->has_det) return fail(CS_ERR_NO_DETECTOR, "cs_screen needs detector parameters");
    const int64_t cc = eff_chunk(m, crops_kind);
    const int64_t ch = n < cc ? n : cc;
    if ((rc = ensure_workspace(m, ch, false))) return rc;
    // Host crops: two staging buffers; the H2D copy of chunk i+1 runs on its own stream while chunk i
    // computes (PCIe at ~50 GB/s carries 3 M cells/s, so the copy hides behind the kernels).  Host results
    // are collected on the device for the whole call and copied back once (18 B per cell).
    const bool host_in = crops_kind == CS_MEM_HOST, host_out = out_kind == CS_MEM_HOST;
    const size_t in_bytes = m->arch.npix * sizeof(float);
    DevBuf* stage[2] = {&m->xin, &m->xin2};
    const bool single = n <= ch;
    if (host_in) {
        if ((rc = m->xin.ensure((size_t)ch * in_bytes)) || (n > ch && (rc = m->xin2.ensure((size_t)ch * in_bytes)))) return rc;
        // a call of one chunk has nothing to overlap: its crops go up on the compute stream itself (no second stream, no event)
        HIPCHK(hipMemcpyAsync(m->xin.p, crops, (size_t)ch * in_bytes, hipMemcpyHostToDevice, single ? m->stream : m->copy_stream));
        if (!single) HIPCHK(hipEventRecord(m->ev_in[0], m->copy_stream));
    }
    // host results: the six arrays back to back in ONE device buffer ([n] double x 2, [n] float x 2, [n] int8 x 2).  A small call
    // (the reference screens one sample of 1e2 .. 1e4 cells per call) fetches them with one copy into pinned memory and hands them
    // out with memcpy -- six copies into pageable memory cost 6 x ~10 us of a 0.3 ms call; a large call copies each array directly.
    const size_t o_off[6] = {0, (size_t)n * 8, (size_t)n * 16, (size_t)n * 20, (size_t)n * 24, (size_t)n * 25};   // sc0 sc1 mse mae pr0 pr1
    const size_t o_total = (size_t)n * 26;
    const bool pack_out = host_out && n <= kPackedResultCells;
    if (host_out) {
        if ((rc = m->o_pack.ensure(o_total))) return rc;
        if (pack_out && m->h_pack_bytes < o_total) {
            if (m->h_pack) { (void)hipHostFree(m->h_pack); m->h_pack = nullptr; m->h_pack_bytes = 0; }
            const size_t cap = (size_t)kPackedResultCells * 26;
            HIPCHK(hipHostMalloc(&m->h_pack, cap, hipHostMallocDefault));
            m->h_pack_bytes = cap;
        }
    }
    char* const o_base = (char*)m->o_pack.p;
    auto dst = [&](auto* user, int which, int64_t off) -> decltype(user) {
        if (!user) return nullptr;
        return host_out ? (decltype(user))(o_base + o_off[which]) + off : user + off;
    };
    // an error inside the pipeline must not return while a copy from the caller's host buffer is still in flight
    auto drain = [&](int code) {
        (void)hipStreamSynchronize(m->copy_stream);
        (void)hipStreamSynchronize(m->stream);
        return code;
    };
    int64_t i = 0;
    for (int64_t off = 0; off < n; off += ch, ++i) {
        const int64_t nc = (n - off) < ch ? (n - off) : ch;
        const int b = (int)(i & 1);
        const float* x = crops + (size_t)off * m->arch.npix;
        if (host_in) {
            if (!single) HIPCHK(hipStreamWaitEvent(m->stream, m->ev_in[b], 0));
            x = stage[b]->as<float>();
        }
        if ((rc = run_convs(m, m->ae, x, nc, 0, m->arch.n_conv - 1, nullptr))) return drain(rc);
        const float* feat = m->act[m->arch.n_enc - 1].as<float>();
        if (!m->shared_encoder) {
            // encoder.keras differs from the autoencoder's encoder half: second encoder pass
            // (improved_detection.py:130), after the decoder has consumed the first pass's features.
            if ((rc = run_convs(m, m->enc, x, nc, 0, m->arch.n_enc - 1, nullptr))) return drain(rc);
            feat = m->act[m->arch.n_enc - 1].as<float>();
        }
        if ((rc = run_tail(m, feat, nc, dst(mse, 2, off), dst(mae, 3, off), dst(cons_score, 0, off),
                           dst(mod_score, 1, off), dst(cons_pred, 4, off), dst(mod_pred, 5, off), true)))
            return drain(rc);
        if (host_in && !single) {
            HIPCHK(hipEventRecord(m->ev_used[b], m->stream));
            const int64_t noff = off + ch;
            if (noff < n) {   // next chunk into the other buffer, once the chunk before this one has released it
                const int64_t nn = (n - noff) < ch ? (n - noff) : ch;
                if (i >= 1) HIPCHK(hipStreamWaitEvent(m->copy_stream, m->ev_used[b ^ 1], 0));
                HIPCHK(hipMemcpyAsync(stage[b ^ 1]->p, crops + (size_t)noff * m->arch.npix, (size_t)nn * in_bytes, hipMemcpyHostToDevice,
                                      m->copy_stream));
                HIPCHK(hipEventRecord(m->ev_in[b ^ 1], m->copy_stream));
            }
        }
    }
    if (pack_out) {
        HIPCHK(hipMemcpyAsync(m->h_pack, o_base, o_total, hipMemcpyDeviceToHost, m->stream));
        if (host_in) HIPCHK(hipStreamSynchronize(m->copy_stream));
        const int erc = end_call(m);                     // synchronises the handle's stream: the packed results are on the host
        if (erc) return guard.done(erc);
        const char* h = (const char*)m->h_pack;
        if (cons_score) memcpy(cons_score, h + o_off[0], (size_t)n * 8);
        if (mod_score) memcpy(mod_score, h + o_off[1], (size_t)n * 8);
        if (mse) memcpy(mse, h + o_off[2], (size_t)n * 4);
        if (mae) memcpy(mae, h + o_off[3], (size_t)n * 4);
        if (cons_pred) memcpy(cons_pred, h + o_off[4], (size_t)n);
        if (mod_pred) memcpy(mod_pred, h + o_off[5], (size_t)n);
        return guard.done(CS_OK);
    }
    if (host_out) {
        if (mse) HIPCHK(hipMemcpyAsync(mse, o_base + o_off[2], (size_t)n * sizeof(float), hipMemcpyDeviceToHost, m->stream));
        if (mae) HIPCHK(hipMemcpyAsync(mae, o_base + o_off[3], (size_t)n * sizeof(float), hipMemcpyDeviceToHost, m->stream));
        if (cons_score) HIPCHK(hipMemcpyAsync(cons_score, o_base + o_off[0], (size_t)n * sizeof(double), hipMemcpyDeviceToHost, m->stream));
        if (mod_score) HIPCHK(hipMemcpyAsync(mod_score, o_base + o_off[1], (size_t)n * sizeof(double), hipMemcpyDeviceToHost, m->stream));
        if (cons_pred) HIPCHK(hipMemcpyAsync(cons_pred, o_base + o_off[4], (size_t)n, hipMemcpyDeviceToHost, m->stream));
        if (mod_pred) HIPCHK(hipMemcpyAsync(mod_pred, o_base + o_off[5], (size_t)n, hipMemcpyDeviceToHost, m->stream));
    }
    if (host_in) HIPCHK(hipStreamSynchronize(m->copy_stream));
    return guard.done(end_call(m));
}

int cs_reconstruct(cs_model* m, const float* crops, int64_t n, int crops_kind, float* recon, float* mse,
                   float* mae, int out_kind)
{
    int rc = begin_call(m);
    if (rc) return rc;
    DrainGuard guard(m);
    if (n < 0) return fail(CS_ERR_INVALID, "n is negative");
    if (n == 0) return guard.done(CS_OK);
    if (!crops) return fail(CS_ERR_INVALID, "crops is NULL");
    if ((rc = check_kind(crops_kind, "crops_kind")) || (rc = check_kind(out_kind, "out_kind"))) return rc;
    const int64_t cc = eff_chunk(m, crops_kind);
    const int64_t ch = n < cc ? n : cc;
    const bool host_recon = recon && out_kind == CS_MEM_HOST;
    if ((rc = ensure_workspace(m, ch, host_recon))) return rc;
    for (int64_t off = 0; off < n; off += ch) {
        const int64_t nc = (n - off) < ch ? (n - off) : ch;
        const float* x;
        if ((rc = stage_in(m, crops, crops_kind, off, nc, m->arch.npix, m->xin, &x))) return rc;
        float* d_rec = !recon ? nullptr : (out_kind == CS_MEM_DEVICE ? recon + (size_t)off * m->arch.npix : m->recon.as<float>());
        if ((rc = run_convs(m, m->ae, x, nc, 0, m->arch.n_conv - 1, d_rec))) return rc;
        float* d_mse = out_ptr(mse, out_kind, off, m->o_mse);
        float* d_mae = out_ptr(mae, out_kind, off, m->o_mae);
        LAUNCH(K_FINALIZE, nc,
               launch_finalize(m->errpart.as<float>(), m->errparts, (int)m->arch.npix, nullptr, nullptr, d_mse, d_mae, nullptr, nullptr,
                               nullptr, nullptr, nc, m->stream));
        if ((rc = stage_out(m, mse, out_kind, off, nc, d_mse))) return rc;
        if ((rc = stage_out(m, mae, out_kind, off, nc, d_mae))) return rc;
        if (host_recon)
            HIPCHK(hipMemcpyAsync(recon + (size_t)off * m->arch.npix, d_rec, (size_t)nc * m->arch.npix * sizeof(float),
                                  hipMemcpyDeviceToHost, m->stream));
        if (crops_kind == CS_MEM_HOST || out_kind == CS_MEM_HOST) HIPCHK(hipStreamSynchronize(m->stream));
    }
    return guard.done(end_call(m));
}

int cs_encode(cs_model* m, const float* crops, int64_t n, int crops_kind, int which, float* features, int out_kind)
{
    int rc = begin_call(m);
    if (rc) return rc;
    DrainGuard guard(m);
    if (n < 0) return fail(CS_ERR_INVALID, "n is negative");
    if (n == 0) return guard.done(CS_OK);
    if (!crops || !features) return fail(CS_ERR_INVALID, "crops/features is NULL");
    if (which != 0 && which != 1) return fail(CS_ERR_INVALID, "which must be 0 (autoencoder) or 1 (encoder.keras)");
    if ((rc = check_kind(crops_kind, "crops_kind")) || (rc = check_kind(out_kind, "out_kind"))) return rc;
    const int64_t cc = eff_chunk(m, crops_kind);
    const int64_t ch = n < cc ? n : cc;
    if ((rc = ensure_workspace(m, ch, false))) return rc;
    const ConvSet& set = (which == 1 && !m->shared_encoder) ? m->enc : m->ae;
    const size_t fl = m->arch.feat();
    for (int64_t off = 0; off < n; off += ch) {
        const int64_t nc = (n - off) < ch ? (n - off) : ch;
        const float* x;
        if ((rc = stage_in(m, crops, crops_kind, off, nc, m->arch.npix, m->xin, &x))) return rc;
        if ((rc = run_convs(m, set, x, nc, 0, m->arch.n_enc - 1, nullptr))) return rc;
        HIPCHK(hipMemcpyAsync(features + (size_t)off * fl, m->act[m->arch.n_enc - 1].p, (size_t)nc * fl * sizeof(float),
                              out_kind == CS_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
    }
    return guard.done(end_call(m));
}

int cs_layer_output(cs_model* m, const float* crops, int64_t n, int crops_kind, int layer, float* out, int out_kind)
{
    int rc = begin_call(m);
    if (rc) return rc;
    DrainGuard guard(m);
    if (n < 0) return fail(CS_ERR_INVALID, "n is negative");
    if (n == 0) return guard.done(CS_OK);
    if (!crops || !out) return fail(CS_ERR_INVALID, "crops/out is NULL");
    const int last = m->arch.n_conv - 1;
    if (layer < 0 || layer > last) return fail(CS_ERR_INVALID, "layer must be in [0,%d]", last);
    if ((rc = check_kind(crops_kind, "crops_kind")) || (rc = check_kind(out_kind, "out_kind"))) return rc;
    const int64_t cc = eff_chunk(m, crops_kind);
    const int64_t ch = n < cc ? n : cc;
    if ((rc = ensure_workspace(m, ch, layer == last))) return rc;
    const size_t fl = m->arch.floats[layer];
    for (int64_t off = 0; off < n; off += ch) {
        const int64_t nc = (n - off) < ch ? (n - off) : ch;
        const float* x;
        if ((rc = stage_in(m, crops, crops_kind, off, nc, m->arch.npix, m->xin, &x))) return rc;
        if ((rc = run_convs(m, m->ae, x, nc, 0, layer, layer == last ? m->recon.as<float>() : nullptr))) return rc;
        const void* src = layer == last ? m->recon.p : m->act[layer].p;
        HIPCHK(hipMemcpyAsync(out + (size_t)off * fl, src, (size_t)nc * fl * sizeof(float),
                              out_kind == CS_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
    }
    return guard.done(end_call(m));
}

int cs_scaler_pca(cs_model* m, const float* features, int64_t n, int in_kind, float* pca_out, int out_kind)
{
    int rc = begin_call(m);
    if (rc) return rc;
    DrainGuard guard(m);
    if (n < 0) return fail(CS_ERR_INVALID, "n is negative");
    if (n == 0) return guard.done(CS_OK);
    if (!features || !pca_out) return fail(CS_ERR_INVALID, "features/pca_out is NULL");
    if ((rc = check_kind(in_kind, "in_kind")) || (rc = check_kind(out_kind, "out_kind"))) return rc;
    if (!m->has_det) return fail(CS_ERR_NO_DETECTOR, "cs_scaler_pca needs detector parameters");
    const int64_t cc = eff_chunk(m, in_kind);
    const int64_t ch = n < cc ? n : cc;
    if ((rc = ensure_workspace(m, ch, false))) return rc;
    for (int64_t off = 0; off < n; off += ch) {
        const int64_t nc = (n - off) < ch ? (n - off) : ch;
        const float* f;
        if ((rc = stage_in(m, features, in_kind, off, nc, (size_t)m->F, m->featE, &f))) return rc;
        if (m->split16)
            LAUNCH(K_SCALER_PCA, nc,
                   launch_scaler_pca_x3(f, m->center.as<float>(), m->scale.as<double>(), m->comps_x3.as<uint16_t>(),
                                        m->mean_proj.as<float>(), m->F, m->fpad, m->C, m->cpad, m->pca.as<float>(), nc, m->stream, m->det_ws.p));
        else
            LAUNCH(K_SCALER_PCA, nc,
                   launch_scaler_pca(f, m->center.as<float>(), m->scale.as<double>(), m->comps.as<float>(),
                                     m->mean_proj.as<float>(), m->F, m->fpad, m->C, m->cpad, m->pca.as<float>(), nc, m->stream));
        HIPCHK(hipMemcpyAsync(pca_out + (size_t)off * m->C, m->pca.p, (size_t)nc * m->C * sizeof(float),
                              out_kind == CS_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
    }
    return guard.done(end_call(m));
}

int cs_svm_decision(cs_model* m, const float* pca, int64_t n, int in_kind, double* cons_dec, double* mod_dec, int out_kind)
{
    int rc = begin_call(m);
    if (rc) return rc;
    DrainGuard guard(m);
    if (n < 0) return fail(CS_ERR_INVALID, "n is negative");
    if (n == 0) return guard.done(CS_OK);
    if (!pca) return fail(CS_ERR_INVALID, "pca is NULL");
    if ((rc = check_kind(in_kind, "in_kind")) || (rc = check_kind(out_kind, "out_kind"))) return rc;
    if (!m->has_det) return fail(CS_ERR_NO_DETECTOR, "cs_svm_decision needs detector parameters");
    const int64_t cc = eff_chunk(m, in_kind);
    const int64_t ch = n < cc ? n : cc;
    if ((rc = ensure_workspace(m, ch, false))) return rc;
    double* outs[2] = {cons_dec, mod_dec};
    for (int64_t off = 0; off < n; off += ch) {
        const int64_t nc = (n - off) < ch ? (n - off) : ch;
        const float* p;
        if ((rc = stage_in(m, pca, in_kind, off, nc, (size_t)m->C, m->pca, &p))) return rc;
        for (int d = 0; d < 2; ++d) {
            if (!outs[d]) continue;
            LAUNCH(K_SVM, nc,
                   launch_ocsvm(p, m->C, m->svm[d].svT.as<double>(), m->svm[d].svn.as<double>(), m->svm[d].coef.as<double>(),
                                m->svm[d].nsv_pad, m->svm[d].gamma, m->svm[d].rho, m->dec[d].as<double>(), nc, m->stream, m->det_ws.p));
            HIPCHK(hipMemcpyAsync(outs[d] + off, m->dec[d].p, (size_t)nc * sizeof(double),
                                  out_kind == CS_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, m->stream));
        }
        HIPCHK(hipStreamSynchronize(m->stream));
    }
    return guard.done(end_call(m));
}

int cs_synth_crops(cs_model* m, uint64_t seed, int64_t first_cell, int64_t n, int32_t npix, float* out_device)
{
    int rc = begin_call(m);
    if (rc) return rc;
    if (n < 0 || npix <= 0) return fail(CS_ERR_INVALID, "n/npix invalid");
    if (n == 0) return CS_OK;
    if (!out_device) return fail(CS_ERR_INVALID, "out_device is NULL");
    LAUNCH(K_SYNTH, n, launch_synth(seed, first_cell, n, npix, out_device, m->stream));
    return end_call(m);
}

// ---- measurement --------------------------------------------------------------------
int cs_profile_enable(cs_model* m, int on)
{
    if (!m) return fail(CS_ERR_INVALID, "model handle is NULL");
    m->prof = on != 0;
    return CS_OK;
}

int cs_profile_reset(cs_model* m)
{
    if (!m) return fail(CS_ERR_INVALID, "model handle is NULL");
    for (int k = 0; k < K_COUNT; ++k) { m->prof_ms[k] = 0; m->prof_launches[k] = 0; m->prof_cells[k] = 0; }
    return CS_OK;
}

int cs_model_wait_stream(cs_model* m, void* hip_stream)
{
    if (!m) return fail(CS_ERR_INVALID, "model handle is NULL");
    HIPCHK(hipSetDevice(m->device));
    return wait_on_stream(m->stream, hip_stream);
}

int cs_profile_kernel_count(void) { return K_COUNT; }

const char* cs_profile_kernel_name(int k) { return (k >= 0 && k < K_COUNT) ? kKernelNames[k] : ""; }

int cs_profile_get(cs_model* m, int k, double* total_ms, int64_t* launches, int64_t* cells, double* flops)
{
    if (!m || k < 0 || k >= K_COUNT) return fail(CS_ERR_INVALID, "bad kernel id");
    if (total_ms) *total_ms = m->prof_ms[k];
    if (launches) *launches = m->prof_launches[k];
    if (cells) *cells = m->prof_cells[k];
    if (flops) {
        double per_cell = 0.0;
        if (k <= K_CONV7_ERR) per_cell = 2.0 * kLayerMacs[k];
        else if (k == K_CONV67_FUSED) per_cell = 2.0 * (kLayerMacs[5] + kLayerMacs[6]);
        else if (k == K_CONV12_FUSED) per_cell = 2.0 * (kLayerMacs[0] + kLayerMacs[1]);
        else if (k == K_SCALER_PCA) per_cell = 2.0 * (double)m->F * m->C;
        else if (k == K_SVM) per_cell = 0.0;  // fp64, reported separately
        *flops = per_cell * (double)m->prof_cells[k];
    }
    return CS_OK;
}

// Matrix-pipe instructions (v_mfma_f32_16x16x4_f32 = 1,024 multiply-adds each) one cell costs in kernel family k with
// the kernels this handle actually runs -- the EXECUTED work behind a roofline fraction (Winograd and the folded
// upsample execute fewer multiply-adds than the layer's algorithmic count; conv1 pads K = 9 to 12).  The same
// number is what SQ_INSTS_MFMA counts per cell (profiles/*_sq_counters.json).  0 for kernels without MFMAs
// (and for the fp64 SVM, which is priced separately).
int cs_profile_mfma_per_cell(cs_model* m, int k, double* mfma)
{
    if (!m || !mfma || k < 0 || k >= K_COUNT) return fail(CS_ERR_INVALID, "bad kernel id");
    double v = 0.0;
    if (m->arch.ref) {
        const bool h2 = m->split16;
        switch (k) {
            case K_CONV1: v = 1536; break;                                  // 256 tiles x 2 slices x 3 K steps
            case K_CONV2: v = 8192; break;                                  // F(2x2,3x3): 16 points x 16 groups x 8 x 4
            case K_CONV3: v = h2 ? 0 : 2048; break;
            case K_CONV4: v = h2 ? 0 : 576; break;                          // the split forms run on v_mfma_f32_16x16x32_f16: cs_profile_bf16_mfma_per_cell
            case K_CONV5: v = h2 ? 0 : 1152; break;                         // F(2x2,2x2) phases: 1/4 of the direct conv's multiply-adds
            case K_CONV6: v = 4608; break;
            case K_CONV67_FUSED: v = h2 ? 512 : 4608 + 512; break;          // conv6 phases (unless on the 16-bit pipe) + conv7's 32 -> 16 contraction
            case K_CONV12_FUSED: v = h2 ? 0 : 4608 + 1536 + 48; break;      // conv2 F(4x4,3x3): 36 points x 4 groups x 8 x 4; conv1 direct
                                                                            // + the discarded fourth row of a cell's last 4-row batch
            case K_SCALER_PCA: v = h2 ? 0.0 : (double)m->fpad * m->cpad / 1024.0; break;
            default: v = 0.0;
        }
    } else if (k <= K_CONV6 || k == K_CONV7_ERR) {
        const int l = k == K_CONV7_ERR ? m->arch.n_conv - 1 : k - K_CONV1;
        if (l < m->arch.n_conv && !(m->split16 && m->gae.x3[l])) {      // split-bf16 layers: cs_profile_bf16_mfma_per_cell
            const int cout_pad = (m->arch.ch[l] + 15) / 16 * 16;
            v = (double)m->arch.gh[l] * m->arch.gw[l] / 16.0 * (cout_pad / 16) * (9.0 * m->arch.cin(l) / 4.0);
        }
    }
    *mfma = v;
    return CS_OK;
}

// The same for the kernels that take the fp32 contraction on the 16-bit matrix pipe (CS_PRECISION_SPLIT16): v_mfma_f32_16x16x32_{f16,bf16}
// instructions (16,384 FLOP each; three (fp16 split) or six (bf16 split) per 16 pixels x 16 filters x 32 channels) per cell.
int cs_profile_bf16_mfma_per_cell(cs_model* m, int k, double* mfma)
{
    if (!m || !mfma || k < 0 || k >= K_COUNT) return fail(CS_ERR_INVALID, "bad kernel id");
    double v = 0.0;
    if (!m->split16) { *mfma = 0.0; return CS_OK; }
    if (k == K_SCALER_PCA && m->has_det) {
        *mfma = (double)(m->cpad / 16) * (m->fpad / 32) * 6.0 / 16.0;      // per 16-cell tile: component tiles x 32-feature blocks x 6 (bf16 split)
        return CS_OK;
    }
    if (m->arch.ref) {                                                     // fp16 split: 3 products per (16 px, 16 filters, 32 channels)
        if (k == K_CONV4) v = 4 * 2 * 9 * 3;                               // 4 tiles x 2 slices x 9 taps
        if (k == K_CONV3) v = 16.0 * 4 * 2 * 2 * 3;                        // 16 points x 4 tile groups x 2 slices x 2 blocks
        if (k == K_CONV12_FUSED) v = 66 * 8 * 2 + 36 * 4 * 4 * 3;          // conv1: 66 conv rows (one pooled row discarded) x 8 (x-tile, slice) x 2 MFMAs;
                                                                           // conv2: 36 points x 4 tile groups x 4 slices x 3
        if (k == K_CONV5) v = 4 * 4 * 4 * 4 * 3 + (m->fuse45 ? 4 * 2 * 9 * 3 : 0);   // 4 phases x 4 tiles x 4 slices x 4 taps (+ conv4 inside the fused kernel)
        if (k == K_CONV67_FUSED) v = 4.0 * 16 * 2 * 4 * 2 * 3;             // 4 phases x 16 tiles x 2 slices x 4 taps x 2 blocks
    } else if (k <= K_CONV6) {
        const int l = k - K_CONV1;
        if (l < m->arch.n_conv && m->gae.x3[l]) {
            const int cout_pad = (m->arch.ch[l] + 15) / 16 * 16;
            const double taps = l > m->arch.n_enc ? 4.0 : 9.0;             // folded upsample: 16 (phase, tap) pairs over a quarter of the grid
            const bool last = l == m->arch.n_conv - 1;
            v = (double)m->arch.gh[l] * m->arch.gw[l] / 16.0 * (cout_pad / 16) * taps * (m->arch.cin(l) / 32.0) * ((m->gae.h2_inv[l] != 0.0f && !last) ? 3.0 : 6.0);
        }
    }
    *mfma = v;
    return CS_OK;
}

}  // extern "C"
