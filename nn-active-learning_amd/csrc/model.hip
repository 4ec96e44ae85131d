// Model plan + orchestration behind the C ABI: shapes, activation / cotangent workspaces
// (concat skips = channel slices of one buffer), weight re-layout into the GEMM engine's
// packed form, and the forward / Fisher launch sequences.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <string>
#include <memory>

#include "alq_internal.h"

namespace alq {

static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int gather_normalize_impl(alq_ctx *, const void *const *, int, int, const int64_t[3], const int64_t[3],
                          const int64_t *, int64_t, const int32_t[3], const double *, int, int, void *);
int score_entropy_impl(alq_ctx *, const float *, int64_t, double *, float *);
int synth_impl(alq_ctx *, uint64_t, int64_t, int64_t, int64_t, float *);
int gather_rows_impl(alq_ctx *, const float *, const int64_t *, int64_t, int64_t, float *);
int debug_view_copy(alq_ctx *, const View &, int, float *);
int debug_f64_copy(alq_ctx *, const double *, long long, float *);
size_t topk_work_bytes_impl(int64_t n);
int topk_impl(alq_ctx *, const double *, int64_t, int64_t, int64_t *, void *);

static void same_pads(int in, int k, int s, int *out, int *lo) {
    *out = (in + s - 1) / s;
    int total = (*out - 1) * s + k - in;
    if (total < 0) total = 0;
    *lo = total / 2;
}

struct Layer {
    alq_layer_t spec;
    int pidx = -1;             // parameterised-layer index t, or -1
    View in, out;              // activation views (in = incl. concatenated skip channels)
    View din, dout;            // cotangent views, same geometry
    int lo[3] = {0, 0, 0};     // SAME pad-before (conv: of the fwd conv; convT: of the conv it transposes)
    bool dense_fc_small = false;
    int64_t F = 0;             // fc: input features
    // weights
    int64_t w_elems = 0, b_elems = 0;
    float *d_bias = nullptr;
    float *d_Wp = nullptr;     // skinny fc: [nout][F] in activation-memory order
    std::vector<Gemm> fwd;        // 1 contraction (conv / fc) or one per output parity class (convT)
    // conv whose output is too wide for the matrix-core engines (NET-B's 96-channel conv: igemm2 / igemm3 hold <= 48, the two-slot
    // engine <= 32 output channels) and would run on the fp32 engine: the same contraction as launches of fwd_co_w output channels
    // each, writing channel slices of the output (round 6)
    std::vector<Gemm> fwd_co;
    int fwd_co_w = 0;
    Igemm4Plan fwd_all;           // convT: every output class from one staged block (igemm4.hip), when eligible
    Gemm bwd;
    bool has_bwd = false;
    bool weights_set = false;
    // workspaces
    uint8_t *argmax = nullptr;
    float *asum = nullptr, *dsum = nullptr;
    float *osum = nullptr;     // channel sums of this layer's OUTPUT (spatial layers): next layers' asum
    bool delta_ready = false;  // backward: the cotangent of our output is already masked and dsum is filled
    bool signs_ready = false;  // Fisher pass: the forward launch wrote the sign field of our output (View::sg)
    bool dsum_partial = false; // backward, first layer: the skip destination has written its share of dsum
    float *fc_partials = nullptr;
    // fc head of a Fisher pass on top of a ReLU conv: its input cotangent is [input > 0] * fc_wv for every patch; the
    // forward pass leaves the signs (one byte per 4 elements), the backward pass of the conv below contracts them
    // directly (igemm4 BITSRC)
    unsigned *fc_maskbits = nullptr;
    float *fc_wv = nullptr;
    unsigned *fc_wv16 = nullptr;       // fc_wv pre-split into fp16 pairs for the fp16x2 contraction of the conv below (BITSRC)
    float fc_wv_amax = 0.f;            // max |W0 - W1| of a two-output head (host side, set with the weights)
    const unsigned *dout_amax = nullptr;   // per-patch max |cotangent of this layer's output| of the running backward pass, or null
    unsigned *amax_fwd = nullptr;          // [max_batch] per-patch max |output| of a forward pass that asked for it
    float dout_vec_amax = 0.f;
    // Static bounds for the fp16x2 contraction of backward launches (no data pass needed): bwd_l1 = max over the input
    // channels of sum_{taps, output channels} |W| (set with the weights), i.e. |cotangent of the input| <= bwd_l1 * max
    // |cotangent of the output|; dout_bound = the bound on this layer's output cotangent in the running Fisher pass
    // (unit cotangent at the logits, chained down by run_backward_main).
    double bwd_l1 = 0;
    float dout_bound = 0.f;
    // flip-safe fused head (the conv under a two-class head): plain fp32 copy of the weights in TF layout [tap][ci][co] for the
    // exact re-evaluation, and max over co of sum_{tap, ci} |W|
    float *d_W32 = nullptr;
    float fwd_l1 = 0.f;
    float out_l1 = 0.f, out_bmax = 0.f;    // |out| <= out_l1 * max |in| + out_bmax (conv: = fwd_l1; conv_transpose: all taps), set with the weights
    unsigned *bound_fwd = nullptr;         // per-patch bound on |out| derived from the first layer's measured maximum (k_fwd_bounds)
    float *fc_part2 = nullptr;         // partial logits per (tile, wave) when the conv below computes them in its epilogue
    int fc_slices2 = 0;
    // plane-sweep engine (c3d.hip) for the conv under the fused two-class head (and its backward): plans on the conv layer,
    // per-(patch, wave) partials of the logit difference / of the head's input sum on the head layer
    C3dPlan c3f, c3b;
    D3dPlan d3f;                           // forward on the row-sweep engine of d3d.hip (NET-C's dec1)
    F3dPlan f3f;                           // forward fused with the max-pool behind it (f3d.hip; NET-C's enc2)
    E3dPlan e3b;                           // backward fused with the pool backward steps on either side (e3d.hip; NET-C's enc2)
    T3dPlan t3f, t3b;                      // row-sweep engine for the stride-2 conv_transpose (t3d.hip), forward / backward-data
    float *c3_part = nullptr, *c3_asum = nullptr;
    unsigned short *fc_wv16c = nullptr;    // fc_wv as fp16 pairs at their true scale, [voxel][h8 | l8] (c3d_presplit_vec), for c3b
    const unsigned short *dout_vec16c = nullptr;   // set on the conv below for one backward pass, like dout_vec16
    const unsigned *dout_bits = nullptr;   // set on the conv below for the duration of one backward pass
    const float *dout_vec = nullptr;
    const unsigned *dout_vec16 = nullptr;
    int fc_slices = 0;
    bool out_is_skip_src = false;
    // convT class tap lists (indices into the k^3 tap enumeration)
    std::vector<std::vector<int>> class_taps;
};

}  // namespace alq

using namespace alq;

struct alq_model {
    alq_ctx *ctx = nullptr;
    int max_batch = 0;
    bool last_call_fisher = false;   // what alq_model_debug_copy may read
    bool last_head_fused = false;    // the last forward pass did not store the last conv's output
    // per-patch max |x| (float bits) of the two producers of the fused-head conv's input, for its fp16x2 contraction
    unsigned *amax_a = nullptr, *amax_b = nullptr, *amax_tiles = nullptr;
    unsigned *flip_cnt = nullptr, *flip_list = nullptr;     // candidates of the flip-safe fused head (igemm4 FCF + F16)
    int flip_cap = 0;
    unsigned *bound_all = nullptr;          // [layer][max_batch] derived per-patch output bounds (float bits), k_fwd_bounds
    unsigned *in_amax = nullptr;            // [max_batch] measured max |x| of every patch of the network input (igemm3's forward fp16 pairs)
    bool v3_fwd_f16 = false;                // some forward conv launch stays on igemm3 and has the fp16-pair twin packed (set at build)
    unsigned *flip_overflow = nullptr;      // marked groups beyond the scan's lists since the model was created: drained by the sweep path of flip_fix_kernel (kernels.hip), none dropped
    int no_flipfix = 0;                                      // ALQ_NO_FLIPFIX at creation (A/B: the head's sign bits as the fp16x2 contraction leaves them)
    size_t amax_tiles_len = 0;
    int in_dims[4] = {1, 1, 1, 1};
    int nclass = 0;
    int L = 0;
    std::vector<Layer> layers;
    std::vector<void *> allocs;
    float *logits = nullptr, *dlogits = nullptr, *post = nullptr;
    double *S = nullptr, *sizes = nullptr, *Apart = nullptr;
    double *Spart = nullptr;       // [L][max_batch][nslab_max] box-dot slab partials
    int *nslab = nullptr;          // [L] slabs actually written per layer
    int nslab_max = 1;
    float *wg_partial = nullptr;   // slab partials of the weight-gradient kernels (grown on demand)
    size_t wg_partial_len = 0;
    float *x_stage = nullptr;      // [max_batch, elems per patch]: rows gathered by the *_rows entry points
    int64_t epp = 0;               // elements per patch
    // Engine-selection knobs, read from the environment ONCE, when this model is created; every call applies the
    // model's own snapshot (alq_debug_set overrides a key for all models until it is set back to -1 / re-set).
    int knobs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int no_f16x2 = 0;
    int no_xcd_order = 0;
    int requested_batch = 0;      // what alq_model_create was asked for (max_batch may be lower: 32-bit tensor offsets)
    int no_bound16 = 0;           // ALQ_NO_BOUND16 at creation: backward launches take their fp16x2 scale from measured per-patch maxima only
    int no_fixed = 0;             // ALQ_NO_FIXED at creation: runtime-constant igemm4 instantiations only
    int no_presplit = 0;           // ALQ_NO_PRESPLIT (A/B): split the fc head's weight-difference vector in the staging part again
    int no_signs = 0;              // ALQ_NO_SIGNS (A/B, bit-identity test): backward launches read ReLU masks from the fp32 activations
    int no_signs0 = 0;             // ALQ_NO_SIGNS0 (A/B): no sign field from the first conv + pool kernel only
    int f16_fwd_mask = -1;         // ALQ_F16_FWD_MASK (diagnostics): forward fp16x2 consumers by layer bit, -1 = default rule
    int f16_fwd_derived = 0;       // layers (bits) whose forward launch takes the fp16x2 split with DERIVED input bounds (see run_forward)
    int no_f16_derived = 0;        // 1 with ALQ_NO_F16_DERIVED=1: those launches stay on bf16x3 (A/B; default since round 5: they take the split)
    float *d_bound_L = nullptr, *d_bound_B = nullptr;      // per layer: out = in * L + B (k_fwd_bounds)
    int *d_bound_src = nullptr;
    int no_light_kernels = 0;      // ALQ_NO_LIGHT_KERNELS (A/B): forward-only passes keep dec1 / enc2 + pool2 on the two-slot engine as until round 5
    int no_c3d = 0;                // ALQ_NO_C3D (A/B): the head conv pair on the two-slot engine (igemm4) as in round 3
    int c3_bwd_rows = 8;           // ALQ_C3D_BWD_ROWS=4 (A/B): the plane-sweep backward kernel in its half-patch form
    bool last_c3 = false;          // the last forward pass ran the head conv on the plane-sweep engine
    bool last_c3_bwd = false;      // ... and the last backward pass its backward
    int no_e3d = 0;                // ALQ_NO_E3D (A/B): pool2 backward, enc2 backward and pool1 backward as three launches as in round 4
    int last_e3b = 0;              // the last backward pass ran them as one launch (e3d.hip)
    int no_d3d = 0;                // ALQ_NO_D3D (A/B): dec1's forward on the two-slot engine as in round 4
    int last_d3f = 0;              // the last forward pass ran it on the row-sweep engine (d3d.hip)
    int no_f3d = 0, last_f3f = 0;  // ALQ_NO_F3D (A/B): enc2's forward on the two-slot engine + the pool as its own launch; the last forward pass ran them fused (f3d.hip)
    int no_d3b = 0, last_d3b = 0;  // ALQ_NO_D3D_BWD (A/B): only the backward launch on the two-slot engine; the last backward pass ran it on d3d.hip
    int no_t3d = 0;                // ALQ_NO_T3D (A/B): conv_transpose launches on the two-slot engine (igemm4) as in round 4
    int last_t3f = 0, last_t3b = 0;   // conv_transpose launches of the last forward / backward pass that ran on the row-sweep engine
    bool last_f16_derived = false; // the last forward pass ran a launch on the fp16x2 split with derived input bounds

    template <typename T>
    int dalloc(T **p, size_t count) {
        void *q = nullptr;
        const size_t bytes = std::max<size_t>(count * sizeof(T), 256);
        if (hipMalloc(&q, bytes) != hipSuccess) {
            set_error("hipMalloc of %zu bytes failed", bytes);
            return ALQ_ENOMEM;
        }
        allocs.push_back(q);
        *p = reinterpret_cast<T *>(q);
        return ALQ_OK;
    }
};

// ------------------------------------------------------------------------------------------
int alq_ctx::prof_begin(int cls, hipEvent_t *e0, hipEvent_t *e1) {
    ProfSlot &s = prof[cls];
    auto get = [&](hipEvent_t *ev) -> int {
        if (!s.pool.empty()) {
            *ev = s.pool.back();
            s.pool.pop_back();
            return ALQ_OK;
        }
        return hipEventCreate(ev) == hipSuccess ? ALQ_OK : ALQ_EHIP;
    };
    if (get(e0) != ALQ_OK || get(e1) != ALQ_OK) return ALQ_EHIP;
    return hipEventRecord(*e0, stream) == hipSuccess ? ALQ_OK : ALQ_EHIP;
}

void alq_ctx::prof_end(int cls, hipEvent_t e0, hipEvent_t e1, double flops) {
    ProfSlot &s = prof[cls];
    (void)hipEventRecord(e1, stream);
    s.pending.emplace_back(e0, e1);
    s.launches += 1;
    s.flops += flops;
}

int alq_ctx::prof_collect() {
    ALQ_HIP(hipStreamSynchronize(stream));
    for (int c = 0; c < PROF_NUM; ++c) {
        ProfSlot &s = prof[c];
        for (auto &pr : s.pending) {
            float ms = 0.f;
            ALQ_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
            s.ms += ms;
            s.pool.push_back(pr.first);
            s.pool.push_back(pr.second);
        }
        s.pending.clear();
    }
    return ALQ_OK;
}

// ------------------------------------------------------------------------------------------
static int upload1(alq_model *m, IgemmPlan *p) {
    if (!p->d_W) ALQ_TRY(m->dalloc(&p->d_W, p->h_W.size()));
    ALQ_HIP(hipMemcpyAsync(p->d_W, p->h_W.data(), p->h_W.size() * sizeof(float), hipMemcpyHostToDevice,
                           m->ctx->stream));
    if (p->smallc) {
        if (!p->d_koff) ALQ_TRY(m->dalloc(&p->d_koff, p->h_koff.size()));
        ALQ_HIP(hipMemcpyAsync(p->d_koff, p->h_koff.data(), p->h_koff.size() * sizeof(int),
                               hipMemcpyHostToDevice, m->ctx->stream));
    }
    ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
    std::vector<float>().swap(p->h_W);
    return ALQ_OK;
}

static bool g_use_v2 = true;
static int g_knob_override[8] = {-1, -1, -1, -1, -1, -1, -1, -1};   // alq_debug_set: >= 0 overrides every model's snapshot

// The kernels' launch helpers read the process-wide g_dbg_knobs / g_no_f16x2; each entry point loads them from the
// model it was called on, so creating another model (or its environment) never changes a live one.
static void apply_knobs(const alq_model *m) {
    for (int k = 0; k < 8; ++k) g_dbg_knobs[k] = g_knob_override[k] >= 0 ? g_knob_override[k] : m->knobs[k];
    g_no_f16x2 = m->no_f16x2;
    g_no_xcd_order = m->no_xcd_order;
    g_no_fixed = m->no_fixed;
}

static int gemm_build(const ConvDesc &d, int max_batch, Gemm *g, const G4Geom *g4 = nullptr) {
    ALQ_TRY(igemm_build_plan(d, max_batch, &g->p1));
    if (g4 && !getenv("ALQ_DISABLE_V4")) {
        G4Geom gg = *g4;
        gg.flops_per_patch = g->p1.flops_per_patch;
        ALQ_TRY(igemm4_build_plan(gg, max_batch, &g->p4));
    }
    ALQ_TRY(igemm2_build_plan(g->p1, &g->p2));
    ALQ_TRY(direct_build_plan(g->p1, &g->pd));
    if (!g_use_v2) { g->p2.ok = false; g->pd.ok = false; }
    ALQ_TRY(igemm3_build_plan(g->p2, &g->p3));
    if (d.ID == 1 && d.IH == 1 && d.IW == 1 && d.tz.size() == 1 && d.sm == 1 && d.so == 1 && !getenv("ALQ_NO_FCGEMM"))
        ALQ_TRY(fcgemm_build_plan(d.Ci, d.Co, &g->pfc));       // a wide fully connected layer
    if (const char *e = getenv("ALQ_DISABLE_V3")) { if (e[0] == '1') g->p3.ok = false; }
    return ALQ_OK;
}

static int set4(alq_model *m, Igemm4Plan *p4, const std::vector<float> &Bmat) {
    igemm4_pack_weights(p4, Bmat);
    hipStream_t st = m->ctx->stream;
    if (!p4->d_tdesc) {
        ALQ_TRY(m->dalloc(&p4->d_tdesc, p4->h_tdesc.size()));
        ALQ_TRY(m->dalloc(&p4->d_sdesc, p4->h_sdesc.size()));
        ALQ_TRY(m->dalloc(&p4->d_pdesc, p4->h_pdesc.size()));
        ALQ_TRY(m->dalloc(&p4->d_ttab, p4->h_ttab.size()));
        ALQ_TRY(m->dalloc(&p4->d_vdesc, p4->h_vdesc.size()));
        ALQ_HIP(hipMemcpyAsync(p4->d_vdesc, p4->h_vdesc.data(), p4->h_vdesc.size() * sizeof(int), hipMemcpyHostToDevice, st));
        ALQ_HIP(hipMemcpyAsync(p4->d_tdesc, p4->h_tdesc.data(), p4->h_tdesc.size() * sizeof(int), hipMemcpyHostToDevice, st));
        ALQ_HIP(hipMemcpyAsync(p4->d_sdesc, p4->h_sdesc.data(), p4->h_sdesc.size() * sizeof(int), hipMemcpyHostToDevice, st));
        ALQ_HIP(hipMemcpyAsync(p4->d_pdesc, p4->h_pdesc.data(), p4->h_pdesc.size() * sizeof(int), hipMemcpyHostToDevice, st));
        ALQ_HIP(hipMemcpyAsync(p4->d_ttab, p4->h_ttab.data(), p4->h_ttab.size() * sizeof(int), hipMemcpyHostToDevice, st));
    }
    unsigned short *dw = reinterpret_cast<unsigned short *>(p4->d_W);
    if (!dw) ALQ_TRY(m->dalloc(&dw, p4->h_W.size()));
    p4->d_W = dw;
    ALQ_HIP(hipMemcpyAsync(dw, p4->h_W.data(), p4->h_W.size() * sizeof(unsigned short), hipMemcpyHostToDevice, st));
    if (!p4->h_W16.empty()) {
        unsigned short *dw16 = reinterpret_cast<unsigned short *>(p4->d_W16);
        if (!dw16) ALQ_TRY(m->dalloc(&dw16, p4->h_W16.size()));
        p4->d_W16 = dw16;
        ALQ_HIP(hipMemcpyAsync(dw16, p4->h_W16.data(), p4->h_W16.size() * sizeof(unsigned short), hipMemcpyHostToDevice, st));
    }
    ALQ_HIP(hipStreamSynchronize(st));
    std::vector<unsigned short>().swap(p4->h_W);
    std::vector<unsigned short>().swap(p4->h_W16);
    if (p4->alt16) ALQ_TRY(set4(m, p4->alt16.get(), Bmat));       // the fp16x2-only twin: own tables, two-piece weights
    return ALQ_OK;
}

static int gemm_set(alq_model *m, Gemm *g, const std::vector<float> &Bmat) {
    if (g->p4.ok) ALQ_TRY(set4(m, &g->p4, Bmat));
    if (g->pfc.ok) {
        fcgemm_pack_weights(&g->pfc, Bmat);
        unsigned short *dw = reinterpret_cast<unsigned short *>(g->pfc.d_W);
        if (!dw) ALQ_TRY(m->dalloc(&dw, g->pfc.h_W.size()));
        g->pfc.d_W = dw;
        ALQ_HIP(hipMemcpyAsync(dw, g->pfc.h_W.data(), g->pfc.h_W.size() * sizeof(unsigned short), hipMemcpyHostToDevice,
                               m->ctx->stream));
        ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
        std::vector<unsigned short>().swap(g->pfc.h_W);
        if (g->pfc_f16 && c3d_subnormals_ok(m->ctx)) {      // the fp16-pair twin for launches with a static input bound (backward, Fisher pass)
            fcgemm_pack_weights_f16(&g->pfc, Bmat);
            unsigned short *dw16 = reinterpret_cast<unsigned short *>(g->pfc.d_W16);
            if (!dw16) ALQ_TRY(m->dalloc(&dw16, g->pfc.h_W16.size()));
            g->pfc.d_W16 = dw16;
            ALQ_HIP(hipMemcpyAsync(dw16, g->pfc.h_W16.data(), g->pfc.h_W16.size() * sizeof(unsigned short), hipMemcpyHostToDevice, m->ctx->stream));
            ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
            std::vector<unsigned short>().swap(g->pfc.h_W16);
        }
    }
    if (g->pd.ok) {      // direct kernel reads the B matrix [K][Co] as it is
        if (!g->pd.d_W) ALQ_TRY(m->dalloc(&g->pd.d_W, Bmat.size()));
        ALQ_HIP(hipMemcpyAsync(g->pd.d_W, Bmat.data(), Bmat.size() * sizeof(float), hipMemcpyHostToDevice, m->ctx->stream));
        ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
    }
    if (g->p2.ok) {
        igemm2_pack_weights(&g->p2, Bmat);
        if (!g->p2.d_tdesc) {
            ALQ_TRY(m->dalloc(&g->p2.d_tdesc, g->p2.h_tdesc.size()));
            ALQ_TRY(m->dalloc(&g->p2.d_sdesc, g->p2.h_sdesc.size()));
            ALQ_HIP(hipMemcpyAsync(g->p2.d_tdesc, g->p2.h_tdesc.data(), g->p2.h_tdesc.size() * sizeof(int),
                                   hipMemcpyHostToDevice, m->ctx->stream));
            ALQ_HIP(hipMemcpyAsync(g->p2.d_sdesc, g->p2.h_sdesc.data(), g->p2.h_sdesc.size() * sizeof(int),
                                   hipMemcpyHostToDevice, m->ctx->stream));
            g->p2.a.tdesc = g->p2.d_tdesc;
            g->p2.a.sdesc = g->p2.d_sdesc;
        }
        if (g->p3.ok) {
            igemm3_pack_weights(g->p2, &g->p3, Bmat);
            unsigned short *dw = reinterpret_cast<unsigned short *>(g->p3.d_W);
            if (!dw) ALQ_TRY(m->dalloc(&dw, g->p3.h_W.size()));
            g->p3.d_W = dw;
            ALQ_HIP(hipMemcpyAsync(dw, g->p3.h_W.data(), g->p3.h_W.size() * sizeof(unsigned short), hipMemcpyHostToDevice,
                                   m->ctx->stream));
            ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
            std::vector<unsigned short>().swap(g->p3.h_W);
            if (g->p3_f16 && !g->p4.ok) {      // (round 6) the fp16-pair twin for launches with a static input bound
                igemm3_pack_weights_f16(g->p2, &g->p3, Bmat);
                unsigned short *dw16 = reinterpret_cast<unsigned short *>(g->p3.d_W16);
                if (!dw16) ALQ_TRY(m->dalloc(&dw16, g->p3.h_W16.size()));
                g->p3.d_W16 = dw16;
                ALQ_HIP(hipMemcpyAsync(dw16, g->p3.h_W16.data(), g->p3.h_W16.size() * sizeof(unsigned short), hipMemcpyHostToDevice, m->ctx->stream));
                ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
                std::vector<unsigned short>().swap(g->p3.h_W16);
            }
        }
        if (!g->p2.d_W) ALQ_TRY(m->dalloc(&g->p2.d_W, g->p2.h_W.size()));
        ALQ_HIP(hipMemcpyAsync(g->p2.d_W, g->p2.h_W.data(), g->p2.h_W.size() * sizeof(float), hipMemcpyHostToDevice,
                               m->ctx->stream));
        ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
        std::vector<float>().swap(g->p2.h_W);
        return ALQ_OK;
    }
    igemm_pack_weights(&g->p1, Bmat);
    return upload1(m, &g->p1);
}

// returns in *fused whether the epilogue fusion request was honoured (only the pipelined kernel can)
static int gemm_launch(alq_ctx *ctx, const Gemm &g, const View &in, const View &out, const float *bias, int relu,
                       int accumulate, int N, int cls, const Igemm2Fuse *fuse = nullptr, bool *fused = nullptr, float fc_in_bound = 0.f) {
    if (g.pd.ok && !accumulate && !(fuse && (fuse->mask || fuse->osumB || fuse->split))) {
        if (fused) *fused = fuse != nullptr;
        return direct_launch(ctx, g.pd, in, out, bias, relu, N, fuse ? fuse->osumA : nullptr, PROF_DIRECT);
    }
    if (g.pfc.ok && !accumulate && !fuse && !g_dbg_knobs[4] && !g_dbg_knobs[5]) {     // wide fc layer: streaming bf16x3 GEMM
        if (fused) *fused = false;
        // forward launches of wide fc layers: fp16 pairs under the per-patch maxima the launch measures (fcgemm.hip, round 6)
        return fcgemm_launch(ctx, g.pfc, in, out, bias, relu, N, cls == PROF_IGEMM_BWD ? PROF_IGEMM3_BWD : PROF_IGEMM3_FWD,
                             (g_no_f16x2 || bias || relu) ? 0.f : fc_in_bound,
                             (cls == PROF_IGEMM_FWD && !g_no_f16x2) ? g.fc_row_amax : nullptr);
    }
    const bool split_view = in.split != 0 || out.split != 0;
    ALQ_REQUIRE(!split_view || g.p4.ok, ALQ_EUNSUPPORTED, "split concat view without a two-slot plan");
    if (g.p4.ok && ((!g_dbg_knobs[4] && !g_dbg_knobs[5]) || split_view)) {      // two-slot bf16x3 engine
        if (fused) *fused = fuse != nullptr;
        return igemm4_launch(ctx, g.p4, in, out, bias, relu, accumulate, N,
                             cls == PROF_IGEMM_BWD ? PROF_IGEMM3_BWD : PROF_IGEMM3_FWD, fuse);
    }
    if (g.p3.ok && !g_dbg_knobs[4]) {      // bf16x3 split on the matrix cores (fp32-equivalent accuracy)
        if (fused) *fused = fuse != nullptr;
        return igemm3_launch(ctx, g.p2, g.p3, in, out, bias, relu, accumulate, N,
                             cls == PROF_IGEMM_BWD ? PROF_IGEMM3_BWD : PROF_IGEMM3_FWD, fuse);
    }
    if (g.p2.ok) {
        if (fused) *fused = fuse != nullptr;
        return igemm2_launch(ctx, g.p2, in, out, bias, relu, accumulate, N, cls, fuse);
    }
    if (fused) *fused = false;
    return igemm_launch(ctx, g.p1, in, out, bias, relu, accumulate, N, cls);
}

static void enum_taps(const int k[3], std::vector<int> *tz, std::vector<int> *ty, std::vector<int> *tx) {
    for (int z = 0; z < k[0]; ++z)
        for (int y = 0; y < k[1]; ++y)
            for (int x = 0; x < k[2]; ++x) {
                tz->push_back(z); ty->push_back(y); tx->push_back(x);
            }
}

static int build_model(alq_model *m, const alq_layer_t *specs, int n_layers) {
    m->layers.resize(n_layers);
    // ---- pass 1: shapes ------------------------------------------------------------------
    struct Shp { int D, H, W, C; };
    std::vector<Shp> outs(n_layers);
    Shp cur = {m->in_dims[0], m->in_dims[1], m->in_dims[2], m->in_dims[3]};
    bool flat = false;
    int pidx = 0;
    std::vector<int> src_of(n_layers, -1);   // dest layer -> src layer
    std::vector<int> dest_of(n_layers, -1);  // src layer -> dest layer
    for (int i = 0; i < n_layers; ++i) {
        Layer &ly = m->layers[i];
        ly.spec = specs[i];
        const alq_layer_t &sp = ly.spec;
        Shp in = cur;
        if (sp.skip_src >= 0) {
            ALQ_REQUIRE(sp.skip_src < i - 1, ALQ_EUNSUPPORTED, "layer %d: skip source %d must be an earlier, non-adjacent layer", i, sp.skip_src);
            ALQ_REQUIRE(dest_of[sp.skip_src] < 0, ALQ_EUNSUPPORTED, "layer %d feeds more than one concat", sp.skip_src);
            const Shp &s = outs[sp.skip_src];
            ALQ_REQUIRE(!flat && s.D == cur.D && s.H == cur.H && s.W == cur.W, ALQ_EUNSUPPORTED,
                        "layer %d: concat of maps of different size (resize_image_with_crop_or_pad) is outside the scored path", i);
            in.C = cur.C + s.C;
            src_of[i] = sp.skip_src;
            dest_of[sp.skip_src] = i;
        }
        Shp o = in;
        switch (sp.type) {
            case ALQ_CONV: {
                ALQ_REQUIRE(!flat, ALQ_EINVAL, "layer %d: conv after fc", i);
                ALQ_REQUIRE(sp.s[0] == 1 && sp.s[1] == 1 && sp.s[2] == 1, ALQ_EUNSUPPORTED, "layer %d: strided conv", i);
                for (int d = 0; d < 3; ++d) {
                    int od;
                    same_pads((&in.D)[d], sp.k[d], 1, &od, &ly.lo[d]);
                }
                o.C = sp.cout;
                ly.pidx = pidx++;
                ly.w_elems = (int64_t)sp.k[0] * sp.k[1] * sp.k[2] * in.C * sp.cout;
                ly.b_elems = sp.cout;
                break;
            }
            case ALQ_CONVT: {
                ALQ_REQUIRE(!flat, ALQ_EINVAL, "layer %d: conv_transpose after fc", i);
                for (int d = 0; d < 3; ++d) {
                    ALQ_REQUIRE(sp.k[d] >= sp.s[d], ALQ_EUNSUPPORTED, "layer %d: conv_transpose kernel < stride", i);
                    int od;
                    same_pads((&in.D)[d] * sp.s[d], sp.k[d], sp.s[d], &od, &ly.lo[d]);
                }
                ALQ_REQUIRE(sp.s[0] == sp.s[1] || in.D == 1, ALQ_EUNSUPPORTED, "layer %d: anisotropic stride", i);
                ALQ_REQUIRE(sp.s[1] == sp.s[2], ALQ_EUNSUPPORTED, "layer %d: anisotropic stride", i);
                o.D = in.D * sp.s[0]; o.H = in.H * sp.s[1]; o.W = in.W * sp.s[2];
                o.C = sp.cout;
                ly.pidx = pidx++;
                ly.w_elems = (int64_t)sp.k[0] * sp.k[1] * sp.k[2] * in.C * sp.cout;
                ly.b_elems = sp.cout;
                break;
            }
            case ALQ_POOL: {
                ALQ_REQUIRE(!flat, ALQ_EINVAL, "layer %d: pool after fc", i);
                for (int d = 0; d < 3; ++d) {
                    ALQ_REQUIRE(sp.k[d] == sp.s[d], ALQ_EUNSUPPORTED, "layer %d: pool window != stride", i);
                    ALQ_REQUIRE(sp.k[0] * sp.k[1] * sp.k[2] <= 255, ALQ_EUNSUPPORTED, "layer %d: pool window too large", i);
                    same_pads((&in.D)[d], sp.k[d], sp.s[d], &(&o.D)[d], &ly.lo[d]);
                }
                break;
            }
            case ALQ_FC: {
                ly.F = (int64_t)in.D * in.H * in.W * in.C;
                o = {1, 1, 1, sp.cout};
                ly.pidx = pidx++;
                ly.w_elems = ly.F * sp.cout;
                ly.b_elems = sp.cout;
                flat = true;
                break;
            }
            default:
                ALQ_REQUIRE(false, ALQ_EINVAL, "layer %d: unknown type %d", i, sp.type);
        }
        outs[i] = o;
        cur = o;
    }
    m->L = pidx;
    ALQ_REQUIRE(n_layers > 0 && specs[n_layers - 1].type == ALQ_FC, ALQ_EUNSUPPORTED,
                "the scored path needs an fc head (get_gradients, NN_extended.py:1025)");
    m->nclass = outs[n_layers - 1].C;
    ALQ_REQUIRE(m->nclass >= 2 && m->nclass <= 64, ALQ_EUNSUPPORTED, "%d classes unsupported", m->nclass);
    {   // The GEMM engines address a tensor with unsigned 32-bit BYTE offsets from its base, and the two halves of a split
        // concat are one allocation: the workspace batch is the largest one every allocation stays below 2^30 floats with
        // (NET-C at 32^3: 2047 patches).  A caller that asks for more gets this many per device pass - alq_model_max_batch
        // reports it and the host walks a larger batch in passes (per-patch results do not depend on the split) - instead
        // of an error at the first launch.
        long long worst = m->epp;                                           // floats per patch of the largest allocation
        for (int i = 0; i < n_layers; ++i) {
            long long e = (long long)outs[i].D * outs[i].H * outs[i].W * outs[i].C;
            if (src_of[i] >= 0) {                                           // consumer of a concat: source + direct producer share one
                const Shp &sh = outs[src_of[i]];
                e = std::max(e, (long long)sh.D * sh.H * sh.W * (outs[src_of[i]].C + outs[i - 1].C));
            }
            worst = std::max(worst, e);
        }
        const long long limit = std::max(1LL, ((1LL << 30) - 64) / std::max(worst, 1LL));
        m->requested_batch = m->max_batch;
        if (m->max_batch > limit) m->max_batch = (int)limit;
    }
    const int NB = m->max_batch;
    ALQ_REQUIRE(m->L <= 16, ALQ_EUNSUPPORTED, "%d parameterised layers > 16", m->L);

    // ---- pass 2: buffers (concat = two producers writing channel slices of one buffer) ----
    auto mkview = [](float *p, const Shp &s, int cs, int c0, int C) {
        View v; v.p = p; v.D = s.D; v.H = s.H; v.W = s.W; v.cs = cs; v.c0 = c0; v.C = C; return v;
    };
    std::vector<View> act(n_layers), dact(n_layers);
    std::vector<View> catv(n_layers), dcatv(n_layers);
    for (int d = 0; d < n_layers; ++d) {
        if (src_of[d] < 0) continue;
        const int s = src_of[d];
        ALQ_REQUIRE(dest_of[d - 1] < 0 && src_of[d] != d - 1, ALQ_EUNSUPPORTED, "layer %d: unsupported skip topology", d);
        const int Cs = outs[s].C, Cp = outs[d - 1].C;
        const Shp &sh = outs[s];
        float *buf, *dbuf;
        const size_t el = (size_t)NB * sh.D * sh.H * sh.W * (Cs + Cp);
        ALQ_TRY(m->dalloc(&buf, el));
        ALQ_TRY(m->dalloc(&dbuf, el));
        unsigned char *sbuf = nullptr;      // sign field of the whole allocation (View::sg): float offset / 4
        if (Cs % 4 == 0 && Cp % 4 == 0) ALQ_TRY(m->dalloc(&sbuf, el / 4));
        // Split concat: when the consumer's forward and backward contractions both run on the two-slot engine, the
        // two producers keep DENSE tensors (the halves of one allocation) and the consumer reads / writes them as
        // two channel groups.  Interleaved slices cost twice the cache lines per staged or stored row (32 of 64
        // bytes used), and the engine's staging and store parts are bound by lines touched.
        bool split_ok = false;
        if (Cs == Cp && Cs % 8 == 0 && specs[d].type == ALQ_CONV && !getenv("ALQ_DISABLE_V4") && !getenv("ALQ_NO_SPLIT")) {
            const Layer &dl = m->layers[d];
            G4Geom gf;
            gf.kind = 0;
            gf.ID = sh.D; gf.IH = sh.H; gf.IW = sh.W; gf.Ci = Cs + Cp;
            gf.OD = outs[d].D; gf.OH = outs[d].H; gf.OW = outs[d].W; gf.Co = specs[d].cout;
            for (int q = 0; q < 3; ++q) { gf.k[q] = specs[d].k[q]; gf.s[q] = specs[d].s[q]; gf.lo[q] = dl.lo[q]; }
            G4Geom gb = gf;
            gb.flipped = true;
            gb.ID = outs[d].D; gb.IH = outs[d].H; gb.IW = outs[d].W; gb.Ci = specs[d].cout;
            gb.OD = sh.D; gb.OH = sh.H; gb.OW = sh.W; gb.Co = Cs + Cp;
            Igemm4Plan tf, tb;
            ALQ_TRY(igemm4_build_plan(gf, NB, &tf));
            ALQ_TRY(igemm4_build_plan(gb, NB, &tb));
            split_ok = tf.ok && tb.ok && !tb.a.pair;
        }
        if (split_ok) {
            const long long half = (long long)NB * sh.D * sh.H * sh.W * Cs;
            act[s] = mkview(buf, sh, Cs, 0, Cs);
            act[d - 1] = mkview(buf + half, sh, Cp, 0, Cp);
            dact[s] = mkview(dbuf, sh, Cs, 0, Cs);
            dact[d - 1] = mkview(dbuf + half, sh, Cp, 0, Cp);
            catv[d] = mkview(buf, sh, Cs, 0, Cs + Cp);
            catv[d].split = Cs; catv[d].delta = half;
            if (sbuf) { act[s].sg = sbuf; act[d - 1].sg = sbuf + half / 4; catv[d].sg = sbuf; }
            dcatv[d] = mkview(dbuf, sh, Cs, 0, Cs + Cp);
            dcatv[d].split = Cs; dcatv[d].delta = half;
            m->layers[s].out_is_skip_src = true;
            continue;
        }
        act[s] = mkview(buf, sh, Cs + Cp, 0, Cs);
        act[d - 1] = mkview(buf, sh, Cs + Cp, Cs, Cp);
        dact[s] = mkview(dbuf, sh, Cs + Cp, 0, Cs);
        dact[d - 1] = mkview(dbuf, sh, Cs + Cp, Cs, Cp);
        catv[d] = mkview(buf, sh, Cs + Cp, 0, Cs + Cp);
        dcatv[d] = mkview(dbuf, sh, Cs + Cp, 0, Cs + Cp);
        if (sbuf) { act[s].sg = sbuf; act[d - 1].sg = sbuf; catv[d].sg = sbuf; }
        m->layers[s].out_is_skip_src = true;
    }
    for (int i = 0; i < n_layers; ++i) {
        if (act[i].p) continue;
        const Shp &sh = outs[i];
        float *buf, *dbuf;
        const size_t el = (size_t)NB * sh.D * sh.H * sh.W * sh.C;
        ALQ_TRY(m->dalloc(&buf, el));
        ALQ_TRY(m->dalloc(&dbuf, el));
        act[i] = mkview(buf, sh, sh.C, 0, sh.C);
        dact[i] = mkview(dbuf, sh, sh.C, 0, sh.C);
        if (sh.C % 4 == 0 && specs[i].type != ALQ_FC && (specs[i].type == ALQ_POOL || specs[i].relu)) ALQ_TRY(m->dalloc(&act[i].sg, el / 4));
    }
    m->logits = act[n_layers - 1].p;
    m->dlogits = dact[n_layers - 1].p;
    ALQ_TRY(m->dalloc(&m->post, (size_t)NB * m->nclass));
    ALQ_TRY(m->dalloc(&m->S, (size_t)NB * m->L));
    ALQ_TRY(m->dalloc(&m->sizes, (size_t)m->L));
    ALQ_TRY(m->dalloc(&m->Apart, (size_t)((NB + 63) / 64) * m->L * m->L));

    // ---- pass 3: per-layer plans ---------------------------------------------------------
    Shp inshape = {m->in_dims[0], m->in_dims[1], m->in_dims[2], m->in_dims[3]};
    std::vector<double> h_sizes(m->L);
    for (int i = 0; i < n_layers; ++i) {
        Layer &ly = m->layers[i];
        const alq_layer_t &sp = ly.spec;
        if (src_of[i] >= 0) {
            ly.in = catv[i];
            ly.din = dcatv[i];
        } else if (i == 0) {
            ly.in = mkview(nullptr, inshape, inshape.C, 0, inshape.C);   // pointer bound per call
            ly.din = View();
        } else {
            ly.in = act[i - 1];
            ly.din = dact[i - 1];
        }
        ly.out = act[i];
        ly.dout = dact[i];
        if (ly.spec.type != ALQ_FC) ALQ_TRY(m->dalloc(&ly.osum, (size_t)NB * ly.out.vox()));
        if (ly.pidx >= 0) {
            h_sizes[ly.pidx] = (double)(ly.w_elems + ly.b_elems);
            ALQ_TRY(m->dalloc(&ly.d_bias, (size_t)ly.b_elems));
            ALQ_TRY(m->dalloc(&ly.asum, (size_t)NB * ly.in.vox()));
            ALQ_TRY(m->dalloc(&ly.dsum, (size_t)NB * ly.out.vox()));
        }
        const bool first_param = (ly.pidx == 0);
        if (sp.type == ALQ_CONV) {
            ConvDesc d;
            d.ID = ly.in.D; d.IH = ly.in.H; d.IW = ly.in.W; d.Ci = ly.in.C;
            d.OD = ly.out.D; d.OH = ly.out.H; d.OW = ly.out.W; d.Co = sp.cout;
            d.MD = d.OD; d.MH = d.OH; d.MW = d.OW;
            enum_taps(sp.k, &d.tz, &d.ty, &d.tx);
            for (size_t t = 0; t < d.tz.size(); ++t) { d.tz[t] -= ly.lo[0]; d.ty[t] -= ly.lo[1]; d.tx[t] -= ly.lo[2]; }
            ly.fwd.resize(1);
            G4Geom g4;
            g4.kind = 0;
            g4.ID = ly.in.D; g4.IH = ly.in.H; g4.IW = ly.in.W; g4.Ci = ly.in.C;
            g4.OD = ly.out.D; g4.OH = ly.out.H; g4.OW = ly.out.W; g4.Co = sp.cout;
            for (int q = 0; q < 3; ++q) { g4.k[q] = sp.k[q]; g4.s[q] = sp.s[q]; g4.lo[q] = ly.lo[q]; }
            ALQ_TRY(gemm_build(d, NB, &ly.fwd[0], &g4));
            // a 2-D window of 25 taps or more: the two-slot engine's tiles re-stage too much halo - NET-B's conv2 (24 -> 32 channels,
            // 5 x 5 at 32^2) takes 673 us per 2048 patches there and ~390 on igemm3 (round 6; ALQ_NO_WIDE2D_RULE=1 = the other arm)
            if (ly.fwd[0].p4.ok && ly.fwd[0].p3.ok && d.ID == 1 && d.tz.size() >= 25 && !getenv("ALQ_NO_WIDE2D_RULE")) ly.fwd[0].p4.ok = false;
            // a forward launch that stays on igemm3, tiles of one patch: the fp16-pair twin of its weights for Fisher / forward-only passes
            // (run_forward, round 6; ALQ_NO_V3_F16_FWD=1: bf16 triples as before)
            if (!ly.fwd[0].p4.ok && !ly.fwd[0].pd.ok && !ly.fwd[0].pfc.ok && ly.fwd[0].p3.ok && ly.fwd[0].p2.a.PT == 1 && sp.skip_src < 0 &&
                !getenv("ALQ_NO_V3_F16_FWD") && !getenv("ALQ_NO_V3_F16")) {
                ly.fwd[0].p3_f16 = true;
                m->v3_fwd_f16 = true;
            }
            if (!ly.fwd[0].p4.ok && !ly.fwd[0].p3.ok && !ly.fwd[0].pd.ok && sp.cout > 32 && !getenv("ALQ_NO_CO_SPLIT")) {
                for (int w : {32, 48, 16}) {
                    if (sp.cout % w || sp.cout <= w) continue;
                    std::vector<Gemm> parts(sp.cout / w);
                    bool ok = true;
                    for (size_t j = 0; j < parts.size() && ok; ++j) {
                        ConvDesc dj = d;
                        dj.Co = w;
                        G4Geom gj = g4;
                        gj.Co = w;
                        ALQ_TRY(gemm_build(dj, NB, &parts[j], &gj));
                        ok = parts[j].p4.ok || parts[j].p3.ok;
                    }
                    if (ok) { ly.fwd_co = std::move(parts); ly.fwd_co_w = w; break; }
                }
            }
            if (!first_param && sp.relu) {
                ALQ_TRY(c3d_fwd_build(ly.in, ly.out, sp.k, ly.lo, sp.s, &ly.c3f));
                ALQ_TRY(c3d_bwd_build(ly.in, ly.out, sp.k, ly.lo, sp.s, &ly.c3b));
                ALQ_TRY(e3d_build(ly.in, ly.out, sp.k, ly.lo, sp.s, &ly.e3b));
                ALQ_TRY(d3d_build(ly.in, ly.out, sp.k, ly.lo, sp.s, &ly.d3f));
                ALQ_TRY(f3d_build(ly.in, ly.out, sp.k, ly.lo, sp.s, &ly.f3f));
            }
            if (!first_param) {
                ConvDesc b;
                b.ID = ly.out.D; b.IH = ly.out.H; b.IW = ly.out.W; b.Ci = sp.cout;
                b.OD = ly.in.D; b.OH = ly.in.H; b.OW = ly.in.W; b.Co = ly.in.C;
                b.MD = b.OD; b.MH = b.OH; b.MW = b.OW;
                enum_taps(sp.k, &b.tz, &b.ty, &b.tx);
                for (size_t t = 0; t < b.tz.size(); ++t) {
                    b.tz[t] = ly.lo[0] - b.tz[t]; b.ty[t] = ly.lo[1] - b.ty[t]; b.tx[t] = ly.lo[2] - b.tx[t];
                }
                G4Geom gb = g4;
                gb.flipped = true;
                gb.ID = ly.out.D; gb.IH = ly.out.H; gb.IW = ly.out.W; gb.Ci = sp.cout;
                gb.OD = ly.in.D; gb.OH = ly.in.H; gb.OW = ly.in.W; gb.Co = ly.in.C;
                ALQ_TRY(gemm_build(b, NB, &ly.bwd, &gb));
                ly.has_bwd = true;
                // a backward launch that stays on igemm3 (no two-slot plan: NET-B's 5 x 5 and 96-channel convs) takes fp16 pairs under the
                // cotangent bound of a Fisher pass like the two-slot launches do (round 6; ALQ_NO_V3_F16=1: bf16 triples as before)
                ly.bwd.p3_f16 = !ly.bwd.p4.ok && ly.bwd.p3.ok && !getenv("ALQ_NO_V3_F16");
            }
        } else if (sp.type == ALQ_CONVT) {
            // y[p] = sum_{q,t: s*q + t - lo = p} x[q] W[t]; output parity class c = p mod s uses the
            // taps t == (c + lo) mod s at input offset (c + lo - t)/s   (<= 0)
            std::vector<int> az, ay, ax;
            enum_taps(sp.k, &az, &ay, &ax);
            const int nclsz = ly.in.D == 1 && sp.s[0] == 1 ? 1 : sp.s[0];
            for (int cz = 0; cz < nclsz; ++cz)
                for (int cy = 0; cy < sp.s[1]; ++cy)
                    for (int cx = 0; cx < sp.s[2]; ++cx) {
                        ConvDesc d;
                        d.ID = ly.in.D; d.IH = ly.in.H; d.IW = ly.in.W; d.Ci = ly.in.C;
                        d.OD = ly.out.D; d.OH = ly.out.H; d.OW = ly.out.W; d.Co = sp.cout;
                        d.MD = d.ID; d.MH = d.IH; d.MW = d.IW;
                        d.so = sp.s[2];
                        d.ooff[0] = cz; d.ooff[1] = cy; d.ooff[2] = cx;
                        std::vector<int> tl;
                        for (size_t t = 0; t < az.size(); ++t) {
                            const int nz = cz + ly.lo[0] - az[t], ny = cy + ly.lo[1] - ay[t], nx = cx + ly.lo[2] - ax[t];
                            if (nz % sp.s[0] || ny % sp.s[1] || nx % sp.s[2]) continue;
                            d.tz.push_back(nz / sp.s[0]); d.ty.push_back(ny / sp.s[1]); d.tx.push_back(nx / sp.s[2]);
                            tl.push_back((int)t);
                        }
                        ALQ_REQUIRE(!tl.empty(), ALQ_EUNSUPPORTED, "layer %d: empty conv_transpose class", i);
                        ly.class_taps.push_back(tl);
                        ly.fwd.emplace_back();
                        G4Geom gc;                // this class alone on the two-slot engine (used when the fused form does not fit)
                        gc.kind = 3;
                        gc.ID = ly.in.D; gc.IH = ly.in.H; gc.IW = ly.in.W; gc.Ci = ly.in.C;
                        gc.OD = ly.out.D; gc.OH = ly.out.H; gc.OW = ly.out.W; gc.Co = sp.cout;
                        for (int q = 0; q < 3; ++q) { gc.k[q] = sp.k[q]; gc.s[q] = sp.s[q]; gc.lo[q] = ly.lo[q]; }
                        gc.cls[0] = cz; gc.cls[1] = cy; gc.cls[2] = cx;
                        ALQ_TRY(gemm_build(d, NB, &ly.fwd.back(), &gc));
                    }
            ALQ_REQUIRE(sp.s[0] == sp.s[2] || (ly.in.D == 1 && sp.s[0] == 1), ALQ_EUNSUPPORTED,
                        "layer %d: conv_transpose stride must be isotropic", i);
            G4Geom g4;
            g4.ID = ly.in.D; g4.IH = ly.in.H; g4.IW = ly.in.W; g4.Ci = ly.in.C;
            g4.OD = ly.out.D; g4.OH = ly.out.H; g4.OW = ly.out.W; g4.Co = sp.cout;
            for (int q = 0; q < 3; ++q) { g4.k[q] = sp.k[q]; g4.s[q] = sp.s[q]; g4.lo[q] = ly.lo[q]; }
            if (!getenv("ALQ_DISABLE_V4")) {
                G4Geom gf = g4;
                gf.kind = 2;
                gf.flops_per_patch = 0;
                for (const Gemm &c : ly.fwd) gf.flops_per_patch += c.p1.flops_per_patch;
                ALQ_TRY(igemm4_build_plan(gf, NB, &ly.fwd_all));
                if (!ly.fwd_all.ok && !getenv("ALQ_NO_CLASS_TILES")) {      // too large to stage once: the classes as tiles of one launch
                    gf.kind = 4;
                    ALQ_TRY(igemm4_build_plan(gf, NB, &ly.fwd_all));
                }
            }
            if (!sp.relu) ALQ_TRY(t3d_build(ly.in, ly.out, sp.k, ly.lo, sp.s, &ly.t3f, &ly.t3b));
            if (first_param) ly.t3b.ok = false;
            if (!first_param) {
                ConvDesc b;
                b.ID = ly.out.D; b.IH = ly.out.H; b.IW = ly.out.W; b.Ci = sp.cout;
                b.OD = ly.in.D; b.OH = ly.in.H; b.OW = ly.in.W; b.Co = ly.in.C;
                b.MD = b.OD; b.MH = b.OH; b.MW = b.OW;
                b.sm = sp.s[2];
                enum_taps(sp.k, &b.tz, &b.ty, &b.tx);
                for (size_t t = 0; t < b.tz.size(); ++t) { b.tz[t] -= ly.lo[0]; b.ty[t] -= ly.lo[1]; b.tx[t] -= ly.lo[2]; }
                G4Geom gb = g4;
                gb.kind = 1;
                gb.ID = ly.out.D; gb.IH = ly.out.H; gb.IW = ly.out.W; gb.Ci = sp.cout;
                gb.OD = ly.in.D; gb.OH = ly.in.H; gb.OW = ly.in.W; gb.Co = ly.in.C;
                ALQ_TRY(gemm_build(b, NB, &ly.bwd, &gb));
                ly.has_bwd = true;
            }
        } else if (sp.type == ALQ_POOL) {
            uint8_t *am;
            ALQ_TRY(m->dalloc(&am, (size_t)NB * ly.out.vox() * ly.out.C));
            ly.argmax = am;
        } else {   // fc
            ALQ_REQUIRE(ly.in.cs == ly.in.C && ly.in.c0 == 0, ALQ_EUNSUPPORTED, "layer %d: fc on a concat slice", i);
            if (sp.cout <= 8) {
                ly.dense_fc_small = true;
                ly.fc_slices = fc_small_slices(ly.F);
                ALQ_TRY(m->dalloc(&ly.d_Wp, (size_t)sp.cout * ly.F));
                ALQ_TRY(m->dalloc(&ly.fc_partials, (size_t)NB * ly.fc_slices * sp.cout));
                if (sp.cout == 2 && ly.F % 1024 == 0 && i == n_layers - 1 && i > 0 && m->layers[i - 1].spec.relu && m->layers[i - 1].pidx > 0 &&
                    m->layers[i - 1].spec.type == ALQ_CONV && m->layers[i - 1].out.C == 8 && !getenv("ALQ_NO_FC_BITS")) {
                    ALQ_TRY(m->dalloc(&ly.fc_maskbits, (size_t)NB * (ly.F / 16)));      // one sign byte per 4 elements
                    ALQ_TRY(m->dalloc(&ly.fc_wv, (size_t)ly.F));
                    ALQ_TRY(m->dalloc(&ly.fc_wv16, (size_t)ly.F));
                    const Layer &cv = m->layers[i - 1];
                    const Igemm4Plan &fp = cv.fwd[0].p4;
                    if (sp.cout == 2 && !sp.relu && fp.ok && fp.NTW == 1 && !fp.multi && fp.a.pair && fp.a.PT == 1 && cv.out.cs == 8 &&
                        cv.out.c0 == 0 && !cv.out.split && cv.osum && !cv.out_is_skip_src && !getenv("ALQ_NO_FC_FUSE") &&
                        // the backward pass must be able to work from the bits alone (bits_ok in run_backward)
                        cv.bwd.p4.ok && cv.bwd.p4.NTW == 1 && !cv.bwd.p4.multi && cv.bwd.p4.a.PT == 1 && !cv.dout.split) {
                        ly.fc_slices2 = fp.a.tpg * 4;
                        ALQ_TRY(m->dalloc(&ly.fc_part2, (size_t)NB * ly.fc_slices2));
                        if (cv.c3f.ok && ly.F == (int64_t)cv.out.vox() * 8) {
                            ALQ_TRY(m->dalloc(&ly.c3_part, (size_t)NB * 4));
                            ALQ_TRY(m->dalloc(&ly.c3_asum, (size_t)NB * 4));
                            if (cv.c3b.ok) ALQ_TRY(m->dalloc(&ly.fc_wv16c, (size_t)ly.F * 2));
                        }
                    }
                }
            } else {
                ConvDesc d;
                d.ID = d.IH = d.IW = 1; d.Ci = (int)ly.F;
                d.OD = d.OH = d.OW = 1; d.Co = sp.cout;
                d.MD = d.MH = d.MW = 1;
                d.tz = {0}; d.ty = {0}; d.tx = {0};
                ly.fwd.resize(1);
                ALQ_TRY(gemm_build(d, NB, &ly.fwd[0]));
                if (ly.fwd[0].pfc.ok && !getenv("ALQ_NO_FC_F16_FWD")) {      // forward launch of a wide fc layer on fp16 pairs (measured per-patch maxima)
                    ly.fwd[0].pfc_f16 = true;
                    ALQ_TRY(m->dalloc(&ly.fwd[0].fc_row_amax, (size_t)m->max_batch));
                }
                if (!first_param) {
                    ConvDesc b = d;
                    b.Ci = sp.cout; b.Co = (int)ly.F;
                    ALQ_TRY(gemm_build(b, NB, &ly.bwd));
                    ly.bwd.pfc_f16 = ly.bwd.pfc.ok && !getenv("ALQ_NO_FC_F16");      // backward launch: fp16 pairs under the static cotangent bound
                    ly.has_bwd = true;
                }
            }
        }
    }
    ALQ_HIP(hipMemcpy(m->sizes, h_sizes.data(), m->L * sizeof(double), hipMemcpyHostToDevice));
    // box-dot slab partials: conv / fc reduce over the OUTPUT grid, conv_transpose over the input grid
    std::vector<int> h_nslab(m->L, 1);
    for (const Layer &ly : m->layers) {
        if (ly.pidx < 0) continue;
        if (ly.spec.type == ALQ_CONV) h_nslab[ly.pidx] = boxdot_conv_slabs(ly.out.D, ly.out.H, ly.out.W, ly.spec.k);
        else if (ly.spec.type == ALQ_CONVT) h_nslab[ly.pidx] = boxdot_convT_slabs(ly.in.D, ly.in.H, ly.in.W, ly.spec.k, ly.spec.s);
        else h_nslab[ly.pidx] = boxdot_slabs(1);
        m->nslab_max = std::max(m->nslab_max, h_nslab[ly.pidx]);
    }
    ALQ_TRY(m->dalloc(&m->nslab, (size_t)m->L));
    ALQ_TRY(m->dalloc(&m->Spart, (size_t)m->L * NB * m->nslab_max));
    ALQ_HIP(hipMemcpy(m->nslab, h_nslab.data(), m->L * sizeof(int), hipMemcpyHostToDevice));
    return ALQ_OK;
}

// dropout of a training / MC forward pass: layers whose OUTPUT is dropped (NN.py:169-171), one keep probability
struct DropSpec {
    unsigned long long layers = 0;     // bit i = layer i
    float keep_prob = 1.f;
    unsigned long long seed = 0;
    long long first_sample = 0;        // id of patch 0 of this call: masks are keyed by sample id, not by batch position
    bool on(int i) const { return keep_prob < 1.f && i < 64 && ((layers >> i) & 1ull); }
};

static View flat_view(const View &v) {
    View f;
    f.p = v.p; f.D = f.H = f.W = 1;
    f.C = (int)(v.vox() * v.C); f.cs = f.C; f.c0 = 0;
    return f;
}

// ------------------------------------------------------------------------------------------
// keep_all (forward-only calls): every activation stays readable (a feature layer was asked for); otherwise the fc
// head of a two-class net is fused into the last conv in forward-only calls too (no channel sums, no sign bytes)
static int run_forward(alq_model *m, const float *d_x, int N, bool with_sums, bool keep_all = false, const DropSpec *drop = nullptr) {
    alq_ctx *ctx = m->ctx;
    const int nl = (int)m->layers.size();
    bool skip_next = false;      // this layer's outputs were produced by the previous layer's kernel
    bool fc_head_fused = false;  // the logits partials of the fc head came out of the previous conv's epilogue
    bool c3_head = false;        // ... of the plane-sweep engine: one partial per (patch, wave), the head's input sum likewise
    m->last_head_fused = false;
    m->last_c3 = false;
    m->last_t3f = 0;
    m->last_d3f = 0;
    m->last_f3f = 0;
    // A conv / conv_transpose launch contracts with the fp16x2 split if it knows max |x| per patch of (every part of)
    // its input ahead of time: the launches that produce those tensors report them (`prod`), the consumers (`cons`)
    // read one scale per tile.  Producers: the first conv + pool kernel and one-patch-per-tile igemm4 launches.
    const bool no16 = g_no_f16x2 != 0;
    auto use_dcp = [&](int i) {      // layer i = first conv, fused with the pool behind it (direct_conv_pool_launch)
        if (i != 0 || nl < 2) return false;
        const Layer &ly = m->layers[0], &nx = m->layers[1];
        const alq_layer_t &sp = ly.spec;
        const View &in = ly.in;
        return ly.spec.type == ALQ_CONV && nx.spec.type == ALQ_POOL && ly.fwd[0].pd.ok && ly.fwd[0].pd.d_W && in.C == 1 && in.cs == 1 &&
               sp.cout == 8 && sp.k[0] == 3 && sp.k[1] == 3 && sp.k[2] == 3 && ly.lo[0] == 1 && ly.lo[1] == 1 && ly.lo[2] == 1 &&
               nx.spec.k[0] == 2 && nx.spec.k[1] == 2 && nx.spec.k[2] == 2 && nx.lo[0] == 0 && nx.lo[1] == 0 && nx.lo[2] == 0 &&
               in.D % 2 == 0 && in.H % 2 == 0 && in.W % 2 == 0 && nx.out.D * 2 == in.D && nx.out.H * 2 == in.H &&
               nx.out.W * 2 == in.W && ((ly.out.cs | ly.out.c0 | nx.out.cs | nx.out.c0) & 3) == 0 && !g_dbg_knobs[7] &&
               !(drop && (drop->on(0) || drop->on(1)));       // a dropped layer 0 output cannot share a kernel with the pool
    };
    std::vector<char> prod(nl, 0), cons(nl, 0);
    const bool light = !with_sums && !keep_all;       // forward-only: fuse objects only for the launches of the fused head
    if ((with_sums || light) && !no16 && !g_dbg_knobs[3] && !g_dbg_knobs[4] && !g_dbg_knobs[5]) {
        auto prod_ok = [&](int p) {
            if (p < 0) return false;
            const Layer &l = m->layers[p];
            if (p == 0 && use_dcp(0)) return true;
            if (!l.osum) return false;
            if (l.spec.type == ALQ_CONV) return l.fwd[0].p4.ok && !l.fwd[0].pd.ok && !l.fwd[0].pfc.ok && l.fwd[0].p4.a.PT == 1;
            if (l.spec.type == ALQ_CONVT) return l.fwd_all.ok && l.fwd_all.a.PT == 1;
            return false;
        };
        size_t need = 0;
        for (int j = 1; j < nl; ++j) {
            const Layer &l = m->layers[j];
            const Igemm4Plan *pl = l.spec.type == ALQ_CONV ? &l.fwd[0].p4 : (l.spec.type == ALQ_CONVT ? &l.fwd_all : nullptr);
            if (!pl || !pl->ok || !pl->d_W16 || pl->multi || pl->a.PT != 1 || !l.osum) continue;
            if (l.spec.type == ALQ_CONV && (l.fwd[0].pd.ok || l.fwd[0].pfc.ok)) continue;
            // Forward launches: only the conv under the fused fc head.  The fp16x2 split rounds at 2^-22 instead of 2^-24; in
            // the forward pass that noise reaches ReLU decisions (measured: with dec1 or up1 on the split one patch in ~12
            // had a ReLU input of a later layer land on the other side of zero, moving two layer scores by 1 % - the same
            // event any two fp32 implementations produce, four times as often).  The head conv has one ReLU behind it;
            // backward launches have none (the masks are fixed by then), so they take the split wherever a variant exists.
            const int s_ = l.spec.skip_src;
            // ... and the conv whose output reaches the head conv through the (linear) conv_transpose only (NET-C's dec1): its
            // input maxima are not measured (an epilogue in two producer launches cost what the split gave) but DERIVED per patch
            // from the first layer's measured maximum through the layers' L1 norms (k_fwd_bounds) - the fp16 pairs keep their 22
            // bits under a bound that is loose by orders of magnitude.  Measured (profiles/r04al_*): bench +2.6 % same-box, launch
            // 1148 -> 875 us; on a 2000-patch batch against the exact-fp32 engine 129 patches with a flipped fragile unit instead
            // of 115 (bf16x3 everywhere: 109; each engine has its own set against fp64).  Default since round 5 (the round-4
            // verdict: a faster, equally accurate path is not withheld for an accounting ratio); ALQ_NO_F16_DERIVED=1 is the A/B arm.
            if (m->f16_fwd_mask < 0 && !m->no_f16_derived && nl <= 16 && ((m->f16_fwd_derived >> j) & 1) && l.spec.type == ALQ_CONV && use_dcp(0) &&
                (s_ >= 0) == (l.in.split != 0) && !(drop && drop->layers)) {
                cons[j] = 2;
                prod[0] = 1;
                continue;
            }
            if (!(j == nl - 2 && m->layers[nl - 1].fc_part2 && l.spec.type == ALQ_CONV) && m->f16_fwd_mask < 0) continue;
            if (l.spec.type == ALQ_CONVT && !pl->fic) continue;                 // MULTI has no F16 variant
            if (!prod_ok(j - 1) || (s_ >= 0 && !prod_ok(s_))) continue;
            if ((s_ >= 0) != (l.in.split != 0)) continue;                       // two parts <-> a split view
            if (m->f16_fwd_mask >= 0 && !((m->f16_fwd_mask >> j) & 1)) continue;      // diagnostics: consumers by layer bit
            cons[j] = 1; prod[j - 1] = 1;
            if (s_ >= 0) prod[s_] = 1;
        }
        for (int p = 0; p < nl; ++p) {
            if (!prod[p]) continue;
            Layer &l = m->layers[p];
            if (!l.amax_fwd) ALQ_TRY(m->dalloc(&l.amax_fwd, (size_t)m->max_batch));
            if (l.spec.type == ALQ_CONV && !(p == 0 && use_dcp(0))) need = std::max(need, (size_t)l.fwd[0].p4.a.tpg * 4);
            if (l.spec.type == ALQ_CONVT) need = std::max(need, (size_t)l.fwd_all.a.tpg * (l.fwd_all.multi ? l.fwd_all.a.ngr : 1) * 4);
            if (l.spec.type == ALQ_CONVT && l.t3f.ok) need = std::max(need, (size_t)16);      // row-sweep engine: one maximum per input plane
        }
        if (need > m->amax_tiles_len) { ALQ_TRY(m->dalloc(&m->amax_tiles, (size_t)m->max_batch * need)); m->amax_tiles_len = need; }
    }
    bool any_derived = false;
    for (int j = 0; j < nl; ++j) any_derived = any_derived || cons[j] == 2;
    m->last_f16_derived = any_derived;
    if (any_derived && !m->bound_all) {
        ALQ_TRY(m->dalloc(&m->bound_all, (size_t)nl * m->max_batch));
        for (int k = 0; k < nl; ++k) m->layers[k].bound_fwd = m->bound_all + (size_t)k * m->max_batch;
    }
    // Forward launches that stay on igemm3 (no two-slot plan: NET-B's conv1 .. conv3) on fp16 pairs, one scale per patch: the
    // measured maximum of every patch of the network input, pushed through the layers' L1 norms, bounds every layer's input.
    // Fisher passes and forward-only passes; not the passes that keep every activation for training, not under dropout.
    const bool v3f16 = m->v3_fwd_f16 && (with_sums || light) && !no16 && !drop && !any_derived && !use_dcp(0) && nl <= 16 &&
                       !g_dbg_knobs[3] && !g_dbg_knobs[4] && !g_dbg_knobs[5] && m->layers[0].in.split == 0 && m->layers[0].in.cs == m->layers[0].in.C;
    if (v3f16) {
        if (!m->in_amax) ALQ_TRY(m->dalloc(&m->in_amax, (size_t)m->max_batch));
        if (!m->bound_all) {
            ALQ_TRY(m->dalloc(&m->bound_all, (size_t)nl * m->max_batch));
            for (int k = 0; k < nl; ++k) m->layers[k].bound_fwd = m->bound_all + (size_t)k * m->max_batch;
        }
        ALQ_TRY(k_rowmax_abs(ctx, d_x, N, (long long)m->layers[0].in.vox() * m->layers[0].in.C, m->in_amax));
        FwdBoundsArgs ba;
        ba.nl = nl;
        ba.from_input = 1;
        for (int k = 0; k < nl && k < 16; ++k) {
            const Layer &lk = m->layers[k];
            const bool par = lk.pidx >= 0 && lk.spec.type != ALQ_FC;
            ba.L[k] = par ? lk.out_l1 : 1.f;
            ba.B[k] = par ? lk.out_bmax : 0.f;
            ba.src2[k] = lk.spec.skip_src;
        }
        ALQ_TRY(k_fwd_bounds(ctx, m->in_amax, N, m->max_batch, ba, m->bound_all));
    }
    auto take_amax = [&](Igemm2Fuse &fz, int j) {      // the input maxima of consumer j
        if (!cons[j]) return;
        const bool dv = cons[j] == 2;
        fz.in_amax = dv ? m->layers[j - 1].bound_fwd : m->layers[j - 1].amax_fwd;
        if (m->layers[j].spec.skip_src >= 0) {
            const Layer &sl = m->layers[m->layers[j].spec.skip_src];
            fz.in_amax2 = dv ? sl.bound_fwd : sl.amax_fwd;
        }
    };
    for (Layer &l : m->layers) l.signs_ready = false;
    // Sign fields (View::sg): in a Fisher pass a forward launch of the two-slot engine also writes one byte per 4 channels
    // with the signs of its ReLU'd output; the backward launches read those instead of the fp32 activations (1/16 of the bytes)
    auto v4_fwd = [&](const Gemm &g, const View &in, const View &out) {      // gemm_launch's choice for a forward launch with sums
        return !g.pd.ok && g.p4.ok && ((!g_dbg_knobs[4] && !g_dbg_knobs[5]) || in.split != 0 || out.split != 0);
    };
    for (int i = 0; i < nl; ++i) {
        Layer &ly = m->layers[i];
        ALQ_REQUIRE(ly.pidx < 0 || ly.weights_set, ALQ_EINVAL, "weights of parameterised layer %d not set", ly.pidx);
        if (skip_next) { skip_next = false; continue; }
        View in = ly.in;
        if (i == 0) in.p = const_cast<float *>(d_x);
        // channel sums of every spatial layer's output ride on the producing kernel's epilogue when it
        // can; they are the `asum` fields of the layers that consume it
        Igemm2Fuse fz;
        fz.osumA = with_sums ? ly.osum : nullptr;
        const bool head_conv = i + 2 == nl && m->layers[nl - 1].fc_part2 && ly.spec.type == ALQ_CONV;
        const bool light_here = light && (prod[i] || cons[i] || head_conv);
        const Igemm2Fuse *fuse = ((with_sums || light_here) && ly.osum && !g_dbg_knobs[3]) ? &fz : nullptr;
        bool fused = false;
        switch (ly.spec.type) {
            case ALQ_CONV: {
                // first conv + the pool behind it in one kernel (one input channel, 3x3x3 -> 8, 2x2x2 windows)
                Layer *nx = i + 1 < nl ? &m->layers[i + 1] : nullptr;
                const alq_layer_t &sp = ly.spec;
                if (use_dcp(i)) {
                    if (prod[i]) ALQ_HIP(hipMemsetAsync(ly.amax_fwd, 0, (size_t)N * sizeof(unsigned), ctx->stream));
                    // (the sign field of the full-resolution output: what the last conv's backward launch reads as its ReLU mask)
                    const bool sg_here = with_sums && !m->no_signs && !m->no_signs0 && sp.relu && ly.out.sg != nullptr;
                    ALQ_TRY(direct_conv_pool_launch(ctx, ly.fwd[0].pd.d_W, in, ly.out, nx->out, ly.d_bias, sp.relu, nx->argmax,
                                                    with_sums ? ly.osum : nullptr, with_sums ? nx->osum : nullptr, N,
                                                    ly.fwd[0].pd.flops_per_patch, prod[i] ? ly.amax_fwd : nullptr,
                                                    sg_here ? ly.out.sg : nullptr, (sg_here && nx->out.sg) ? nx->out.sg : nullptr));
                    ly.signs_ready = sg_here;
                    nx->signs_ready = sg_here && nx->out.sg != nullptr;      // (the pool's output: sign of the window maximum)
                    if (any_derived) {      // per-patch bounds on every later layer's output from this layer's measured maximum
                        FwdBoundsArgs ba;
                        ba.nl = nl;
                        for (int k = 0; k < nl && k < 16; ++k) {
                            const Layer &lk = m->layers[k];
                            const bool par = lk.pidx >= 0 && lk.spec.type != ALQ_FC;
                            ba.L[k] = par ? lk.out_l1 : 1.f;
                            ba.B[k] = par ? lk.out_bmax : 0.f;
                            ba.src2[k] = lk.spec.skip_src;
                        }
                        ALQ_TRY(k_fwd_bounds(ctx, ly.amax_fwd, N, m->max_batch, ba, m->bound_all));
                    }
                    fused = true;
                    skip_next = true;
                    break;
                }
                if (fuse && nx && i + 2 == nl && nx->fc_part2 && !g_dbg_knobs[4] && !g_dbg_knobs[5]) {
                    // the fc head is this layer's only consumer in a Fisher pass: logits partials + sign bytes from the
                    // epilogue, the tensor itself is not stored
                    fz.fc_W = nx->fc_wv; fz.fc_F = nx->F; fz.fc_part = nx->fc_part2; fz.fc_bits = with_sums ? nx->fc_maskbits : nullptr;
                    take_amax(fz, i);
                    // flip-safe head: only where the launch will contract with the fp16x2 split (input maxima known) and the
                    // sign bits are wanted (Fisher pass); stride-1 conv on one dense tensor or a split concat of two
                    const bool flipfix = with_sums && fz.fc_bits && fz.in_amax && !g_no_f16x2 && !m->no_flipfix && ly.d_W32 && ly.fwd_l1 > 0.f &&
                                         sp.s[0] == 1 && sp.s[1] == 1 && sp.s[2] == 1 && in.c0 == 0 && nx->F % 64 == 0 &&      // (whole 16-byte words of sign bytes per patch)
                                         ((in.split == 0 && in.cs == in.C) || (in.split > 0 && in.cs == in.split && in.C == 2 * in.split));
                    if (flipfix) {
                        if (!m->flip_list) {
                            m->flip_cap = flip_list_len(m->max_batch);      // kernels.hip: per-patch segments of FLIP_PER_BLOCK slots
                            ALQ_TRY(m->dalloc(&m->flip_cnt, (size_t)flip_segments(m->max_batch)));
                            ALQ_TRY(m->dalloc(&m->flip_list, (size_t)m->flip_cap));
                            ALQ_TRY(m->dalloc(&m->flip_overflow, (size_t)1));
                            ALQ_HIP(hipMemsetAsync(m->flip_overflow, 0, sizeof(unsigned), ctx->stream));
                        }
                        fz.flip_cnt = m->flip_cnt; fz.flip_list = m->flip_list; fz.flip_cap = m->flip_cap; fz.flip_l1 = ly.fwd_l1; fz.flip_bias_nonzero = ly.out_bmax > 0.f;
                    }
                    // the plane-sweep engine (c3d.hip) where its geometry applies and both per-patch input maxima are known
                    c3_head = ly.c3f.ok && ly.c3f.d_W && nx->c3_part && fz.in_amax && fz.in_amax2 && !g_no_f16x2 && !m->no_c3d;
                    if (c3_head)
                        ALQ_TRY(c3d_fwd_launch(ctx, ly.c3f, in, ly.d_bias, N, fz.in_amax, fz.in_amax2, nx->fc_wv, nx->c3_part,
                                               with_sums ? nx->c3_asum : nullptr, reinterpret_cast<unsigned char *>(fz.fc_bits),
                                               flipfix ? std::ldexp(ly.fwd_l1, -24) : 0.f));
                    else
                        ALQ_TRY(igemm4_launch(ctx, ly.fwd[0].p4, in, ly.out, ly.d_bias, ly.spec.relu, 0, N, PROF_IGEMM3_FWD, &fz));
                    if (flipfix)
                        ALQ_TRY(k_flip_fix(ctx, m->flip_list, m->flip_cnt, m->flip_cap, N, in.p, in.split ? in.p + in.delta : nullptr,
                                           in.split ? in.split : in.C, in.split ? in.C - in.split : 0, in.D, in.H, in.W, sp.k[0], sp.k[1], sp.k[2],
                                           ly.lo[0], ly.lo[1], ly.lo[2], ly.d_W32, ly.d_bias, sp.cout,
                                           reinterpret_cast<unsigned char *>(nx->fc_maskbits), nx->F, m->flip_overflow));
                    fused = true;
                    fc_head_fused = true;
                    m->last_head_fused = true;
                    m->last_c3 = c3_head;
                    break;
                }
                if (fuse) take_amax(fz, i);
                // the row-sweep engine (d3d.hip) where its geometry applies, both per-patch input bounds are known and nothing but the tensor,
                // its channel sums and its sign field is wanted
                // (round 6: forward-only passes too - the entropy filter over a pool is most of a query round - without sums / sign field)
                if (fuse && ly.d3f.ok && ly.d3f.d_Whi && !m->no_d3d && fz.in_amax && fz.in_amax2 && !prod[i] && !g_no_f16x2 && !g_dbg_knobs[4] && !g_dbg_knobs[5] &&
                    ((with_sums && fz.osumA) || (light && !m->no_light_kernels)) && in.split == 16 && !(drop && drop->on(i))) {
                    const bool sgd = with_sums && !m->no_signs && ly.spec.relu && ly.out.sg;
                    ALQ_TRY(d3d_fwd_launch(ctx, ly.d3f, N, in.p, in.p + in.delta, fz.in_amax, fz.in_amax2, ly.d_bias, ly.spec.relu ? 1 : 0, ly.out.p,
                                           sgd ? ly.out.sg : nullptr, fz.osumA));
                    ly.signs_ready = sgd;
                    fused = true;
                    m->last_d3f = 1;
                    break;
                }
                // enc2 + the pool behind it in one launch (f3d.hip): fp16 pairs under the first layer's measured maximum
                if (fuse && ly.f3f.ok && ly.f3f.d_Whi && !m->no_f3d && cons[i] == 2 && fz.in_amax && !fz.in_amax2 && !prod[i] && !g_no_f16x2 && !g_dbg_knobs[4] && !g_dbg_knobs[5] &&
                    ((with_sums && fz.osumA) || (light && !m->no_light_kernels)) && sp.relu && in.split == 0 && !(drop && (drop->on(i) || drop->on(i + 1))) && nx && nx->spec.type == ALQ_POOL &&
                    nx->spec.k[0] == 2 && nx->spec.k[1] == 2 && nx->spec.k[2] == 2 && nx->lo[0] == 0 && nx->lo[1] == 0 && nx->lo[2] == 0 && nx->out.D == 8 && nx->out.H == 8 &&
                    nx->out.W == 8 && nx->out.C == 16 && nx->out.cs == 16 && nx->out.c0 == 0 && !nx->out.split && nx->argmax && !prod[i + 1]) {
                    const bool sgd = with_sums && !m->no_signs && ly.out.sg;
                    ALQ_TRY(f3d_fwd_launch(ctx, ly.f3f, N, in.p, fz.in_amax, ly.d_bias, ly.out.p, sgd ? ly.out.sg : nullptr, fz.osumA, nx->out.p, nx->argmax,
                                           with_sums ? nx->osum : nullptr));
                    ly.signs_ready = sgd;
                    nx->signs_ready = false;
                    fused = true;
                    skip_next = true;
                    m->last_f3f = 1;
                    break;
                }
                if (fuse && prod[i]) fz.out_amax = m->amax_tiles;
                if (!ly.fwd_co.empty() && !prod[i] && !cons[i] && !g_dbg_knobs[4] && !g_dbg_knobs[5]) {
                    // a wide conv as launches over slices of its output channels (no epilogue fusion: channel sums and ReLU masks the plain way)
                    // (round 6) with per-patch bounds on the input at hand (v3f16: igemm3's forward launches above this layer asked for
                    // them) the slices take the two-slot engine's fp16-pair form too: one scale per patch, three products
                    Igemm2Fuse sz;
                    const bool s16 = v3f16 && i >= 1 && in.split == 0 && ly.spec.skip_src < 0 && !getenv("ALQ_NO_CO_SPLIT_F16");
                    if (s16) sz.in_amax = m->layers[i - 1].bound_fwd;
                    for (size_t j = 0; j < ly.fwd_co.size(); ++j) {
                        View oj = ly.out;
                        oj.c0 = ly.out.c0 + (int)j * ly.fwd_co_w;
                        oj.C = ly.fwd_co_w;
                        bool f1 = false;
                        ALQ_TRY(gemm_launch(ctx, ly.fwd_co[j], in, oj, ly.d_bias + j * ly.fwd_co_w, ly.spec.relu, 0, N, PROF_IGEMM_FWD, s16 ? &sz : nullptr, &f1));
                    }
                    fused = false;
                    ly.signs_ready = false;
                    break;
                }
                const bool sg_here = with_sums && fuse && !m->no_signs && ly.spec.relu && ly.out.sg && v4_fwd(ly.fwd[0], in, ly.out);
                if (sg_here) fz.sign_out = ly.out.sg;
                {
                    // igemm3's fp16-pair instantiation for a forward launch without a two-slot plan (NET-B's conv1 .. conv3): one scale per
                    // patch from the bound on this layer's input (measured maximum of the network input pushed through the L1 norms)
                    Igemm2Fuse hz;
                    const Igemm2Fuse *fu = fuse;
                    if (v3f16 && !ly.fwd[0].p4.ok && !ly.fwd[0].pd.ok && !ly.fwd[0].pfc.ok && ly.fwd[0].p3.ok && ly.fwd[0].p3.d_W16 && ly.fwd[0].p2.a.PT == 1 &&
                        in.split == 0 && ly.spec.skip_src < 0) {
                        if (fuse) hz = *fuse;
                        hz.in_amax = i == 0 ? m->in_amax : m->layers[i - 1].bound_fwd;
                        hz.in_amax2 = nullptr;
                        fu = &hz;
                    }
                    bool f2 = false;
                    ALQ_TRY(gemm_launch(ctx, ly.fwd[0], in, ly.out, ly.d_bias, ly.spec.relu, 0, N, PROF_IGEMM_FWD, fu, &f2));
                    fused = fuse ? f2 : false;
                }
                ly.signs_ready = sg_here;
                if (fuse && prod[i]) ALQ_TRY(k_rowmax_u32(ctx, m->amax_tiles, ly.fwd[0].p4.a.tpg * 4, N, ly.amax_fwd));
                break;
            }
            case ALQ_CONVT: {
                // the row-sweep engine (t3d.hip): bf16 triples like the two-slot launch it replaces, nothing but the tensor, its channel
                // sums and its per-patch maximum to produce (no ReLU behind a conv_transpose of this geometry: no sign field)
                if (ly.t3f.ok && ly.t3f.d_W && !m->no_t3d && !g_dbg_knobs[3] && !g_dbg_knobs[4] && !g_dbg_knobs[5] && !cons[i] && !ly.spec.relu &&
                    !(drop && drop->on(i)) && (!with_sums || ly.osum) && !(ly.t3f.kind == 8 && prod[i])) {
                    const bool want_amax = prod[i] != 0;
                    ALQ_TRY(t3d_fwd_launch(ctx, ly.t3f, in, ly.out, ly.d_bias, N, with_sums ? ly.osum : nullptr, want_amax ? m->amax_tiles : nullptr));
                    if (want_amax) ALQ_TRY(k_rowmax_u32(ctx, m->amax_tiles, 16, N, ly.amax_fwd));
                    ly.signs_ready = false;
                    fused = true;
                    m->last_t3f += 1;
                    break;
                }
                if (ly.fwd_all.ok && !g_dbg_knobs[4] && !g_dbg_knobs[5]) {
                    const bool want_amax = fuse && prod[i];
                    if (fuse) take_amax(fz, i);
                    if (want_amax) fz.out_amax = m->amax_tiles;
                    const bool sg_here = with_sums && fuse && !m->no_signs && ly.spec.relu && ly.out.sg;
                    if (sg_here) fz.sign_out = ly.out.sg;
                    ALQ_TRY(igemm4_launch(ctx, ly.fwd_all, in, ly.out, ly.d_bias, ly.spec.relu, 0, N, PROF_IGEMM3_FWD, fuse));
                    ly.signs_ready = sg_here;
                    if (want_amax)
                        ALQ_TRY(k_rowmax_u32(ctx, m->amax_tiles, ly.fwd_all.a.tpg * (ly.fwd_all.multi ? ly.fwd_all.a.ngr : 1) * 4, N, ly.amax_fwd));
                    fused = fuse != nullptr;
                    break;
                }
                bool all = true;
                for (auto &p : ly.fwd) {
                    bool f1 = false;
                    ALQ_TRY(gemm_launch(ctx, p, in, ly.out, ly.d_bias, ly.spec.relu, 0, N, PROF_IGEMM_FWD, fuse, &f1));
                    all = all && f1;
                }
                fused = all;
                break;
            }
            case ALQ_POOL:
                ALQ_TRY(k_pool_fwd(ctx, in, ly.out, ly.argmax, ly.spec.k, ly.lo, N, with_sums ? ly.osum : nullptr, &fused));
                break;
            case ALQ_FC:
                if (with_sums) {
                    // sum of all inputs of the fc layer, per patch
                    const bool prev_spatial = i > 0 && m->layers[i - 1].osum != nullptr;
                    if (c3_head) ALQ_TRY(k_rowsum_field(ctx, ly.c3_asum, 4, N, ly.asum));
                    else if (prev_spatial) ALQ_TRY(k_rowsum_field(ctx, m->layers[i - 1].osum, m->layers[i - 1].out.vox(), N, ly.asum));
                    else ALQ_TRY(k_chansum(ctx, flat_view(in), ly.asum, N));
                }
                if (ly.dense_fc_small && fc_head_fused && c3_head) {
                    ALQ_TRY(k_fc_small_finish_diff(ctx, ly.c3_part, 4, ly.d_bias, N, ly.out.p));
                } else if (ly.dense_fc_small && fc_head_fused) {
                    ALQ_TRY(k_fc_small_finish_diff(ctx, ly.fc_part2, ly.fc_slices2, ly.d_bias, N, ly.out.p));
                } else if (ly.dense_fc_small) {
                    ALQ_TRY(k_fc_small_fwd(ctx, in.p, ly.F, ly.d_Wp, ly.spec.cout, N, ly.fc_partials, ly.fc_slices,
                                           with_sums ? ly.fc_maskbits : nullptr));
                    ALQ_TRY(k_fc_small_finish(ctx, ly.fc_partials, ly.fc_slices, ly.d_bias, ly.spec.cout,
                                              ly.spec.relu, N, ly.out.p));
                } else {
                    ALQ_TRY(gemm_launch(ctx, ly.fwd[0], flat_view(in), ly.out, ly.d_bias, ly.spec.relu, 0, N,
                                        PROF_IGEMM_FWD));
                }
                break;
        }
        if (drop && drop->on(i)) {
            ALQ_REQUIRE(keep_all && !with_sums, ALQ_EINVAL, "dropout only in passes that keep every activation");
            ALQ_TRY(k_dropout(ctx, ly.spec.type == ALQ_FC ? flat_view(ly.out) : ly.out, N, drop->first_sample, drop->seed, i, drop->keep_prob));
        }
        if (with_sums && ly.osum && !fused) ALQ_TRY(k_chansum(ctx, ly.out, ly.osum, N));
        if (with_sums && i == 0 && ly.pidx == 0 && ly.spec.type != ALQ_FC && in.C > 1)
            ALQ_TRY(k_chansum(ctx, in, ly.asum, N));     // channel sums of the network input
    }
    return ALQ_OK;
}

// While one of these lives, launches through ctx go to the side stream, ordered after everything already on the main
// stream.  Without a side stream (or when the fork fails) they simply stay on the main stream.
struct SideStream {
    alq_ctx *c;
    hipStream_t main;
    bool on = false;
    explicit SideStream(alq_ctx *ctx) : c(ctx), main(ctx->stream) {
        if (!c->side || c->side_off) return;
        if (hipEventRecord(c->ev_fork, main) != hipSuccess || hipStreamWaitEvent(c->side, c->ev_fork, 0) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        c->stream = c->side;
        c->side_used = true;
        on = true;
    }
    ~SideStream() { if (on) c->stream = main; }
    SideStream(const SideStream &) = delete;
    SideStream &operator=(const SideStream &) = delete;
};

// the main stream waits for what the side stream was given since the last join
static int side_join(alq_ctx *c) {
    if (!c->side || !c->side_used) return ALQ_OK;
    c->side_used = false;
    ALQ_HIP(hipEventRecord(c->ev_join, c->side));
    ALQ_HIP(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    return ALQ_OK;
}

static int run_backward_main(alq_model *m, const float *d_x, int N);
static int run_backward(alq_model *m, const float *d_x, int N) {
    const int rc = run_backward_main(m, d_x, N);
    const int rj = side_join(m->ctx);       // also after a failure: nothing may still run beside the caller's next step
    return rc != ALQ_OK ? rc : rj;
}

static int run_backward_main(alq_model *m, const float *d_x, int N) {
    alq_ctx *ctx = m->ctx;
    const int nl = (int)m->layers.size();
    ALQ_REQUIRE(m->nclass == 2, ALQ_EUNSUPPORTED, "Fisher scoring is binary (PW_NNAL.py:766), got %d classes", m->nclass);
    ALQ_TRY(k_fill_unit_cotangent(ctx, m->dlogits, N));
    for (Layer &l : m->layers) { l.delta_ready = false; l.dsum_partial = false; l.dout_bits = nullptr; l.dout_vec = nullptr; l.dout_vec16 = nullptr; l.dout_vec16c = nullptr; l.dout_amax = nullptr; }
    m->last_c3_bwd = false;
    m->last_t3b = 0;
    const bool v4_on = !g_dbg_knobs[4] && !g_dbg_knobs[5];
    {   // bounds on every layer's output cotangent under the unit cotangent (+1, -1): |d out| of the head = 1; a two-class
        // head hands max |W0 - W1| down, a conv / conv_transpose its L1 bound, a pool passes the bound on, a ReLU mask
        // cannot raise it; a tensor with several consumers (skip source) collects the sum.  One fp16x2 scale per LAUNCH
        // follows from it (igemm4: Igemm2Fuse::in_bound) - no per-patch maxima, no extra pass, batch-invariant results.
        for (Layer &l : m->layers) l.dout_bound = 0.f;
        m->layers[nl - 1].dout_bound = 1.f;
        for (int i = nl - 1; i >= 1; --i) {
            const Layer &ly = m->layers[i];
            double inb = ly.dout_bound;
            if (ly.spec.type == ALQ_FC)      // the head: max |W0 - W1|; a hidden fc layer: its L1 bound like a conv's (set with the weights)
                inb = (ly.spec.cout == 2 && ly.fc_wv_amax > 0.f && i == nl - 1) ? (double)ly.fc_wv_amax : (i < nl - 1 ? ly.bwd_l1 * ly.dout_bound : 0.0);
            else if (ly.spec.type != ALQ_POOL) inb = ly.bwd_l1 * ly.dout_bound;
            if (!(inb > 0.0) || ly.dout_bound <= 0.f) inb = 0.0;       // unknown upstream: no bound below either
            auto add = [&](Layer &p) { p.dout_bound = (p.dout_bound < 0.f || inb <= 0.0) ? -1.f : p.dout_bound + (float)(inb * (1.0 + 1e-6)); };
            add(m->layers[i - 1]);
            if (ly.spec.skip_src >= 0) add(m->layers[ly.spec.skip_src]);
        }
    }
    // a pool whose producer is the first parameterised layer: 2x2(x2) windows tiling the input exactly
    auto pool_first_ok = [&](const Layer &pl, const Layer &src) {
        return pl.spec.type == ALQ_POOL && src.pidx == 0 && src.spec.type != ALQ_FC && pl.spec.k[1] == 2 && pl.spec.k[2] == 2 &&
               (pl.spec.k[0] == 1 || pl.spec.k[0] == 2) && pl.lo[0] == 0 && pl.lo[1] == 0 && pl.lo[2] == 0 &&
               src.out.D == pl.out.D * pl.spec.k[0] && src.out.H == pl.out.H * 2 && src.out.W == pl.out.W * 2 &&
               (pl.out.C == 4 || pl.out.C == 8 || pl.out.C == 16) && ((pl.dout.cs | pl.dout.c0 | pl.out.cs | pl.out.c0) & 3) == 0 &&
               src.spec.relu && !g_dbg_knobs[6];
    };
    int e3_conv = -1;      // the conv whose backward ran fused with the pool backward steps around it (e3d.hip), or -1
    m->last_e3b = 0;
    m->last_d3b = 0;
    for (int i = nl - 1; i >= 0; --i) {
        Layer &ly = m->layers[i];
        const bool prev_is_src = (i > 0 && m->layers[i - 1].out_is_skip_src && ly.spec.skip_src < 0);
        // [pool (i)] <- [ReLU conv (i - 1), a skip source whose consumer wrote its cotangent] <- [pool (i - 2)] <- [first conv (i - 3)]: one
        // launch produces both channel-sum fields (e3d.hip); the conv's own box-filter dot product runs as usual when the loop gets there
        if (ly.spec.type == ALQ_POOL && i >= 3 && !m->no_e3d && v4_on && !g_no_f16x2 && !m->no_bound16 && !m->no_signs && !g_dbg_knobs[2] && !g_dbg_knobs[6]) {
            Layer &cv = m->layers[i - 1], &p1 = m->layers[i - 2], &c0 = m->layers[i - 3];
            const bool geo = cv.spec.type == ALQ_CONV && cv.e3b.ok && cv.e3b.d_Whi && cv.spec.relu && cv.signs_ready && cv.out.sg && prev_is_src &&
                             p1.spec.type == ALQ_POOL && pool_first_ok(p1, c0) && c0.dsum_partial && p1.signs_ready && p1.out.sg && p1.spec.k[0] == 2 &&
                             ly.spec.k[0] == 2 && ly.spec.k[1] == 2 && ly.spec.k[2] == 2 && ly.lo[0] == 0 && ly.lo[1] == 0 && ly.lo[2] == 0 &&
                             ly.dout.D == 8 && ly.dout.H == 8 && ly.dout.W == 8 && ly.dout.C == 16 && ly.dout.cs == 16 && ly.dout.c0 == 0 && !ly.dout.split &&
                             cv.dout.cs == 16 && cv.dout.c0 == 0 && !cv.dout.split && cv.out.cs == 16 && cv.out.c0 == 0 && p1.out.cs == 8 && p1.out.c0 == 0 &&
                             p1.out.C == 8 && c0.out.D == 32 && cv.dout_bound > 0.f && cv.dsum && c0.dsum;
            if (geo) {
                ALQ_TRY(e3d_bwd_launch(ctx, cv.e3b, N, cv.dout.p, ly.dout.p, ly.argmax, cv.out.sg, p1.argmax, p1.out.sg, cv.dsum, c0.dsum, cv.dout_bound));
                cv.delta_ready = true;
                c0.delta_ready = true;
                e3_conv = i - 1;
                m->last_e3b = 1;
                continue;
            }
        }
        if (e3_conv >= 0 && i == e3_conv - 1) continue;      // the pool below the fused conv: done
        // Every backward op is the LAST writer of its direct input's cotangent (a skip destination
        // wrote its slice earlier), so it can finish that tensor: ReLU-grad mask + channel sums.
        Layer *prev = i > 0 ? &m->layers[i - 1] : nullptr;
        const bool prev_param = prev && prev->pidx >= 0 && prev->spec.type != ALQ_FC;
        if (ly.spec.type == ALQ_POOL && prev_param && pool_first_ok(ly, *prev) && (prev->dsum_partial || !prev_is_src)) {
            ALQ_TRY(k_pool_bwd_first(ctx, ly.dout, ly.out, ly.argmax, ly.spec.k, prev->out.D, prev->out.H, prev->out.W, N,
                                     prev->dsum, prev->dsum_partial ? 1 : 0, ly.signs_ready ? 1 : 0));
            prev->delta_ready = true;
            continue;
        }
        if (ly.spec.type == ALQ_POOL) {
            bool fused = false;
            ALQ_TRY(k_pool_bwd(ctx, ly.dout, ly.din, ly.argmax, ly.spec.k, ly.lo, N, prev_is_src ? 1 : 0,
                               (prev_param && prev->spec.relu) ? &prev->out : nullptr, prev_param ? prev->dsum : nullptr,
                               &fused, /*store_din=*/!(prev_param && prev->pidx == 0),      // layer 0 only needs the sums
                               /*use_signs=*/(prev_param && prev->signs_ready) ? 1 : 0));
            if (prev_param && fused) prev->delta_ready = true;
            continue;
        }
        // cotangent w.r.t. the pre-activation + its channel sum (unless the producer already did it)
        const bool isfc = ly.spec.type == ALQ_FC;
        if (!ly.delta_ready) {
            View dv = isfc ? flat_view(ly.dout) : ly.dout;
            View av = isfc ? flat_view(ly.out) : ly.out;
            ALQ_TRY(k_mask_chansum(ctx, dv, ly.spec.relu ? &av : nullptr, ly.dsum, N));
        }
        // channel sums of the layer input: the producers' osum fields (two for a concat input)
        const float *as1 = ly.asum, *as2 = nullptr;
        if (!isfc) {
            if (i == 0) as1 = ly.in.C == 1 ? d_x : ly.asum;
            else as1 = m->layers[i - 1].osum;
            if (ly.spec.skip_src >= 0) as2 = m->layers[ly.spec.skip_src].osum;
        }
        double *Sdst = m->Spart + (size_t)ly.pidx * m->max_batch * m->nslab_max;
        {   // reads this layer's sums, writes its own slice of Spart: beside the contraction launched below
            SideStream beside(ctx);
            if (ly.spec.type == ALQ_CONVT) {
                ALQ_TRY(k_boxdot_convT(ctx, ly.dsum, as1, as2, ly.in.D, ly.in.H, ly.in.W, ly.spec.k, ly.spec.s, ly.lo, N, Sdst, m->nslab_max));
            } else if (isfc) {
                const int one[3] = {1, 1, 1}, zero[3] = {0, 0, 0};
                ALQ_TRY(k_boxdot_conv(ctx, ly.dsum, ly.asum, nullptr, 1, 1, 1, one, zero, N, Sdst, m->nslab_max));
            } else {
                ALQ_TRY(k_boxdot_conv(ctx, ly.dsum, as1, as2, ly.out.D, ly.out.H, ly.out.W, ly.spec.k, ly.lo, N, Sdst, m->nslab_max));
            }
        }
        if (ly.pidx == 0) break;   // nothing upstream needs a cotangent
        if (i == e3_conv) continue;      // its backward-data pass ran inside the fused launch
        const int acc = prev_is_src ? 1 : 0;   // the skip destination has already written this slice
        bool fused = false;
        if (isfc) {
            ALQ_REQUIRE(!acc, ALQ_EUNSUPPORTED, "layer %d: fc consumer of a skip source", i);
            const bool bits_ok = ly.dense_fc_small && ly.fc_maskbits && prev_param && prev->out.cs == prev->out.C &&
                                 prev->out.c0 == 0 && v4_on && prev->bwd.p4.ok && prev->bwd.p4.NTW == 1 && !prev->bwd.p4.multi &&
                                 prev->bwd.p4.a.PT == 1 && !prev->dout.split && !prev_is_src &&
                                 // ... and that launch must not accumulate (a skip source right below the conv): the
                                 // accumulating instantiations of the engine are the plain ones
                                 !(i >= 2 && m->layers[i - 2].out_is_skip_src && prev->spec.skip_src < 0);
            if (bits_ok) {
                // every patch has the same head cotangent (the unit cotangent): nothing of the size of the conv's output
                // is written; the conv's backward contraction reads [sign] * wv (wv = W0 - W1, set with the weights)
                {   // the sums feed only the layer's box-filter dot products (same stream, later)
                    SideStream beside(ctx);
                    ALQ_TRY(k_fc_small_dsum_bits(ctx, ly.fc_maskbits, ly.fc_wv, ly.F, N, prev->dsum));
                }
                prev->dout_bits = ly.fc_maskbits;
                prev->dout_vec = ly.fc_wv;
                prev->dout_vec16 = ly.fc_wv16;
                prev->dout_vec16c = ly.fc_wv16c;
                prev->dout_vec_amax = ly.spec.cout == 2 ? ly.fc_wv_amax : 0.f;
                fused = true;
            } else if (ly.dense_fc_small) {
                const bool can = prev_param && prev->out.cs == prev->out.C;
                ALQ_TRY(k_fc_small_bwd(ctx, ly.dout.p, ly.spec.cout, ly.d_Wp, ly.F, N, ly.din.p,
                                       (can && prev->spec.relu) ? prev->out.p : nullptr, can ? prev->dsum : nullptr,
                                       can ? prev->out.C : 0, &fused));
            } else {
                // a wide fc layer's backward GEMM on fp16 pairs under the static bound on its input cotangent (fcgemm.hip, F16)
                ALQ_TRY(gemm_launch(ctx, ly.bwd, ly.dout, flat_view(ly.din), nullptr, 0, 0, N, PROF_IGEMM_BWD, nullptr, nullptr,
                                    (ly.dout_bound > 0.f && !m->no_bound16) ? ly.dout_bound : 0.f));
            }
        } else {
            Igemm2Fuse fz;
            const Igemm2Fuse *fuse = nullptr;
            if (prev_param && !g_dbg_knobs[2]) {
                const int Cs = ly.spec.skip_src >= 0 ? m->layers[ly.spec.skip_src].out.C : 0;   // concat: [src | prev]
                if (prev->spec.relu) {
                    fz.mask = prev->out.p; fz.mask_cs = prev->out.cs; fz.mask_c0 = prev->out.c0; fz.mask_from = Cs;
                    if (prev->signs_ready) fz.mask_bits = prev->out.sg;
                }
                if (Cs > 0) { fz.split = Cs; fz.osumB = prev->dsum; } else { fz.osumA = prev->dsum; }
                // the skip source is the first parameterised layer and sits in front of a pool: nothing needs its
                // cotangent except the channel sums, so mask and sum its columns here and do not store them
                if (Cs > 0 && (v4_on || ly.din.split) && ly.bwd.p4.ok) {
                    Layer &sl = m->layers[ly.spec.skip_src];
                    const bool next_pool = ly.spec.skip_src + 1 < nl && pool_first_ok(m->layers[ly.spec.skip_src + 1], sl);
                    const bool sliced = sl.out.p == prev->out.p && sl.out.cs == prev->out.cs && prev->out.c0 == sl.out.c0 + Cs;
                    const bool splitv = ly.in.split == Cs && sl.out.p == ly.in.p;
                    if (next_pool && (sliced || splitv) && (Cs & 3) == 0) {
                        fz.mask = sl.out.p; fz.mask_cs = sl.out.cs; fz.mask_c0 = sl.out.c0; fz.mask_from = 0;
                        fz.mask_bits = (sl.signs_ready && (!prev->spec.relu || prev->signs_ready)) ? sl.out.sg : nullptr;
                        if (splitv) { fz.mask_split = Cs; fz.mask_delta = ly.in.delta; }
                        if (!prev->spec.relu) fz.mask_to = Cs;
                        fz.osumA = sl.dsum; fz.store_from = Cs;
                        sl.dsum_partial = true;
                    }
                }
                fuse = &fz;
            }
            if (ly.dout_bits) {       // the cotangent of this layer's output exists only as mask bits and one vector
                fz.in_bits = ly.dout_bits; fz.in_vec = ly.dout_vec; fz.in_vec_amax = ly.dout_vec_amax;
                fz.in_vec16 = m->no_presplit ? nullptr : ly.dout_vec16;
                const unsigned short *vec16c = ly.dout_vec16c;
                ly.dout_bits = nullptr; ly.dout_vec = nullptr; ly.dout_vec16 = nullptr; ly.dout_vec16c = nullptr;
                const bool honoured = fuse != nullptr;
                // the plane-sweep engine (c3d.hip): the NET-C pattern - channels 0..7 = the skip source (first conv, ReLU: masked by
                // its sign field and only summed), channels 8..15 = a layer without ReLU (stored and summed)
                const bool c3 = ly.c3b.ok && ly.c3b.d_W && vec16c && !m->no_c3d && !g_no_f16x2 && fuse && !acc && fz.in_vec_amax > 0.f &&
                                fz.split == 8 && fz.store_from == 8 && fz.mask_bits && fz.mask_from == 0 && fz.mask_to == 8 && fz.osumA && fz.osumB &&
                                ly.din.split == 8 && ly.din.cs == 8 && ly.din.C == 16;
                m->last_c3_bwd = c3;
                if (c3) {
                    int ex = 0;
                    (void)std::frexp(fz.in_vec_amax, &ex);
                    ALQ_TRY(c3d_bwd_launch(ctx, ly.c3b, N, reinterpret_cast<const unsigned char *>(fz.in_bits), vec16c, 14 - ex, fz.mask_bits,
                                           ly.din.p + ly.din.delta, fz.osumA, fz.osumB, m->c3_bwd_rows));
                } else {
                    ALQ_TRY(igemm4_launch(ctx, ly.bwd.p4, ly.dout, ly.din, nullptr, 0, acc, N, PROF_IGEMM3_BWD, &fz));
                }
                fused = honoured;
            } else {
                // Hand the per-patch maxima of what this launch stores to the backward launch below it when that one has
                // an fp16x2 variant (two column tiles, or one with the prefetch on the contracting side).  Not from the
                // mask-bit launch above: its staging side is the critical one and the maxima cost more than they gave.
                const bool p4_here = fuse && v4_on && ly.bwd.p4.ok && !ly.bwd.p4.multi && ly.bwd.p4.a.PT == 1 && !g_no_f16x2;
                bool hand = false;
                // (with a static bound for the launch below, nothing needs to be measured here)
                if (p4_here && prev_param && !acc && !(prev->dout_bound > 0.f && !m->no_bound16) && prev->pidx > 0 && prev->bwd.p4.ok && !prev->bwd.p4.multi && prev->bwd.p4.a.PT == 1 &&
                    prev->bwd.p4.d_W16 && !prev->dout.split &&
                    ((prev->bwd.p4.NTW == 1 && prev->bwd.p4.fic) || prev->bwd.p4.NTW == 2)) {
                    const size_t len = (size_t)ly.bwd.p4.a.tpg * 4;
                    if (!m->amax_a) { ALQ_TRY(m->dalloc(&m->amax_a, (size_t)m->max_batch)); ALQ_TRY(m->dalloc(&m->amax_b, (size_t)m->max_batch)); }
                    if (m->amax_tiles_len < len) { ALQ_TRY(m->dalloc(&m->amax_tiles, (size_t)m->max_batch * len)); m->amax_tiles_len = len; }
                    fz.out_amax = m->amax_tiles;
                    fz.amax_from = fz.split > 0 && fz.split < (1 << 29) ? fz.split : 0;       // the slice the layer below reads
                    hand = true;
                }
                // (not for an accumulating launch - a conv right behind a skip source: those run the plain bf16x3 instantiation,
                // igemm4_launch_impl's `no16`, and must not be routed to an fp16x2-only twin plan)
                if (p4_here && !acc && ly.dout_amax) fz.in_amax = ly.dout_amax;
                if (p4_here && !acc && !fz.in_amax && ly.dout_bound > 0.f && !m->no_bound16) fz.in_bound = ly.dout_bound;
                // a launch without an epilogue request (nothing parameterised below it to mask or sum for) still gets its
                // fp16x2 scale: an otherwise empty request carries the bound (no mask, no sums: the plain epilogue runs)
                if (!fuse && !acc && v4_on && ly.bwd.p4.ok && !ly.bwd.p4.multi && ly.bwd.p4.a.PT == 1 && !g_no_f16x2 && !ly.dout_amax &&
                    ly.dout_bound > 0.f && !m->no_bound16 && !g_dbg_knobs[2]) {
                    fz = Igemm2Fuse();
                    fz.in_bound = ly.dout_bound;
                    fuse = &fz;
                }
                // the same bound for a launch that runs on igemm3's fp16-pair instantiation (no two-slot plan)
                if (!ly.bwd.p4.ok && ly.bwd.p3.ok && ly.bwd.p3.d_W16 && !acc && !g_no_f16x2 && ly.dout_bound > 0.f && !m->no_bound16 && !g_dbg_knobs[2] &&
                    !g_dbg_knobs[4] && !g_dbg_knobs[5] && !ly.bwd.pd.ok && !ly.bwd.pfc.ok) {
                    if (!fuse) { fz = Igemm2Fuse(); fuse = &fz; }
                    fz.in_bound = ly.dout_bound;
                }
                const unsigned *mine = ly.dout_amax;
                ly.dout_amax = nullptr;
                // the row-sweep engine (t3d.hip) for the backward-data pass of a stride-2 conv_transpose: fp16 pairs under the static
                // bound like the two-slot launch it replaces; the producer below is a ReLU conv whose sign field masks the result
                const bool t3 = ly.spec.type == ALQ_CONVT && ly.t3b.ok && ly.t3b.d_W && !m->no_t3d && v4_on && !g_no_f16x2 && !acc && fuse && !hand &&
                                fz.in_bound > 0.f && !fz.in_amax && fz.split == 0 && fz.store_from == 0 && fz.osumA && !fz.osumB && fz.mask_from == 0 &&
                                (!fz.mask || fz.mask_bits) && !ly.dout.split && !ly.din.split && prev_param;
                // the plane-sweep kernel of d3d.hip for the backward-data pass of the conv over a split concat (NET-C's dec1): fp16 pairs under the static
                // bound like the two-slot launch it replaces; channels [0, 16) = the skip source's cotangent (stored), [16, 32) = the producer's (stored + summed)
                const bool d3 = ly.spec.type == ALQ_CONV && ly.d3f.ok && ly.d3f.d_Bhi && !m->no_d3d && !m->no_d3b && v4_on && !g_no_f16x2 && !acc && fuse && !hand &&
                                fz.in_bound > 0.f && !fz.in_amax && fz.split == 16 && fz.store_from == 0 && !fz.mask && !fz.osumA && fz.osumB &&
                                !ly.dout.split && ly.dout.cs == 16 && ly.dout.c0 == 0 && ly.din.split == 16 && ly.din.cs == 16 && ly.din.c0 == 0 && ly.din.C == 32;
                if (t3) {
                    ALQ_TRY(t3d_bwd_launch(ctx, ly.t3b, ly.dout, ly.din, N, fz.in_bound, fz.mask ? fz.mask_bits : nullptr, fz.osumA));
                    fused = true;
                    m->last_t3b += 1;
                } else if (d3) {
                    ALQ_TRY(d3d_bwd_launch(ctx, ly.d3f, N, ly.dout.p, fz.in_bound, ly.din.p, ly.din.p + ly.din.delta, fz.osumB));
                    fused = true;
                    m->last_d3b = 1;
                } else
                ALQ_TRY(gemm_launch(ctx, ly.bwd, ly.dout, ly.din, nullptr, 0, acc, N, PROF_IGEMM_BWD, fuse, &fused));
                if (hand) {
                    // two buffers alternate down the chain: this launch may have read the other one
                    unsigned *dst = mine == m->amax_a ? m->amax_b : m->amax_a;
                    ALQ_TRY(k_rowmax_u32(ctx, m->amax_tiles, ly.bwd.p4.a.tpg * 4, N, dst));
                    prev->dout_amax = dst;
                }
            }
        }
        if (prev_param && fused) prev->delta_ready = true;
    }
    return ALQ_OK;
}

// General backward-data pass for a cotangent already in m->dlogits ([N, c], w.r.t. the logits as the forward pass left
// them): every layer's masked pre-activation cotangent ends up in its `dout` view (what the weight gradients need), no
// shortcut of the Fisher pass is taken (no unit cotangent, no sign-byte input, every tensor stored).  The forward pass
// must have kept every activation (run_forward(..., keep_all = true)).
static int run_backward_general(alq_model *m, int N, const DropSpec *drop) {
    alq_ctx *ctx = m->ctx;
    const int nl = (int)m->layers.size();
    for (int i = nl - 1; i >= 0; --i) {
        Layer &ly = m->layers[i];
        const bool isfc = ly.spec.type == ALQ_FC;
        const bool prev_is_src = (i > 0 && m->layers[i - 1].out_is_skip_src && ly.spec.skip_src < 0);
        if (drop && drop->on(i))       // out = act * keep / keep_prob: the cotangent takes the same factor
            ALQ_TRY(k_dropout(ctx, isfc ? flat_view(ly.dout) : ly.dout, N, drop->first_sample, drop->seed, i, drop->keep_prob));
        if (ly.spec.type == ALQ_POOL) {
            if (i == 0) break;
            bool fused = false;
            ALQ_TRY(k_pool_bwd(ctx, ly.dout, ly.din, ly.argmax, ly.spec.k, ly.lo, N, prev_is_src ? 1 : 0, nullptr, nullptr, &fused, 1));
            continue;
        }
        View dv = isfc ? flat_view(ly.dout) : ly.dout;
        View av = isfc ? flat_view(ly.out) : ly.out;
        ALQ_TRY(k_mask_chansum(ctx, dv, ly.spec.relu ? &av : nullptr, ly.dsum, N));      // ReLU-grad mask in place (+ sums, unused)
        if (ly.pidx == 0) break;      // the first parameterised layer: nothing below needs a cotangent
        const int acc = prev_is_src ? 1 : 0;
        if (isfc) {
            ALQ_REQUIRE(!acc, ALQ_EUNSUPPORTED, "layer %d: fc consumer of a skip source", i);
            if (ly.dense_fc_small) {
                bool fused = false;
                ALQ_TRY(k_fc_small_bwd(ctx, ly.dout.p, ly.spec.cout, ly.d_Wp, ly.F, N, ly.din.p, nullptr, nullptr, 0, &fused));
            } else {
                // no bound: ly.dout_bound belongs to the unit cotangent of a Fisher pass (run_backward_main), this
                // cotangent is arbitrary (loss scale, dropout factors) - the bf16-triple path, whatever ran before
                ALQ_TRY(gemm_launch(ctx, ly.bwd, ly.dout, flat_view(ly.din), nullptr, 0, 0, N, PROF_IGEMM_BWD, nullptr, nullptr, 0.f));
            }
        } else {
            ALQ_TRY(gemm_launch(ctx, ly.bwd, ly.dout, ly.din, nullptr, 0, acc, N, PROF_IGEMM_BWD));
        }
    }
    return ALQ_OK;
}

// Weight + bias gradients of every parameterised layer from the tensors a general backward pass left behind, written as
// one flat vector per sample (or one for the batch): [W_0, b_0, W_1, b_1, ...] in the reference's variable order and
// TF layouts (conv [k.., ci, co], conv_transpose [k.., co, ci], fc [out, in] with `in` in flatten order, NN.py:272-318).
static int run_param_grads(alq_model *m, const float *d_x, int N, int sum_n, float *d_out, long long P) {
    alq_ctx *ctx = m->ctx;
    long long need = 0;
    for (Layer &ly : m->layers) {
        if (ly.pidx < 0 || ly.spec.type == ALQ_FC) continue;
        const bool convt = ly.spec.type == ALQ_CONVT;
        need = std::max(need, wgrad_partial_floats(convt ? ly.in : ly.dout, convt ? ly.dout : ly.in, ly.spec.k) * N);
    }
    if ((size_t)need > m->wg_partial_len) {
        ALQ_TRY(m->dalloc(&m->wg_partial, (size_t)need));
        m->wg_partial_len = (size_t)need;
    }
    long long off = 0;
    const long long stride = sum_n ? 0 : P;
    for (size_t i = 0; i < m->layers.size(); ++i) {
        Layer &ly = m->layers[i];
        if (ly.pidx < 0) continue;
        View in = ly.in;
        if (i == 0) in.p = const_cast<float *>(d_x);
        const int one[3] = {1, 1, 1};
        if (ly.spec.type == ALQ_CONV) {
            ALQ_TRY(k_wgrad(ctx, ly.dout, in, ly.spec.k, one, ly.lo, N, sum_n, m->wg_partial, d_out + off, stride));
            ALQ_TRY(k_bgrad(ctx, ly.dout, N, sum_n, d_out + off + ly.w_elems, stride));
        } else if (ly.spec.type == ALQ_CONVT) {
            ALQ_TRY(k_wgrad(ctx, in, ly.dout, ly.spec.k, ly.spec.s, ly.lo, N, sum_n, m->wg_partial, d_out + off, stride));
            ALQ_TRY(k_bgrad(ctx, ly.dout, N, sum_n, d_out + off + ly.w_elems, stride));
        } else {
            ALQ_TRY(k_fc_wgrad(ctx, ly.dout.p, in, ly.spec.cout, N, sum_n, d_out + off, stride));
            ALQ_TRY(k_bgrad(ctx, flat_view(ly.dout), N, sum_n, d_out + off + ly.w_elems, stride));
        }
        off += ly.w_elems + ly.b_elems;
    }
    ALQ_REQUIRE(off == P, ALQ_EINVAL, "parameter count mismatch");
    return ALQ_OK;
}

static int make_drop(const alq_model *m, float keep_prob, uint64_t seed, int64_t first_sample, const int32_t *h_layers, int n_layers,
                     DropSpec *d) {
    ALQ_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, ALQ_EINVAL, "keep_prob %g outside (0, 1]", (double)keep_prob);
    ALQ_REQUIRE(n_layers == 0 || h_layers, ALQ_EINVAL, "dropout layer list missing");
    d->keep_prob = keep_prob; d->seed = seed; d->first_sample = first_sample; d->layers = 0;
    for (int i = 0; i < n_layers; ++i) {
        ALQ_REQUIRE(h_layers[i] >= 0 && h_layers[i] < (int)m->layers.size() && h_layers[i] < 64, ALQ_EINVAL, "dropout layer %d", h_layers[i]);
        d->layers |= 1ull << h_layers[i];
    }
    return ALQ_OK;
}

// =========================================================================================== C ABI
extern "C" {

int64_t alq_model_num_params(const alq_model *m) {
    if (!m) return ALQ_EINVAL;
    int64_t p = 0;
    for (const Layer &ly : m->layers)
        if (ly.pidx >= 0) p += ly.w_elems + ly.b_elems;
    return p;
}

int alq_forward_dropout(alq_model *m, const float *d_x, int N, float keep_prob, uint64_t seed, int64_t first_sample,
                        const int32_t *h_drop_layers, int n_drop_layers, float *d_post, int64_t *d_pred) {
    ALQ_REQUIRE(m && d_x, ALQ_EINVAL, "alq_forward_dropout: null argument");
    ALQ_REQUIRE(N >= 0 && N <= m->max_batch, ALQ_EINVAL, "alq_forward_dropout: N=%d exceeds max_batch=%d", N, m->max_batch);
    if (N == 0) return ALQ_OK;
    ALQ_HIP(hipSetDevice(m->ctx->device));
    DropSpec ds;
    ALQ_TRY(make_drop(m, keep_prob, seed, first_sample, h_drop_layers, n_drop_layers, &ds));
    m->last_call_fisher = false;
    apply_knobs(m);
    ALQ_TRY(run_forward(m, d_x, N, false, /*keep_all=*/true, &ds));
    ALQ_TRY(k_softmax(m->ctx, m->logits, m->nclass, N, d_post ? d_post : m->post, d_pred));
    return ALQ_OK;
}

int alq_param_grads(alq_model *m, const float *d_x, int N, int mode, int cls, const int32_t *d_labels, float loss_scale,
                    float keep_prob, uint64_t seed, int64_t first_sample, const int32_t *h_drop_layers, int n_drop_layers,
                    int per_sample, float *d_grads, float *d_post, double *d_loss) {
    ALQ_REQUIRE(m && d_x && d_grads, ALQ_EINVAL, "alq_param_grads: null argument");
    ALQ_REQUIRE(N >= 1 && N <= m->max_batch, ALQ_EINVAL, "alq_param_grads: N=%d outside [1, max_batch=%d]", N, m->max_batch);
    ALQ_REQUIRE(mode == 0 || (mode == 1 && d_labels), ALQ_EINVAL, "alq_param_grads: mode %d / labels", mode);
    ALQ_REQUIRE(mode != 0 || (cls >= 0 && cls < m->nclass), ALQ_EINVAL, "alq_param_grads: class %d", cls);
    ALQ_HIP(hipSetDevice(m->ctx->device));
    DropSpec ds;
    ALQ_TRY(make_drop(m, keep_prob, seed, first_sample, h_drop_layers, n_drop_layers, &ds));
    m->last_call_fisher = false;
    apply_knobs(m);
    ALQ_TRY(run_forward(m, d_x, N, false, /*keep_all=*/true, &ds));
    float *post = d_post ? d_post : m->post;
    ALQ_TRY(k_softmax(m->ctx, m->logits, m->nclass, N, post, nullptr));
    if (d_loss && mode == 1) ALQ_TRY(k_ce_loss(m->ctx, post, m->nclass, N, d_labels, d_loss));
    ALQ_TRY(k_logit_cotangent(m->ctx, post, m->nclass, N, mode, cls, d_labels, loss_scale, m->dlogits));
    ALQ_TRY(run_backward_general(m, N, &ds));
    return run_param_grads(m, d_x, N, per_sample ? 0 : 1, d_grads, alq_model_num_params(m));
}

int alq_sgd_step(alq_ctx *ctx, float *d_theta, const float *d_grad, int64_t n, float lr) {
    ALQ_REQUIRE(ctx && (n == 0 || (d_theta && d_grad)) && n >= 0, ALQ_EINVAL, "alq_sgd_step: bad argument");
    ALQ_HIP(hipSetDevice(ctx->device));
    return n ? k_sgd(ctx, d_theta, d_grad, n, lr) : ALQ_OK;
}

int alq_adam_step(alq_ctx *ctx, float *d_theta, const float *d_grad, float *d_m, float *d_v, int64_t n, float lr, float beta1,
                  float beta2, float eps, int64_t t) {
    ALQ_REQUIRE(ctx && (n == 0 || (d_theta && d_grad && d_m && d_v)) && n >= 0 && t >= 1, ALQ_EINVAL, "alq_adam_step: bad argument");
    ALQ_HIP(hipSetDevice(ctx->device));
    const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)t)) / (1.0 - std::pow((double)beta1, (double)t));
    return n ? k_adam(ctx, d_theta, d_grad, d_m, d_v, n, (float)lr_t, beta1, beta2, eps) : ALQ_OK;
}

int alq_sq_accum(alq_ctx *ctx, const float *d_grads, int64_t per_sample_len, int N, double *d_acc) {
    ALQ_REQUIRE(ctx && d_grads && d_acc && per_sample_len >= 1 && N >= 1, ALQ_EINVAL, "alq_sq_accum: bad argument");
    ALQ_HIP(hipSetDevice(ctx->device));
    return k_sq_accum(ctx, d_grads, per_sample_len, N, d_acc);
}

int alq_shrink_sum(alq_ctx *ctx, const float *d_grads, int N, int64_t P, const int64_t *h_layer_elems, int L, double *d_out) {
    ALQ_REQUIRE(ctx && d_grads && h_layer_elems && d_out && N >= 1 && L >= 1 && L <= 64, ALQ_EINVAL, "alq_shrink_sum: bad argument");
    long long off[65];
    off[0] = 0;
    for (int t = 0; t < L; ++t) {
        ALQ_REQUIRE(h_layer_elems[t] >= 1, ALQ_EINVAL, "alq_shrink_sum: layer %d has %lld elements", t, (long long)h_layer_elems[t]);
        off[t + 1] = off[t] + h_layer_elems[t];
    }
    ALQ_REQUIRE(off[L] == P, ALQ_EINVAL, "alq_shrink_sum: the layers hold %lld elements, a gradient row %lld", off[L], (long long)P);
    ALQ_HIP(hipSetDevice(ctx->device));
    return k_shrink_sum(ctx, d_grads, N, P, off, L, d_out);
}

int alq_fisher_classes(alq_ctx *ctx, const double *d_g, const double *d_w, const double *d_diag, int N, int c, int L, double *d_A) {
    ALQ_REQUIRE(ctx && d_g && d_w && d_diag && d_A && N >= 1 && c >= 1 && L >= 1, ALQ_EINVAL, "alq_fisher_classes: bad argument");
    ALQ_HIP(hipSetDevice(ctx->device));
    return k_fisher_classes(ctx, d_g, d_w, d_diag, N, c, L, d_A);
}

const char *alq_last_error(void) { return g_err; }
int alq_version(void) { return 1; }

int alq_ctx_create(int device, void *stream, alq_ctx **out) {
    ALQ_REQUIRE(out != nullptr, ALQ_EINVAL, "alq_ctx_create: null out");
    int count = 0;
    ALQ_HIP(hipGetDeviceCount(&count));
    ALQ_REQUIRE(device >= 0 && device < count, ALQ_EINVAL, "device %d not present (%d visible)", device, count);
    ALQ_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    ALQ_HIP(hipGetDeviceProperties(&prop, device));
    ALQ_REQUIRE(std::strncmp(prop.gcnArchName, "gfx950", 6) == 0, ALQ_EUNSUPPORTED,
                "libalq is built for gfx950 (MI355X) only, device %d is %s", device, prop.gcnArchName);
    alq_ctx *c = new alq_ctx();
    c->device = device;
    if (prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount;
    c->stream = reinterpret_cast<hipStream_t>(stream);
    if (hipMalloc(&c->param_block, ALQ_PARAM_BLOCK_BYTES) != hipSuccess) {
        delete c;
        set_error("alq_ctx_create: hipMalloc failed");
        return ALQ_ENOMEM;
    }
    if (!std::getenv("ALQ_NO_SIDE_STREAM")) {      // diagnostic: everything on one stream
        if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (c->side) (void)hipStreamDestroy(c->side);
            if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
            c->side = nullptr; c->ev_fork = nullptr; c->ev_join = nullptr;
        }
    }
    *out = c;
    return ALQ_OK;
}

int alq_ctx_destroy(alq_ctx *ctx) {
    if (!ctx) return ALQ_OK;
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->side) {
        (void)hipStreamSynchronize(ctx->side);
        (void)hipStreamDestroy(ctx->side);
        (void)hipEventDestroy(ctx->ev_fork);
        (void)hipEventDestroy(ctx->ev_join);
    }
    (void)alq_comm_destroy(ctx);
    for (int c = 0; c < PROF_NUM; ++c) {
        for (auto &pr : ctx->prof[c].pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        for (auto e : ctx->prof[c].pool) (void)hipEventDestroy(e);
    }
    (void)hipFree(ctx->param_block);
    delete ctx;
    return ALQ_OK;
}

int alq_ctx_use_side_stream(alq_ctx *ctx, int on) {
    ALQ_REQUIRE(ctx != nullptr, ALQ_EINVAL, "null ctx");
    ctx->side_off = on == 0;
    return ALQ_OK;
}

int alq_ctx_set_stream(alq_ctx *ctx, void *stream) {
    ALQ_REQUIRE(ctx != nullptr, ALQ_EINVAL, "null ctx");
    hipStream_t next = reinterpret_cast<hipStream_t>(stream);
    if (next != ctx->stream) {
        // the library's hidden state (activation workspaces, partial sums, side-stream work joined into the OLD stream) is
        // ordered on the old stream only: the new stream waits for everything enqueued there so far
        ALQ_HIP(hipSetDevice(ctx->device));
        hipEvent_t ev;
        ALQ_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e1 = hipEventRecord(ev, ctx->stream);
        hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(next, ev, 0) : e1;
        (void)hipEventDestroy(ev);
        ALQ_REQUIRE(e2 == hipSuccess, ALQ_EHIP, "alq_ctx_set_stream: %s", hipGetErrorString(e2));
        ctx->stream = next;
    }
    return ALQ_OK;
}

int alq_ctx_synchronize(alq_ctx *ctx) {
    ALQ_REQUIRE(ctx != nullptr, ALQ_EINVAL, "null ctx");
    ALQ_HIP(hipStreamSynchronize(ctx->stream));
    return ALQ_OK;
}

int alq_model_create(alq_ctx *ctx, const alq_layer_t *layers, int n_layers, const int32_t in_dims[4],
                     int max_batch, alq_model **out) {
    ALQ_REQUIRE(ctx && layers && in_dims && out, ALQ_EINVAL, "alq_model_create: null argument");
    ALQ_REQUIRE(n_layers >= 1 && n_layers <= 64 && max_batch >= 1, ALQ_EINVAL, "alq_model_create: bad sizes");
    ALQ_HIP(hipSetDevice(ctx->device));
    alq_model *m = new alq_model();
    m->ctx = ctx;
    m->max_batch = max_batch;
    {
        const char *e = getenv("ALQ_DISABLE_V2");   // diagnostics: force the general GEMM kernel (plan-time only)
        g_use_v2 = !(e && e[0] == '1');
        m->no_f16x2 = getenv("ALQ_NO_F16X2") != nullptr;
        m->no_xcd_order = getenv("ALQ_NO_XCD_ORDER") != nullptr;
        m->no_fixed = getenv("ALQ_NO_FIXED") != nullptr;
        m->no_bound16 = getenv("ALQ_NO_BOUND16") != nullptr;
        m->no_flipfix = getenv("ALQ_NO_FLIPFIX") != nullptr;
        m->no_presplit = getenv("ALQ_NO_PRESPLIT") != nullptr;
        m->no_c3d = getenv("ALQ_NO_C3D") != nullptr;
        // 7 (default): the 27 taps packed into 7 k-steps (c3d_bwd7_kernel); 8: the 9-k-step kernel of round 4; 4: its half-patch form
        m->no_light_kernels = getenv("ALQ_NO_LIGHT_KERNELS") != nullptr;
        { const char *e = getenv("ALQ_C3D_BWD_ROWS"); m->c3_bwd_rows = (e && atoi(e) == 4) ? 4 : ((e && atoi(e) == 8) ? 8 : 7); }
        m->no_signs = getenv("ALQ_NO_SIGNS") != nullptr;
        m->no_signs0 = getenv("ALQ_NO_SIGNS0") != nullptr;
        if (const char *f = getenv("ALQ_F16_FWD_MASK")) m->f16_fwd_mask = atoi(f);
        m->no_t3d = getenv("ALQ_NO_T3D") ? 1 : 0;
        m->no_e3d = getenv("ALQ_NO_E3D") ? 1 : 0;
        m->no_d3d = getenv("ALQ_NO_D3D") ? 1 : 0;
        m->no_f3d = getenv("ALQ_NO_F3D") ? 1 : 0;
        m->no_d3b = getenv("ALQ_NO_D3D_BWD") ? 1 : 0;
        {   // default since round 5: on.  ALQ_NO_F16_DERIVED=1 (or ALQ_F16_DERIVED=0) keeps that launch on bf16 triples (A/B)
            const char *e = getenv("ALQ_F16_DERIVED"), *n = getenv("ALQ_NO_F16_DERIVED");
            m->no_f16_derived = ((e && atoi(e) == 0) || (n && atoi(n) == 1)) ? 1 : 0;
        }
        static const char *names[8] = {"ALQ_DEBUG_REPEAT", "ALQ_DEBUG_FLAGS", "ALQ_NO_BWD_FUSE", "ALQ_NO_FWD_FUSE", "ALQ_NO_V3", "ALQ_NO_V4",
                                       "ALQ_NO_POOL_FIRST", "ALQ_NO_CONV_POOL"};
        for (int k = 0; k < 8; ++k) {
            const char *v = getenv(names[k]);
            if (v) m->knobs[k] = atoi(v);
        }
    }
    for (int i = 0; i < 4; ++i) m->in_dims[i] = in_dims[i];
    m->epp = (int64_t)in_dims[0] * in_dims[1] * in_dims[2] * in_dims[3];
    const int rc = build_model(m, layers, n_layers);
    if (rc != ALQ_OK) {
        alq_model_destroy(m);
        return rc;
    }
    if (const char *t = getenv("ALQ_G4_TUNE")) {
        // diagnostic: per-launch balance of the two-slot engine, "f<layer>:fic=<0|1>,epi=<0..2>;b<layer>:..." (f = forward
        // launch of the layer, b = its backward-data launch); every setting computes the same bits
        std::string all(t);
        size_t pos = 0;
        while (pos < all.size()) {
            size_t end = all.find(';', pos);
            if (end == std::string::npos) end = all.size();
            const std::string item = all.substr(pos, end - pos);
            pos = end + 1;
            if (item.size() < 3 || (item[0] != 'f' && item[0] != 'b')) continue;
            const size_t colon = item.find(':');
            if (colon == std::string::npos) continue;
            const int li = atoi(item.substr(1, colon - 1).c_str());
            if (li < 0 || li >= (int)m->layers.size()) continue;
            Layer &ly = m->layers[li];
            Igemm4Plan *pl = item[0] == 'b' ? &ly.bwd.p4 : (ly.spec.type == ALQ_CONVT ? &ly.fwd_all : (ly.fwd.empty() ? nullptr : &ly.fwd[0].p4));
            if (!pl) continue;
            const size_t fi = item.find("fic="), ei = item.find("epi=");
            if (fi != std::string::npos) pl->tune_fic = atoi(item.c_str() + fi + 4);
            if (ei != std::string::npos) pl->tune_epi = atoi(item.c_str() + ei + 4);
        }
    }
    {   // the conv in front of [conv_transpose without ReLU -> conv under a fused two-class head]: forward launch on the fp16x2 split
        // with derived input bounds (run_forward)
        const int nl = (int)m->layers.size();
        if (nl >= 4 && nl <= 16 && m->layers[nl - 1].fc_part2 && m->layers[nl - 2].spec.type == ALQ_CONV &&
            m->layers[nl - 3].spec.type == ALQ_CONVT && !m->layers[nl - 3].spec.relu && m->layers[nl - 4].spec.type == ALQ_CONV &&
            m->layers[nl - 4].spec.relu)
            m->f16_fwd_derived = 1 << (nl - 4);
        for (int i = 1; i + 1 < nl; ++i)      // enc2 of NET-C: its fused kernel (f3d.hip) contracts fp16 pairs under the first layer's measured maximum
            if (m->layers[i].f3f.ok && m->f16_fwd_derived) m->f16_fwd_derived |= 1 << i;
        if (const char *e = getenv("ALQ_F16_DERIVED_MASK")) m->f16_fwd_derived = atoi(e);      // (study: other forward launches on fp16 pairs, by layer bit)
    }
    *out = m;
    return ALQ_OK;
}

int alq_model_destroy(alq_model *m) {
    if (!m) return ALQ_OK;
    (void)hipStreamSynchronize(m->ctx->stream);
    for (void *p : m->allocs) (void)hipFree(p);
    delete m;
    return ALQ_OK;
}

int alq_model_num_param_layers(const alq_model *m) { return m ? m->L : ALQ_EINVAL; }
int alq_model_max_batch(const alq_model *m) { return m ? m->max_batch : ALQ_EINVAL; }

int alq_model_param_sizes(const alq_model *m, int t, int64_t *w_elems, int64_t *b_elems) {
    ALQ_REQUIRE(m != nullptr, ALQ_EINVAL, "null model");
    for (const Layer &ly : m->layers)
        if (ly.pidx == t) {
            if (w_elems) *w_elems = ly.w_elems;
            if (b_elems) *b_elems = ly.b_elems;
            return ALQ_OK;
        }
    set_error("no parameterised layer %d", t);
    return ALQ_EINVAL;
}

int alq_model_layer_out_elems(const alq_model *m, int layer_idx, int64_t *elems) {
    ALQ_REQUIRE(m && elems && layer_idx >= 0 && layer_idx < (int)m->layers.size(), ALQ_EINVAL, "bad layer index");
    *elems = m->layers[layer_idx].out.elems();
    return ALQ_OK;
}

int alq_model_set_weights(alq_model *m, int t, const float *W, const float *b) {
    ALQ_REQUIRE(m && W && b, ALQ_EINVAL, "alq_model_set_weights: null argument");
    ALQ_HIP(hipSetDevice(m->ctx->device));
    Layer *lyp = nullptr;
    for (Layer &l : m->layers)
        if (l.pidx == t) lyp = &l;
    ALQ_REQUIRE(lyp != nullptr, ALQ_EINVAL, "no parameterised layer %d", t);
    Layer &ly = *lyp;
    const alq_layer_t &sp = ly.spec;
    // Which kernels get their weights packed below depends on whether the matrix cores keep fp16 subnormals: the answer is a
    // property of the device, probed once per context.  A probe that could not RUN must not select engines (it used to read as
    // "flushes subnormals" for this call only: the plans of one layer then differed from the others' for good): retry, then fail.
    for (int tries = 0; tries < 3 && m->ctx->f16_subnormal_mfma < 0; ++tries) (void)c3d_subnormals_ok(m->ctx);
    ALQ_REQUIRE(m->ctx->f16_subnormal_mfma >= 0, ALQ_EHIP, "alq_model_set_weights: the fp16-subnormal probe of the matrix cores could not run (device error)");
    ALQ_HIP(hipMemcpyAsync(ly.d_bias, b, ly.b_elems * sizeof(float), hipMemcpyHostToDevice, m->ctx->stream));
    const int Ci = ly.in.C, Co = sp.cout;
    const int ntaps = sp.k[0] * sp.k[1] * sp.k[2];
    if (sp.type == ALQ_CONV || sp.type == ALQ_CONVT) {
        // conv W[tap][ci][co], conv_transpose W[tap][co][ci]: L1 norm of everything that multiplies into input channel ci
        double best = 0;
        for (int ci = 0; ci < Ci; ++ci) {
            double s = 0;
            for (int tp = 0; tp < ntaps; ++tp)
                for (int co = 0; co < Co; ++co)
                    s += std::fabs((double)(sp.type == ALQ_CONV ? W[((size_t)tp * Ci + ci) * Co + co] : W[((size_t)tp * Co + co) * Ci + ci]));
            best = std::max(best, s);
        }
        ly.bwd_l1 = best;
        // |out[.., co]| <= (sum over everything that multiplies into channel co) * max |in| + |b[co]|: for a conv_transpose all taps
        // are counted (an output point sees a subset of them: the bound is only looser)
        double ob = 0, bm = 0;
        for (int co = 0; co < Co; ++co) {
            double t = 0;
            for (int tp = 0; tp < ntaps; ++tp)
                for (int ci = 0; ci < Ci; ++ci)
                    t += std::fabs((double)(sp.type == ALQ_CONV ? W[((size_t)tp * Ci + ci) * Co + co] : W[((size_t)tp * Co + co) * Ci + ci]));
            ob = std::max(ob, t);
            bm = std::max(bm, std::fabs((double)b[co]));
        }
        ly.out_l1 = (float)(ob * (1.0 + 1e-6));
        ly.out_bmax = (float)(bm * (1.0 + 1e-6));
    }
    if (sp.type == ALQ_CONV) {
        double best = 0;
        for (int co = 0; co < Co; ++co) {
            double s = 0;
            for (int tp = 0; tp < ntaps; ++tp)
                for (int ci = 0; ci < Ci; ++ci) s += std::fabs((double)W[((size_t)tp * Ci + ci) * Co + co]);
            best = std::max(best, s);
        }
        ly.fwd_l1 = (float)(best * (1.0 + 1e-6));
        const int li = (int)(lyp - &m->layers[0]);
        if (li + 2 == (int)m->layers.size() && m->layers[li + 1].fc_part2) {      // the conv under a fused two-class head
            if (!ly.d_W32) ALQ_TRY(m->dalloc(&ly.d_W32, (size_t)ly.w_elems));
            ALQ_HIP(hipMemcpyAsync(ly.d_W32, W, ly.w_elems * sizeof(float), hipMemcpyHostToDevice, m->ctx->stream));
        }
    }
    if (sp.type == ALQ_CONV) {
        // TF [tap][ci][co] is already the fwd B matrix [(tap, ci)][co]
        std::vector<float> B(W, W + ly.w_elems);
        ALQ_TRY(gemm_set(m, &ly.fwd[0], B));
        for (size_t j = 0; j < ly.fwd_co.size(); ++j) {      // the output-channel slices of a wide conv: columns [j w, (j + 1) w) of B
            const int w = ly.fwd_co_w, K = ntaps * Ci;
            std::vector<float> Bj((size_t)K * w);
            for (int k = 0; k < K; ++k)
                for (int c = 0; c < w; ++c) Bj[(size_t)k * w + c] = B[(size_t)k * Co + j * w + c];
            ALQ_TRY(gemm_set(m, &ly.fwd_co[j], Bj));
        }
        if (ly.c3f.ok) {
            // one accumulator (pieces at their true scale) only where the matrix cores honour fp16 subnormals
            if (ly.c3f.oneacc && !c3d_subnormals_ok(m->ctx)) ly.c3f.oneacc = 0;
            c3d_fwd_pack(&ly.c3f, B);
            unsigned short *dw = reinterpret_cast<unsigned short *>(ly.c3f.d_W);
            if (!dw) ALQ_TRY(m->dalloc(&dw, ly.c3f.h_W.size()));
            ly.c3f.d_W = dw;
            ALQ_HIP(hipMemcpyAsync(dw, ly.c3f.h_W.data(), ly.c3f.h_W.size() * sizeof(unsigned short), hipMemcpyHostToDevice, m->ctx->stream));
            ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
        }
        if (ly.c3b.ok && ly.has_bwd && c3d_subnormals_ok(m->ctx)) {      // (the backward kernel exists in the one-accumulator form only)
            c3d_bwd_pack(&ly.c3b, B);
            unsigned short *dw = reinterpret_cast<unsigned short *>(ly.c3b.d_W);
            if (!dw) ALQ_TRY(m->dalloc(&dw, ly.c3b.h_W.size()));
            ly.c3b.d_W = dw;
            ALQ_HIP(hipMemcpyAsync(dw, ly.c3b.h_W.data(), ly.c3b.h_W.size() * sizeof(unsigned short), hipMemcpyHostToDevice, m->ctx->stream));
            c3d_bwd7_pack(&ly.c3b, B);          // the 7-k-step form of the same weights (c3d_bwd7_kernel)
            unsigned short *dw7 = reinterpret_cast<unsigned short *>(ly.c3b.d_W7);
            if (!dw7) ALQ_TRY(m->dalloc(&dw7, ly.c3b.h_W7.size()));
            ly.c3b.d_W7 = dw7;
            ALQ_HIP(hipMemcpyAsync(dw7, ly.c3b.h_W7.data(), ly.c3b.h_W7.size() * sizeof(unsigned short), hipMemcpyHostToDevice, m->ctx->stream));
            ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
        }
        if (ly.e3b.ok && ly.has_bwd && c3d_subnormals_ok(m->ctx)) {      // (fp16 pairs at their true scale: the one-accumulator form)
            e3d_pack(&ly.e3b, W);
            unsigned short *dh = reinterpret_cast<unsigned short *>(ly.e3b.d_Whi), *dl = reinterpret_cast<unsigned short *>(ly.e3b.d_Wlo);
            if (!dh) { ALQ_TRY(m->dalloc(&dh, ly.e3b.h_Whi.size())); ALQ_TRY(m->dalloc(&dl, ly.e3b.h_Wlo.size())); }
            ly.e3b.d_Whi = dh; ly.e3b.d_Wlo = dl;
            ALQ_HIP(hipMemcpyAsync(dh, ly.e3b.h_Whi.data(), ly.e3b.h_Whi.size() * sizeof(unsigned short), hipMemcpyHostToDevice, m->ctx->stream));
            ALQ_HIP(hipMemcpyAsync(dl, ly.e3b.h_Wlo.data(), ly.e3b.h_Wlo.size() * sizeof(unsigned short), hipMemcpyHostToDevice, m->ctx->stream));
            ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
        }
        if (ly.f3f.ok && !c3d_subnormals_ok(m->ctx)) ly.f3f.ok = false;      // (one-accumulator fp16 pairs)
        if (ly.f3f.ok) {
            f3d_pack(&ly.f3f, W);
            unsigned short *dh = reinterpret_cast<unsigned short *>(ly.f3f.d_Whi), *dl = reinterpret_cast<unsigned short *>(ly.f3f.d_Wlo);
            if (!dh) { ALQ_TRY(m->dalloc(&dh, ly.f3f.h_Whi.size())); ALQ_TRY(m->dalloc(&dl, ly.f3f.h_Wlo.size())); }
            ly.f3f.d_Whi = dh; ly.f3f.d_Wlo = dl;
            ALQ_HIP(hipMemcpyAsync(dh, ly.f3f.h_Whi.data(), ly.f3f.h_Whi.size() * sizeof(unsigned short), hipMemcpyHostToDevice, m->ctx->stream));
            ALQ_HIP(hipMemcpyAsync(dl, ly.f3f.h_Wlo.data(), ly.f3f.h_Wlo.size() * sizeof(unsigned short), hipMemcpyHostToDevice, m->ctx->stream));
            ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
        }
        if (ly.d3f.ok) {
            d3d_pack(&ly.d3f, W);
            unsigned short *dh = reinterpret_cast<unsigned short *>(ly.d3f.d_Whi), *dl = reinterpret_cast<unsigned short *>(ly.d3f.d_Wlo);
            if (!dh) { ALQ_TRY(m->dalloc(&dh, ly.d3f.h_Whi.size())); ALQ_TRY(m->dalloc(&dl, ly.d3f.h_Wlo.size())); }
            ly.d3f.d_Whi = dh; ly.d3f.d_Wlo = dl;
            ALQ_HIP(hipMemcpyAsync(dh, ly.d3f.h_Whi.data(), ly.d3f.h_Whi.size() * sizeof(unsigned short), hipMemcpyHostToDevice, m->ctx->stream));
            ALQ_HIP(hipMemcpyAsync(dl, ly.d3f.h_Wlo.data(), ly.d3f.h_Wlo.size() * sizeof(unsigned short), hipMemcpyHostToDevice, m->ctx->stream));
            if (ly.has_bwd) {
                d3d_bwd_pack(&ly.d3f, W);
                unsigned short *bh = reinterpret_cast<unsigned short *>(ly.d3f.d_Bhi), *bl = reinterpret_cast<unsigned short *>(ly.d3f.d_Blo);
                if (!bh) { ALQ_TRY(m->dalloc(&bh, ly.d3f.h_Bhi.size())); ALQ_TRY(m->dalloc(&bl, ly.d3f.h_Blo.size())); }
                ly.d3f.d_Bhi = bh; ly.d3f.d_Blo = bl;
                ALQ_HIP(hipMemcpyAsync(bh, ly.d3f.h_Bhi.data(), ly.d3f.h_Bhi.size() * sizeof(unsigned short), hipMemcpyHostToDevice, m->ctx->stream));
                ALQ_HIP(hipMemcpyAsync(bl, ly.d3f.h_Blo.data(), ly.d3f.h_Blo.size() * sizeof(unsigned short), hipMemcpyHostToDevice, m->ctx->stream));
            }
            ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
        }
        if (ly.has_bwd) {
            std::vector<float> Bb((size_t)ntaps * Co * Ci);
            for (int tp = 0; tp < ntaps; ++tp)
                for (int ci = 0; ci < Ci; ++ci)
                    for (int co = 0; co < Co; ++co)
                        Bb[((size_t)tp * Co + co) * Ci + ci] = W[((size_t)tp * Ci + ci) * Co + co];
            ALQ_TRY(gemm_set(m, &ly.bwd, Bb));
        }
    } else if (sp.type == ALQ_CONVT) {
        // TF [tap][co][ci]; the two-slot plans index taps in the full k^3 enumeration, the older engines by class
        std::vector<float> Bfull((size_t)ntaps * Ci * Co);
        for (int tp = 0; tp < ntaps; ++tp)
            for (int ci = 0; ci < Ci; ++ci)
                for (int co = 0; co < Co; ++co)
                    Bfull[((size_t)tp * Ci + ci) * Co + co] = W[((size_t)tp * Co + co) * Ci + ci];
        for (size_t c = 0; c < ly.fwd.size(); ++c) {
            const std::vector<int> &tl = ly.class_taps[c];
            std::vector<float> B((size_t)tl.size() * Ci * Co);
            for (size_t j = 0; j < tl.size(); ++j)
                for (int ci = 0; ci < Ci; ++ci)
                    for (int co = 0; co < Co; ++co)
                        B[((size_t)j * Ci + ci) * Co + co] = W[((size_t)tl[j] * Co + co) * Ci + ci];
            const bool p4ok = ly.fwd[c].p4.ok && !ly.fwd_all.ok;     // per-class two-slot plan: only without the fused form
            ly.fwd[c].p4.ok = false;
            ALQ_TRY(gemm_set(m, &ly.fwd[c], B));
            if (p4ok) {
                ly.fwd[c].p4.ok = true;
                ALQ_TRY(set4(m, &ly.fwd[c].p4, Bfull));
            }
        }
        if (ly.fwd_all.ok) ALQ_TRY(set4(m, &ly.fwd_all, Bfull));
        auto up3 = [&](T3dPlan *pl) -> int {
            unsigned short *dw = reinterpret_cast<unsigned short *>(pl->d_W);
            if (!dw) ALQ_TRY(m->dalloc(&dw, pl->h_W.size()));
            pl->d_W = dw;
            ALQ_HIP(hipMemcpyAsync(dw, pl->h_W.data(), pl->h_W.size() * sizeof(unsigned short), hipMemcpyHostToDevice, m->ctx->stream));
            ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
            std::vector<unsigned short>().swap(pl->h_W);
            return ALQ_OK;
        };
        if (ly.t3f.ok) { if (ly.t3f.kind == 8) t3d8_fwd_pack(&ly.t3f, W); else t3d_fwd_pack(&ly.t3f, W); ALQ_TRY(up3(&ly.t3f)); }
        if (ly.t3b.ok && ly.has_bwd && ly.t3b.kind == 8) { t3d8_bwd_pack(&ly.t3b, W); ALQ_TRY(up3(&ly.t3b)); }      // (two accumulators)
        else if (ly.t3b.ok && ly.has_bwd && c3d_subnormals_ok(m->ctx)) { t3d_bwd_pack(&ly.t3b, W); ALQ_TRY(up3(&ly.t3b)); }      // (one-accumulator fp16 pairs)
        if (ly.has_bwd) {
            std::vector<float> Bb(W, W + ly.w_elems);   // [(tap, co)][ci] as stored
            ALQ_TRY(gemm_set(m, &ly.bwd, Bb));
        }
    } else {
        // fc: TF W[o][f_tf]; activation memory order f_mem = ((d*H+h)*W+w)*C+c, reference flatten
        // order f_tf = ((c*W+w)*H+h)*D+d (tf.transpose = full axis reversal, NN.py:296-301)
        const int D = ly.in.D, H = ly.in.H, Wd = ly.in.W, C = ly.in.C;
        const int64_t F = ly.F;
        std::vector<float> Wp((size_t)Co * F);
        for (int d = 0; d < D; ++d)
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < Wd; ++w)
                    for (int c = 0; c < C; ++c) {
                        const int64_t fm = (((int64_t)d * H + h) * Wd + w) * C + c;
                        const int64_t ft = (((int64_t)c * Wd + w) * H + h) * D + d;
                        for (int o = 0; o < Co; ++o) Wp[(size_t)o * F + fm] = W[(size_t)o * F + ft];
                    }
        {   // |cotangent of input f| <= sum_o |W[o][f]| * max |cotangent of the output|: the layer's L1 bound for the chain of static
            // fp16x2 bounds of the backward pass (like bwd_l1 of a conv)
            std::vector<double> col((size_t)F, 0.0);
            for (int o = 0; o < Co; ++o)
                for (int64_t f = 0; f < F; ++f) col[(size_t)f] += std::fabs((double)W[(size_t)o * F + f]);
            double best = 0;
            for (int64_t f = 0; f < F; ++f) best = std::max(best, col[(size_t)f]);
            ly.bwd_l1 = best;
        }
        if (ly.dense_fc_small) {
            ly.fc_wv_amax = 0.f;
            if (Co == 2 && !ly.fc_wv)      // a two-class head that is not fused into a conv: still the bound its input cotangent obeys under the unit cotangent
                for (int64_t f = 0; f < F; ++f) ly.fc_wv_amax = std::max(ly.fc_wv_amax, std::fabs((0.f + Wp[(size_t)f]) - Wp[(size_t)F + f]));
            if (Co == 2 && ly.fc_wv) {      // the head's input cotangent under the unit cotangent (+1, -1): W0 - W1, set with the weights
                std::vector<float> wv((size_t)F);
                for (int64_t f = 0; f < F; ++f) {
                    wv[(size_t)f] = (0.f + Wp[(size_t)f]) - Wp[(size_t)F + f];
                    ly.fc_wv_amax = std::max(ly.fc_wv_amax, std::fabs(wv[(size_t)f]));
                }
                ALQ_HIP(hipMemcpyAsync(ly.fc_wv, wv.data(), wv.size() * sizeof(float), hipMemcpyHostToDevice, m->ctx->stream));
                ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
                if (ly.fc_wv16 && ly.fc_wv_amax > 0.f && F % 4 == 0) {
                    // the same vector as fp16 pairs of x * 2^e, e = 14 - exponent(max |x|) (the scale igemm4_launch derives from
                    // fc_wv_amax): per 4 consecutive values [h0 h1 | h2 h3 | l0 l1 | l2 l3], h = fp16(x 2^e), l = fp16((x 2^e - h) 2^11)
                    int ex = 0;
                    (void)std::frexp(ly.fc_wv_amax, &ex);
                    const int e = 14 - ex;
                    std::vector<unsigned> sp((size_t)F);
                    auto hb = [](float v) { const _Float16 h = (_Float16)v; unsigned short b; std::memcpy(&b, &h, 2); return (unsigned)b; };
                    for (int64_t f = 0; f < F; f += 4) {
                        unsigned h[4], l[4];
                        for (int k = 0; k < 4; ++k) {
                            const float xs = std::ldexp(wv[(size_t)f + k], e);
                            const _Float16 hh = (_Float16)xs;
                            h[k] = hb(xs);
                            l[k] = hb(std::ldexp(xs - (float)hh, 11));
                        }
                        sp[(size_t)f] = h[0] | (h[1] << 16); sp[(size_t)f + 1] = h[2] | (h[3] << 16);
                        sp[(size_t)f + 2] = l[0] | (l[1] << 16); sp[(size_t)f + 3] = l[2] | (l[3] << 16);
                    }
                    ALQ_HIP(hipMemcpyAsync(ly.fc_wv16, sp.data(), sp.size() * sizeof(unsigned), hipMemcpyHostToDevice, m->ctx->stream));
                    ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
                    if (ly.fc_wv16c && F % 8 == 0) {      // the same scale, pieces at their true scale, per voxel [h8 | l8] (plane-sweep backward)
                        std::vector<unsigned short> sc;
                        c3d_presplit_vec(wv.data(), F, e, &sc);
                        ALQ_HIP(hipMemcpyAsync(ly.fc_wv16c, sc.data(), sc.size() * sizeof(unsigned short), hipMemcpyHostToDevice, m->ctx->stream));
                        ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
                    }
                }
            }
            ALQ_HIP(hipMemcpyAsync(ly.d_Wp, Wp.data(), Wp.size() * sizeof(float), hipMemcpyHostToDevice, m->ctx->stream));
            ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
        } else {
            std::vector<float> B((size_t)F * Co);
            for (int64_t f = 0; f < F; ++f)
                for (int o = 0; o < Co; ++o) B[(size_t)f * Co + o] = Wp[(size_t)o * F + f];
            ALQ_TRY(gemm_set(m, &ly.fwd[0], B));
            if (ly.has_bwd) ALQ_TRY(gemm_set(m, &ly.bwd, Wp));   // [(o)][f_mem]
        }
    }
    ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
    ly.weights_set = true;
    return ALQ_OK;
}

int alq_gather_normalize(alq_ctx *ctx, const void *const *d_vols, int mm, int vol_is_f64,
                         const int64_t pad_dims[3], const int64_t *d_inds, int64_t n,
                         const int32_t ps[3], const double *h_stats, int quirk, int out_is_f64, void *d_out) {
    ALQ_REQUIRE(ctx && d_vols && pad_dims && ps && d_out && (n == 0 || d_inds), ALQ_EINVAL, "alq_gather_normalize: null argument");
    ALQ_REQUIRE(mm >= 1 && mm <= 16, ALQ_EINVAL, "alq_gather_normalize: %d modalities", mm);
    ALQ_REQUIRE(quirk == 2 || h_stats, ALQ_EINVAL, "alq_gather_normalize: stats missing");
    if (n == 0) return ALQ_OK;
    int64_t orig[3];
    for (int d = 0; d < 3; ++d) {
        ALQ_REQUIRE(ps[d] >= 1 && (ps[d] & 1), ALQ_EINVAL, "patch dims must be odd (patch_utils.py:1119-1121), got %d", ps[d]);
        orig[d] = pad_dims[d] - 2 * ((ps[d] - 1) / 2);
        ALQ_REQUIRE(orig[d] >= 1, ALQ_EINVAL, "padded volume smaller than the patch");
    }
    ALQ_HIP(hipSetDevice(ctx->device));
    // small device-side parameter block: m pointers + 2m doubles
    struct Block { const void *ptrs[16]; double stats[32]; } hb;
    std::memset(&hb, 0, sizeof(hb));
    for (int j = 0; j < mm; ++j) {
        hb.ptrs[j] = d_vols[j];
        if (h_stats) { hb.stats[2 * j] = h_stats[2 * j]; hb.stats[2 * j + 1] = h_stats[2 * j + 1]; }
    }
    static_assert(sizeof(Block) <= ALQ_PARAM_BLOCK_BYTES, "parameter block too small");
    Block *db = reinterpret_cast<Block *>(ctx->param_block);
    // pageable source: the copy is staged before the call returns, and it is stream-ordered
    // behind any kernel of an earlier call that still reads the block
    ALQ_HIP(hipMemcpyAsync(db, &hb, sizeof(Block), hipMemcpyHostToDevice, ctx->stream));
    return gather_normalize_impl(ctx, db->ptrs, mm, vol_is_f64, pad_dims, orig, d_inds, n, ps, db->stats, quirk,
                                 out_is_f64, d_out);
}

int alq_forward(alq_model *m, const float *d_x, int N, float *d_post, int64_t *d_pred, float *d_feat,
                int feature_layer_idx) {
    ALQ_REQUIRE(m && d_x, ALQ_EINVAL, "alq_forward: null argument");
    ALQ_REQUIRE(N >= 0 && N <= m->max_batch, ALQ_EINVAL, "alq_forward: N=%d exceeds max_batch=%d", N, m->max_batch);
    if (N == 0) return ALQ_OK;
    ALQ_HIP(hipSetDevice(m->ctx->device));
    m->last_call_fisher = false;
    apply_knobs(m);
    ALQ_TRY(run_forward(m, d_x, N, false, /*keep_all=*/d_feat != nullptr));
    ALQ_TRY(k_softmax(m->ctx, m->logits, m->nclass, N, d_post ? d_post : m->post, d_pred));
    if (d_feat) {
        ALQ_REQUIRE(feature_layer_idx >= 0 && feature_layer_idx < (int)m->layers.size(), ALQ_EINVAL, "bad feature layer");
        const View &v = m->layers[feature_layer_idx].out;
        ALQ_REQUIRE(v.cs == v.C, ALQ_EUNSUPPORTED, "feature layer is a concat slice");
        ALQ_HIP(hipMemcpyAsync(d_feat, v.p, (size_t)N * v.elems() * sizeof(float), hipMemcpyDeviceToDevice, m->ctx->stream));
    }
    return ALQ_OK;
}

int alq_score_entropy(alq_ctx *ctx, const float *d_p1, int64_t n, double *d_absdev, float *d_H) {
    ALQ_REQUIRE(ctx && (n == 0 || d_p1), ALQ_EINVAL, "alq_score_entropy: null argument");
    if (n == 0) return ALQ_OK;
    ALQ_HIP(hipSetDevice(ctx->device));
    return score_entropy_impl(ctx, d_p1, n, d_absdev, d_H);
}

size_t alq_topk_work_bytes(int64_t n) { return topk_work_bytes_impl(n); }

int alq_topk_uncertain(alq_ctx *ctx, const double *d_keys, int64_t n, int64_t B, int64_t *d_out_idx, void *d_work) {
    ALQ_REQUIRE(ctx && (n == 0 || (d_keys && d_work)) && (B == 0 || d_out_idx), ALQ_EINVAL, "alq_topk_uncertain: null argument");
    ALQ_HIP(hipSetDevice(ctx->device));
    return topk_impl(ctx, d_keys, n, B, d_out_idx, d_work);
}

int alq_fisher(alq_model *m, const float *d_x, int N, const float *d_p1_in, double diag_load, float *d_p1_out,
               double *d_g0, double *d_g1, double *d_A, double *d_trace, double *d_Asum) {
    ALQ_REQUIRE(m && d_x, ALQ_EINVAL, "alq_fisher: null argument");
    ALQ_REQUIRE(N >= 0 && N <= m->max_batch, ALQ_EINVAL, "alq_fisher: N=%d exceeds max_batch=%d", N, m->max_batch);
    ALQ_HIP(hipSetDevice(m->ctx->device));
    if (N == 0) {
        if (d_Asum) ALQ_HIP(hipMemsetAsync(d_Asum, 0, sizeof(double) * m->L * m->L, m->ctx->stream));
        return ALQ_OK;
    }
    struct SkipGuard {       // sampled profiling: only every prof_every-th pass records events
        alq_ctx *c;
        explicit SkipGuard(alq_ctx *ctx) : c(ctx) { c->prof_skip = c->prof_on && (c->prof_pass++ % c->prof_every) != 0; }
        ~SkipGuard() { c->prof_skip = false; }
    } guard(m->ctx);
    m->last_call_fisher = true;
    apply_knobs(m);
    ALQ_TRY(run_forward(m, d_x, N, true));
    ALQ_TRY(k_softmax(m->ctx, m->logits, m->nclass, N, m->post, nullptr));
    ALQ_TRY(run_backward(m, d_x, N));
    int nblocks = 0;
    ALQ_TRY(k_fisher_finalize(m->ctx, m->Spart, m->nslab, m->nslab_max, m->max_batch, m->S, m->L, m->sizes, m->post,
                              d_p1_in, N, diag_load, d_p1_out, d_g0, d_g1, d_A, d_trace, m->Apart, &nblocks));
    if (d_Asum) ALQ_TRY(k_reduce_Asum(m->ctx, m->Apart, nblocks, m->L * m->L, d_Asum));
    return ALQ_OK;
}

// rows of a resident pool -> the model's staging buffer (one 16-byte-wide copy kernel, no caller-side temporary)
static int stage_rows(alq_model *m, const float *d_pool, const int64_t *d_rows, int N) {
    ALQ_REQUIRE(d_pool && d_rows, ALQ_EINVAL, "null pool / row list");
    if (!m->x_stage) ALQ_TRY(m->dalloc(&m->x_stage, (size_t)m->max_batch * m->epp));
    return gather_rows_impl(m->ctx, d_pool, d_rows, N, m->epp, m->x_stage);
}

int alq_forward_rows(alq_model *m, const float *d_pool, const int64_t *d_rows, int N, float *d_post, int64_t *d_pred,
                     float *d_feat, int feature_layer_idx) {
    ALQ_REQUIRE(m != nullptr, ALQ_EINVAL, "alq_forward_rows: null model");
    ALQ_REQUIRE(N >= 0 && N <= m->max_batch, ALQ_EINVAL, "alq_forward_rows: N=%d exceeds max_batch=%d", N, m->max_batch);
    if (N == 0) return ALQ_OK;
    ALQ_HIP(hipSetDevice(m->ctx->device));
    ALQ_TRY(stage_rows(m, d_pool, d_rows, N));
    return alq_forward(m, m->x_stage, N, d_post, d_pred, d_feat, feature_layer_idx);
}

int alq_fisher_rows(alq_model *m, const float *d_pool, const int64_t *d_rows, int N, const float *d_p1_in, double diag_load,
                    float *d_p1_out, double *d_g0, double *d_g1, double *d_A, double *d_trace, double *d_Asum) {
    ALQ_REQUIRE(m != nullptr, ALQ_EINVAL, "alq_fisher_rows: null model");
    ALQ_REQUIRE(N >= 0 && N <= m->max_batch, ALQ_EINVAL, "alq_fisher_rows: N=%d exceeds max_batch=%d", N, m->max_batch);
    ALQ_HIP(hipSetDevice(m->ctx->device));
    if (N > 0) ALQ_TRY(stage_rows(m, d_pool, d_rows, N));
    return alq_fisher(m, N > 0 ? m->x_stage : d_pool, N, d_p1_in, diag_load, d_p1_out, d_g0, d_g1, d_A, d_trace, d_Asum);
}

int alq_topk_merge(const double *h_keys, const int64_t *h_idx, int64_t n, int64_t B, int64_t *h_out_idx,
                   int64_t *n_out) {
    ALQ_REQUIRE(n >= 0 && B >= 0 && (n == 0 || (h_keys && h_idx)) && (B == 0 || h_out_idx) && n_out, ALQ_EINVAL,
                "alq_topk_merge: bad argument");
    // Keys are compared by BIT PATTERN, like the device top-B (topk.hip): |p - .5| is non-negative, so the unsigned
    // order of the bits is the numeric order, and a NaN key (NaN posterior of a NaN / inf patch on some rank) sorts
    // deterministically behind every number instead of breaking the strict weak ordering of operator<.
    std::vector<std::pair<uint64_t, int64_t>> v;
    v.reserve((size_t)n);
    for (int64_t i = 0; i < n; ++i)
        if (h_idx[i] >= 0) {
            double k = h_keys[i] == 0.0 ? 0.0 : h_keys[i];      // -0.0 -> +0.0
            uint64_t b;
            std::memcpy(&b, &k, sizeof(b));
            v.emplace_back(b, h_idx[i]);
        }
    std::sort(v.begin(), v.end());          // key bits, then global index
    const int64_t m = std::min<int64_t>(B, (int64_t)v.size());
    for (int64_t i = 0; i < m; ++i) h_out_idx[i] = v[(size_t)i].second;
    *n_out = m;
    return ALQ_OK;
}

static const char *kProfNames[PROF_NUM] = {"igemm_fwd", "igemm_bwd", "elementwise", "reduce", "fc_small",
                                           "igemm3_fwd", "igemm3_bwd", "direct_conv", "igemm_f16x2"};

int alq_prof_enable(alq_ctx *ctx, int on) {
    ALQ_REQUIRE(ctx != nullptr, ALQ_EINVAL, "null ctx");
    ctx->prof_on = on != 0;
    ctx->prof_every = on > 1 ? on : 1;
    ctx->prof_pass = 0;
    return ALQ_OK;
}

int alq_prof_reset(alq_ctx *ctx) {
    ALQ_REQUIRE(ctx != nullptr, ALQ_EINVAL, "null ctx");
    ALQ_TRY(ctx->prof_collect());
    for (int c = 0; c < PROF_NUM; ++c) {
        ctx->prof[c].ms = 0; ctx->prof[c].launches = 0; ctx->prof[c].flops = 0;
    }
    return ALQ_OK;
}

int alq_prof_num_classes(void) { return PROF_NUM; }
const char *alq_prof_class_name(int cls) { return (cls >= 0 && cls < PROF_NUM) ? kProfNames[cls] : ""; }

int alq_prof_read(alq_ctx *ctx, int cls, double *ms, int64_t *launches, double *flops) {
    ALQ_REQUIRE(ctx && cls >= 0 && cls < PROF_NUM, ALQ_EINVAL, "alq_prof_read: bad class");
    ALQ_TRY(ctx->prof_collect());
    if (ms) *ms = ctx->prof[cls].ms;
    if (launches) *launches = ctx->prof[cls].launches;
    if (flops) *flops = ctx->prof[cls].flops;
    return ALQ_OK;
}

int alq_model_debug_copy(alq_model *m, int layer_idx, int what, int N, float *d_out, int64_t *elems_out) {
    ALQ_REQUIRE(m && d_out && N >= 1 && N <= m->max_batch, ALQ_EINVAL, "alq_model_debug_copy: bad argument");
    ALQ_HIP(hipSetDevice(m->ctx->device));
    if (what == 4) {
        if (elems_out) *elems_out = (int64_t)N * m->L;
        return debug_f64_copy(m->ctx, m->S, (long long)N * m->L, d_out);
    }
    if (what == 5) {       // per (tile, wave) partials of the logit difference from the fused fc head (last pass)
        const Layer &head = m->layers.back();
        ALQ_REQUIRE(head.fc_part2 && m->last_head_fused, ALQ_EUNSUPPORTED, "the last pass did not run the fused fc head");
        const int ns = m->last_c3 ? 4 : head.fc_slices2;      // plane-sweep engine: one partial per (patch, wave)
        if (elems_out) *elems_out = (int64_t)N * ns;
        ALQ_HIP(hipMemcpyAsync(d_out, m->last_c3 ? head.c3_part : head.fc_part2, (size_t)N * ns * sizeof(float), hipMemcpyDeviceToDevice,
                               m->ctx->stream));
        return ALQ_OK;
    }
    ALQ_REQUIRE(layer_idx >= 0 && layer_idx < (int)m->layers.size(), ALQ_EINVAL, "bad layer index");
    const Layer &ly = m->layers[layer_idx];
    if (what == 0 || what == 1) {
        // the last conv under a fused fc head: a Fisher pass stores neither its output nor the cotangent of it
        const Layer &head = m->layers.back();
        ALQ_REQUIRE(!(layer_idx + 2 == (int)m->layers.size() && head.spec.type == ALQ_FC &&
                      (what == 0 ? m->last_head_fused
                                 : (m->last_call_fisher && head.fc_maskbits != nullptr && !g_dbg_knobs[4] && !g_dbg_knobs[5]))),
                    ALQ_EUNSUPPORTED, "layer %d: this tensor is not materialised in a Fisher pass (fc head fused into the layer: "
                    "create the model under ALQ_NO_FC_BITS=1 to keep it)", layer_idx);
        const View &v = what == 0 ? ly.out : ly.dout;
        if (elems_out) *elems_out = (int64_t)N * v.elems();
        return debug_view_copy(m->ctx, v, N, d_out);
    }
    ALQ_REQUIRE(ly.pidx >= 0 && (what == 2 || what == 3), ALQ_EINVAL, "layer has no channel-sum fields");
    const bool isfc = ly.spec.type == ALQ_FC;
    const int64_t e = (int64_t)N * (isfc ? 1 : (what == 2 ? ly.in.vox() : ly.out.vox()));
    if (elems_out) *elems_out = e;
    ALQ_HIP(hipMemcpyAsync(d_out, what == 2 ? ly.asum : ly.dsum, e * sizeof(float), hipMemcpyDeviceToDevice,
                           m->ctx->stream));
    return ALQ_OK;
}

int alq_model_engine_info(alq_model *m, int what) {
    ALQ_REQUIRE(m && ((what >= 0 && what <= 3) || (what >= 5 && what <= 13)), ALQ_EINVAL, "alq_model_engine_info: bad argument");
    if (what == 13) return m->last_c3_bwd ? m->c3_bwd_rows : 0;      // form of the head conv's backward kernel: 7 = 27 taps in 7 k-steps, 8 / 4 = the 9-k-step kernel
    if (what == 6) return m->last_f16_derived ? 1 : 0;
    if (what == 7) return m->last_t3f;        // conv_transpose launches of the last forward pass on the row-sweep engine (t3d.hip)
    if (what == 8) return m->last_t3b;        // ... of the last backward pass
    if (what == 12) return m->last_f3f;       // the last forward pass ran enc2 + pool2 as one launch (f3d.hip)
    if (what == 11) return m->last_d3b;       // the last backward pass ran dec1's backward-data launch on the plane-sweep kernel (d3d.hip)
    if (what == 10) return m->last_d3f;       // the last forward pass ran dec1 on the row-sweep engine (d3d.hip)
    if (what == 9) return m->last_e3b;        // the last backward pass ran enc2's backward fused with both pool backward steps (e3d.hip)
    ALQ_HIP(hipSetDevice(m->ctx->device));
    if (what == 0) return c3d_subnormals_ok(m->ctx);
    if (what == 1) return m->last_c3 ? 1 : 0;
    if (what == 2) return m->last_c3_bwd ? 1 : 0;
    if (what == 5) {       // flip-safe head: marked groups dropped by a full list segment since the model was created
        if (!m->flip_overflow) return 0;
        unsigned h = 0;
        ALQ_HIP(hipMemcpyAsync(&h, m->flip_overflow, sizeof(unsigned), hipMemcpyDeviceToHost, m->ctx->stream));
        ALQ_HIP(hipStreamSynchronize(m->ctx->stream));
        return h > 0x7fffffffu ? 0x7fffffff : (int)h;
    }
    for (const Layer &ly : m->layers)
        if (ly.c3f.ok) return ly.c3f.oneacc;
    return 0;
}

int alq_debug_set(int key, int value) {
    ALQ_REQUIRE(key >= 0 && key < 8, ALQ_EINVAL, "alq_debug_set: bad key");
    // an explicit override of every model's snapshot; 0 restores "whatever the model was created with" for the keys
    // whose environment default is 0 (all of them unless the ALQ_* diagnostics variables were set at creation)
    g_knob_override[key] = value > 0 ? value : -1;
    g_dbg_knobs[key] = value;
    return ALQ_OK;
}

int alq_debug_set_stamp_buffer(void *d_buf) {
    g_igemm2_dbg = reinterpret_cast<unsigned long long *>(d_buf);
    return ALQ_OK;
}

int alq_synth_patches(alq_ctx *ctx, uint64_t seed, int64_t first_id, int64_t n, int64_t epp, float *d_out) {
    ALQ_REQUIRE(ctx && (n == 0 || d_out), ALQ_EINVAL, "alq_synth_patches: null argument");
    if (n == 0) return ALQ_OK;
    ALQ_HIP(hipSetDevice(ctx->device));
    return synth_impl(ctx, seed, first_id, n, epp, d_out);
}

}  // extern "C"
