// Internal declarations shared by the translation units of libalq.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "alq.h"

namespace alq {

void set_error(const char *fmt, ...);

#define ALQ_HIP(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            ::alq::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,               \
                             hipGetErrorString(e_));                                      \
            return ALQ_EHIP;                                                              \
        }                                                                                 \
    } while (0)

#define ALQ_REQUIRE(cond, code, ...)                                                      \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            ::alq::set_error(__VA_ARGS__);                                                \
            return (code);                                                                \
        }                                                                                 \
    } while (0)

#define ALQ_TRY(expr)                                                                     \
    do {                                                                                  \
        int rc_ = (expr);                                                                 \
        if (rc_ != ALQ_OK) return rc_;                                                    \
    } while (0)

// A channels-last activation tensor [N, D, H, W, cs] of which channels c0 .. c0+C-1 are "ours"
// (concat skips are realised by two producers writing disjoint channel slices of one buffer).
struct View {
    float *p = nullptr;
    int D = 1, H = 1, W = 1;
    int cs = 0, c0 = 0, C = 0;
    // split concat (igemm4 only): channels [0, split) are a dense tensor at p, channels [split, C) a dense tensor
    // of the same row stride cs at p + delta; split = 0: one strided tensor
    int split = 0;
    long long delta = 0;
    // sign field of the tensor (Fisher pass): one byte per 4 consecutive channels, bit k of the low nibble = (value k > 0), at
    // byte (float offset from p) / 4 - what a backward launch needs of a ReLU'd activation (16 x fewer bytes than the floats)
    unsigned char *sg = nullptr;
    int64_t vox() const { return (int64_t)D * H * W; }
    int64_t elems() const { return vox() * C; }
};

// igemm_* = the fp32 engines (v1 / v2), igemm3_* = the bf16x3 matrix-core engine, direct = the VALU first-layer conv
enum ProfClass { PROF_IGEMM_FWD = 0, PROF_IGEMM_BWD = 1, PROF_ELEMWISE = 2, PROF_REDUCE = 3,
                 PROF_FC_SMALL = 4, PROF_IGEMM3_FWD = 5, PROF_IGEMM3_BWD = 6, PROF_DIRECT = 7,
                 PROF_IGEMM_F16 = 8 /* igemm4 launches on the fp16x2 split (3 products) */, PROF_NUM = 9 };

struct ProfSlot {
    double ms = 0;
    int64_t launches = 0;
    double flops = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<hipEvent_t> pool;
};

}  // namespace alq

// ---- the 16-byte-store data hazard (device code) -------------------------------------------------------------------------
// A `buffer_store_dwordx3/x4` reads its data registers AFTER it has issued.  The compiler's hazard recogniser knows the case
// (GCNHazardRecognizer::createsVALUHazard, "VMEM store of more than 64 bits followed by a VALU write of its vdata": 2 wait
// states on gfx940+), but only for stores WITHOUT a register soffset; for a store whose scalar offset is an SGPR - every
// plane / row-sweep store of d3d / f3d / t3d / t3d8b / c3d - it inserts nothing.  Round 5 saw exactly that case corrupt
// outputs once a second wave shared the SIMD (the rows held a later vector result; HISTORY.md 13).  ALQ_STORE_HOLD(data),
// placed right behind such a store, keeps the data registers the store's for ALQ_STORE_HOLD_STATES wait states (the table
// entry's figure for its sibling): they are inputs of the statement, so nothing may be allocated into them before it, and
// the "memory" clobber keeps the store in front of it.  tools/isa_store_hazard.py checks the RESULT in the device assembly
// of every build (csrc/build.sh): no such store may have a writer of its data registers within fewer wait states.
#define ALQ_STORE_HOLD_STATES 2
#ifdef ALQ_HOLD_NOMEM      // (timing experiment: what the "memory" clobber costs the schedule)
#define ALQ_STORE_HOLD(...) asm volatile("s_nop 1" ::__VA_ARGS__)
#else
#define ALQ_STORE_HOLD(...) asm volatile("s_nop 1" ::__VA_ARGS__ : "memory")
#endif

#define ALQ_PARAM_BLOCK_BYTES 512

struct alq_ctx {
    int device = 0;
    int num_cus = 256;         // compute units of the device (alq_ctx_create): what the persistent-grid launches size themselves by
    hipStream_t stream = nullptr;
    bool prof_on = false;
    int prof_every = 1;        // time the launches of every prof_every-th alq_fisher pass (event pairs cost ~6 % when on every launch)
    long long prof_pass = 0;
    bool prof_skip = false;    // this pass is not sampled
    // Side stream of the Fisher pass: the HBM-bound statistics kernels of a layer (box-filter dot products, the head's
    // channel sums) need nothing from the backward contraction launched right after them, so they run beside it
    // (fork / join with the two events; null when ALQ_NO_SIDE_STREAM was set at creation: everything on `stream`)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool side_used = false;
    bool side_off = false;     // alq_ctx_use_side_stream(ctx, 0): statistics kernels stay on the main stream (a caller that overlaps whole passes on two contexts)
    void *param_block = nullptr;   // small device buffer for per-call parameters (gather)
    void *comm = nullptr;          // RCCL communicator of this rank (comm.hip), or null
    int comm_rank = 0, comm_world = 1;
    int f16_subnormal_mfma = -1;   // c3d_subnormals_ok: -1 not probed yet on this context's device, else 0 / 1
    alq::ProfSlot prof[alq::PROF_NUM];
    int prof_begin(int cls, hipEvent_t *e0, hipEvent_t *e1);
    void prof_end(int cls, hipEvent_t e0, hipEvent_t e1, double flops);
    int prof_collect();
};

namespace alq {

struct ProfScope {
    alq_ctx *ctx;
    int cls;
    double flops;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool active = false;
    ProfScope(alq_ctx *c, int k, double f) : ctx(c), cls(k), flops(f) {
        if (ctx->prof_on && !ctx->prof_skip) active = (ctx->prof_begin(cls, &e0, &e1) == ALQ_OK);
    }
    ~ProfScope() {
        if (active) ctx->prof_end(cls, e0, e1, flops);
    }
};

// ------------------------------------------------------------------ implicit-GEMM engine
// C[m, n] = sum_k A(m, k) * B[k, n] on v_mfma_f32_16x16x4_f32, where
//   m  = (patch, point of an "M grid" MD x MH x MW),
//   k  = (tap, input channel): A(m, k) = in[patch, m*sm + tapoff(tap), ci]  (0 outside the tensor),
//   n  = output channel, written at out[patch, m*so + ooff, n].
// conv fwd, conv bwd-data, conv_transpose fwd (one launch per output parity class),
// conv_transpose bwd-data and fc are all this GEMM with different tap tables.
constexpr int IG_MAXTAPS = 28;
constexpr int IG_MAXK_SMALL = 1024;

struct IgemmArgs {
    const float *in;
    float *out;
    const float *W;     // packed [chunk][nblock][NTW][KC][16]
    const float *bias;  // [Co] or null
    int in_cs, in_c0, Ci, ID, IH, IW;
    int out_cs, out_c0, Co, OD, OH, OW;
    int MD, MH, MW;
    int sm, so, ooffz, ooffy, ooffx;
    int PT, TZ, TY, TX, HZ, HY, HX;
    int rows;                     // PT*TZ*TY*TX live GEMM rows of the 256-row tile
    int minz, miny, minx;
    int ntaps, nchunks, NB;
    int tilesZ, tilesY, tilesX;
    int N, relu, accumulate;
    int K;                        // SMALLC: ntaps * Ci
    int tapoff[IG_MAXTAPS];       // LDS float offset of each tap's shifted window (non-SMALLC)
    const int *koff;              // SMALLC: device table, LDS float offset of flattened (tap, ci)
};

struct IgemmPlan {
    IgemmArgs a;
    int CB = 8, NTW = 1;
    bool smallc = false;
    size_t lds_bytes = 0;
    dim3 grid;
    double flops_per_patch = 0;  // 2 * MACs of the GEMM actually described (no padding waste)
    std::vector<float> h_W;      // packed weights (host), uploaded to d_W
    float *d_W = nullptr;
    std::vector<int> h_koff;     // SMALLC tap/channel offset table, uploaded to d_koff
    int *d_koff = nullptr;
};

struct ConvDesc {
    // geometry of one GEMM: taps are INPUT offsets relative to m*sm
    int ID, IH, IW, Ci;
    int OD, OH, OW, Co;
    int MD, MH, MW;
    int sm = 1, so = 1, ooff[3] = {0, 0, 0};
    std::vector<int> tz, ty, tx;  // tap offsets
};

int igemm_build_plan(const ConvDesc &d, int max_batch, IgemmPlan *plan);
// Bmat(k = tap*Ci + ci, n) supplied by callback -> packed layout
void igemm_pack_weights(IgemmPlan *plan, const std::vector<float> &Bmat /* [K][Co] row-major */);
int igemm_launch(alq_ctx *ctx, const IgemmPlan &plan, const View &in, const View &out,
                 const float *bias, int relu, int accumulate, int N, int prof_cls);

// ------------------------------------------------------------------ pipelined engine (igemm2.hip)
struct Igemm2Args {
    const float *in;
    float *out;
    const float *W;      // packed [chunk][tap][tile][lane][2]
    const float *bias;
    const float *mask;   // ReLU-grad mask source (activation of the destination layer) or null
    float *osumA, *osumB;   // channel sums of the produced rows: columns < split -> A, others -> B
    int mask_cs, mask_c0, mask_from, split;
    int in_cs, in_c0, Ci, ID, IH, IW;
    int out_cs, out_c0, Co, OD, OH, OW;
    int MD, MH, MW;
    int sm, so, ooffz, ooffy, ooffx;
    int PT, TZ, TY, TX, HZ, HY, HX, rows;
    int minz, miny, minx;
    int ntaps, nchunks;
    int tilesZ, tilesY, tilesX;
    int N, relu, accumulate;
    int t0, tsz, tsy, tsx, tnx, tny;   // tap box: LDS offset = t0 + iz*tsz + iy*tsy + ix*tsx
    int dbg_repeat, dbg_flags;
    unsigned long long *dbg;   // phase stamps (diagnostic build), else null
    // host-built tables (igemm2_build_plan): no division / bounds arithmetic per tile in the kernel
    const int *tdesc;          // [tiles_per_group][8]: in_org_vox, out_org_vox, class | full<<8, mz0, my0, mx0, -, -
    const int *sdesc;          // [nhv][4]: rel_vox, valid-mask lo, valid-mask hi, packed (pt,hz,hy,hx)
    int tpg;                   // tiles per patch group
    int in_bytes;              // bytes of the input tensor for N patches (buffer descriptor range)
    int in_pstride, out_pstride;   // voxels per patch group step: PT*ID*IH*IW, PT*OD*OH*OW
    // igemm3's fp16-pair instantiation (launches with a host-known bound on their input): x * f16_sc = h + l 2^-11,
    // result = (c + cl 2^-11) * f16_inv; a bias enters the accumulator as bias * f16_bias_sc
    float f16_sc = 0.f, f16_inv = 0.f, f16_bias_sc = 0.f;
    // ... or ONE scale per patch from a bound on max |input| of every patch (float bits; forward launches: measured input maximum
    // pushed through the layers' L1 norms, k_fwd_bounds): tiles of one patch only (PT = 1); f16_wexp = the weights' exponent
    const unsigned *f16_bound = nullptr;
    int f16_wexp = 0;
};

struct Igemm2Plan {
    Igemm2Args a;
    bool ok = false;
    bool wres = false;   // all weights resident in LDS (vs one 8-channel chunk staged per step)
    int NTW = 1;
    int wgs_per_cu = 1;
    size_t lds_bytes = 0;
    double flops_per_patch = 0;
    std::vector<float> h_W;
    float *d_W = nullptr;
    std::vector<int> h_tdesc, h_sdesc;
    int *d_tdesc = nullptr, *d_sdesc = nullptr;
};

struct Igemm2Fuse {
    const float *mask = nullptr;   // activation whose sign masks output columns >= mask_from (ReLU grad)
    int mask_cs = 0, mask_c0 = 0, mask_from = 0;
    int mask_to = 1 << 30;   // igemm4 only: columns in [mask_from, mask_to) are masked
    float *osumA = nullptr, *osumB = nullptr;
    int split = 0;       // 0: all columns -> osumA; else columns < split -> osumA, others -> osumB
    int store_from = 0;  // columns below this are masked and summed but not stored (igemm4 only)
    int mask_split = 0;  // igemm4 only: the mask source is a split concat (see View): columns >= mask_split at mask + mask_delta
    long long mask_delta = 0;
    // igemm4 only: the sign field of the mask tensor (View::sg of the view `mask` points into): read instead of the floats.
    // sign_out: the sign field of the launch's OUTPUT view, written by the epilogue of a forward launch (after its ReLU)
    const unsigned char *mask_bits = nullptr;
    unsigned char *sign_out = nullptr;
    // igemm4 only: the GEMM input is not a stored tensor but in[n, j] = [bit j of patch n] * in_vec[j]
    const unsigned *in_bits = nullptr;
    const float *in_vec = nullptr;
    float in_vec_amax = 0.f;   // max |in_vec| when the host knows it (> 0 enables the fp16x2 contraction of that launch)
    const unsigned *in_vec16 = nullptr;   // in_vec pre-split for that contraction: per 4 values [h01 | h23 | l01 | l23] fp16 pairs of x * 2^(14 - exp(amax))
    // igemm4 only: max |x| of tensors as float bits (non-negative, so unsigned order = float order).  out_amax
    // [N][tiles per patch * groups * 4] receives the maxima of the STORED output columns per (tile, group, wave)
    // (k_rowmax_u32 folds them per patch); in_amax / in_amax2 [N] describe the (two parts of the) input tensor and
    // enable the fp16x2 contraction with one scale per patch.
    unsigned *out_amax = nullptr;
    int amax_from = 0;         // only columns >= amax_from count for out_amax (the slice the next launch reads)
    const unsigned *in_amax = nullptr, *in_amax2 = nullptr;
    // igemm4 only: a host-known bound on max |x| of the stored input tensor (e.g. a cotangent under the unit cotangent, bounded
    // through the weights' L1 norms): enables the fp16x2 contraction with ONE scale for the launch.  A bound far above the
    // values costs no accuracy: (h, l) fp16 pairs carry 22 bits over 29 binades and an absolute 2^-36 of the scaled range below
    float in_bound = 0.f;
    // igemm4 only (pair form, 8 output channels, one patch per tile): the output feeds nothing but a 2-output fc head -
    // the epilogue emits per (tile, wave) partials of the logit difference against fc_W [fc_F] = W0 - W1
    // (activation-memory order) and the sign byte of every voxel instead of storing the tensor
    const float *fc_W = nullptr;
    long long fc_F = 0;
    float *fc_part = nullptr;
    unsigned *fc_bits = nullptr;
    // flip-safe fused head (fp16x2 contraction only): see Igemm4Args::flip_tau; flip_l1 = max over the output channels of the L1
    // norm of their weights
    unsigned *flip_cnt = nullptr, *flip_list = nullptr;
    int flip_cap = 0;
    float flip_l1 = 0.f;
    bool flip_bias_nonzero = false;     // igemm4's fused head: with a non-zero bias an exact +0 is marked too (see the kernel)
};

extern unsigned long long *g_igemm2_dbg;
extern int g_dbg_knobs[8];
extern int g_no_f16x2;        // ALQ_NO_F16X2, read when a model is created: bf16x3 split in every launch
extern int g_no_fixed;        // ALQ_NO_FIXED (A/B runs, bit-identity test): igemm4 launches use the runtime-constant instantiation only
extern int g_no_xcd_order;    // ALQ_NO_XCD_ORDER (A/B runs): igemm4 tiles in dispatch order
int igemm2_build_plan(const IgemmPlan &p1, Igemm2Plan *p2);
void igemm2_pack_weights(Igemm2Plan *p2, const std::vector<float> &Bmat);
int igemm2_launch(alq_ctx *ctx, const Igemm2Plan &plan, const View &in, const View &out, const float *bias,
                  int relu, int accumulate, int N, int prof_cls, const Igemm2Fuse *fuse);

// ------------------------------------------------------------------ bf16x3 split engine (igemm3.hip)
struct Igemm3Plan {
    bool ok = false;
    bool wres = false;
    int NTW = 1;
    int wgs_per_cu = 1;
    size_t lds_bytes = 0;
    std::vector<unsigned short> h_W;   // [chunk][s][piece][tile][lane][8] bf16 bits
    void *d_W = nullptr;
    // the fp16-pair twin (round 6): pieces h, l x 2^11 of W x 2^w16_exp, same layout with two pieces; packed on request
    // (igemm3_pack_weights_f16) for the Gemms whose launches come with a bound on their input (backward launches of a Fisher pass)
    std::vector<unsigned short> h_W16;
    void *d_W16 = nullptr;
    int w16_exp = 0;
};
int igemm3_build_plan(const Igemm2Plan &p2, Igemm3Plan *p3);
void igemm3_pack_weights(const Igemm2Plan &p2, Igemm3Plan *p3, const std::vector<float> &Bmat);
void igemm3_pack_weights_f16(const Igemm2Plan &p2, Igemm3Plan *p3, const std::vector<float> &Bmat);
int igemm3_launch(alq_ctx *ctx, const Igemm2Plan &p2, const Igemm3Plan &plan, const View &in, const View &out,
                  const float *bias, int relu, int accumulate, int N, int prof_cls, const Igemm2Fuse *fuse);

// ------------------------------------------------------------------ two-slot bf16x3 engine (igemm4.hip)
// One 512-thread workgroup per CU; its two 4-wave halves each own a tile slot and run in anti-phase
// (one stages / stores while the other contracts), sharing one resident copy of the weights.
// A launch is a small "program": per tile a list of staging phases (conv: 8-channel chunks; conv_transpose
// bwd-data: (z, y) parity class x chunk) feeding one accumulator, or - MULTI - one staging phase of all
// chunks followed by several output groups (conv_transpose fwd: every output parity class from one read).
struct Igemm4Args {
    const float *in;
    float *out;
    const void *W;          // packed bf16 pieces, all units resident in LDS
    const float *bias;
    const float *mask;
    float *osumA, *osumB;
    const int *tdesc;       // [tiles per group][8]: in_org_vox, out_org_vox, flags, lohi_zy, lohi_x, (mz0, my0, mx0) bytes, first phase, -
    const int *sdesc;       // [nslots][4]: rel_vox, (pt,hz,hy,hx) packed, LDS byte offset, channel offset
    const int *pdesc;       // [phases or groups][8]: in_off_vox, chunk, S, w_off, tap row, out_off_vox, -, -
    const int *ttab;        // [tap rows][G4_MAXS][4] LDS byte offsets of the taps of k-step s, lane group q
    const int *vdesc;       // [256]: GEMM row -> (pt, z, y, x) of its M-grid point inside the tile, -1 = unused row
    int in_cs, in_c0, out_cs, out_c0, Co;
    int mask_cs, mask_c0, mask_from, mask_to, split;
    int N, PT, tpg, rows;
    int PX, PYX, PZ;        // LDS row pitches of the staged block: per y line, per z plane, per patch
    int smz, smy, smx;      // M-grid point -> halo index multipliers
    int soz, soy, sox, OD, OH, OW;
    int MD, MH, MW;
    int nph, ngr, NP;       // staging phases per tile; MULTI: output groups, planes
    int nslots, plane_bytes;
    int in_pstride, out_pstride, in_bytes;
    int relu, accumulate, pair, store_from;
    int cls_ok;             // halo validity by tile class (sdesc[1] = one bit per class) instead of by coordinate ranges
    int in_split_ch, in_delta, out_split, out_delta, mask_split, mask_delta;   // split-concat views (floats)
    int tt_ints, pd_off, td_off, wbytes, abytes;   // table block in LDS (ints): tap table, then phase, then tile descriptors
    int dbg_repeat;
    unsigned long long *dbg;
    const unsigned *src_bits;   // BITSRC: `in` is one patch-independent vector, masked per patch by these bits
    int bits_pstride, bits_bytes;   // floats per patch of the tensor the bits describe; size of the bit field
    const float *fc_W;          // FCF: fused 2-output fc head (see Igemm2Fuse)
    float *fc_part;
    unsigned char *fc_bits;
    int fc_F;
    int f16_ein;                // F16: the staged input is scaled by 2^f16_ein before the fp16 split (launch-wide scale)
    int f16_ew;                 // F16: scale exponent of the packed weights
    const unsigned *in_amax, *in_amax2;   // F16: per-patch max |x| of the input part(s), float bits -> one scale per tile
    unsigned *out_amax;         // any variant: max |stored output| per (tile, group, wave), float bits
    int amax_from;
    int wp;                     // weight pieces per k-step in LDS (3: bf16x3 layout, 2: fp16x2-only plan)
    int src_presplit;           // BITSRC + F16: `in` holds the vector already split into fp16 pairs [h01|h23|l01|l23] per 4 channels
    int xcd_order;              // 1: logical workgroup id = (XCD, slot) instead of the dispatch id (see the kernel)
    // (appended: the position of an argument changes how the compiler packs the scalar loads of ALL kernels - a field in
    // the middle cost the mask-bit launch 13 %)
    int zreuse;             // 1: k-step 3 * iz + j reads, for row blocks 2 / 3 of a wave, what k-step 3 * (iz + 1) + j reads for row blocks 0 / 1 (fragment reuse, igemm4.hip)
    // FCF + F16 (flip-safe fused head): a 4-channel group holding a finished pre-activation with |value| < flip_tau * 2^(14 - scale
    // exponent) - closer to zero than the fp16x2 contraction's error bound - gets bit 4 of its sign byte set (flip_list non-null
    // switches it on; the list itself is filled by k_flip_fix's scan, which then re-evaluates the marked groups exactly)
    float flip_tau;
    int flip_cap;
    unsigned *flip_cnt;
    unsigned *flip_list;
    // sign fields (View::sg): the ReLU-grad mask as one byte per 4 channels instead of the fp32 activations (mask_bits set: `mask`
    // is not read), and the sign field of the output written after the ReLU of a forward launch (sign_out)
    const unsigned char *mask_bits;
    unsigned char *sign_out;
};

struct G4Geom {
    int kind = 0;             // 0 stride-1 conv (fwd, or bwd-data when flipped), 1 conv_transpose bwd-data, 2 conv_transpose fwd (all classes),
                              // 3 conv_transpose fwd, the one output parity class `cls`,
                              // 4 conv_transpose fwd, every class as its own tiles of one launch (input restaged per class)
    int cls[3] = {0, 0, 0};
    int ID = 1, IH = 1, IW = 1, Ci = 0;     // GEMM input tensor
    int OD = 1, OH = 1, OW = 1, Co = 0;     // GEMM output tensor
    int k[3] = {1, 1, 1}, s[3] = {1, 1, 1}, lo[3] = {0, 0, 0};
    bool flipped = false;
    double flops_per_patch = 0;
};

struct Igemm4Plan {
    bool ok = false;
    Igemm4Args a;
    int NTW = 1;
    int wp = 3;               // weight pieces per k-step (see Igemm4Args::wp)
    int xw = 0;               // > 0: conflict-free fragment layout found for the chosen tile
    // fp16x2-only twin of the plan (two weight pieces in LDS instead of three): with the LDS that frees, tiles whose
    // bf16x3 plan had to fall back to the bank-conflicted natural row order get a conflict-free layout.  Used by
    // igemm4_launch whenever the launch contracts with the fp16x2 split; same tiles per patch, rows and form as `this`.
    std::shared_ptr<Igemm4Plan> alt16;
    bool multi = false;
    bool fic = false;         // issue the prefetch from the contracting side (stage-bound plans)
    // per-launch balance of the two sides of a tick (igemm4.hip, `FIC` / `EPI`): -1 = the kernel family's default
    int tune_fic = -1;        // 0 / 1: where the prefetch is issued (one-column-tile plans without a mask-bit source)
    int tune_epi = -1;        // 0..2: row blocks of a finished tile written back by the contracting side
    size_t lds_bytes = 0;
    double flops_per_patch = 0;
    int Ci = 0, Co = 0;
    struct Unit { int chunk = 0, S = 0, w_off = 0; std::vector<int> tap; };   // tap[(s*4+q)*2+half] = enum tap or -1
    std::vector<Unit> units;
    std::vector<int> h_tdesc, h_sdesc, h_pdesc, h_ttab, h_vdesc;
    int *d_tdesc = nullptr, *d_sdesc = nullptr, *d_pdesc = nullptr, *d_ttab = nullptr, *d_vdesc = nullptr;
    std::vector<unsigned short> h_W;
    void *d_W = nullptr;
    // the same weights as fp16 pairs (h, l * 2^11) of W * 2^w16_exp, for the F16 variant (one column tile, not MULTI)
    std::vector<unsigned short> h_W16;
    void *d_W16 = nullptr;
    int w16_exp = 0;
};
int igemm4_build_plan(const G4Geom &g, int max_batch, Igemm4Plan *plan, int wp = 3);
void igemm4_pack_weights(Igemm4Plan *plan, const std::vector<float> &Bmat /* [(enum tap, ci)][co] */);
int igemm4_launch(alq_ctx *ctx, const Igemm4Plan &plan, const View &in, const View &out, const float *bias,
                  int relu, int accumulate, int N, int prof_cls, const Igemm2Fuse *fuse);

// ------------------------------------------------------------------ plane-sweep engine for the head conv pair (c3d.hip)
// One workgroup per patch sweeps the z planes with three rotating accumulator plane sets in registers; 32^3 patches, 3x3x3,
// forward: 16 (two dense 8-channel tensors) -> 8 channels with the two-class head fused; backward: 8 (sign bytes x W0 - W1) -> 16.
struct C3dPlan {
    bool ok = false;
    int D = 0;
    int w_exp = 0;                        // scale exponent of the packed fp16 weight pairs
    int oneacc = 1;                       // pieces at their true scale, one accumulator (c3d.hip); 0: l scaled by 2^11, two accumulators
    double flops_per_patch = 0;
    std::vector<unsigned short> h_W;      // [k-step][piece][lane][8] fp16 bits
    void *d_W = nullptr;
    std::vector<unsigned short> h_W7;     // backward, 7-k-step form: [A0 A1 A2 B0 B1 B2 S][piece][lane][8] (c3d_bwd7_pack)
    void *d_W7 = nullptr;
};
void c3d_bwd7_pack(C3dPlan *plan, const std::vector<float> &Bmat);      // after c3d_bwd_pack (same scale exponent)
int c3d_subnormals_ok(alq_ctx *ctx);      // 1: the matrix cores keep fp16 subnormal operands (needed by the one-accumulator form)
int c3d_fwd_build(const View &in, const View &out, const int k[3], const int lo[3], const int s[3], C3dPlan *plan);
void c3d_fwd_pack(C3dPlan *plan, const std::vector<float> &Bmat /* [(tap, ci)][co] */);
int c3d_fwd_launch(alq_ctx *ctx, const C3dPlan &plan, const View &in, const float *bias, int N, const unsigned *amaxA, const unsigned *amaxB,
                   const float *fc_W, float *fc_part, float *asum_part, unsigned char *fc_bits, float flip_tau);
// backward of that conv in a Fisher pass: input = [sign bytes of its output] x one pre-split vector (c3d_presplit_vec), output
// channels 0..7 masked (maskA: sign field, or null) and summed per voxel, channels 8..15 stored (dB) and summed
int c3d_bwd_build(const View &fwd_in, const View &fwd_out, const int k[3], const int lo[3], const int s[3], C3dPlan *plan);
void c3d_bwd_pack(C3dPlan *plan, const std::vector<float> &Bmat_fwd /* the forward conv's [(tap, ci)][co] */);
void c3d_presplit_vec(const float *v, long long F, int e, std::vector<unsigned short> *out);
int c3d_bwd_launch(alq_ctx *ctx, const C3dPlan &plan, int N, const unsigned char *bits, const void *vec16, int e_in, const unsigned char *maskA,
                   float *dB, float *sumA, float *sumB, int rows_per_wave = 8);

// ------------------------------------------------------------------ row-sweep engine for the stride-2 conv_transpose (t3d.hip)
// 3x3x3 / stride 2 conv_transpose 16 -> 8 channels at 16^3 -> 32^3 (NET-C's up2) and its backward-data pass: every wave sweeps the
// rows of one input plane on its own (no workgroup barrier), weights in registers, an accumulator = one 1 KB output row.
struct T3dPlan {
    bool ok = false;
    int kind = 0;                         // 0: 16 -> 8 channels at 16^3 (up2); 8: 32 -> 16 channels at 8^3 (up1; backward: t3d8b.hip)
    int w_exp = 0;                        // backward: scale exponent of the packed fp16 weight pairs
    double flops_per_patch = 0;
    std::vector<unsigned short> h_W;      // forward [9][3 pieces][64][8] bf16 bits; backward [9][2 pieces][64][8] fp16 bits
    void *d_W = nullptr;
};
int t3d_build(const View &in, const View &out, const int k[3], const int lo[3], const int s[3], T3dPlan *fwd, T3dPlan *bwd);
void t3d_fwd_pack(T3dPlan *plan, const float *W /* TF filter [tap][co][ci] */);
void t3d_bwd_pack(T3dPlan *plan, const float *W);
void t3d8_fwd_pack(T3dPlan *plan, const float *W /* [tap][16][32] */);
void t3d8_bwd_pack(T3dPlan *plan, const float *W /* [tap][16][32] */);      // (t3d8b.hip)
int t3d8_bwd_launch(alq_ctx *ctx, const T3dPlan &plan, const View &dout, const View &din, int N, float in_bound, const unsigned char *mask_bits, float *dsum);
int t3d_fwd_launch(alq_ctx *ctx, const T3dPlan &plan, const View &in, const View &out, const float *bias, int N, float *osum, unsigned *out_amax);
int t3d_bwd_launch(alq_ctx *ctx, const T3dPlan &plan, const View &dout, const View &din, int N, float in_bound, const unsigned char *mask_bits,
                   float *dsum);

// ------------------------------------------------------------------ enc2 backward fused with both pool backward steps (e3d.hip)
// NET-C in a Fisher pass: [pool2 backward + skip cotangent + ReLU mask + channel sums] -> [3x3x3 conv 16 -> 8 backward-data at
// 16^3, fp16 pairs under the static bound] -> [pool1 backward into enc1's channel-sum field] in one launch.
struct E3dPlan {
    bool ok = false;
    int w_exp = 0;
    double flops_per_patch = 0;
    std::vector<unsigned short> h_Whi, h_Wlo;      // [9 (dz, dy)][2 K steps][64][8] fp16 bits
    void *d_Whi = nullptr, *d_Wlo = nullptr;
};
int e3d_build(const View &in, const View &out, const int k[3], const int lo[3], const int s[3], E3dPlan *plan);
void e3d_pack(E3dPlan *plan, const float *W /* TF conv filter [tap][ci 8][co 16] */);
int e3d_bwd_launch(alq_ctx *ctx, const E3dPlan &plan, int N, const float *skip, const float *dpool, const unsigned char *am2, const unsigned char *sg2,
                   const unsigned char *am1, const unsigned char *sg1, float *dsum2, float *dsum1, float in_bound);

// NET-C's dec1 forward (3x3x3 conv 32 -> 16 channels at 16^3 over a split concat, fp16 pairs at per-patch scales): row sweep at one
// wave per SIMD with all weight fragments in registers (d3d.hip)
struct D3dPlan {
    bool ok = false;
    int w_exp = 0;
    double flops_per_patch = 0;
    std::vector<unsigned short> h_Whi, h_Wlo;      // [27 taps][64][8] fp16 bits
    void *d_Whi = nullptr, *d_Wlo = nullptr;
    std::vector<unsigned short> h_Bhi, h_Blo;      // backward: [2 row blocks][3 dz][5 K steps][64][8]
    void *d_Bhi = nullptr, *d_Blo = nullptr;
};
void d3d_bwd_pack(D3dPlan *plan, const float *W);      // (after d3d_pack: the same weight scale)
int d3d_bwd_launch(alq_ctx *ctx, const D3dPlan &plan, int N, const float *dout, float in_bound, float *dinA, float *dinB, float *sumB);
int d3d_build(const View &in, const View &out, const int k[3], const int lo[3], const int s[3], D3dPlan *plan);
void d3d_pack(D3dPlan *plan, const float *W /* TF conv filter [tap][ci 32][co 16] */);
int d3d_fwd_launch(alq_ctx *ctx, const D3dPlan &plan, int N, const float *inA, const float *inB, const unsigned *amaxA, const unsigned *amaxB,
                   const float *bias, int relu, float *out, unsigned char *sg, float *osum);

// NET-C's enc2 forward (3x3x3 conv 8 -> 16 channels at 16^3, fp16 pairs at per-patch scales) + the 2x2x2 max-pool behind it in one launch (f3d.hip)
struct F3dPlan {
    bool ok = false;
    int w_exp = 0;
    double flops_per_patch = 0;
    std::vector<unsigned short> h_Whi, h_Wlo;      // [9 (dz, dy)][64][8] fp16 bits
    void *d_Whi = nullptr, *d_Wlo = nullptr;
};
int f3d_build(const View &in, const View &out, const int k[3], const int lo[3], const int s[3], F3dPlan *plan);
void f3d_pack(F3dPlan *plan, const float *W /* TF conv filter [tap][ci 8][co 16] */);
int f3d_fwd_launch(alq_ctx *ctx, const F3dPlan &plan, int N, const float *in, const unsigned *amax, const float *bias, float *out, unsigned char *sg,
                   float *osum, float *pout, unsigned char *parg, float *posum);

// ------------------------------------------------------------------ direct first-layer conv (direct.hip)
struct DirectArgs {
    const float *in;
    float *out;
    const float *W;      // [ntaps*Ci][Co] (the fwd B matrix as is)
    const float *bias;
    float *osum;
    int in_cs, in_c0, Ci, ID, IH, IW;
    int out_cs, out_c0, Co, OD, OH, OW;
    int MD, MH, MW;
    int PT, TZ, TY, TX, HZ, HY, HX, rows;
    int minz, miny, minx;
    int ntaps, tnx, tny, tnz;
    int tilesZ, tilesY, tilesX;
    int N, relu;
};

struct DirectPlan {
    DirectArgs a;
    bool ok = false;
    size_t lds_bytes = 0;
    double flops_per_patch = 0;
    float *d_W = nullptr;
};

int direct_build_plan(const IgemmPlan &p1, DirectPlan *dp);
int direct_launch(alq_ctx *ctx, const DirectPlan &dp, const View &in, const View &out, const float *bias, int relu,
                  int N, float *osum, int prof_cls);
// first conv (1 -> 8 channels, 3x3x3) fused with the 2x2x2 max-pool behind it: both outputs, arg-max, both sums
int direct_conv_pool_launch(alq_ctx *ctx, const float *d_W, const View &in, const View &out, const View &pout,
                            const float *bias, int relu, uint8_t *argmax, float *osum, float *posum, int N,
                            double flops_per_patch, unsigned *amax = nullptr, unsigned char *sg = nullptr, unsigned char *psg = nullptr);

// ------------------------------------------------------------------ wide fc layers (fcgemm.hip)
struct FcGemmPlan {
    bool ok = false;
    int K = 0, N = 0;
    std::vector<unsigned short> h_W;   // [feature tile][k-step][piece][k-group][feature row][8] bf16 bits
    void *d_W = nullptr;
    std::vector<unsigned short> h_W16; // the same as fp16 pairs (2 pieces) of w 2^w_exp, for launches with a static input bound
    void *d_W16 = nullptr;
    int w_exp = 0;
    int form = 0;                      // fp16-pair launches: 0 = 128-wide tiles, tall wave tiles (default), 1 = 64 x 64 wave tiles (ALQ_FC_SQUARE), 2 = 64-wide tiles (ALQ_FC_BN64); read when the plan is built
};
int fcgemm_build_plan(int K, int N, FcGemmPlan *plan);
void fcgemm_pack_weights(FcGemmPlan *plan, const std::vector<float> &Bmat /* [K][N] */);
void fcgemm_pack_weights_f16(FcGemmPlan *plan, const std::vector<float> &Bmat /* [K][N] */);
// in_bound > 0 (no bias): fp16 pairs under that static bound; row_amax (device, [M], scratch): fp16 pairs under the per-row maxima the
// launch measures into it first; neither: bf16 triples
int fcgemm_launch(alq_ctx *ctx, const FcGemmPlan &plan, const View &in, const View &out, const float *bias, int relu,
                  int M, int prof_cls, float in_bound = 0.f, unsigned *row_amax = nullptr);

// one contraction = general plan + (when eligible) pipelined plan / direct first-layer plan
struct Gemm {
    IgemmPlan p1;
    Igemm2Plan p2;
    Igemm3Plan p3;
    Igemm4Plan p4;
    FcGemmPlan pfc;
    bool pfc_f16 = false;      // also pack the fp16-pair twin of pfc's weights (backward launches; forward launches of wide fc layers)
    bool p3_f16 = false;       // also pack the fp16-pair twin of p3's weights: a backward Gemm that runs on igemm3 (no two-slot plan)
    unsigned *fc_row_amax = nullptr;   // forward launch of a wide fc layer on fp16 pairs: [max_batch] measured input maxima (scratch)
    DirectPlan pd;
};

// ------------------------------------------------------------------ other kernels
// osum (optional): channel sums of the pooled output, *fused says whether the kernel produced them
int k_pool_fwd(alq_ctx *, const View &in, const View &out, uint8_t *argmax, const int w[3],
               const int lo[3], int N, float *osum = nullptr, bool *fused = nullptr);
// mask_act / dsum (both or neither): apply the ReLU-grad mask of the input layer's activation to the
// finished cotangent and emit its channel sums; *fused tells whether the kernel could do it
// first parameterised layer behind a pool: only the channel sums of its masked cotangent are needed.  Scatters
// dout (masked by pooled activation > 0, i.e. the ReLU of the arg-max element) into the 2x2(x2) window sums;
// accumulate = the skip destination has already written its part of the field.
int k_pool_bwd_first(alq_ctx *, const View &dout, const View &pool_out, const uint8_t *argmax, const int w[3],
                     int ID, int IH, int IW, int N, float *dsum, int accumulate, int use_signs = 0);   // use_signs: pool_out.sg holds the signs
int k_pool_bwd(alq_ctx *, const View &dout, const View &din, const uint8_t *argmax,
               const int w[3], const int lo[3], int N, int accumulate, const View *mask_act = nullptr,
               float *dsum = nullptr, bool *fused = nullptr, int store_din = 1, int use_signs = 0);   // use_signs: mask_act->sg holds the signs
int k_chansum(alq_ctx *, const View &in, float *field, int N);
int k_mask_chansum(alq_ctx *, const View &dact, const View *act_or_null, float *field, int N);
int boxdot_slabs(long long vox);
int boxdot_conv_slabs(int D, int H, int W, const int k[3]);
int boxdot_convT_slabs(int ID, int IH, int IW, const int k[3], const int s[3]);
// asum2 (optional): second channel-sum field added to asum (input = concat of two producers)
int k_boxdot_conv(alq_ctx *, const float *dsum, const float *asum, const float *asum2, int D, int H, int W,
                  const int k[3], const int lo[3], int N, double *Spart, int nslab_max);
int k_boxdot_convT(alq_ctx *, const float *dsum, const float *asum, const float *asum2, int ID, int IH, int IW,
                   const int k[3], const int s[3], const int lo[3], int N, double *Spart, int nslab_max);
// maskbits (optional, F % 1024 == 0): act > 0 as one bit per element
int k_fc_small_fwd(alq_ctx *, const float *act, int64_t F, const float *Wp, int nout, int N,
                   float *partials, int nslices, unsigned *maskbits = nullptr);
// fc head under the unit cotangent: per-voxel sums of [sign] * wv over 8 channels (wv = W0 - W1)
int k_fc_small_dsum_bits(alq_ctx *, const unsigned *maskbits, const float *wv, int64_t F, int N, float *dsum);
int k_fc_small_finish(alq_ctx *, const float *partials, int nslices, const float *bias, int nout,
                      int relu, int N, float *out);
// two-class head from partials of the logit DIFFERENCE z0 - z1: out[n] = (sum + b0 - b1, 0) - the same posteriors
int k_fc_small_finish_diff(alq_ctx *, const float *partials, int nslices, const float *bias, int N, float *out);
// mask_act / dsum / C (optional): rows are [voxel][C]; masks by act > 0 and emits per-voxel channel sums
int k_fc_small_bwd(alq_ctx *, const float *delta, int nout, const float *Wp, int64_t F, int N,
                   float *dact, const float *mask_act = nullptr, float *dsum = nullptr, int C = 0,
                   bool *fused = nullptr);
int k_rowsum_field(alq_ctx *, const float *field, int64_t len, int N, float *out);
// bound[k][p] = max(bound[k - 1][p], bound[src2[k]][p]) * L[k] + B[k] for k = 1 .. nl - 1, bound[0] = the first layer's measured maximum
// (float bits, as the producers' epilogues leave them): what a forward launch with the fp16x2 split takes as its input maxima when
// nothing measured them (model.hip, run_forward)
struct FwdBoundsArgs { int nl; float L[16]; float B[16]; int src2[16]; int from_input = 0; };
int k_rowmax_abs(alq_ctx *, const float *x, int N, long long K, unsigned *out);      // out[n] = bits of max |x[n][:]|
int k_fwd_bounds(alq_ctx *, const unsigned *amax0, int N, int stride, const FwdBoundsArgs &a, unsigned *bound_all);
int flip_segments(int N);      // scan segments / list slots k_flip_fix needs for N patches
int flip_list_len(int N);
int k_flip_fix(alq_ctx *, unsigned *list, unsigned *cnt, int cap, int N, const float *inA, const float *inB, int CA, int CB,
               int D, int H, int W, int kz, int ky, int kx, int lz, int ly, int lx, const float *W32, const float *bias, int Co,
               unsigned char *bits, long long F, unsigned *overflow = nullptr);
int k_rowmax_u32(alq_ctx *, const unsigned *in, int len, int N, unsigned *out);      // out[n] = max_k in[n][k]
int k_softmax(alq_ctx *, const float *logits, int c, int N, float *post_cN, int64_t *pred);
int k_fill_unit_cotangent(alq_ctx *, float *dlogits, int N);
int k_fisher_finalize(alq_ctx *, const double *Spart, const int *nslab, int nslab_max, int max_batch, double *S,
                      int L, const double *sizes, const float *post_cN, const float *p1_branch, int N,
                      double diag_load, float *p1_out, double *g0, double *g1, double *A, double *trace,
                      double *Apart, int *nblocks_out);
int k_reduce_Asum(alq_ctx *, const double *Apart, int nblocks, int LL, double *Asum);
int fc_small_slices(int64_t F);

// ------------------------------------------------------------------ parameter gradients / training (train.hip)
int k_dropout(alq_ctx *, const View &t, int N, long long first_sample, unsigned long long seed, int layer, float keep_prob);
int k_logit_cotangent(alq_ctx *, const float *post_cN, int c, int N, int mode, int cls, const int *labels, float scale,
                      float *dlogits);
int k_ce_loss(alq_ctx *, const float *post_cN, int c, int N, const int *labels, double *d_out);
long long wgrad_partial_floats(const View &U, const View &V, const int k[3]);
int k_wgrad(alq_ctx *, const View &U, const View &V, const int k[3], const int s[3], const int lo[3], int N, int sum_n,
            float *partial, float *d_out, long long out_stride);
int k_bgrad(alq_ctx *, const View &delta, int N, int sum_n, float *d_out, long long out_stride);
int k_fc_wgrad(alq_ctx *, const float *delta, const View &a, int nout, int N, int sum_n, float *d_out, long long out_stride);
int k_sgd(alq_ctx *, float *theta, const float *g, long long n, float lr);
int k_adam(alq_ctx *, float *theta, const float *g, float *m, float *v, long long n, float lr_t, float b1, float b2, float eps);
int k_sq_accum(alq_ctx *, const float *g, long long per, int N, double *acc);
int k_shrink_sum(alq_ctx *, const float *g, int N, long long P, const long long *off, int L, double *out);
int k_fisher_classes(alq_ctx *, const double *g, const double *w, const double *diag, int N, int c, int L, double *A);

}  // namespace alq
