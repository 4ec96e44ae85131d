// Two-slot bf16x3 implicit-GEMM engine (v4).
//
// Same arithmetic as igemm3.hip (every fp32 operand = hi + mid + lo bf16, six piece products on
// v_mfma_f32_16x16x32_bf16, fp32 accumulate), different machine mapping:
//
//  * ONE 512-thread workgroup per CU.  Its two 4-wave halves each own a tile slot (a halo block in LDS,
//    256 GEMM rows) and run in ANTI-PHASE: in every tick one half contracts its staged block on the
//    matrix cores while the other half writes back its previous tile, splits and stages its next block
//    and issues the prefetch for the one after; one barrier per tick (half 1 simply runs the common loop
//    body one barrier late).  With two independent 256-thread
//    workgroups per CU (igemm3) the two stay in phase - both fight for LDS in the contraction and both
//    leave it idle while staging.
//  * the weights of the whole launch are resident in LDS once per CU (not once per workgroup).
//  * a launch is a small host-built program, so one kernel covers
//      - stride-1 convolutions, forward and backward-data: staging phases = 8-channel chunks;
//      - conv_transpose backward-data (a stride-2 gather): staging phases = (z, y) parity class x chunk,
//        each a dense sub-grid of the cotangent with its own tap box, all feeding one accumulator;
//      - conv_transpose forward, all output parity classes from ONE staged block (MULTI): the block is
//        staged once, then every class contracts its own taps and stores its own output voxels;
//      - "pair" forms for 8 output channels: the 16 MFMA rows are two x-adjacent output voxels
//        (or two x parity classes) x 8 channels, so no half of the tile is padding.
//  * tile / phase descriptors and the tap table live in LDS, halo validity is a per-tile index range per
//    dimension, fragment reads are bank-conflict-free (host-built row permutation + padded pitches),
//    the prefetch is unconditional at one site of a straight-line loop body (see the tick loop), and the
//    two producers of a concat may hand over two dense tensors (split views).
//  * variants of the same kernel (template flags, one instantiation per combination in use):
//      SUMS    channel sums of the output in the epilogue (selector MFMA);
//      FIC     the prefetch is issued at the start of the contraction instead of the end of the staging part, for
//              plans whose staging part is the longer one (a tick is max(stage of one half, contraction of the other));
//      BITSRC  the input is [sign byte] * one patch-independent vector instead of a tensor (backward of the conv under
//              a fc head in a Fisher pass);
//      FCF     the output feeds only a 2-output fc head: the epilogue reduces it against W0 - W1 and writes sign bytes
//              instead of the tensor.
#include "alq_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <type_traits>
#include <vector>

namespace alq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#ifdef ALQ_STAMPS
#define STAMP4(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define PHASE4_END(idx)                        \
    do {                                       \
        unsigned long long t_now_;             \
        STAMP4(t_now_);                        \
        ph[idx] += t_now_ - t_last;            \
        t_last = t_now_;                       \
    } while (0)
#else
// keeps the compiler from moving memory operations across the part boundaries (measured: without it the
// restructured loop ran 35 % slower than the stamped diagnostic build of the same source)
#define PHASE4_END(idx) __builtin_amdgcn_sched_barrier(0)
#endif

constexpr int G4_ROWB = 48;      // bytes per staged voxel row: [hi8 | mid8 | lo8] bf16
constexpr int G4_NSLOT = 8;      // 16-byte staging slots per thread of a half
constexpr int G4_MAXS = 10;      // k-steps (4 taps each) per unit
// byte offset of a parked staging slot: beyond any buffer (offsets are UNSIGNED 32-bit: tensors up to 4 GB - 256 B)
constexpr int G4_OOB = (int)0xffffff00u;

// x -> (hi, rem): hi = bf16(x) round-to-nearest packed pairwise, rem = x - hi (exact).  Rounding to nearest
// matters: a truncating split biases the dropped piece products to one sign, the bias accumulates over K and
// (measured, tools/gpu_accuracy.py) is enough to flip ReLU masks like a plain fp32 summation-order change does.
__device__ inline unsigned g4_split2(float &a, float &b) {
    const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    a -= __builtin_bit_cast(float, hb << 16);
    b -= __builtin_bit_cast(float, hb & 0xffff0000u);
    return hb;
}
// last piece: the remainders have at most 8 significant bits left, so their high halves are exact
__device__ inline unsigned g4_pack2(float a, float b) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}

// BITSRC: the GEMM input is not a stored tensor: in[n, j] = [sign of element j of patch n in a.src_bits: one byte per 4
// channels, bit c of byte j / 4 for element j = 4 (j / 4) + c] * a.in[j] (the cotangent
// of a fc head's input under a patch-independent head cotangent; one patch per tile).  The staging part loads the
// vector instead of the tensor plus one mask word per slot and clears the masked elements before the split.
// FCF: the output tensor is consumed only by a 2-output fc head (pair form, one patch per tile): the epilogue
// multiplies the finished values with the head's weight difference W0 - W1 (same memory order as the output: the
// posteriors of two classes depend on the logit difference only), reduces them to one partial per (tile, wave) -
// summed in fixed order by fc_small_finish_diff - and writes the signs of its 4 channels as one byte for the
// backward pass (each lane its own byte: 4 channels, no cross-lane step); nothing else of the tensor is stored.
// The dot product of the FCF epilogue is plain C++: scalar multiplies and adds (this file is compiled with
// -fno-slp-vectorize, build.sh: packed fp32 math issues slower beside another wave's MFMAs; the build checks the device
// assembly for v_pk_mul / v_pk_fma).  Round 1's "wrong partials with packed multiplies" was refuted in round 2 (packed and
// scalar builds are bit-identical, profiles/r02_fcf_diag.txt); plain C++ keeps every read of an MFMA result (f16_combine)
// under the compiler's hazard recognizer.  test_writeback_inside_and_after_the_tick_loop_agree stays as the guard.
// The contraction into fmas is spelled out: left to -ffp-contract the compiler picks, per instantiation, which product of
// a.x * b.x + a.y * b.y becomes the fma addend, and the constant-folded and the runtime instantiation of one launch stopped
// agreeing in the last bit of the logits once the epilogue around this call differed.
__device__ inline float g4_dot4(const f32x4 &a, const f32x4 &b) {
    const float t0 = __builtin_fmaf(a.y, b.y, a.x * b.x);
    const float t1 = __builtin_fmaf(a.w, b.w, a.z * b.z);
    return t0 + t1;
}

// F16 (one column tile): fp16x2 split instead of bf16x3 - x * 2^e = h + l * 2^-11 with fp16 h, l (weights alike, packed
// by the host), h.h into `acc`, h.l + l.h into `accl`, result (acc + accl * 2^-11) * 2^-(e_in + e_w): THREE MFMAs and
// two LDS pieces per operand instead of six and three, at the accuracy of a plain fp32 GEMM for operands within 2^28 of
// the scale (tools/study_split_precision.py).  Needs max |x| of the input ahead of the launch: available for free
// where the input is [sign] * one host-known vector (BITSRC).

// launch constants a specialised instantiation may fold (everything that depends only on the layer geometry and the fusion
// choices, not on the batch, the buffers or the weights), and the pointers whose presence it may assume
#define G4_FIXED_INTS(X) X(in_cs) X(in_c0) X(out_cs) X(out_c0) X(Co) X(mask_cs) X(mask_c0) X(mask_from) X(mask_to) X(split) X(PT) X(tpg) \
    X(rows) X(PX) X(PYX) X(PZ) X(smz) X(smy) X(smx) X(soz) X(soy) X(sox) X(OD) X(OH) X(OW) X(MD) X(MH) X(MW) X(nph) X(ngr) X(NP) X(nslots) \
    X(plane_bytes) X(in_pstride) X(out_pstride) X(relu) X(accumulate) X(pair) X(store_from) X(cls_ok) X(in_split_ch) X(out_split) \
    X(mask_split) X(tt_ints) X(pd_off) X(td_off) X(wbytes) X(abytes) X(dbg_repeat) X(bits_pstride) X(fc_F) X(amax_from) X(wp) \
    X(src_presplit) X(xcd_order) X(zreuse)
#define G4_FIXED_PTRS(X) X(bias) X(mask) X(osumA) X(osumB) X(out_amax) X(in_amax) X(in_amax2) X(fc_bits) X(dbg) X(flip_list) X(mask_bits) X(sign_out)

// Launch-constant traits of a kernel instantiation.  G4Runtime (the default): every constant is read from the argument
// block.  A generated G4F_<n> (igemm4_fixed.inc, tools/gen_igemm4_fixed.py) states the constants of ONE launch of a known
// network - geometry, table offsets, fusion switches, which optional pointers are present - so that the compiler folds them:
// the generic kernel keeps ~80 arguments live in 102 SGPRs and moves the overflow through v_readlane / v_writelane inside
// the tick loop (the issue slots of an issue-bound kernel); with the constants folded the spills and the address
// multiplications disappear.  Same code path, same arithmetic, same bits.  igemm4_launch_impl picks a G4F_<n> only when
// EVERY listed constant of the launch equals the trait's (g4_matches), otherwise the runtime instantiation runs.
struct G4Runtime {
    static constexpr bool fixed = false;
#define X(f) static constexpr int f = 0;
    G4_FIXED_INTS(X)
#undef X
#define X(f) static constexpr bool has_##f = false;
    G4_FIXED_PTRS(X)
#undef X
};
#define AF(f) (G::fixed ? (int)G::f : a.f)
#define AHAS(p) (G::fixed ? (bool)G::has_##p : (a.p != nullptr))

template <int V> using IC = std::integral_constant<int, V>;
__device__ inline float g4_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ inline unsigned g4_pack_h2(_Float16 a, _Float16 b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

template <int NTW, bool MULTI, bool SUMS, bool BITSRC = false, bool FCF = false, bool FIC = false, bool F16 = false, int EPI = -1, bool ZRE = false,
          bool ACC = false, class G = G4Runtime>
__global__ __launch_bounds__(512, 1) void igemm4_kernel(const Igemm4Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds4[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform by construction: keep it scalar
    const int h = wave >> 2;          // which half (tile slot)
    const int hw = wave & 3;          // wave within the half
    const int ht = tid & 255;         // thread within the half
    const int lrow = lane & 15;
    const int lq = lane >> 4;

    int *Tl = reinterpret_cast<int *>(lds4);
    char *Wl = lds4 + AF(tt_ints) * 4;
    char *Al = Wl + AF(wbytes) + h * AF(abytes);
    {
        const char *Wg = reinterpret_cast<const char *>(a.W);
        // eight 16-byte loads in flight per thread (the rolled copy paid one L2 latency per 8 KB of weights: ~15 us
        // at the head of a launch with 86 KB resident)
        for (int i0 = tid * 16; i0 < AF(wbytes); i0 += 8 * 512 * 16) {
            i32x4 w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 512 * 16;
                w8[u] = (i < AF(wbytes)) ? *reinterpret_cast<const i32x4 *>(Wg + i) : i32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 512 * 16;
                if (i < AF(wbytes)) *reinterpret_cast<i32x4 *>(Wl + i) = w8[u];
            }
        }
        for (int i = tid; i < AF(tt_ints); i += 512) Tl[i] = a.ttab[i];
    }

    // tile / phase descriptors live in LDS next to the tap table: a broadcast ds_read costs ~100 cycles where a
    // scalar load from L2 cost 500+ per tile under load (phase stamps), and the values need no SGPRs
    auto ld4 = [&](int int_off) __attribute__((always_inline)) { return *reinterpret_cast<const i32x4 *>(Tl + int_off); };

    // ---- staging slots of this thread --------------------------------------------------------------
    int s_rel[G4_NSLOT], s_pk[G4_NSLOT], s_lds[G4_NSLOT];
#pragma unroll
    for (int it = 0; it < G4_NSLOT; ++it) {
        const int slot = ht + it * 256;
        int rel = 0x20000000, pk = AF(cls_ok) ? 0 : 0x00ff0000, ld = -1; // unused slot: past the buffer on the fast path; no class bit / hz = 255 on the checked ones
        if (slot < AF(nslots)) {
            const int4 sd = *reinterpret_cast<const int4 *>(a.sdesc + slot * 4);
            rel = sd.x * AF(in_cs) + AF(in_c0) + sd.w;
            pk = sd.y;
            ld = sd.z;
        }
        s_rel[it] = rel; s_pk[it] = pk; s_lds[it] = ld;
    }

    // ---- per-lane row geometry ------------------------------------------------------------------------
    const bool pair = AF(pair) != 0;
    const int cl = pair ? (lq & 1) * 4 : lq * 4;       // first of this lane's 4 output channels (tile 0)
    // GEMM row -> M-grid point through a host-built table: the host orders the 16 points of every MFMA
    // column block so that the 16 lanes of each ds_read_b128 lane group hit 16 different LDS rows mod 16
    int vbase[4], evox[4], eoff[4], vpk[4];
    bool erow_ok = true;
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
        const int e = a.vdesc[(hw * 4 + ms) * 16 + lrow];
        const int ee = e < 0 ? 0 : e;
        const int pt = ee >> 24, z = (ee >> 16) & 255, y = (ee >> 8) & 255, x = ee & 255;
        vbase[ms] = (pt * AF(PZ) + z * AF(smz) * AF(PYX) + y * AF(smy) * AF(PX) + x * AF(smx)) * G4_ROWB;
        evox[ms] = ((pt * AF(OD) + z * AF(soz)) * AF(OH) + y * AF(soy)) * AF(OW) + x * AF(sox) + (pair ? (lq >> 1) : 0);
        eoff[ms] = evox[ms] * AF(out_cs) + AF(out_c0);
        vpk[ms] = e;
        erow_ok = erow_ok && e >= 0;
    }
    // column -> float offset inside an output / mask row (a split concat keeps its second part delta floats away)
    int coff[NTW], mcoff[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int c = nt * 16 + cl;
        coff[nt] = (AF(out_split) && c >= AF(out_split)) ? a.out_delta + c - AF(out_split) : c;
        const int mc = c - AF(mask_from);
        mcoff[nt] = (AF(mask_split) && c >= AF(mask_split)) ? a.mask_delta + c - AF(mask_split) : mc;
    }
    f32x4 bias4[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        bias4[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = nt * 16 + cl;
        if (AHAS(bias) && c < AF(Co)) bias4[nt] = *reinterpret_cast<const f32x4 *>(a.bias + c);
    }

    // ---- this half's share of the tile list ------------------------------------------------------------
    const int pgroups = (a.N + AF(PT) - 1) / AF(PT);
    const int total = pgroups * AF(tpg);
    const int nwork = gridDim.x * 2;
    // XCD-aware work order: consecutive workgroup ids land on different XCDs (8 of them, each with its own L2), so with
    // me = 2 * blockIdx.x + h the tiles of one patch - which share halo planes - were spread over all eight L2s.
    // Logical id = (XCD, slot within the XCD): at every step an XCD owns a run of gridDim.x / 4 consecutive tiles
    // (dec2 at 32^3: exactly one patch's 64 tiles), so a halo re-read hits the L2 that already holds it.
    // Placement-independent for correctness (a bijection of the block ids); the id -> XCD map is only assumed for speed.
    int bx = blockIdx.x;
    if (AF(xcd_order)) {
        const int per = gridDim.x >> 3;
        bx = (bx & 7) * per + (bx >> 3);
    }
    const int me = bx * 2 + h;
    const int n_mine = me < total ? (total - me + nwork - 1) / nwork : 0;
    const int oth = me ^ 1;
    const int n_oth = oth < total ? (total - oth + nwork - 1) / nwork : 0;
    const int gp = nwork / AF(tpg), gl = nwork % AF(tpg);
    int fpg = me / AF(tpg), fl = me % AF(tpg);
    auto advance_cursor = [&]() __attribute__((always_inline)) {
        fl += gl;
        const int c = fl >= AF(tpg);
        fl -= c ? AF(tpg) : 0;
        fpg += gp + c;
    };

    int f_out = 0, f_full = 0, f_l = 0, f_g = 0;      // tile the prefetch cursor points at
    int c_out = 0, c_full = 0, c_l = 0, c_g = 0;      // tile being contracted
    int f_pdb = 0, c_pdb = 0;                         // first phase descriptor of that tile (several tap sets in one launch)
    unsigned f_m = 0, f_m2 = 0;                       // F16 with per-patch scales: max |x| (float bits) of the cursor tile's patch (two parts of a concat)
    int c_e = a.f16_ein;                              // F16: scale exponent of the tile being staged / contracted
    int p_out = 0, p_full = 0, p_l = 0, p_g = 0;      // tile whose results wait in registers
    bool have_pend = false;
    // MULTI (one column tile): the group BEFORE the last one also waits in registers for the staging side.  With every
    // group but the last written back between the contractions, the contracting side of the second conv_transpose
    // carried three of the four epilogues of a tile (phase stamps: contraction 46 %, staging 20 % + 27 % waiting).
    constexpr bool PEND2 = MULTI && NTW == 1;
    bool have_pend2 = false;
    int p2_out = 0;

    int goff[G4_NSLOT];
    auto park = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < G4_NSLOT; ++it) goff[it] = G4_OOB;
    };
    auto locate = [&]() __attribute__((always_inline)) {
        const i32x4 t0 = ld4(AF(td_off) + fl * 8);
        const i32x4 t1 = ld4(AF(td_off) + fl * 8 + 4);
        const int org = (t0.x + fpg * AF(in_pstride)) * AF(in_cs);
        const int tflags = __builtin_amdgcn_readfirstlane(t0.z);
        f_out = t0.y + fpg * AF(out_pstride);
        f_pdb = __builtin_amdgcn_readfirstlane(t1.z);
        f_full = (tflags & 1) && (fpg + 1) * AF(PT) <= a.N;
        f_l = fl; f_g = fpg;
        if constexpr (F16) {       // consumed one tick later (stage, a_ph == 0)
            if (AHAS(in_amax)) {
                const int pn = fpg < a.N ? fpg : 0;
                // the two loads stay in flight until the tile is staged (one tick): combining them here met their whole
                // latency once per tile
                f_m = a.in_amax[pn];
                f_m2 = AHAS(in_amax2) ? a.in_amax2[pn] : 0u;
            }
        }
        if ((tflags & 2) && (fpg + 1) * AF(PT) <= a.N) {      // whole halo inside the tensor: no per-slot checks
#pragma unroll
            for (int it = 0; it < G4_NSLOT; ++it) goff[it] = (int)((unsigned)(org + s_rel[it]) * 4u);
            return;
        }
        if (AF(cls_ok)) {         // one patch per tile, <= 3 validity classes per dimension: s_pk holds a bit per class
            const int cls = __builtin_amdgcn_readfirstlane(t1.w);
            const bool pin = fpg < a.N;
#pragma unroll
            for (int it = 0; it < G4_NSLOT; ++it)
                goff[it] = (pin && (((unsigned)s_pk[it] >> cls) & 1u)) ? (int)((unsigned)(org + s_rel[it]) * 4u) : G4_OOB;
            return;
        }
        const unsigned zy = (unsigned)t0.w, xx = (unsigned)t1.x;
        const unsigned loz = zy & 255u, nz = ((zy >> 8) & 255u) - loz;
        const unsigned loy = (zy >> 16) & 255u, ny = (zy >> 24) - loy;
        const unsigned lox = xx & 255u, nx = ((xx >> 8) & 255u) - lox;
        const int pbase = fpg * AF(PT);
#pragma unroll
        for (int it = 0; it < G4_NSLOT; ++it) {
            const unsigned pk = (unsigned)s_pk[it];
            const bool ok = (((pk >> 16) & 255u) - loz) < nz && (((pk >> 8) & 255u) - loy) < ny &&
                            ((pk & 255u) - lox) < nx && pbase + (int)(pk >> 24) < a.N;
            goff[it] = (ok && s_lds[it] >= 0) ? (int)((unsigned)(org + s_rel[it]) * 4u) : G4_OOB;
        }
    };

    // Where the prefetch of the next phase is issued (FIC).  A tick is max(stage of one half, contraction of the other);
    // with short contractions (1-2 k-steps per phase: conv_transpose classes; 8 input channels and a heavy epilogue:
    // the last conv's backward) the staging part is the longer one (phase stamps: 465 k vs 353 k cycles per half,
    // 47 % vs 20 % for the first conv_transpose), so its ~1 k cycles of load issue per phase move to the start of the
    // contraction - still ONE site per kernel: the loads then have the contraction to land in.
    constexpr bool FETCH_IN_CONTRACT = BITSRC || FIC;
    // row blocks of a tile written back by the contracting side: with three products instead of six the contraction of
    // the F16 kernel is the shorter part again (phase stamps: 302 k vs 445 k cycles per half)
    // (EPI >= 0: the launch's own balance, Igemm4Plan::tune_epi)
    // Default 0 since the epilogue meets its loads with one wait: the split that balanced the mask-bit fp16x2 launch in
    // round 1 (764 -> 627 us at batch 512) now costs it 3 % (tools/tune_sens.sh: 2634 -> 2553 us per 2000 patches without).
    constexpr int EPI_SPLIT = EPI >= 0 ? EPI : 0;
    static_assert(!(FCF || MULTI) || EPI_SPLIT == 0, "the fused-head and the multi-group epilogues are not split");
    f32x4 R[G4_NSLOT];
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.in), 0, a.in_bytes, 0x00020000);
    unsigned Rb[BITSRC ? G4_NSLOT : 1];
    const __amdgpu_buffer_rsrc_t bits_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned *>(a.src_bits), 0, BITSRC ? a.bits_bytes : 0, 0x00020000);
    // the ONE prefetch site: unconditional loads, a slot without work points past the buffer
    auto fetch = [&](int ph) __attribute__((always_inline)) {
        int soff = 0;
        if constexpr (!MULTI) {
            const i32x4 pd = ld4(AF(pd_off) + (f_pdb + ph) * 8);
            int chl = pd.y, extra = 0;
            if (AF(in_split_ch) && chl >= AF(in_split_ch)) { chl -= AF(in_split_ch); extra = a.in_delta; }     // second part of a split concat
            soff = __builtin_amdgcn_readfirstlane((int)((unsigned)(pd.x * AF(in_cs) + chl * 8 + extra) * 4u));
        }
#pragma unroll
        for (int it = 0; it < G4_NSLOT; ++it)
            R[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, goff[it], soff, 0));
        if constexpr (BITSRC) {
            const unsigned pbase = (unsigned)f_g * (unsigned)AF(bits_pstride);
#pragma unroll
            for (int it = 0; it < G4_NSLOT; ++it) {
                const unsigned e = (unsigned)(goff[it] + soff) >> 2;          // float index inside the patch (huge when parked)
                Rb[it] = (unsigned char)__builtin_amdgcn_raw_buffer_load_b8(bits_rsrc, (int)((pbase + e) >> 2), 0, 0);
            }
        }
    };
    auto stash = [&]() __attribute__((always_inline)) {
        const float f16_sc = __builtin_ldexpf(1.f, F16 ? c_e : 0), f16_sc11 = __builtin_ldexpf(1.f, F16 ? c_e + 11 : 0);
#pragma unroll
        for (int it = 0; it < G4_NSLOT; ++it) {
            if (s_lds[it] >= 0) {
                float v0 = R[it].x, v1 = R[it].y, v2 = R[it].z, v3 = R[it].w;
                if constexpr (BITSRC) {
                    if (!(F16 && AF(src_presplit))) {
                        const unsigned nib = Rb[it];       // one sign byte (low nibble) per 4 channels
                        v0 = (nib & 1u) ? v0 : 0.f; v1 = (nib & 2u) ? v1 : 0.f;
                        v2 = (nib & 4u) ? v2 : 0.f; v3 = (nib & 8u) ? v3 : 0.f;
                    }
                }
                if constexpr (BITSRC && F16) {
                    if (AF(src_presplit)) {
                        // the vector arrives ALREADY split (model.hip packs [h01 | h23 | l01 | l23] per 4 channels with the
                        // launch-wide scale when the weights are set: it is the same for every patch), so a slot is four
                        // ANDs with the sign masks instead of a select + split per value (~30 -> ~10 VALU per slot; this
                        // launch runs 7.7 VALU instructions per MFMA and the staging part is its long pole)
                        const unsigned nib = Rb[it];
                        const unsigned b0 = (unsigned)__builtin_amdgcn_sbfe((int)nib, 0, 1), b1 = (unsigned)__builtin_amdgcn_sbfe((int)nib, 1, 1);
                        const unsigned b2 = (unsigned)__builtin_amdgcn_sbfe((int)nib, 2, 1), b3 = (unsigned)__builtin_amdgcn_sbfe((int)nib, 3, 1);
                        const unsigned m01 = (b0 & 0xffffu) | (b1 & 0xffff0000u), m23 = (b2 & 0xffffu) | (b3 & 0xffff0000u);
                        const i32x4 r = __builtin_bit_cast(i32x4, R[it]);
                        char *dst = Al + s_lds[it];
                        *reinterpret_cast<uint2 *>(dst) = uint2{(unsigned)r.x & m01, (unsigned)r.y & m23};
                        *reinterpret_cast<uint2 *>(dst + 16) = uint2{(unsigned)r.z & m01, (unsigned)r.w & m23};
                        continue;
                    }
                }
                if constexpr (F16) {
                    // x * 2^e = h + l * 2^-11.  Same values as ldexp / scalar converts (every step but the two roundings to
                    // fp16 is exact), fewer instructions: the scales are two wave-uniform multipliers, the fp16 roundings
                    // go through v_cvt_pk_f16_f32 (round to nearest, two values and the packing in one instruction) and
                    // (x 2^e - h) 2^11 is one fma.  The staging part is ALU-issue bound (phase stamps): ~30 -> ~20 per slot.
                    const f16x2 h01 = __builtin_convertvector(f32x2{v0 * f16_sc, v1 * f16_sc}, f16x2);
                    const f16x2 h23 = __builtin_convertvector(f32x2{v2 * f16_sc, v3 * f16_sc}, f16x2);
                    const f32x2 g01 = __builtin_convertvector(h01, f32x2), g23 = __builtin_convertvector(h23, f32x2);
                    const f16x2 l01 = __builtin_convertvector(f32x2{__builtin_fmaf(g01.x, -2048.f, v0 * f16_sc11),
                                                                    __builtin_fmaf(g01.y, -2048.f, v1 * f16_sc11)}, f16x2);
                    const f16x2 l23 = __builtin_convertvector(f32x2{__builtin_fmaf(g23.x, -2048.f, v2 * f16_sc11),
                                                                    __builtin_fmaf(g23.y, -2048.f, v3 * f16_sc11)}, f16x2);
                    char *dst = Al + s_lds[it];
                    *reinterpret_cast<uint2 *>(dst) = uint2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
                    *reinterpret_cast<uint2 *>(dst + 16) = uint2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
                    continue;
                }
                uint2 hi, mid, lo;
                hi.x = g4_split2(v0, v1);  hi.y = g4_split2(v2, v3);
                mid.x = g4_split2(v0, v1); mid.y = g4_split2(v2, v3);
                lo.x = g4_pack2(v0, v1);   lo.y = g4_pack2(v2, v3);
                char *dst = Al + s_lds[it];
                *reinterpret_cast<uint2 *>(dst) = hi;
                *reinterpret_cast<uint2 *>(dst + 16) = mid;
                *reinterpret_cast<uint2 *>(dst + 32) = lo;
            }
        }
    };

    // ---------------- epilogue of one tile (bias already in the accumulators) --------------------------
    f32x4 acc[4][NTW];
    f32x4 acc2[(MULTI && NTW == 1) ? 4 : 1];      // MULTI: the second pending group (PEND2 below)
    f32x4 accl[F16 ? 4 : 1][NTW];      // F16: the h.l + l.h products (weight 2^-11)
    char *outb = reinterpret_cast<char *>(a.out);
    const char *maskb = reinterpret_cast<const char *>(a.mask);
    // ReLU-grad mask values of the pending tile: loaded before the staging work of the same tick so that their
    // latency is behind ~1 us of split / LDS-store instructions when the epilogue consumes them
    constexpr bool MASK_PF = (NTW == 1);      // with two column tiles the 32 extra registers spill: load in the epilogue
    f32x4 Mk[MASK_PF ? 4 : 1][NTW];
    auto load_mask = [&](int q_out, int q_full, int q_l, int q_g) __attribute__((always_inline)) {
        if constexpr (MASK_PF) {
        int mz0 = 0, my0 = 0, mx0 = 0;
        if (!q_full) {
            const i32x4 t1 = ld4(AF(td_off) + q_l * 8 + 4);
            mz0 = t1.y & 255; my0 = (t1.y >> 8) & 255; mx0 = (t1.y >> 16) & 255;
        }
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
            bool live = erow_ok;
            if (!q_full) {
                const int e = vpk[ms];
                const int pt = e >> 24, z = (e >> 16) & 255, y = (e >> 8) & 255, x = e & 255;
                live = e >= 0 && q_g * AF(PT) + pt < a.N && mz0 + z < AF(MD) && my0 + y < AF(MH) && mx0 + x < AF(MW);
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int c = nt * 16 + cl;
                Mk[ms][nt] = f32x4{1.f, 1.f, 1.f, 1.f};
                if (AHAS(mask_bits)) Mk[ms][nt].x = __builtin_bit_cast(float, 15u);      // the sign nibble travels in .x
                if (live && c < AF(Co) && c >= AF(mask_from) && c < AF(mask_to)) {
                    const int mo = (q_out + evox[ms]) * AF(mask_cs) + AF(mask_c0) + mcoff[nt];
                    if (AHAS(mask_bits)) Mk[ms][nt].x = __builtin_bit_cast(float, (unsigned)a.mask_bits[(unsigned)mo >> 2]);
                    else Mk[ms][nt] = *reinterpret_cast<const f32x4 *>(maskb + ((unsigned)mo * 4u));
                }
            }
        }
        }
    };
    // FCF: the head's weight difference for the 4 row blocks of a tile, loaded at the start of the tile's last
    // contraction like the ReLU-grad mask values (their L2 latency, exposed in the epilogue, cost as much as the
    // fusion saved)
    f32x4 fwv[FCF ? 4 : 1];
    auto fcw_prefetch = [&](int q_out, int q_full, int q_l, int q_g) __attribute__((always_inline)) {
        if constexpr (FCF) {
            int mz0 = 0, my0 = 0, mx0 = 0;
            if (!q_full) {
                const i32x4 t1 = ld4(AF(td_off) + q_l * 8 + 4);
                mz0 = t1.y & 255; my0 = (t1.y >> 8) & 255; mx0 = (t1.y >> 16) & 255;
            }
            const int jb = (q_out - q_g * AF(out_pstride)) * AF(out_cs);
#pragma unroll
            for (int ms = 0; ms < 4; ++ms) {
                bool live = erow_ok;
                if (!q_full) {
                    const int e = vpk[ms];
                    const int z = (e >> 16) & 255, y = (e >> 8) & 255, x = e & 255;
                    live = e >= 0 && q_g < a.N && mz0 + z < AF(MD) && my0 + y < AF(MH) && mx0 + x < AF(MW);
                }
                fwv[ms] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (live) fwv[ms] = *reinterpret_cast<const f32x4 *>(a.fc_W + (jb + eoff[ms] + coff[0]));
            }
        }
    };
    // F16: the two scaled accumulators of a finished tile -> its fp32 result (+ bias).  Runs at the end of the tile's
    // last contraction: with three products the contracting side is the shorter one (phase stamps)
    auto f16_combine = [&]() __attribute__((always_inline)) {
        if constexpr (F16) {
            const float inv = __builtin_ldexpf(1.f, -(c_e + a.f16_ew));
#pragma unroll
            for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    f32x4 &c = acc[ms][nt];
                    const f32x4 d = accl[ms][nt], b4 = bias4[nt];
                    c.x = g4_fma(g4_fma(d.x, 0x1p-11f, c.x), inv, b4.x);
                    c.y = g4_fma(g4_fma(d.y, 0x1p-11f, c.y), inv, b4.y);
                    c.z = g4_fma(g4_fma(d.z, 0x1p-11f, c.z), inv, b4.z);
                    c.w = g4_fma(g4_fma(d.w, 0x1p-11f, c.w), inv, b4.w);
                }
        }
    };
    // Row blocks [MS0, MS1) of a finished tile (compile-time range).  The whole tile is normally written back by the
    // staging part of the next tick; a kernel whose staging part is the longer one writes its first EPI_SPLIT row
    // blocks at the end of the contraction instead (the channel-sum accumulator travels in sacc_k).
    f32x4 sacc_k = f32x4{0.f, 0.f, 0.f, 0.f};
    float p_tau = 0.f;                // FCF + F16: flagging threshold of the pending tile (its patch's scale), see Igemm4Args::flip_tau
    float amx_k = 0.f;                // a.out_amax: max |stored value| of the row blocks written so far
    int flush_grp = 0;                // a.out_amax: output group of the running flush (MULTI)
    auto flush = [&](int q_out, int q_full, int q_l, int q_g, auto MS0, auto MS1) __attribute__((always_inline)) {
        constexpr int m0 = decltype(MS0)::value, m1 = decltype(MS1)::value;
        const int obase_e = q_out * AF(out_cs);
        int mz0 = 0, my0 = 0, mx0 = 0;
        if (!q_full) {
            const i32x4 t1 = ld4(AF(td_off) + q_l * 8 + 4);
            mz0 = t1.y & 255; my0 = (t1.y >> 8) & 255; mx0 = (t1.y >> 16) & 255;
        }
        // channel sums: ONE accumulator for the four row blocks.  The selector puts set A / set B of row block ms
        // into MFMA rows (0, 4), (1, 5), (8, 12), (9, 13), i.e. lane group q ends up with  .x / .y = the sum of
        // set (q & 1) for row blocks 2 * (q >> 1) and 2 * (q >> 1) + 1:  two 64-lane stores instead of eight
        // 16-lane ones (those, each waiting on its own MFMA, were 4 % of a pass).
        f32x4 sacc = m0 == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : sacc_k;
        float amx = m0 == 0 ? 0.f : amx_k;
        bool livem[4];
        float fs0 = 0.f;
        unsigned fbyte[FCF ? 4 : 1];
        bool fon[FCF ? 4 : 1];
        const int jbase = (q_out - q_g * AF(out_pstride)) * AF(out_cs);      // FCF: float offset of the tile inside its patch
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
            bool live = erow_ok;
            if (!q_full) {
                const int e = vpk[ms];
                const int pt = e >> 24, z = (e >> 16) & 255, y = (e >> 8) & 255, x = e & 255;
                live = e >= 0 && q_g * AF(PT) + pt < a.N && mz0 + z < AF(MD) && my0 + y < AF(MH) && mx0 + x < AF(MW);
            }
            livem[ms] = live;
        }
        // ReLU-grad mask values that were not prefetched (two column tiles): ALL loads of the flush first.  Loaded one
        // row block at a time inside the loop below, each met its full memory latency (eight times per tile: the flush
        // was 55 % of the two-column-tile backward launches - timing of a build without it).
        f32x4 Mq[(MASK_PF || FCF) ? 1 : 4][NTW];
        if constexpr (!MASK_PF && !FCF) {
            if (AHAS(mask) || AHAS(mask_bits)) {
#pragma unroll
                for (int ms = m0; ms < m1; ++ms)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const int c = nt * 16 + cl;
                        Mq[ms][nt] = f32x4{1.f, 1.f, 1.f, 1.f};
                        if (AHAS(mask_bits)) Mq[ms][nt].x = __builtin_bit_cast(float, 15u);
                        if (livem[ms] && c < AF(Co) && c >= AF(mask_from) && c < AF(mask_to)) {
                            const int mo = (q_out + evox[ms]) * AF(mask_cs) + AF(mask_c0) + mcoff[nt];
                            if (AHAS(mask_bits)) Mq[ms][nt].x = __builtin_bit_cast(float, (unsigned)a.mask_bits[(unsigned)mo >> 2]);
                            else Mq[ms][nt] = *reinterpret_cast<const f32x4 *>(maskb + ((unsigned)mo * 4u));
                        }
                    }
            }
        }
        // ONE wait for every older load (mask values, prefetched or just issued; the tile prefetch, a contraction old).
        // vmcnt retires in order and the compiler, unable to count across the branches below, met the first use of a
        // mask value in EVERY row block with vmcnt(0) - which also waits for the stores of the row block before it:
        // three or four store round trips per tile (a third of the last conv's backward launch).  After this wait no
        // load is outstanding, so the row blocks below only issue stores.
        if constexpr (!FCF) __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0), expcnt / lgkmcnt untouched
#pragma unroll
        for (int ms = m0; ms < m1; ++ms) {
            const bool live = livem[ms];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int c = nt * 16 + cl;
                f32x4 val = acc[ms][nt];
                // a fused head (FCF): igemm4_launch guarantees the pair form on exactly 8 channels with ReLU, no mask, no
                // accumulation, no output maxima and nothing of the tensor stored - those launch constants are folded at
                // compile time (each test of one keeps a kernel argument in SGPRs, of which this kernel has too few)
                const bool on = live && (FCF || c < AF(Co));
                unsigned unsure = 0u;      // FCF + F16: bit 4 of the sign byte (see below)
                if (on) {
                    f32x4 *dst = reinterpret_cast<f32x4 *>(outb + ((unsigned)(obase_e + eoff[ms] + coff[nt]) * 4u));
                    // accumulating launches (a skip source whose direct consumer is a conv) have their own instantiations: the
                    // read-modify-write path in every kernel cost 1 % of a pass it never ran in (same-box A/B of a build without it)
                    if constexpr (ACC) { if (AF(accumulate)) val += *dst; }      // (the one packed fp32 op of the kernel: v_pk_add_f32)
                    if constexpr (FCF && F16) {
                        // Flip-safe head: the fp16x2 contraction rounds every operand at 2^-22, four times the noise of an fp32
                        // GEMM, and a ReLU input within that noise of zero would get its SIGN (the mask bit the backward pass
                        // works from) decided by it.  Such values are rare (~30 of 262144 per patch): a 4-channel group holding
                        // one is marked in bit 4 of its sign byte (the consumers read bits 0-3), k_flip_fix collects the marked
                        // groups and re-evaluates them exactly from the stored fp32 inputs.  No atomics here: a returning atomic
                        // in half of the tiles' epilogues cost the launch 10 %.
                        if (AHAS(flip_list)) {
                            const float mn = fminf(fminf(__builtin_fabsf(val.x), __builtin_fabsf(val.y)),
                                                   fminf(__builtin_fabsf(val.z), __builtin_fabsf(val.w)));
                            // (exactly +0 under ZERO biases: an all-zero window, +0 in the exact evaluation too - not marked; under non-zero
                            // biases an exact +0 is acc x inv == -bias or two flushed pieces, and the exact value may be a tiny positive:
                            // marked.  The host passes the threshold negated for non-zero biases: no further kernel argument.)
                            const float tau = __builtin_fabsf(p_tau);
                            unsure = (mn < tau && (mn > 0.f || p_tau < 0.f)) ? 16u : 0u;
                        }
                    }
                    if (FCF || AF(relu)) {
                        val.x = __builtin_amdgcn_fmed3f(val.x, 0.f, __builtin_inff());
                        val.y = __builtin_amdgcn_fmed3f(val.y, 0.f, __builtin_inff());
                        val.z = __builtin_amdgcn_fmed3f(val.z, 0.f, __builtin_inff());
                        val.w = __builtin_amdgcn_fmed3f(val.w, 0.f, __builtin_inff());
                    }
                    if (!FCF && (AHAS(mask) || AHAS(mask_bits))) {
                        f32x4 mk;
                        if constexpr (MASK_PF) mk = Mk[ms][nt];
                        else mk = Mq[ms][nt];
                        if (AHAS(mask_bits)) {
                            const unsigned nb = __builtin_bit_cast(unsigned, mk.x);
                            val.x = (nb & 1u) ? val.x : 0.f; val.y = (nb & 2u) ? val.y : 0.f;
                            val.z = (nb & 4u) ? val.z : 0.f; val.w = (nb & 8u) ? val.w : 0.f;
                        } else {
                            val.x = mk.x > 0.f ? val.x : 0.f; val.y = mk.y > 0.f ? val.y : 0.f;
                            val.z = mk.z > 0.f ? val.z : 0.f; val.w = mk.w > 0.f ? val.w : 0.f;
                        }
                    }
                    if (!FCF && AHAS(sign_out))       // the output's sign field (forward launch: after the ReLU)
                        a.sign_out[(unsigned)(obase_e + eoff[ms] + coff[nt]) >> 2] =
                            (unsigned char)((val.x > 0.f ? 1u : 0u) | (val.y > 0.f ? 2u : 0u) | (val.z > 0.f ? 4u : 0u) | (val.w > 0.f ? 8u : 0u));
                    if (!FCF && c >= AF(store_from)) {
                        __builtin_nontemporal_store(val, dst);
                        if (AHAS(out_amax) && c >= AF(amax_from))
                            amx = fmaxf(fmaxf(amx, fmaxf(__builtin_fabsf(val.x), __builtin_fabsf(val.y))),
                                        fmaxf(__builtin_fabsf(val.z), __builtin_fabsf(val.w)));
                    }
                }
                if constexpr (FCF) {
                    // the sign bytes go out after the loop: no store between the weight loads above and their use
                    unsigned nib = 0;
                    if (on) {
                        fs0 += g4_dot4(val, fwv[ms]);
                        nib = (val.x > 0.f ? 1u : 0u) | (val.y > 0.f ? 2u : 0u) | (val.z > 0.f ? 4u : 0u) | (val.w > 0.f ? 8u : 0u) | unsure;
                    }
                    fbyte[ms] = nib;
                    fon[ms] = on;
                }
                if constexpr (SUMS) {
                    if (!on) val = f32x4{0.f, 0.f, 0.f, 0.f};
                    // selector MFMA: result row 0 = sum over the lane groups of set A, row 1 = set B
                    const bool inA = pair ? (lq < 2) : (c < AF(split));
                    constexpr int rowA[4] = {0, 1, 8, 9};
                    const float sel = (lrow == rowA[ms]) ? (inA ? 1.f : 0.f) : ((lrow == rowA[ms] + 4) ? (inA ? 0.f : 1.f) : 0.f);
                    sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(sel, (val.x + val.y) + (val.z + val.w), sacc, 0, 0, 0);
                }
            }
        }
        if constexpr (m1 < 4) {       // the rest of the tile follows in the staging part
            sacc_k = sacc;
            amx_k = amx;
            return;
        }
        if (!FCF && AHAS(out_amax)) {      // one plain store per wave, tile and group; k_rowmax_u32 folds the slots of a patch
            int v = __builtin_bit_cast(int, amx);
#define G4_ROW_SHR_MAX(n)                                                                                              \
    v = __builtin_bit_cast(int, fmaxf(__builtin_bit_cast(float, v),                                                    \
                                      __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x110 + (n), 0xf, 0xf, true))))
            G4_ROW_SHR_MAX(1); G4_ROW_SHR_MAX(2); G4_ROW_SHR_MAX(4); G4_ROW_SHR_MAX(8);
#undef G4_ROW_SHR_MAX
            const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 15));
            const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 31));
            const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 47));
            const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 63));
            const int ngr = MULTI ? AF(ngr) : 1;
            if (lane == 0)
                a.out_amax[((size_t)(q_g * AF(tpg) + q_l) * ngr + flush_grp) * 4 + hw] =
                    __builtin_bit_cast(unsigned, fmaxf(fmaxf(r0, r1), fmaxf(r2, r3)));
        }
        if constexpr (SUMS) {
            // lane group q: set (q & 1), row blocks m0 = 2 * (q >> 1) (in .x) and m0 + 1 (in .y)
            const bool setB = (lq & 1) != 0;
            float *base = pair ? a.osumA + (setB ? 1 : 0) : (setB ? a.osumB : a.osumA);
            const bool hi2 = lq >= 2;
            const int ev0 = hi2 ? evox[2] : evox[0], ev1 = hi2 ? evox[3] : evox[1];
            const bool l0 = hi2 ? livem[2] : livem[0], l1 = hi2 ? livem[3] : livem[1];
            // pair form: evox of this lane carries + (q >> 1) for the lanes' own voxel; the sums belong to the pair's
            // first voxel (+ 1 through `base` for the second)
            const int fix = pair ? (lq >> 1) : 0;
            if (pair ? AHAS(osumA) : (setB ? AHAS(osumB) : AHAS(osumA))) {
                if (l0) base[q_out + ev0 - fix] = sacc.x;
                if (l1) base[q_out + ev1 - fix] = sacc.y;
            }
        }
        if constexpr (FCF) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ms = 0; ms < 4; ++ms)
                if (fon[ms] && AHAS(fc_bits))
                    a.fc_bits[(size_t)q_g * (AF(fc_F) >> 2) + ((jbase + eoff[ms] + coff[0]) >> 2)] = (unsigned char)fbyte[ms];
            // wave sum without the LDS crossbar (six dependent ds_bpermute round trips were ~1 k cycles per tile): prefix
            // sums inside each row of 16 lanes with DPP shifts, then the four row totals through scalar registers
            int v = __builtin_bit_cast(int, fs0);
#define G4_ROW_SHR_ADD(n)                                                                                              \
    v = __builtin_bit_cast(int, __builtin_bit_cast(float, v) +                                                         \
                                    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x110 + (n), 0xf, 0xf, true)))
            G4_ROW_SHR_ADD(1); G4_ROW_SHR_ADD(2); G4_ROW_SHR_ADD(4); G4_ROW_SHR_ADD(8);
#undef G4_ROW_SHR_ADD
            const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 15));
            const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 31));
            const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 47));
            const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 63));
            if (lane == 0) a.fc_part[(size_t)(q_g * AF(tpg) + q_l) * 4 + hw] = (r0 + r1) + (r2 + r3);
        }
    };

    // ---------------- one unit: S k-steps of 4 taps x 8 channels, fragment reads one half-step ahead ----
    auto unit = [&](int S, const char *Wc, const char *Ab, int trow) __attribute__((always_inline)) {
        const int *tt = Tl + trow * (G4_MAXS * 4) + lq;
        bf16x8 Wa[3][NTW], Wb[NTW >= 2 ? 1 : 3][NTW], Xa[3][2], Xb[3][2];
        auto rdW = [&](bf16x8 (&Wf)[3][NTW], int s) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    Wf[p][nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const i32x4 *>(Wc + ((s * 3 + p) * NTW + nt) * 1024));
        };
        auto rdX = [&](bf16x8 (&X)[3][2], int mh, int to) {
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                const char *row = Ab + vbase[mh + m2] + to;
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    X[p][m2] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const i32x4 *>(row + 16 * p));
            }
        };
        // six piece products, smallest weights first, the two row blocks alternating (independent chains)
        auto mm = [&](const bf16x8 (&Wf)[3][NTW], const bf16x8 (&X)[3][2], int mh) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                f32x4 c0 = acc[mh][nt], c1 = acc[mh + 1][nt];
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[1][nt], X[1][0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[1][nt], X[1][1], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[2][nt], X[0][0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[2][nt], X[0][1], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[0][nt], X[2][0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[0][nt], X[2][1], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[1][nt], X[0][0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[1][nt], X[0][1], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[0][nt], X[1][0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[0][nt], X[1][1], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[0][nt], X[0][0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[0][nt], X[0][1], c1, 0, 0, 0);
                acc[mh][nt] = c0; acc[mh + 1][nt] = c1;
            }
        };
        int t_cur = tt[0];
        rdX(Xa, 0, t_cur);
        if constexpr (NTW >= 2) {
            // two column tiles: one weight fragment set only (a second set spills registers); its six reads are
            // exposed once per k-step of 48 MFMAs
            for (int s = 0; s < S; ++s) {
                const int s1 = s + 1 < S ? s + 1 : s;
                const int t_nxt = tt[s1 * 4];
                rdW(Wa, s);
                rdX(Xb, 2, t_cur);
                __builtin_amdgcn_sched_barrier(0);
                mm(Wa, Xa, 0);
                __builtin_amdgcn_sched_barrier(0);
                rdX(Xa, 0, t_nxt);
                __builtin_amdgcn_sched_barrier(0);
                mm(Wa, Xb, 2);
                __builtin_amdgcn_sched_barrier(0);
                t_cur = t_nxt;
            }
        } else {
            rdW(Wa, 0);
            auto kstep = [&](const bf16x8 (&Wc_)[3][NTW], bf16x8 (&Wn_)[3][NTW], int s) {
                const int s1 = s + 1 < S ? s + 1 : s;
                const int t_nxt = tt[s1 * 4];
                rdX(Xb, 2, t_cur);
                __builtin_amdgcn_sched_barrier(0);
                mm(Wc_, Xa, 0);
                __builtin_amdgcn_sched_barrier(0);
                rdW(Wn_, s1);
                rdX(Xa, 0, t_nxt);
                __builtin_amdgcn_sched_barrier(0);
                mm(Wc_, Xb, 2);
                __builtin_amdgcn_sched_barrier(0);
                t_cur = t_nxt;
            };
            int s = 0;
            for (; s + 1 < S; s += 2) {
                kstep(Wa, Wb, s);
                kstep(Wb, Wa, s + 1);
            }
            if (s < S) kstep(Wa, Wb, s);
        }
    };
    auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ms = 0; ms < 4; ++ms)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                if constexpr (F16) {      // scaled sums: the bias joins in the epilogue
                    acc[ms][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    accl[ms][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    acc[ms][nt] = bias4[nt];
                }
            }
    };
    // F16 twin of `unit` (one column tile): two fragment pieces per operand, three products
    auto unit16 = [&](int S, const char *Wc, const char *Ab, int trow) __attribute__((always_inline)) {
        if constexpr (F16) {
            const int *tt = Tl + trow * (G4_MAXS * 4) + lq;
            const int wpieces = AF(wp);       // 3: the fp16 pieces sit in the first two of the bf16x3 slots; 2: an fp16x2-only plan
            f16x8 Wa[2][NTW], Wb[NTW >= 2 ? 1 : 2][NTW], Xa[2][2], Xb[2][2];
            auto rdW = [&](f16x8 (&Wf)[2][NTW], int s) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        Wf[p][nt] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(Wc + ((s * wpieces + p) * NTW + nt) * 1024));
            };
            auto rdX = [&](f16x8 (&X)[2][2], int mh, int to) {
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    const char *row = Ab + vbase[mh + m2] + to;
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        X[p][m2] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(row + 16 * p));
                }
            };
            auto mm = [&](const f16x8 (&Wf)[2][NTW], const f16x8 (&X)[2][2], int mh) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    f32x4 c0 = acc[mh][nt], c1 = acc[mh + 1][nt], d0 = accl[mh][nt], d1 = accl[mh + 1][nt];
                    d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[1][nt], X[0][0], d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[1][nt], X[0][1], d1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[0][nt], X[0][0], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[0][nt], X[0][1], c1, 0, 0, 0);
                    d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[0][nt], X[1][0], d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[0][nt], X[1][1], d1, 0, 0, 0);
                    acc[mh][nt] = c0; acc[mh + 1][nt] = c1; accl[mh][nt] = d0; accl[mh + 1][nt] = d1;
                }
            };
            int t_cur = tt[0];
#if defined(ALQ_ZRE_DEEP) && ALQ_ZRE_DEEP
            if constexpr (!(ZRE && NTW == 1)) rdX(Xa, 0, t_cur);
#else
            rdX(Xa, 0, t_cur);
#endif
            if constexpr (NTW >= 2) {       // one weight fragment set, as in `unit`
                for (int s = 0; s < S; ++s) {
                    const int s1 = s + 1 < S ? s + 1 : s;
                    const int t_nxt = tt[s1 * 4];
                    rdW(Wa, s);
                    rdX(Xb, 2, t_cur);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wa, Xa, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    rdX(Xa, 0, t_nxt);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wa, Xb, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    t_cur = t_nxt;
                }
            } else if constexpr (ZRE) {
                // Fragment reuse: k-step 3 * iz + j (the host packs the taps plane by plane) reads for row blocks 2 / 3 what
                // k-step 3 * (iz + 1) + j reads for row blocks 0 / 1, so per column j of three k-steps the fragments are read
                // four times instead of six.  Same products, summed in the order (j, iz) instead of (iz, j).
#if defined(ALQ_ZRE_DEEP) && ALQ_ZRE_DEEP       // opt-in (-DALQ_ZRE_DEEP=1): same-box kernel A/B 2405 -> 2476 us, i.e. NO gain - the fragment reads'
                // latency is not what holds this contraction back (the two waves of a SIMD add their issue cycles)
                // The three columns unrolled over THREE fragment register sets: fragment n = 4 j + i (i = 0: rows 0 / 1 at
                // plane 0, 1: rows 2 / 3 at plane 0 = rows 0 / 1 at plane 1, 2: ... plane 1 / 2, 3: rows 2 / 3 at plane 2) lives
                // in set n mod 3, whose previous tenant n - 3 has just had its last use when n is issued.  Every fragment
                // read is then issued at least TWO six-MFMA groups (>= 192 cycles) ahead of its first use, every weight read
                // four; with two sets the distance was one group (96 cycles), less than the LDS round trip while the other
                // half stages (PMC: a third of the wave cycles of this launch waiting, LDS conflicts or not).
                f16x8 Xs[3][2][2], Wc2[2][NTW];
                // the nine tap offsets of the phase in ONE batch of table reads ahead of the first fragment read: a table read
                // between two fragment reads is met with lgkmcnt(0) (the counter retires in order), i.e. it drains every
                // fragment read in flight and the distance gained above is lost
                int tapv[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) tapv[q] = tt[q * 4];
                auto tj = [&](int j, int iz) __attribute__((always_inline)) { return tapv[3 * iz + j]; };
                auto issue = [&](auto N) __attribute__((always_inline)) {          // fragment n = 4 j + i into set n mod 3
                    constexpr int n = decltype(N)::value, j = n >> 2, i = n & 3;
                    if constexpr (j < 3) rdX(Xs[n % 3], i == 0 ? 0 : 2, tj(j, i == 0 ? 0 : i - 1));
                };
                rdW(Wa, 0);
                issue(IC<0>{});
                issue(IC<1>{});
                rdW(Wb, 3);
                auto column = [&](auto J) __attribute__((always_inline)) {
                    constexpr int j = decltype(J)::value, n0 = 4 * j;
                    issue(IC<n0 + 2>{});
                    rdW(Wc2, 6 + j);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wa, Xs[n0 % 3], 0);
                    __builtin_amdgcn_sched_barrier(0);
                    issue(IC<n0 + 3>{});
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wa, Xs[(n0 + 1) % 3], 2);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (j < 2) rdW(Wa, j + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wb, Xs[(n0 + 1) % 3], 0);
                    __builtin_amdgcn_sched_barrier(0);
                    issue(IC<n0 + 4>{});
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wb, Xs[(n0 + 2) % 3], 2);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (j < 2) rdW(Wb, 3 + j + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wc2, Xs[(n0 + 2) % 3], 0);
                    __builtin_amdgcn_sched_barrier(0);
                    issue(IC<n0 + 5>{});
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wc2, Xs[(n0 + 3) % 3], 2);
                    __builtin_amdgcn_sched_barrier(0);
                };
                column(IC<0>{});
                column(IC<1>{});
                column(IC<2>{});
#else
                f16x8 Wc2[2][NTW];                      // third weight set: fixed roles, so the three columns are a rolled loop
                rdW(Wa, 0);
#pragma unroll 1
                for (int j = 0; j < 3; ++j) {
                    // enters with Xa = the fragments of k-step j for row blocks 0 / 1 in flight, Wa = its weights
                    const int t0 = tt[j * 4], t1 = tt[(3 + j) * 4], t2 = tt[(6 + j) * 4];
                    const int jn = j + 1 < 3 ? j + 1 : j;
                    const int t0n = tt[jn * 4];
                    rdX(Xb, 2, t0);                      // row blocks 2 / 3 at plane 0 = row blocks 0 / 1 at plane 1
                    rdW(Wb, 3 + j);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wa, Xa, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    rdX(Xa, 2, t1);                      // row blocks 2 / 3 at plane 1 = row blocks 0 / 1 at plane 2
                    rdW(Wc2, 6 + j);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wa, Xb, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    rdW(Wa, jn);                         // the next column's first weights
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wb, Xb, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    rdX(Xb, 2, t2);                      // row blocks 2 / 3 at plane 2
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wb, Xa, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wc2, Xa, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    rdX(Xa, 0, t0n);                     // the next column's first fragments
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wc2, Xb, 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
            } else {
                rdW(Wa, 0);
                auto kstep = [&](const f16x8 (&Wc_)[2][NTW], f16x8 (&Wn_)[2][NTW], int s) {
                    const int s1 = s + 1 < S ? s + 1 : s;
                    const int t_nxt = tt[s1 * 4];
                    rdX(Xb, 2, t_cur);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wc_, Xa, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    rdW(Wn_, s1);
                    rdX(Xa, 0, t_nxt);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(Wc_, Xb, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    t_cur = t_nxt;
                };
                int s = 0;
                for (; s + 1 < S; s += 2) {
                    kstep(Wa, Wb, s);
                    kstep(Wb, Wa, s + 1);
                }
                if (s < S) kstep(Wa, Wb, s);
            }
        }
    };

    // ---------------- tick loop --------------------------------------------------------------------------
    // Both halves run the same straight-line body  [contract | barrier | stage | barrier];  half 1 starts one
    // barrier late, so that one half contracts while the other stages.  No branch on the tick parity: with one
    // the prefetched registers lived on two control-flow paths and the compiler merged them with copies - which
    // wait for the loads - at the head of every contraction (igemm3.hip has the same lesson).
    // Waits: vmcnt retires in issue order and the compiler waits conservatively (vmcnt(0)), so every wait must
    // only meet operations that are at least one tick old.  Per half the order is
    //   stage(i):    split (prefetch of tick i-1, mask loads of contract(i-1)) -> epilogue stores -> prefetch
    //   contract(i): mask loads of the tile (last phase only) -> MFMAs
    const int n_ph = n_mine * AF(nph);     // staging phases of this half
    const int n_iter = (n_mine > n_oth ? n_mine : n_oth) * AF(nph);
    int a_i = 0, a_ph = 0;               // next phase to stage (counter, index within the tile)
    int b_i = 0, b_ph = 0;               // next phase to contract
    __syncthreads();                     // the descriptor / tap tables (and weights) copied above are read from here on
    if (n_mine > 0) locate(); else park();
    fetch(0);
#ifdef ALQ_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last;
    STAMP4(t_last);
#endif
    auto stage = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(0);
        PHASE4_END(5);
        int nph = 0;
        if (a_i < n_ph) {
            if (a_ph == 0) {
                c_out = f_out; c_full = f_full; c_l = f_l; c_g = f_g; c_pdb = f_pdb;
                if constexpr (F16) {
                    if (AHAS(in_amax)) {      // max |x| < 2^ex  ->  scale 2^(14 - ex); an all-zero patch keeps 0
                        const unsigned fm = max(f_m, f_m2);
                        const int ex = (int)((fm >> 23) & 255u) - 126;
                        c_e = fm ? 14 - ex : 0;
                    }
                }
            }
            stash();
            PHASE4_END(0);
            if (a_ph == 0 && have_pend) {
                flush(p_out, p_full, p_l, p_g, IC<EPI_SPLIT>{}, IC<4>{});
                have_pend = false;
                if constexpr (PEND2) {
                    if (have_pend2) {
#pragma unroll
                        for (int ms = 0; ms < 4; ++ms) acc[ms][0] = acc2[ms];
                        flush_grp = AF(ngr) - 2;
                        flush(p2_out, p_full, p_l, p_g, IC<0>{}, IC<4>{});
                        flush_grp = AF(ngr) - 1;
                        have_pend2 = false;
                    }
                }
            }
            PHASE4_END(1);
            nph = a_ph + 1;
            if (nph == AF(nph)) {
                nph = 0;
                advance_cursor();
                if (a_i + 1 < n_ph) locate(); else park();
            }
            a_ph = nph;
            ++a_i;
        }
        if constexpr (!FETCH_IN_CONTRACT) fetch(nph);       // unconditional: parked slots read nothing
        PHASE4_END(2);
    };
    auto contract = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(3);       // the contracting wave goes first on its SIMD; staging fills the gaps
        PHASE4_END(4);
        // still the ONE prefetch site of the kernel, on the side that has the slack (see FETCH_IN_CONTRACT)
        if constexpr (FETCH_IN_CONTRACT) fetch(a_i < n_ph ? a_ph : 0);
        if (b_i < a_i) {
            if constexpr (!MULTI) {
                const i32x4 pdA = ld4(AF(pd_off) + (c_pdb + b_ph) * 8), pdB = ld4(AF(pd_off) + (c_pdb + b_ph) * 8 + 4);
                const int pd[5] = {0, 0, __builtin_amdgcn_readfirstlane(pdA.z), pdA.w, pdB.x};
                if constexpr (MASK_PF) { if (!FCF && b_ph == AF(nph) - 1 && (AHAS(mask) || AHAS(mask_bits))) load_mask(c_out, c_full, c_l, c_g); }
                // the contracting side has the slack (phase stamps: with the prefetch in the staging part that part was
                // the longer one and the other half waited for it at the barrier)
                if constexpr (FCF) { if (b_ph == AF(nph) - 1) fcw_prefetch(c_out, c_full, c_l, c_g); }
                if (b_ph == 0) init_acc();
                for (int rep = 0; rep <= AF(dbg_repeat); ++rep) {
                    if constexpr (F16) unit16(pd[2], Wl + pd[3] + lane * 16, Al, pd[4]);
                    else unit(pd[2], Wl + pd[3] + lane * 16, Al, pd[4]);
                }
                if (b_ph == AF(nph) - 1) {
                    if constexpr (FCF && F16) p_tau = __builtin_ldexpf(a.flip_tau, 14 - c_e);      // max |x| of the patch < 2^(14 - c_e)
                    f16_combine();
                    if constexpr (EPI_SPLIT > 0) flush(c_out, c_full, c_l, c_g, IC<0>{}, IC<EPI_SPLIT>{});
                    have_pend = true;
                    p_out = c_out; p_full = c_full; p_l = c_l; p_g = c_g;
                }
                b_ph = b_ph + 1 == AF(nph) ? 0 : b_ph + 1;
            } else {
                for (int g = 0; g < AF(ngr); ++g) {
                    const i32x4 gdA = ld4(AF(pd_off) + g * 8), gdB = ld4(AF(pd_off) + g * 8 + 4);
                    const int gd[6] = {0, 0, __builtin_amdgcn_readfirstlane(gdA.z), gdA.w, gdB.x, gdB.y};
                    init_acc();
                    const int ub = gd[2] * (3 * NTW * 1024);
                    for (int rep = 0; rep <= AF(dbg_repeat); ++rep)
                        for (int p = 0; p < AF(NP); ++p)
                            unit(gd[2], Wl + gd[3] + p * ub + lane * 16, Al + p * AF(plane_bytes), gd[4]);
                    bool kept = false;
                    if constexpr (PEND2) {
                        if (g + 2 == AF(ngr)) {
#pragma unroll
                            for (int ms = 0; ms < 4; ++ms) acc2[ms] = acc[ms][0];
                            have_pend2 = true;
                            p2_out = c_out + gd[5];
                            kept = true;
                        }
                    }
                    if (kept) {
                    } else if (g + 1 < AF(ngr)) {
                        flush_grp = g;
                        flush(c_out + gd[5], c_full, c_l, c_g, IC<0>{}, IC<4>{});
                        flush_grp = AF(ngr) - 1;       // the group left pending
                    } else {
                        have_pend = true;
                        p_out = c_out + gd[5]; p_full = c_full; p_l = c_l; p_g = c_g;
                    }
                }
            }
            ++b_i;
            PHASE4_END(3);
        }
    };
    if (h == 1) __syncthreads();
    stage();
    __syncthreads();
    for (int it = 0; it < n_iter; ++it) {
        contract();
        __syncthreads();
        stage();
        __syncthreads();
    }
    if (h == 0) __syncthreads();
    if (have_pend) {
        flush(p_out, p_full, p_l, p_g, IC<EPI_SPLIT>{}, IC<4>{});
        if constexpr (PEND2) {
            if (have_pend2) {
#pragma unroll
                for (int ms = 0; ms < 4; ++ms) acc[ms][0] = acc2[ms];
                flush_grp = AF(ngr) - 2;
                flush(p2_out, p_full, p_l, p_g, IC<0>{}, IC<4>{});
            }
        }
    }
#ifdef ALQ_STAMPS
    PHASE4_END(6);
    if (a.dbg && (tid & 255) == 0)
        for (int i = 0; i < 8; ++i) a.dbg[(bx * 2 + h) * 8 + i] = ph[i];
#endif
}

// ======================================================================================================
// host: plan builder
// ======================================================================================================
static int g4_pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static int g4_floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
static int g4_ceildiv(int a, int b) { return -g4_floordiv(-a, b); }

namespace {
struct DimGeo {
    int I = 1, O = 1, M = 1;     // input / output / M-grid extent
    int hs = 1;                  // input coordinate step per halo index
    int sm = 1;                  // M-grid point -> halo index multiplier
    int bm = 1, bo = 0;          // input coordinate of halo index 0 of the tile at m0: m0*bm + bo
    int span = 1;                // halo indices covered by the taps of one M-grid point
    int so = 1;                  // M-grid point -> output coordinate multiplier
};
struct Box { int b[3] = {0, 0, 0}, n[3] = {1, 1, 1}; };
}  // namespace

int igemm4_build_plan(const G4Geom &g, int max_batch, Igemm4Plan *plan, int wp) {
    const int WP = wp == 2 ? 2 : 3;       // weight pieces kept in LDS per k-step: 3 = bf16x3 (and its fp16x2 twin in the same slots), 2 = fp16x2 only
    plan->ok = false;
    plan->units.clear();
    if (g.Ci % 8 != 0 || g.Co % 4 != 0 || g.Ci < 8) return ALQ_OK;
    const int I[3] = {g.ID, g.IH, g.IW}, O[3] = {g.OD, g.OH, g.OW};
    const int NT = (g.Co + 15) / 16;
    if (NT > 2) return ALQ_OK;
    const int NTW = NT;
    const int NCH = g.Ci / 8;
    DimGeo dg[3];
    bool pair = false;
    const int *k = g.k, *s = g.s, *lo = g.lo;

    // class structure per dimension (kinds 1, 2)
    int amin[3] = {0, 0, 0}, amax[3] = {0, 0, 0};
    auto a_range = [&](int d, int r, int *a0, int *a1) {      // kind 1: taps t = s*a + r + lo in [0, k)
        *a0 = g4_ceildiv(-r - lo[d], s[d]);
        *a1 = g4_floordiv(k[d] - 1 - r - lo[d], s[d]);
    };
    auto n_range = [&](int d, int c, int *n0, int *n1) {      // kind 2: taps t = c + lo - s*n in [0, k)
        *n0 = g4_ceildiv(c + lo[d] - k[d] + 1, s[d]);
        *n1 = g4_floordiv(c + lo[d], s[d]);
    };
    if (g.kind == 0) {
        for (int d = 0; d < 3; ++d) {
            if (s[d] != 1 || I[d] != O[d]) return ALQ_OK;
            dg[d].I = I[d]; dg[d].O = O[d]; dg[d].M = O[d];
            dg[d].bo = g.flipped ? lo[d] - (k[d] - 1) : -lo[d];
            dg[d].span = k[d];
        }
        // pair form: 8 output channels, two x-adjacent voxels per 16 MFMA rows, (k+1)-wide x window
        if (g.Co == 8 && O[2] % 2 == 0 && O[2] >= 4) {
            pair = true;
            dg[2].M = O[2] / 2; dg[2].sm = 2; dg[2].bm = 2; dg[2].so = 2; dg[2].span = k[2] + 1;
        }
    } else if (g.kind == 1) {
        for (int d = 0; d < 3; ++d) {
            if (I[d] != O[d] * s[d] || k[d] < s[d]) return ALQ_OK;
            dg[d].I = I[d]; dg[d].O = O[d]; dg[d].M = O[d];
            if (d < 2) {
                int lo_a = 1 << 30, hi_a = -(1 << 30);
                for (int r = 0; r < s[d]; ++r) {
                    int a0, a1;
                    a_range(d, r, &a0, &a1);
                    if (a1 < a0) return ALQ_OK;
                    lo_a = std::min(lo_a, a0); hi_a = std::max(hi_a, a1);
                }
                amin[d] = lo_a; amax[d] = hi_a;
                dg[d].hs = s[d]; dg[d].sm = 1; dg[d].bm = s[d]; dg[d].bo = s[d] * lo_a; dg[d].span = hi_a - lo_a + 1;
            } else {
                dg[d].hs = 1; dg[d].sm = s[d]; dg[d].bm = s[d]; dg[d].bo = -lo[d]; dg[d].span = k[d];
            }
        }
    } else if (g.kind == 2 || g.kind == 4) {
        for (int d = 0; d < 3; ++d) {
            if (O[d] != I[d] * s[d] || k[d] < s[d]) return ALQ_OK;
            dg[d].I = I[d]; dg[d].O = O[d]; dg[d].M = I[d];
            int lo_n = 1 << 30, hi_n = -(1 << 30);
            for (int c = 0; c < s[d]; ++c) {
                int n0, n1;
                n_range(d, c, &n0, &n1);
                if (n1 < n0) return ALQ_OK;
                lo_n = std::min(lo_n, n0); hi_n = std::max(hi_n, n1);
            }
            amin[d] = lo_n; amax[d] = hi_n;
            dg[d].bo = lo_n; dg[d].span = hi_n - lo_n + 1; dg[d].so = s[d];
        }
        pair = (g.kind == 2 && g.Co == 8 && s[2] == 2);
    } else if (g.kind == 3) {
        for (int d = 0; d < 3; ++d) {
            if (O[d] != I[d] * s[d] || k[d] < s[d] || g.cls[d] < 0 || g.cls[d] >= s[d]) return ALQ_OK;
            dg[d].I = I[d]; dg[d].O = O[d]; dg[d].M = I[d];
            int n0, n1;
            n_range(d, g.cls[d], &n0, &n1);
            if (n1 < n0) return ALQ_OK;
            amin[d] = n0; amax[d] = n1;
            dg[d].bo = n0; dg[d].span = n1 - n0 + 1; dg[d].so = s[d];
        }
    } else {
        return ALQ_OK;
    }
    if (pair && NTW != 1) return ALQ_OK;

    // ---- tap boxes of every unit row (a row = one tap table: conv 1, conv_transpose one per class) ------
    struct URow {
        Box box;
        std::function<int(int, int, int, int)> tapof;     // (iz, iy, ix, half) -> tap of the k^3 enumeration or -1
        int ntaps = 0, S = 0;
        int in_off[3] = {0, 0, 0}, out_off[3] = {0, 0, 0};
        std::vector<int> tap;                              // [(s*4+q)*2+half] after the k-step assignment
        std::vector<int> toff;                             // [s*4+q] LDS row offset
    };
    std::vector<URow> rows;
    auto enum_tap = [=](int tz, int ty, int tx) {
        if (tz < 0 || tz >= k[0] || ty < 0 || ty >= k[1] || tx < 0 || tx >= k[2]) return -1;
        return (tz * k[1] + ty) * k[2] + tx;
    };
    const bool flipped = g.flipped;
    const int kk[3] = {k[0], k[1], k[2]}, ss[3] = {s[0], s[1], s[2]}, ll[3] = {lo[0], lo[1], lo[2]};
    if (g.kind == 0) {
        URow r;
        for (int d = 0; d < 3; ++d) { r.box.b[d] = 0; r.box.n[d] = dg[d].span; }
        const int bo[3] = {dg[0].bo, dg[1].bo, dg[2].bo};
        r.tapof = [=](int iz, int iy, int ix, int half) {
            if (half && !pair) return -1;
            const int o[3] = {bo[0] + iz, bo[1] + iy, bo[2] + ix - (half ? 1 : 0)};
            int t[3];
            for (int d = 0; d < 3; ++d) t[d] = flipped ? ll[d] - o[d] : o[d] + ll[d];
            return enum_tap(t[0], t[1], t[2]);
        };
        rows.push_back(r);
    } else if (g.kind == 1) {
        for (int rz = 0; rz < s[0]; ++rz)
            for (int ry = 0; ry < s[1]; ++ry) {
                URow r;
                int a0z, a1z, a0y, a1y;
                a_range(0, rz, &a0z, &a1z);
                a_range(1, ry, &a0y, &a1y);
                r.box.b[0] = a0z - amin[0]; r.box.n[0] = a1z - a0z + 1;
                r.box.b[1] = a0y - amin[1]; r.box.n[1] = a1y - a0y + 1;
                r.box.b[2] = 0; r.box.n[2] = k[2];
                r.in_off[0] = rz; r.in_off[1] = ry;
                r.tapof = [=](int iz, int iy, int ix, int half) {
                    if (half) return -1;
                    return enum_tap(ss[0] * (a0z + iz) + rz + ll[0], ss[1] * (a0y + iy) + ry + ll[1], ix);
                };
                rows.push_back(r);
            }
    } else if (g.kind == 3) {
        URow r;
        for (int d = 0; d < 3; ++d) { r.box.b[d] = 0; r.box.n[d] = dg[d].span; r.out_off[d] = g.cls[d]; }
        const int c0 = g.cls[0], c1 = g.cls[1], c2 = g.cls[2], nz0 = amin[0], ny0 = amin[1], nx0 = amin[2];
        r.tapof = [=](int iz, int iy, int ix, int half) {
            if (half) return -1;
            return enum_tap(c0 + ll[0] - ss[0] * (nz0 + iz), c1 + ll[1] - ss[1] * (ny0 + iy), c2 + ll[2] - ss[2] * (nx0 + ix));
        };
        rows.push_back(r);
    } else {
        const int ncx = pair ? 1 : s[2];
        for (int cz = 0; cz < s[0]; ++cz)
            for (int cy = 0; cy < s[1]; ++cy)
                for (int cx = 0; cx < ncx; ++cx) {
                    URow r;
                    int n0[3], n1[3];
                    n_range(0, cz, &n0[0], &n1[0]);
                    n_range(1, cy, &n0[1], &n1[1]);
                    if (pair) {
                        int m0, m1;
                        n_range(2, 0, &n0[2], &n1[2]);
                        n_range(2, 1, &m0, &m1);
                        n0[2] = std::min(n0[2], m0); n1[2] = std::max(n1[2], m1);
                    } else {
                        n_range(2, cx, &n0[2], &n1[2]);
                    }
                    for (int d = 0; d < 3; ++d) { r.box.b[d] = n0[d] - amin[d]; r.box.n[d] = n1[d] - n0[d] + 1; }
                    r.out_off[0] = cz; r.out_off[1] = cy; r.out_off[2] = cx;
                    const int nz0 = n0[0], ny0 = n0[1], nx0 = n0[2];
                    r.tapof = [=](int iz, int iy, int ix, int half) {
                        if (half && !pair) return -1;
                        const int ccx = pair ? half : cx;
                        return enum_tap(cz + ll[0] - ss[0] * (nz0 + iz), cy + ll[1] - ss[1] * (ny0 + iy),
                                        ccx + ll[2] - ss[2] * (nx0 + ix));
                    };
                    rows.push_back(r);
                }
    }
    (void)kk;
    for (URow &r : rows) {
        r.ntaps = r.box.n[0] * r.box.n[1] * r.box.n[2];
        if ((r.ntaps + 3) / 4 + 1 > G4_MAXS) return ALQ_OK;
    }
    const bool multi = (g.kind == 2);
    const int ncls = g.kind == 4 ? (int)rows.size() : 1;       // kind 4: every class is its own set of tiles
    const int nstage = multi ? 1 : (g.kind == 4 ? NCH : (int)rows.size() * NCH);      // staging phases per tile
    const int NPs = multi ? NCH : 1;                             // planes staged per phase
    const size_t tt_ints = (size_t)rows.size() * G4_MAXS * 4;

    // ---- LDS layout of the staged block ----------------------------------------------------------------
    // Every ds_read_b128 lane group is 8 lanes of one tap + 8 lanes of the neighbouring tap of the k-step;
    // a 48-byte row occupies 3 of 16 16-byte bank granules, so the group is conflict-free iff its 16 rows
    // differ mod 16.  The 16 M-grid points of a column block are therefore chosen as a complete residue
    // system of LDS rows mod 16 (16 x-neighbours, or 8 x-neighbours on two lines whose pitch is padded to
    // 8 mod 16; with an x multiplier of 2: two lines an odd pitch apart), the even rows on lanes 0-3 / 12-15,
    // the odd rows on lanes 4-11, and the two taps sharing a lane group are paired with an even row distance.
    struct Lay { int PX = 0, PYX = 0, PZ = 0, xw = 0, ud = -1; };
    auto roundup_mod = [](int v, int r, int m) { int w = v; while (((w % m) + m) % m != r) ++w; return w; };
    auto layout = [&](int PT, const int T[3], const int H[3], bool natural = false) {
        Lay L;
        L.PX = H[2]; L.PYX = H[1] * L.PX; L.PZ = H[0] * L.PYX;
        const bool unit_zy = dg[0].sm == 1 && dg[1].sm == 1;
        if (!unit_zy || natural) return L;
        if (dg[2].sm == 1) {
            if (T[2] % 16 == 0) { L.xw = 16; return L; }
            if (T[2] != 8) return L;
            L.xw = 8;
            if (T[0] >= 2) { L.ud = 0; L.PYX = roundup_mod(H[1] * L.PX, 8, 16); }
            else if (PT >= 2) { L.ud = 3; L.PZ = 0; }
            else if (T[1] >= 2) { L.ud = 1; L.PX = roundup_mod(H[2], 8, 16); L.PYX = H[1] * L.PX; }
            else { L.xw = 0; return L; }
            L.PZ = H[0] * L.PYX;
            if (L.ud == 3) L.PZ = roundup_mod(L.PZ, 8, 16);
        } else if (dg[2].sm == 2) {
            if (T[2] == 4 && T[1] >= 2 && T[0] >= 2 && T[1] % 2 == 0 && T[0] % 2 == 0) {
                // 4 x-neighbours (rows 0, 2, 4, 6) on two lines an ODD pitch apart and two planes a pitch of 8 mod 16 apart:
                // {0,2,4,6} + {0, odd, 8, odd + 8} is a complete residue system mod 16, half of it even
                L.xw = 4; L.ud = 4;
                L.PX = H[2] | 1;
                L.PYX = roundup_mod(H[1] * L.PX, 8, 16);
                L.PZ = H[0] * L.PYX;
                return L;
            }
            if (T[2] % 8 != 0) return L;
            L.xw = 8;
            if (T[1] >= 2) { L.ud = 1; L.PX = H[2] | 1; L.PYX = H[1] * L.PX; }
            else if (T[0] >= 2) { L.ud = 0; L.PYX = (H[1] * L.PX) | 1; }
            else if (PT >= 2) { L.ud = 3; }
            else { L.xw = 0; return L; }
            L.PZ = H[0] * L.PYX;
            if (L.ud == 3) L.PZ |= 1;
        }
        return L;
    };

    // k-steps of all tap rows under a layout: taps are paired by the parity of their LDS row offset
    auto ksteps = [&](const Lay &L) {
        int sum = 0;
        for (const URow &r : rows) {
            int ne = 0, no = 0;
            for (int t = 0; t < r.ntaps; ++t) {
                const int ix = t % r.box.n[2], iy = (t / r.box.n[2]) % r.box.n[1], iz = t / (r.box.n[2] * r.box.n[1]);
                const int off = (r.box.b[0] + iz) * L.PYX + (r.box.b[1] + iy) * L.PX + (r.box.b[2] + ix);
                ((off & 1) ? no : ne)++;
            }
            sum += ((ne + 1) / 2 + (no + 1) / 2 + 1) / 2;
        }
        return sum;
    };

    // ---- tile search ---------------------------------------------------------------------------------------
    double best = 1e300;
    int bt[4] = {0, 0, 0, 0};
    bool bt_nat = false;
    for (int TX = 1; TX <= 256; TX <<= 1) {
        if (TX > g4_pow2ceil(dg[2].M)) break;
        for (int TY = 1; TX * TY <= 256; TY <<= 1) {
            if (TY > g4_pow2ceil(dg[1].M)) break;
            for (int TZ = 1; TX * TY * TZ <= 256; TZ <<= 1) {
                if (TZ > g4_pow2ceil(dg[0].M)) break;
                for (int PT = 256 / (TX * TY * TZ); PT >= 1; PT >>= 1) {
                    if (PT > 1 && (TX < dg[2].M || TY < dg[1].M || TZ < dg[0].M)) continue;
                    if (PT > 127) continue;
                    const int T[3] = {TZ, TY, TX};
                    int H[3];
                    bool okh = true;
                    for (int d = 0; d < 3; ++d) { H[d] = (T[d] - 1) * dg[d].sm + dg[d].span; okh = okh && H[d] <= 254; }
                    if (!okh) continue;
                    if (PT * TX * TY * TZ < 16) continue;
                    const long long nhv = (long long)PT * H[0] * H[1] * H[2];
                    if (nhv * 2 * NPs > 256 * G4_NSLOT) continue;
                    Lay L = layout(PT, T, H);
                    int sumS_t = ksteps(L);
                    const size_t tpg_t = (size_t)((dg[0].M + TZ - 1) / TZ) * ((dg[1].M + TY - 1) / TY) * ((dg[2].M + TX - 1) / TX) * ncls;
                    const size_t tab_ints = tt_ints + (size_t)(multi ? rows.size() : rows.size() * NCH) * 8 + tpg_t * 8;
                    size_t lds = tab_ints * 4 + (size_t)sumS_t * NCH * WP * NTW * 1024 + 2 * (size_t)PT * L.PZ * NPs * G4_ROWB;
                    bool nat = false;
                    if (lds > 160 * 1024 - 256 && L.xw > 0) {      // the padded conflict-free layout does not fit: natural row order
                        L = layout(PT, T, H, true);
                        sumS_t = ksteps(L);
                        lds = tab_ints * 4 + (size_t)sumS_t * NCH * WP * NTW * 1024 + 2 * (size_t)PT * L.PZ * NPs * G4_ROWB;
                        nat = true;
                    }
                    if (lds > 160 * 1024 - 256) continue;
                    const double tiles = std::ceil((double)max_batch / PT) * std::ceil((double)dg[0].M / TZ) *
                                         std::ceil((double)dg[1].M / TY) * std::ceil((double)dg[2].M / TX);
                    double cost = tiles * ((double)sumS_t * NCH * 420.0 + (double)(multi ? NPs : nstage) * 1.2 * (double)nhv + 600.0);
                    if (L.xw == 0) cost *= 1.5;        // bank-conflicted fragment reads
                    if (cost < best) { best = cost; bt[0] = PT; bt[1] = TZ; bt[2] = TY; bt[3] = TX; bt_nat = nat; }
                }
            }
        }
    }
    if (best > 1e299) return ALQ_OK;
    const int PT = bt[0], T[3] = {bt[1], bt[2], bt[3]};
    int H[3], tiles[3];
    for (int d = 0; d < 3; ++d) {
        H[d] = (T[d] - 1) * dg[d].sm + dg[d].span;
        tiles[d] = (dg[d].M + T[d] - 1) / T[d];
    }
    const int nhv = PT * H[0] * H[1] * H[2];
    const int nrows = PT * T[0] * T[1] * T[2];
    if (multi && nrows < 192) return ALQ_OK;         // a mostly empty tile: the per-class launches do better
    Lay L = layout(PT, T, H, bt_nat);

    // ---- GEMM row -> M-grid point table ---------------------------------------------------------------------
    plan->h_vdesc.assign(256, -1);
    auto lds_row = [&](int pt, int z, int y, int x) {
        return pt * L.PZ + z * dg[0].sm * L.PYX + y * dg[1].sm * L.PX + x * dg[2].sm;
    };
    {
        std::vector<int> order;      // packed points, 16 per column block
        bool okperm = L.xw > 0;
        if (okperm) {
            const int ext[4] = {T[0], T[1], T[2], PT};
            for (int pt = 0; pt < PT && okperm; pt += (L.ud == 3 ? 2 : 1))
                for (int z = 0; z < T[0] && okperm; z += ((L.ud == 0 || L.ud == 4) ? 2 : 1))
                    for (int y = 0; y < T[1] && okperm; y += ((L.ud == 1 || L.ud == 4) ? 2 : 1))
                        for (int xc = 0; xc < T[2] && okperm; xc += L.xw) {
                            std::vector<int> ev, od;
                            std::vector<bool> seen(16, false);
                            const int nu = 16 / L.xw;
                            for (int u = 0; u < nu; ++u)
                                for (int xi = 0; xi < L.xw; ++xi) {
                                    const int p2 = pt + (L.ud == 3 ? u : 0), z2 = z + (L.ud == 0 ? u : (L.ud == 4 ? (u >> 1) : 0)),
                                              y2 = y + (L.ud == 1 ? u : (L.ud == 4 ? (u & 1) : 0));
                                    const int R = lds_row(p2, z2, y2, xc + xi);
                                    const int pk = (p2 << 24) | (z2 << 16) | (y2 << 8) | (xc + xi);
                                    if (seen[R & 15]) okperm = false;
                                    seen[R & 15] = true;
                                    ((R & 1) ? od : ev).push_back(pk);
                                }
                            if (ev.size() != 8 || od.size() != 8) okperm = false;
                            if (!okperm) break;
                            int blk[16];
                            for (int i = 0; i < 8; ++i) {
                                blk[i < 4 ? i : 8 + i] = ev[i];      // lanes 0-3, 12-15
                                blk[4 + i] = od[i];                   // lanes 4-11
                            }
                            order.insert(order.end(), blk, blk + 16);
                        }
            (void)ext;
            if ((int)order.size() != nrows) okperm = false;
        }
        if (!okperm) {                // natural order (small or odd tiles): correct, possibly bank-conflicted
            order.clear();
            L = Lay();
            L.PX = H[2]; L.PYX = H[1] * L.PX; L.PZ = H[0] * L.PYX;
            for (int pt = 0; pt < PT; ++pt)
                for (int z = 0; z < T[0]; ++z)
                    for (int y = 0; y < T[1]; ++y)
                        for (int x = 0; x < T[2]; ++x) order.push_back((pt << 24) | (z << 16) | (y << 8) | x);
        }
        for (size_t i = 0; i < order.size() && i < 256; ++i) plan->h_vdesc[i] = order[i];
    }
    const int plane_rows = PT * L.PZ;

    // ---- k-steps: taps paired by the parity of their LDS row offset -------------------------------------------
    int sumS = 0;
    // Fragment reuse (kernel: `zreuse`): a 3 x 3 x 4 tap box (pair form) on the natural row order, whose waves hold two row
    // blocks one plane above the other two.  The taps are then packed plane by plane - three k-steps per plane, the same
    // (y, x) taps at the same lane groups in every plane - so that k-step 3 * iz + j of the upper row blocks reads exactly
    // what k-step 3 * (iz + 1) + j of the lower ones reads.  Checked on the finished tables below.
    const bool plane_pack = !multi && rows.size() == 1 && pair && g.kind == 0 && L.xw == 0 && (L.PYX % 2) == 0 && PT == 1 &&
                            rows[0].box.n[0] == 3 && rows[0].box.n[1] == 3 && rows[0].box.n[2] == 4 && nrows == 256 &&
                            !getenv("ALQ_NO_ZREUSE");
    for (URow &r : rows) {
        std::vector<std::pair<int, int>> seq;          // lane-group order; t = -1: padding (zero weights)
        const int nplanes = plane_pack ? r.box.n[0] : 1;
        const int per_plane = r.ntaps / nplanes;
        for (int pl = 0; pl < nplanes; ++pl) {
            std::vector<std::pair<int, int>> ev, od;      // (box position t, row offset)
            for (int t = pl * per_plane; t < (pl + 1) * per_plane; ++t) {
                const int ix = t % r.box.n[2], iy = (t / r.box.n[2]) % r.box.n[1], iz = t / (r.box.n[2] * r.box.n[1]);
                const int off = (r.box.b[0] + iz) * L.PYX + (r.box.b[1] + iy) * L.PX + (r.box.b[2] + ix);
                ((off & 1) ? od : ev).push_back({t, off});
            }
            for (auto *lst : {&ev, &od})
                for (size_t i = 0; i < lst->size(); i += 2) {
                    seq.push_back((*lst)[i]);
                    seq.push_back(i + 1 < lst->size() ? (*lst)[i + 1] : std::make_pair(-1, (*lst)[i].second));
                }
        }
        if (seq.size() % 4) { const int o = seq.back().second; seq.push_back({-1, o}); seq.push_back({-1, o}); }
        r.S = (int)seq.size() / 4;
        if (r.S > G4_MAXS) return ALQ_OK;
        r.tap.assign((size_t)r.S * 4 * 2, -1);
        r.toff.assign((size_t)r.S * 4, 0);
        for (size_t i = 0; i < seq.size(); ++i) {
            r.toff[i] = seq[i].second;
            if (seq[i].first < 0) continue;
            const int t = seq[i].first;
            const int ix = t % r.box.n[2], iy = (t / r.box.n[2]) % r.box.n[1], iz = t / (r.box.n[2] * r.box.n[1]);
            for (int half = 0; half < 2; ++half) r.tap[i * 2 + half] = r.tapof(iz, iy, ix, half);
        }
        sumS += r.S;
    }
    const size_t wbytes = (size_t)sumS * NCH * WP * NTW * 1024;

    Igemm4Args &a = plan->a;
    std::memset(&a, 0, sizeof(a));
    a.Co = g.Co;
    a.PT = PT; a.tpg = tiles[0] * tiles[1] * tiles[2] * ncls;
    a.rows = nrows;
    a.PX = L.PX; a.PYX = L.PYX; a.PZ = L.PZ;
    a.smz = dg[0].sm; a.smy = dg[1].sm; a.smx = dg[2].sm;
    a.soz = dg[0].so; a.soy = dg[1].so; a.sox = dg[2].so;
    a.OD = O[0]; a.OH = O[1]; a.OW = O[2];
    a.MD = dg[0].M; a.MH = dg[1].M; a.MW = dg[2].M;
    a.nph = multi ? 1 : nstage;
    a.ngr = multi ? (int)rows.size() : 1;
    a.NP = NPs;
    a.nslots = nhv * 2 * NPs;
    a.plane_bytes = plane_rows * G4_ROWB;
    a.in_pstride = PT * I[0] * I[1] * I[2];
    a.out_pstride = PT * O[0] * O[1] * O[2];
    a.pair = pair ? 1 : 0;
    a.zreuse = 0;
    if (plane_pack && rows[0].S == 9) {
        bool ok = true;
        const URow &r = rows[0];
        for (int iz = 0; iz < 2 && ok; ++iz)
            for (int j = 0; j < 3 && ok; ++j)
                for (int q = 0; q < 4; ++q)
                    if (r.toff[((iz + 1) * 3 + j) * 4 + q] - r.toff[(iz * 3 + j) * 4 + q] != L.PYX * dg[0].sm) ok = false;
        for (int hw = 0; hw < 4 && ok; ++hw)
            for (int m = 0; m < 2 && ok; ++m)
                for (int lr = 0; lr < 16; ++lr) {
                    const int lo_e = plan->h_vdesc[(hw * 4 + m) * 16 + lr], hi_e = plan->h_vdesc[(hw * 4 + 2 + m) * 16 + lr];
                    if (lo_e < 0 || hi_e != lo_e + (1 << 16)) ok = false;       // same point, one M-grid plane up
                }
        a.zreuse = ok ? 1 : 0;
    }
    a.tt_ints = (int)tt_ints;
    a.wbytes = (int)wbytes;
    a.abytes = plane_rows * NPs * G4_ROWB;
    a.split = 1 << 30;

    // tap table + units + phase / group descriptors
    plan->h_ttab.assign(tt_ints, 0);
    for (size_t r = 0; r < rows.size(); ++r)
        for (size_t i = 0; i < rows[r].toff.size(); ++i) plan->h_ttab[r * G4_MAXS * 4 + i] = rows[r].toff[i] * G4_ROWB;
    int w_off = 0;
    plan->h_pdesc.clear();
    if (!multi) {
        for (size_t r = 0; r < rows.size(); ++r)
            for (int ch = 0; ch < NCH; ++ch) {
                Igemm4Plan::Unit u;
                u.chunk = ch; u.S = rows[r].S; u.w_off = w_off; u.tap = rows[r].tap;
                plan->units.push_back(u);
                const int in_off = (rows[r].in_off[0] * I[1] + rows[r].in_off[1]) * I[2] + rows[r].in_off[2];
                const int pd[8] = {in_off, ch, u.S, w_off, (int)r, 0, 0, 0};
                plan->h_pdesc.insert(plan->h_pdesc.end(), pd, pd + 8);
                w_off += u.S * WP * NTW * 1024;
            }
    } else {
        for (size_t r = 0; r < rows.size(); ++r) {
            const int out_off = (rows[r].out_off[0] * O[1] + rows[r].out_off[1]) * O[2] + rows[r].out_off[2];
            const int gd[8] = {0, 0, rows[r].S, w_off, (int)r, out_off, 0, 0};
            plan->h_pdesc.insert(plan->h_pdesc.end(), gd, gd + 8);
            for (int ch = 0; ch < NCH; ++ch) {
                Igemm4Plan::Unit u;
                u.chunk = ch; u.S = rows[r].S; u.w_off = w_off; u.tap = rows[r].tap;
                plan->units.push_back(u);
                w_off += u.S * WP * NTW * 1024;
            }
        }
    }
    // tiles.  Halo validity classes: along every dimension the tiles' valid index ranges [lo, hi) usually take at most
    // three values (first tile, interior, last tile); then a tile is one of 27 classes and a staging slot carries one
    // validity bit per class instead of its halo coordinates (locate(): one bit test per slot instead of six compares).
    std::vector<std::pair<int, int>> dcls[3];
    plan->h_tdesc.assign((size_t)a.tpg * 8, 0);
    if (dg[0].M > 255 || dg[1].M > 255 || dg[2].M > 255) return ALQ_OK;       // tile origins are packed in bytes
    for (int cl = 0; cl < ncls; ++cl)
    for (int tz = 0; tz < tiles[0]; ++tz)
        for (int ty = 0; ty < tiles[1]; ++ty)
            for (int tx = 0; tx < tiles[2]; ++tx) {
                int *td = &plan->h_tdesc[((((size_t)cl * tiles[0] + tz) * tiles[1] + ty) * tiles[2] + tx) * 8];
                const int m0[3] = {tz * T[0], ty * T[1], tx * T[2]};
                int base[3], lohi[3][2];
                bool full = nrows == 256;
                for (int d = 0; d < 3; ++d) {
                    base[d] = m0[d] * dg[d].bm + dg[d].bo;
                    int l = g4_ceildiv(-base[d], dg[d].hs), hgh = g4_ceildiv(dg[d].I - base[d], dg[d].hs);
                    // kind 1 sampled dims: the class offset r < hs never changes validity because I is a multiple of hs
                    l = std::max(0, std::min(l, H[d])); hgh = std::max(l, std::min(hgh, H[d]));
                    lohi[d][0] = l; lohi[d][1] = hgh;
                    full = full && m0[d] + T[d] <= dg[d].M;
                }
                td[0] = (base[0] * I[1] + base[1]) * I[2] + base[2];
                td[1] = (m0[0] * dg[0].so * O[1] + m0[1] * dg[1].so) * O[2] + m0[2] * dg[2].so;
                if (g.kind == 3) td[1] += (g.cls[0] * O[1] + g.cls[1]) * O[2] + g.cls[2];
                if (g.kind == 4) td[1] += (rows[cl].out_off[0] * O[1] + rows[cl].out_off[1]) * O[2] + rows[cl].out_off[2];
                bool inside = true;       // every halo index of the tile valid
                for (int d = 0; d < 3; ++d) inside = inside && lohi[d][0] == 0 && lohi[d][1] == H[d];
                td[2] = (full ? 1 : 0) | (inside ? 2 : 0);
                td[3] = lohi[0][0] | (lohi[0][1] << 8) | (lohi[1][0] << 16) | (lohi[1][1] << 24);
                td[4] = lohi[2][0] | (lohi[2][1] << 8);
                td[5] = m0[0] | (m0[1] << 8) | (m0[2] << 16);
                td[6] = g.kind == 4 ? cl * NCH : 0;          // first phase descriptor of this tile
                int cls = 0;
                for (int d = 0; d < 3; ++d) {
                    const std::pair<int, int> pr(lohi[d][0], lohi[d][1]);
                    size_t k = 0;
                    while (k < dcls[d].size() && dcls[d][k] != pr) ++k;
                    if (k == dcls[d].size()) dcls[d].push_back(pr);
                    cls = cls * 3 + (int)k;
                }
                td[7] = cls;                                 // meaningful when every dimension has <= 3 classes
            }
    // staging slots
    const bool cls_ok = PT == 1 && dcls[0].size() <= 3 && dcls[1].size() <= 3 && dcls[2].size() <= 3 && !getenv("ALQ_NO_HALO_CLASSES");
    a.cls_ok = cls_ok ? 1 : 0;
    plan->h_sdesc.assign((size_t)a.nslots * 4, 0);
    for (int pl = 0; pl < NPs; ++pl)
        for (int pt = 0; pt < PT; ++pt)
            for (int hz = 0; hz < H[0]; ++hz)
                for (int hy = 0; hy < H[1]; ++hy)
                    for (int hx = 0; hx < H[2]; ++hx) {
                        const int hv = ((pt * H[0] + hz) * H[1] + hy) * H[2] + hx;
                        const int lrow = pt * L.PZ + hz * L.PYX + hy * L.PX + hx;
                        for (int half = 0; half < 2; ++half) {
                            int *sd = &plan->h_sdesc[(((size_t)pl * nhv + hv) * 2 + half) * 4];
                            sd[0] = ((pt * I[0] + hz * dg[0].hs) * I[1] + hy * dg[1].hs) * I[2] + hx * dg[2].hs;
                            sd[1] = (pt << 24) | (hz << 16) | (hy << 8) | hx;
                            if (cls_ok) {       // bit ((cz * 3 + cy) * 3 + cx): the slot is inside the tensor for tiles of that class
                                const int hh[3] = {hz, hy, hx};
                                int mask = 0;
                                for (int c = 0; c < 27; ++c) {
                                    const int cc[3] = {c / 9, (c / 3) % 3, c % 3};
                                    bool ok = true;
                                    for (int d = 0; d < 3; ++d)
                                        ok = ok && cc[d] < (int)dcls[d].size() && hh[d] >= dcls[d][cc[d]].first && hh[d] < dcls[d][cc[d]].second;
                                    if (ok) mask |= 1 << c;
                                }
                                sd[1] = mask;
                            }
                            sd[2] = (pl * plane_rows + lrow) * G4_ROWB + half * 8;
                            sd[3] = pl * 8 + half * 4;
                        }
                    }
    // one table block, copied to LDS by the kernel: [tap table | phase / group descriptors | tile descriptors]
    a.pd_off = (int)plan->h_ttab.size();
    plan->h_ttab.insert(plan->h_ttab.end(), plan->h_pdesc.begin(), plan->h_pdesc.end());
    a.td_off = (int)plan->h_ttab.size();
    plan->h_ttab.insert(plan->h_ttab.end(), plan->h_tdesc.begin(), plan->h_tdesc.end());
    a.tt_ints = (int)plan->h_ttab.size();
    plan->NTW = NTW;
    plan->wp = WP;
    a.wp = WP;
    plan->xw = L.xw;
    plan->multi = multi;
    // short contractions per phase (see FIC in the kernel): the conv_transpose classes, and a plain conv whose tile is ONE
    // phase - stash, epilogue and tile lookup then all fall into every staging part (the 8 -> 16 channel conv at 16^3:
    // 401 -> 357 us per 2000 patches with the prefetch issued from the contracting side; the 32 -> 16 channel one with
    // four phases per tile LOSES 4 % the same way: tools/tune_sens.sh)
    plan->fic = NTW == 1 && (g.kind == 4 || g.kind == 1 || (g.kind == 0 && a.nph == 1 && !pair)) && !getenv("ALQ_NO_FIC");
    plan->Ci = g.Ci; plan->Co = g.Co;
    plan->lds_bytes = (size_t)a.tt_ints * 4 + wbytes + 2 * (size_t)a.abytes;
    if (plan->lds_bytes > 160 * 1024) return ALQ_OK;
    plan->flops_per_patch = g.flops_per_patch;
    plan->ok = true;
    plan->alt16.reset();
    if (WP == 3 && !multi && L.xw == 0 && !getenv("ALQ_NO_ALT16")) {
        // This plan reads its fragments with bank conflicts (no conflict-free layout fits beside three weight pieces: the
        // pair-form 16 -> 8 channel convs at 32^3 / 16^3, 3-way conflicts on every X fragment read, PMC 51 - 56 % of the LDS-active
        // cycles).  Launches that contract with the fp16x2 split need two pieces only: with that budget the tile gets the
        // padded 2 x 2 x 4 layout and the twin is conflict-free (PMC: 56 % -> 8 %, LDS active 72 % -> 39 % of CU-busy).
        // Rounds 1 - 2 measured no gain from it (the launches were bound by their instruction streams: 155.7 k vs 154.9 k
        // patches/s) and kept it opt-in; with the launch constants folded and the 16 -> 8 backward launch on the fp16x2 split it
        // pays (round 3, same-box kernel A/B per 2000 patches: fused-head conv 2415 -> 2311 us, enc2 backward 306 -> 246 us, the
        // twelve launches 9337 -> 9206 us, bench 182.6 k -> 184.5 k patches/s): default on, ALQ_NO_ALT16=1 switches it off.
        auto alt = std::make_shared<Igemm4Plan>();
        if (igemm4_build_plan(g, max_batch, alt.get(), 2) == ALQ_OK && alt->ok && alt->xw > 0 && alt->a.PT == a.PT &&
            alt->a.tpg == a.tpg && alt->a.rows == a.rows && alt->a.pair == a.pair && alt->NTW == NTW && alt->fic == plan->fic &&
            alt->a.nph == a.nph)
            plan->alt16 = alt;
    }
    if (getenv("ALQ_G4_VERBOSE"))
        fprintf(stderr, "[igemm4] kind %d%s Ci %d Co %d M %dx%dx%d: tile %d x (%d,%d,%d) halo (%d,%d,%d) pitches %d/%d/%d xw %d ud %d, "
                "%d tap rows, k-steps/tile %d, phases %d groups %d planes %d, NTW %d%s, LDS %zu B (W %zu, %d pieces)\n",
                g.kind, g.flipped ? " flipped" : "", g.Ci, g.Co, dg[0].M, dg[1].M, dg[2].M, PT, T[0], T[1], T[2], H[0], H[1], H[2],
                L.PX, L.PYX, L.PZ, L.xw, L.ud, (int)rows.size(), sumS * NCH, a.nph, a.ngr, a.NP, NTW, pair ? " pair" : "",
                plan->lds_bytes, wbytes, WP);
    return ALQ_OK;
}

static unsigned short g4_bf16_rne(float x) {
    unsigned u;
    std::memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float g4_bf16_to_f(unsigned short hb) {
    const unsigned u = (unsigned)hb << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// packed layout per unit: [s][piece][16-col tile][lane][8] bf16; lane = (q = lane>>4 -> tap 4s+q, row = lane&15)
void igemm4_pack_weights(Igemm4Plan *plan, const std::vector<float> &Bmat) {
    const int NTW = plan->NTW, Ci = plan->Ci, Co = plan->Co;
    const bool pair = plan->a.pair != 0;
    const int WP = plan->wp;
    const bool only16 = WP == 2;                  // an fp16x2-only plan: two pieces per k-step, no bf16 copy
    if (!only16) plan->h_W.assign((size_t)plan->a.wbytes / 2, 0);
    const bool w16 = !plan->multi;
    float amax = 0.f;
    for (float w : Bmat) amax = std::max(amax, std::fabs(w));
    int ex = 0;
    if (amax > 0.f) (void)std::frexp(amax, &ex);       // amax < 2^ex
    plan->w16_exp = 14 - ex;
    if (w16) plan->h_W16.assign((size_t)plan->a.wbytes / 2, 0);
    for (const Igemm4Plan::Unit &u : plan->units)
        for (int s = 0; s < u.S; ++s)
            for (int nt = 0; nt < NTW; ++nt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int q = lane >> 4, r = lane & 15;
                    const int half = pair ? (r >> 3) : 0;
                    const int co = pair ? (r & 7) : nt * 16 + r;
                    const int tap = u.tap[(size_t)(s * 4 + q) * 2 + half];
                    for (int j = 0; j < 8; ++j) {
                        float w = 0.f;
                        if (tap >= 0 && co < Co) w = Bmat[((size_t)tap * Ci + u.chunk * 8 + j) * Co + co];
                        if (w16) {      // fp16 pair of w * 2^w16_exp: h, then the remainder times 2^11 (both round to nearest)
                            const float ws = std::ldexp(w, plan->w16_exp);
                            const _Float16 h = (_Float16)ws;
                            const _Float16 l = (_Float16)std::ldexp(ws - (float)h, 11);
                            unsigned short hb, lb;
                            std::memcpy(&hb, &h, 2);
                            std::memcpy(&lb, &l, 2);
                            plan->h_W16[(size_t)u.w_off / 2 + ((((size_t)s * WP + 0) * NTW + nt) * 64 + lane) * 8 + j] = hb;
                            plan->h_W16[(size_t)u.w_off / 2 + ((((size_t)s * WP + 1) * NTW + nt) * 64 + lane) * 8 + j] = lb;
                        }
                        for (int p = 0; p < 3 && !only16; ++p) {
                            const unsigned short hb = g4_bf16_rne(w);
                            w -= g4_bf16_to_f(hb);
                            plan->h_W[(size_t)u.w_off / 2 + ((((size_t)s * 3 + p) * NTW + nt) * 64 + lane) * 8 + j] = hb;
                        }
                    }
                }
}


static void g4_dump_args(const char *fmt, const Igemm4Args &a, int f0, int f1, int f2, int f3, int f4, int f5, int f6, int f7, int f8, int f9) {
    static std::vector<std::string> seen;
    std::string line;
    char buf[256];
    snprintf(buf, sizeof buf, fmt, f0, f1, f2, f3, f4, f5, f6, f7, f8, f9);
    line = std::string("G4ARGS flags ") + buf + " |";
#define X(f) snprintf(buf, sizeof buf, " " #f "=%d", a.f); line += buf;
    G4_FIXED_INTS(X)
#undef X
#define X(f) snprintf(buf, sizeof buf, " has_" #f "=%d", a.f ? 1 : 0); line += buf;
    G4_FIXED_PTRS(X)
#undef X
    for (const std::string &s : seen) if (s == line) return;
    seen.push_back(line);
    fprintf(stderr, "%s\n", line.c_str());
}

template <int NTW, bool MULTI, bool SUMS, bool BITSRC, bool FCF, bool FIC, bool F16, int EPI, bool ZRE, bool ACC, class G>
static int launch4_k(alq_ctx *ctx, const Igemm4Plan &plan, const Igemm4Args &a, unsigned grid) {
    auto kfn = igemm4_kernel<NTW, MULTI, SUMS, BITSRC, FCF, FIC, F16, EPI, ZRE, ACC, G>;
    if (plan.lds_bytes > 64 * 1024)
        ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)plan.lds_bytes));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), plan.lds_bytes, ctx->stream, a);
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

// every constant a trait states equals the launch's (and the optional pointers are present / absent as stated)
#ifdef ALQ_STAMPS
#define G4_STAMPED 1      // the diagnostic build stamps the folded instantiations too (its a.dbg is set)
#else
#define G4_STAMPED 0
#endif
template <class G>
static bool g4_matches(const Igemm4Args &a) {
    bool ok = true;
#define X(f) ok = ok && a.f == G::f;
    G4_FIXED_INTS(X)
#undef X
#define X(f) ok = ok && ((a.f != nullptr) == G::has_##f || (G4_STAMPED && std::strcmp(#f, "dbg") == 0));
    G4_FIXED_PTRS(X)
#undef X
    return ok;
}

#include "igemm4_fixed.inc"

template <int NTW, bool MULTI, bool SUMS, bool BITSRC = false, bool FCF = false, bool FIC = false, bool F16 = false, int EPI = -1, bool ZRE = false,
          bool ACC = false>
static int launch4_s(alq_ctx *ctx, const Igemm4Plan &plan, const Igemm4Args &a, unsigned grid) {
    {   // ALQ_DUMP_ARGS=1: the launch constants of every distinct launch, as input for tools/gen_igemm4_fixed.py
        static const bool dump = getenv("ALQ_DUMP_ARGS") != nullptr;
        if (dump) g4_dump_args("%d %d %d %d %d %d %d %d %d %d", a, NTW, (int)MULTI, (int)SUMS, (int)BITSRC, (int)FCF, (int)FIC, (int)F16, EPI, (int)ZRE, (int)ACC);
    }
    if (!g_no_fixed) {
        bool done = false;
        const int rc = launch4_fixed<NTW, MULTI, SUMS, BITSRC, FCF, FIC, F16, EPI, ZRE, ACC>(ctx, plan, a, grid, &done);
        if (done) return rc;
    }
    return launch4_k<NTW, MULTI, SUMS, BITSRC, FCF, FIC, F16, EPI, ZRE, ACC, G4Runtime>(ctx, plan, a, grid);
}

template <int NTW, bool MULTI>
static int launch4_t(alq_ctx *ctx, const Igemm4Plan &plan, const Igemm4Args &a, unsigned grid) {
    return (a.osumA || a.osumB) ? launch4_s<NTW, MULTI, true>(ctx, plan, a, grid)
                                : launch4_s<NTW, MULTI, false>(ctx, plan, a, grid);
}

// launches with channel sums (the Fisher pass) whose plan carries its own balance (tune_fic / tune_epi)
template <int NTW, bool BITSRC, bool FIC, bool F16>
static int launch4_epi(alq_ctx *ctx, const Igemm4Plan &plan, const Igemm4Args &a, unsigned grid, int epi) {
    switch (epi) {
        case 0: return launch4_s<NTW, false, true, BITSRC, false, FIC, F16, 0>(ctx, plan, a, grid);
        case 1: return launch4_s<NTW, false, true, BITSRC, false, FIC, F16, 1>(ctx, plan, a, grid);
        case 2: return launch4_s<NTW, false, true, BITSRC, false, FIC, F16, 2>(ctx, plan, a, grid);
        default: return launch4_s<NTW, false, true, BITSRC, false, FIC, F16>(ctx, plan, a, grid);
    }
}

static int igemm4_launch_impl(alq_ctx *ctx, const Igemm4Plan &plan, const View &in, const View &out, const float *bias,
                              int relu, int accumulate, int N, int prof_cls, const Igemm2Fuse *fuse);

// will this launch contract with the fp16x2 split?  (the conditions under which igemm4_launch_impl sets `f16`)
static bool g4_wants_f16(const Igemm4Plan &plan, const View &in, const Igemm2Fuse *fuse, int accumulate) {
    // the same conditions as `f16` / `no16` in igemm4_launch_impl: an accumulating launch never contracts with the split, so it
    // must never be routed to an fp16x2-only twin plan (whose weights exist in that form only)
    if (!fuse || g_no_f16x2 || accumulate || !plan.d_W16 || plan.multi) return false;
    if (fuse->in_bits) return fuse->in_vec_amax > 0.f;
    if (fuse->fc_W) return fuse->in_amax != nullptr;
    if (fuse->in_bound > 0.f && !fuse->in_amax) return plan.a.PT == 1 && plan.NTW <= 2;
    return fuse->in_amax != nullptr && plan.a.PT == 1 && (!in.split || fuse->in_amax2) && plan.NTW <= 2;
}

int igemm4_launch(alq_ctx *ctx, const Igemm4Plan &plan, const View &in, const View &out, const float *bias,
                  int relu, int accumulate, int N, int prof_cls, const Igemm2Fuse *fuse) {
    if (plan.alt16 && plan.alt16->ok && plan.alt16->d_W16 && g4_wants_f16(*plan.alt16, in, fuse, accumulate))
        return igemm4_launch_impl(ctx, *plan.alt16, in, out, bias, relu, accumulate, N, prof_cls, fuse);
    return igemm4_launch_impl(ctx, plan, in, out, bias, relu, accumulate, N, prof_cls, fuse);
}

static int igemm4_launch_impl(alq_ctx *ctx, const Igemm4Plan &plan, const View &in, const View &out, const float *bias,
                              int relu, int accumulate, int N, int prof_cls, const Igemm2Fuse *fuse) {
    Igemm4Args a = plan.a;
    ALQ_REQUIRE(in.C == plan.Ci && out.C == plan.Co, ALQ_EINVAL, "igemm4: channel counts do not match the plan");
    ALQ_REQUIRE(out.D == a.OD && out.H == a.OH && out.W == a.OW, ALQ_EINVAL, "igemm4: output view mismatch");
    ALQ_REQUIRE(plan.d_W && plan.d_tdesc, ALQ_EINVAL, "igemm4: weights not set");
    // unsigned 32-bit byte offsets from the tensor base (float indices stay below 2^30, so they fit an int as well)
    ALQ_REQUIRE(in.delta + (long long)N * in.vox() * in.cs < (1LL << 30) - 64 && out.delta + (long long)N * out.vox() * out.cs < (1LL << 30) - 64,
                ALQ_EUNSUPPORTED, "igemm4: tensor exceeds the 32-bit byte-offset range (lower the batch)");
    ALQ_REQUIRE(!in.split || (!plan.multi && in.split % 8 == 0 && in.cs == in.split && in.C == 2 * in.split && in.c0 == 0),
                ALQ_EUNSUPPORTED, "igemm4: unsupported split input");
    ALQ_REQUIRE(!out.split || (!a.pair && out.split % 4 == 0 && out.cs == out.split && out.C == 2 * out.split && out.c0 == 0),
                ALQ_EUNSUPPORTED, "igemm4: unsupported split output");
    ALQ_REQUIRE(in.cs % 4 == 0 && in.c0 % 4 == 0 && out.cs % 4 == 0 && out.c0 % 4 == 0, ALQ_EUNSUPPORTED,
                "igemm4: channel slice not 16-byte aligned");
    a.in = in.p; a.in_cs = in.cs; a.in_c0 = in.c0;
    a.out = out.p; a.out_cs = out.cs; a.out_c0 = out.c0;
    a.W = plan.d_W; a.bias = bias; a.relu = relu; a.accumulate = accumulate; a.N = N;
    a.tdesc = plan.d_tdesc; a.sdesc = plan.d_sdesc; a.pdesc = plan.d_pdesc; a.ttab = plan.d_ttab; a.vdesc = plan.d_vdesc;
    a.in_bytes = (int)(unsigned)((in.delta + (long long)N * in.vox() * in.cs) * 4);
    a.in_split_ch = in.split / 8; a.in_delta = (int)in.delta;
    a.out_split = out.split; a.out_delta = (int)out.delta;
    a.mask_split = 0; a.mask_delta = 0;
    a.src_bits = nullptr; a.bits_pstride = 0; a.bits_bytes = 0; a.src_presplit = 0;
    a.f16_ein = 0; a.f16_ew = plan.w16_exp; a.in_amax = nullptr; a.in_amax2 = nullptr; a.out_amax = nullptr;
    bool f16 = false;
    // an accumulating launch runs the plain bf16x3 instantiation of its column-tile count (see ACC in the kernel)
    const bool no16 = g_no_f16x2 != 0 || accumulate != 0;
    a.amax_from = 0;
    if (fuse && fuse->out_amax) {
        ALQ_REQUIRE(a.PT == 1, ALQ_EUNSUPPORTED, "igemm4: output maxima need one patch per tile");
        a.out_amax = fuse->out_amax;
        a.amax_from = fuse->amax_from;
    }
    // a stored input tensor with per-patch maxima: fp16x2 where a variant of the plan's shape exists
    if (fuse && fuse->in_amax && !fuse->in_bits && !fuse->fc_W && plan.d_W16 && !plan.multi && a.PT == 1 && !no16 &&
        (!in.split || fuse->in_amax2) && plan.NTW <= 2) {
        a.in_amax = fuse->in_amax; a.in_amax2 = fuse->in_amax2;
        a.W = plan.d_W16;
        f16 = true;
    }
    // a stored input tensor with a host-known bound on its magnitude: fp16x2 with one launch-wide scale
    if (!f16 && fuse && fuse->in_bound > 0.f && !fuse->in_amax && !fuse->in_bits && !fuse->fc_W && plan.d_W16 && !plan.multi &&
        a.PT == 1 && !no16 && plan.NTW <= 2) {
        int ex = 0;
        (void)std::frexp(fuse->in_bound, &ex);       // bound < 2^ex
        a.f16_ein = 14 - ex;
        a.W = plan.d_W16;
        f16 = true;
    }
    a.fc_W = nullptr; a.fc_part = nullptr; a.fc_bits = nullptr; a.fc_F = 0;
    a.flip_tau = 0.f; a.flip_cap = 0; a.flip_cnt = nullptr; a.flip_list = nullptr;
    if (fuse && fuse->in_bits) {      // masked-vector input: `in` only gives the geometry
        ALQ_REQUIRE(plan.NTW == 1 && !plan.multi && a.PT == 1 && !in.split && in.c0 == 0 && in.cs == in.C && fuse->in_vec &&
                        ((long long)in.vox() * in.cs) % 32 == 0,
                    ALQ_EUNSUPPORTED, "igemm4: masked-vector input needs a one-patch, one-column-tile plan on a dense tensor");
        a.in = fuse->in_vec;
        a.in_pstride = 0;
        a.in_bytes = (int)((long long)in.vox() * in.cs * 4);
        a.src_bits = fuse->in_bits;
        a.bits_pstride = (int)((long long)in.vox() * in.cs);
        a.bits_bytes = (int)((long long)N * in.vox() * in.cs / 4);
        if (fuse->in_vec_amax > 0.f && plan.d_W16 && !no16) {       // scale known ahead of the launch: fp16x2 contraction
            int ex = 0;
            (void)std::frexp(fuse->in_vec_amax, &ex);
            a.f16_ein = 14 - ex;
            a.W = plan.d_W16;
            f16 = true;
            if (fuse->in_vec16) {      // the vector pre-split with exactly this scale (same 16 bytes per 4 channels)
                a.in = reinterpret_cast<const float *>(fuse->in_vec16);
                a.src_presplit = 1;
            }
        }
    }
    a.dbg = nullptr;
    if (g_igemm2_dbg) {   // diagnostic: stamp only the launch whose ordinal (since the buffer was set) is ALQ_STAMP_ONLY
        static int want = -2;
        if (want == -2) { const char *e = getenv("ALQ_STAMP_ONLY"); want = e ? atoi(e) : -1; }
        static int ordinal = 0;
        static unsigned long long *last = nullptr;
        if (last != g_igemm2_dbg) { last = g_igemm2_dbg; ordinal = 0; }
        if (want < 0 || ordinal == want) a.dbg = g_igemm2_dbg;
        ++ordinal;
    }
    a.dbg_repeat = g_dbg_knobs[0];
    a.split = 1 << 30;
    if (fuse) {
        ALQ_REQUIRE(fuse->split % 4 == 0 && fuse->mask_cs % 4 == 0 && fuse->mask_c0 % 4 == 0 && fuse->mask_from % 4 == 0,
                    ALQ_EUNSUPPORTED, "igemm4: fused epilogue needs 4-channel aligned slices");
        ALQ_REQUIRE(!(plan.multi && fuse->mask), ALQ_EUNSUPPORTED, "igemm4: no mask in the multi-output form");
        ALQ_REQUIRE(!a.pair || (!fuse->osumB && fuse->split == 0), ALQ_EUNSUPPORTED,
                    "igemm4: the pair form sums all 8 channels of a voxel");
        a.mask = fuse->mask; a.mask_cs = fuse->mask_cs; a.mask_c0 = fuse->mask_c0; a.mask_from = fuse->mask_from;
        // the sign field stands in for the floats (same index space / 4; the slices are 4-channel aligned, checked above)
        a.mask_bits = fuse->mask ? fuse->mask_bits : nullptr;
        if (a.mask_bits) a.mask = nullptr;
        a.sign_out = fuse->sign_out;
        ALQ_REQUIRE(!a.sign_out || (!accumulate && !fuse->fc_W && fuse->store_from == 0 && !out.split), ALQ_EUNSUPPORTED,
                    "igemm4: a sign field is written for a plainly stored output only");
        a.mask_to = fuse->mask_to;
        a.mask_split = fuse->mask_split; a.mask_delta = (int)fuse->mask_delta;
        ALQ_REQUIRE(!a.mask_split || a.mask_from == 0, ALQ_EUNSUPPORTED, "igemm4: a split mask covers all columns");
        a.osumA = fuse->osumA; a.osumB = fuse->osumB;
        a.split = fuse->split > 0 ? fuse->split : (1 << 30);
        a.store_from = fuse->store_from;
        ALQ_REQUIRE(a.store_from % 4 == 0 && (a.store_from == 0 || !accumulate), ALQ_EUNSUPPORTED,
                    "igemm4: store_from needs a 4-aligned column and no accumulation");
        if (fuse->fc_W) {
            ALQ_REQUIRE(plan.NTW == 1 && !plan.multi && a.pair && a.PT == 1 && out.C == 8 && out.cs == 8 && out.c0 == 0 && !out.split &&
                            !accumulate && !fuse->mask && fuse->fc_part && relu && !fuse->out_amax &&
                            fuse->fc_F == (long long)out.vox() * 8,
                        ALQ_EUNSUPPORTED, "igemm4: fused fc head needs the pair form on a dense 8-channel output");
            a.fc_W = fuse->fc_W; a.fc_F = (int)fuse->fc_F; a.fc_part = fuse->fc_part;
            a.fc_bits = reinterpret_cast<unsigned char *>(fuse->fc_bits);
            a.store_from = out.C;
            if (fuse->flip_list && fuse->fc_bits && fuse->flip_l1 > 0.f) {
                // The fp16x2 contraction's error: each operand is off by <= 2^-22 relative, the l.l product is dropped, the
                // accumulation is fp32, so |error| <= 2^-20 * max |x| * (L1 norm of the output channel's weights) in the WORST case
                // (every term at the patch's maximum, every rounding error of one sign) - three orders of magnitude above the
                // rms error 2^-22 sqrt(sum x^2 w^2) of independent roundings, and a threshold there marks ~150 groups per 32^3
                // patch (the fix-up then costs 160 us per 2000 patches).  The threshold is 1/16 of the worst case = ~160 rms
                // errors: ~10 groups per patch, 15 us.
                a.flip_tau = std::ldexp(fuse->flip_l1, -24) * (fuse->flip_bias_nonzero ? -1.f : 1.f);
                a.flip_cnt = fuse->flip_cnt; a.flip_list = fuse->flip_list; a.flip_cap = fuse->flip_cap;
            }
        }
    }
    const int pgroups = (N + a.PT - 1) / a.PT;
    const long long total = (long long)pgroups * a.tpg;
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((total + 1) / 2, 256));
    a.xcd_order = (grid % 8 == 0 && !g_no_xcd_order) ? 1 : 0;
    if (a.fc_W && fuse->in_amax && plan.d_W16 && !no16) {       // the fused-head conv with per-patch input maxima: fp16x2
        a.in_amax = fuse->in_amax; a.in_amax2 = fuse->in_amax2;
        a.W = plan.d_W16;
        f16 = true;
    }
    if (!f16) { a.flip_list = nullptr; a.flip_cnt = nullptr; a.flip_cap = 0; a.flip_tau = 0.f; }
    ALQ_REQUIRE(plan.wp == 3 || f16, ALQ_EINVAL, "igemm4: an fp16x2-only plan was asked for a bf16x3 launch");
    ProfScope ps(ctx, f16 ? (int)PROF_IGEMM_F16 : prof_cls, plan.flops_per_patch * N);
    if (accumulate) {
        ALQ_REQUIRE(!plan.multi && !a.src_bits && !a.fc_W, ALQ_EUNSUPPORTED, "igemm4: accumulation into the output only in the plain conv form");
        const bool sums = a.osumA || a.osumB;
        if (plan.NTW == 1)
            return sums ? launch4_s<1, false, true, false, false, false, false, -1, false, true>(ctx, plan, a, grid)
                        : launch4_s<1, false, false, false, false, false, false, -1, false, true>(ctx, plan, a, grid);
        return sums ? launch4_s<2, false, true, false, false, false, false, -1, false, true>(ctx, plan, a, grid)
                    : launch4_s<2, false, false, false, false, false, false, -1, false, true>(ctx, plan, a, grid);
    }
    if (plan.multi) {
        if (plan.NTW == 1) return launch4_t<1, true>(ctx, plan, a, grid);
        return launch4_t<2, true>(ctx, plan, a, grid);
    }
    if ((plan.tune_fic >= 0 || plan.tune_epi >= 0) && (a.osumA || a.osumB)) {
        const int epi = plan.tune_epi;
        if (a.fc_W) {        // the fused head: only the prefetch side moves
            const bool fic = plan.tune_fic >= 0 ? plan.tune_fic != 0 : f16;
            if (f16) return fic ? launch4_s<1, false, true, false, true, true, true>(ctx, plan, a, grid)
                                : launch4_s<1, false, true, false, true, false, true>(ctx, plan, a, grid);
            return fic ? launch4_s<1, false, true, false, true, true, false>(ctx, plan, a, grid)
                       : launch4_s<1, false, true, false, true, false, false>(ctx, plan, a, grid);
        }
        if (a.src_bits)
            return f16 ? launch4_epi<1, true, false, true>(ctx, plan, a, grid, epi) : launch4_epi<1, true, false, false>(ctx, plan, a, grid, epi);
        if (plan.NTW == 2)
            return f16 ? launch4_epi<2, false, false, true>(ctx, plan, a, grid, epi) : launch4_epi<2, false, false, false>(ctx, plan, a, grid, epi);
        const bool fic = plan.tune_fic >= 0 ? plan.tune_fic != 0 : plan.fic;
        if (fic) return f16 ? launch4_epi<1, false, true, true>(ctx, plan, a, grid, epi) : launch4_epi<1, false, true, false>(ctx, plan, a, grid, epi);
        return f16 ? launch4_epi<1, false, false, true>(ctx, plan, a, grid, epi) : launch4_epi<1, false, false, false>(ctx, plan, a, grid, epi);
    }
    if (a.src_bits && f16)
        return (a.osumA || a.osumB) ? launch4_s<1, false, true, true, false, false, true>(ctx, plan, a, grid)
                                    : launch4_s<1, false, false, true, false, false, true>(ctx, plan, a, grid);
    if (a.src_bits)
        return (a.osumA || a.osumB) ? launch4_s<1, false, true, true>(ctx, plan, a, grid)
                                    : launch4_s<1, false, false, true>(ctx, plan, a, grid);
    // the prefetch of the fused-head conv is issued from the staging side again: with its launch constants folded and one
    // wait per epilogue the staging part became the shorter one (phase stamps: contraction 46 %, staging 36 % + 14 %
    // waiting; tools/tune_sens.sh: 2670 -> 2593 us per 2000 patches)
    if (a.fc_W && f16 && a.zreuse)      // one tap row of nine k-steps per phase (checked when the plan was built): the fragment-reuse loop
        return a.osumA ? launch4_s<1, false, true, false, true, false, true, -1, true>(ctx, plan, a, grid)
                       : launch4_s<1, false, false, false, true, false, true, -1, true>(ctx, plan, a, grid);
    if (a.fc_W && f16)
        return a.osumA ? launch4_s<1, false, true, false, true, false, true>(ctx, plan, a, grid)
                       : launch4_s<1, false, false, false, true, false, true>(ctx, plan, a, grid);
    if (a.fc_W)
        return a.osumA ? launch4_s<1, false, true, false, true>(ctx, plan, a, grid)
                       : launch4_s<1, false, false, false, true>(ctx, plan, a, grid);
    if (f16 && !a.src_bits && !a.fc_W) {
        if (plan.NTW == 2)
            return (a.osumA || a.osumB) ? launch4_s<2, false, true, false, false, false, true>(ctx, plan, a, grid)
                                        : launch4_s<2, false, false, false, false, false, true>(ctx, plan, a, grid);
        if (plan.fic)
            return (a.osumA || a.osumB) ? launch4_s<1, false, true, false, false, true, true>(ctx, plan, a, grid)
                                        : launch4_s<1, false, false, false, false, true, true>(ctx, plan, a, grid);
        return (a.osumA || a.osumB) ? launch4_s<1, false, true, false, false, false, true>(ctx, plan, a, grid)
                                    : launch4_s<1, false, false, false, false, false, true>(ctx, plan, a, grid);
    }
    if (plan.NTW == 1 && plan.fic)
        return (a.osumA || a.osumB) ? launch4_s<1, false, true, false, false, true>(ctx, plan, a, grid)
                                    : launch4_s<1, false, false, false, false, true>(ctx, plan, a, grid);
    if (plan.NTW == 1) return launch4_t<1, false>(ctx, plan, a, grid);
    return launch4_t<2, false>(ctx, plan, a, grid);
}

}  // namespace alq
