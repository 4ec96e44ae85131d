// Plane-sweep engine for the 3x3x3 convolutions around the two-class head (NET-C: dec2, 16 -> 8 channels at 32^3, and its
// backward, 8 -> 16): the two launches that were 47 % of the two-slot engine's time (igemm4.hip) at 0.24 - 0.29 of the 16-bit
// MFMA peak.  Replaces the tf.nn.conv3d call site NN_extended.py:416-426 for that layer (forward) and the corresponding
// node of tf.gradients (NN_extended.py:1029-1035) (backward), same arithmetic as the fp16x2 variant of igemm4
// (x 2^e = h + l 2^-11, products h.h | h.l + l.h on v_mfma_f32_16x16x32_f16, fp32 accumulate).
//
// What is different from the tile engine, and why (MI355X_MICROARCH.md: "Two waves per SIMD", LDS table):
//  * INPUT-STATIONARY Z SWEEP.  One 256-thread workgroup per CU owns a whole patch and walks its z planes; wave w owns
//    the output rows y = 8w .. 8w+7 (all 32 x).  An MFMA column block is one x row (16 x-pairs), so a fragment read from
//    LDS for input row (z, y) and k-step s feeds every (dz, dy) tap that touches it: up to 3 x 3 x 3 products = 27 MFMAs per
//    two ds_read_b128 (the tile engine: 3 per two reads; its contraction was LDS-read and issue bound).  The three output
//    planes z-1, z, z+1 an input plane contributes to live in three rotating accumulator sets (3 x 8 rows x 2 x 4 VGPRs).
//  * WEIGHTS IN REGISTERS (18 k-steps x 2 pieces x 4 VGPRs = 144 of the 512 a one-wave-per-SIMD kernel owns): no weight
//    fragment reads at all, no LDS for them.
//  * EVERY INPUT ELEMENT IS STAGED ONCE PER PATCH: the x halo is two zero slots per row image, the y halo two zero rows per
//    plane image, the z halo is no plane at all; nothing is re-read from HBM or re-split (the tile engine staged and
//    split every halo voxel 2.3 x).  Two plane images (2 x 74 KB) alternate; the split of plane z+1 is spread over the
//    MFMA stream of plane z.
//  * ONE WAVE PER SIMD, no second wave competing for the issue port; the epilogue of plane z-2 (fused two-class head:
//    logit-difference partials, sign bytes, flip marks, the head's input sum) is spread over the rows of step z, each
//    row just before its accumulators are restarted.
//  * compile-time geometry: every LDS offset is an instruction immediate, no descriptor tables, no per-slot EXEC masks.
#include "alq_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace alq {

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int V> using IC = std::integral_constant<int, V>;
// __builtin_amdgcn_sched_barrier mask: VALU (2), SALU (4), all DS (0x80 | 0x100 | 0x200), transcendentals (0x400) may cross;
// MFMA (8) and vector memory (0x10 | 0x20 | 0x40) may not
#ifndef C3_SCHED_MASK
#define C3_SCHED_MASK 0
#endif
// forward kernel: staging units (1 KB of fp32 per wave) in flight, and how many rows ahead the head's weight difference is
// fetched.  vmcnt retires in order, so the wait for a weight-difference slice (an L2 hit) also waits for every staging load
// issued before it: the HBM latency the sweep tolerates is C3_WD + 1 rows of MFMAs (~860 cycles each), not the staging depth
#ifndef C3_PF
#define C3_PF 8
#endif
#ifndef C3_WD
#define C3_WD 4
#endif
static_assert(C3_PF == 8 || C3_PF == 16, "the ring of staging registers must divide the 16 units of a plane");
static_assert(C3_WD >= 1 && C3_WD <= 8, "weight-difference slices are fetched 1..8 rows ahead");
// timing experiments only (tools/probe/c3d_bench.hip): 1 no staging, 2 no epilogue in the sweep steps, 4 no barrier, 8 staging loads
// always from the first plane (L2 hits), 16 no sign-byte stores, 32 no weight-difference loads, 64 no LDS writes of the staging
#ifndef C3_ABL
#define C3_ABL 0
#endif
#ifndef C3_PIPE
#define C3_PIPE 2
#endif
// what may fill the gap behind an MFMA in the sweep's 1 : C3_PIPE pattern: VALU (0x002) + SALU (0x004) + vector memory (0x010) + LDS
// (0x080).  With VALU alone the address arithmetic, loads and fragment reads piled up in a few gaps (a gap with up to two fillers is
// free, each further one costs its issue cycles - tools/probe/mfma_chain_probe.hip): -4 % cycles per step, of which the power-limited
// clock gives back about a third
#ifndef C3_FILL_MASK
#define C3_FILL_MASK 0x096
#endif
#ifndef C3_PIN_SUMS
#define C3_PIN_SUMS 1
#endif
#ifndef C3_PEEL
#define C3_PEEL 1
#endif
#ifndef C3_EPI2
#define C3_EPI2 1
#endif
#ifndef C3_BPIPE
#define C3_BPIPE 2
#endif
// the 7-k-step backward kernel: non-MFMA instructions behind each MFMA of the sweep's pattern, weights pinned to accumulation registers
#ifndef C3_B7PIPE
#define C3_B7PIPE 4
#endif
#ifndef C3_B7PINW
#define C3_B7PINW 0
#endif
// timing experiments only (results wrong): 1 no epilogue in the sweep steps, 2 no staging (loads + LDS writes), 4 no S k-step,
// 8 no hold behind the 16-byte stores
#ifndef C3_B7ABL
#define C3_B7ABL 0
#endif

// LDS image of one input plane (bytes).  Slot = 8 fp16 channels of one tensor at one voxel, one piece (h or l).
constexpr int C3_TEN = 34 * 16;          // slots x = -1 .. 32 of one (row, piece, tensor); the two outer ones stay zero
constexpr int C3_PIECE = 2 * C3_TEN;     // two tensors (channels 0..7 | 8..15 of the concat)
constexpr int C3_ROW = 2 * C3_PIECE;     // two pieces
constexpr int C3_PLANE = 34 * C3_ROW;    // rows y = -1 .. 32; the two outer ones stay zero
constexpr int C3_LDS = 2 * C3_PLANE;     // two planes alternate: 147,968 of the 163,840 bytes
static_assert(C3_LDS <= 160 * 1024, "two plane images must fit the CU's LDS");
}  // namespace

struct C3FwdArgs {
    const float *inA, *inB;        // dense [N, D, 32, 32, 8] fp32: channels 0..7 / 8..15 of the conv's 16-channel input
    const void *W;                 // [18 k-steps][2 pieces][64 lanes][8] fp16 (c3d_fwd_pack)
    const float *bias;             // [8]
    const unsigned *amaxA, *amaxB; // [N] max |x| per patch of the two tensors (float bits)
    const float *fc_W;             // [D * 32 * 32 * 8] W0 - W1 of the two-class head, activation-memory order
    float *fc_part;                // [N][4] logit-difference partials, one per wave
    float *asum_part;              // [N][4] sum of the ReLU'd output per wave (the head's input sum), or null
    unsigned char *fc_bits;        // [N][D * 32 * 32 * 2] sign byte per 4 channels (bit 4: flip mark), or null
    int N, D;
    int e_w;                       // scale exponent of the packed weights
    float flip_tau;                // > 0: mark 4-channel groups holding |pre-activation| < flip_tau * 2^(14 - e_patch)
    unsigned long long *clk;       // diagnostic builds (-DC3_CLK): per workgroup {shader cycles, 100 MHz ticks} of the whole kernel
};

// SUMS: Fisher pass (sign bytes, flip marks, the head's input sum); otherwise forward only (logit partials).
// ONEACC: both pieces at their true scale (l = x 2^e - h, not times 2^11) and all three products in ONE accumulator: half the
// accumulator registers (96 instead of 192 of the 512) and no combine step.  A small l is then an fp16 subnormal (absolute
// resolution 2^-24 of the scaled range, i.e. 2^-38 of the patch's maximum): the MFMA of gfx950 does not flush them (checked on
// the device by alq_c3d_selftest), and an element that small contributes 2^-38 max |x| |w| of error - far below the 2^-24
// relative rounding of the large elements' products.
template <bool SUMS, bool ONEACC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void c3d_fwd_kernel(const C3FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int D = a.D;
    const unsigned plane_f = 32u * 32u * 8u;                 // floats per plane of one 8-channel tensor
    const unsigned patch_f = (unsigned)D * plane_f;

    for (int i = tid * 16; i < C3_LDS; i += 256 * 16) *reinterpret_cast<i32x4 *>(lds + i) = i32x4{0, 0, 0, 0};

    // weights: A operand (rows = 2 x-adjacent output voxels x 8 channels), resident in registers for the whole launch
    f16x8 Wh[18], Wl[18];
    {
        const i32x4 *Wg = reinterpret_cast<const i32x4 *>(a.W);
#pragma unroll
        for (int k = 0; k < 18; ++k) {
            Wh[k] = __builtin_bit_cast(f16x8, Wg[(k * 2 + 0) * 64 + lane]);
            Wl[k] = __builtin_bit_cast(f16x8, Wg[(k * 2 + 1) * 64 + lane]);
        }
        // Start the weights' lives in ACCUMULATION registers: only MFMAs read them, and with them (144) + the accumulators (96)
        // in the AGPR half the 256 architectural VGPRs are left to the staging / epilogue arithmetic.  Left to itself the
        // allocator keeps half of them in VGPRs, fills that half and spills - and a scratch reload inside the sweep waits
        // for every staging load in flight (vmcnt retires in order).
#ifndef C3_NO_PIN
#pragma unroll
        for (int k = 0; k < 18; ++k) {
            i32x4 h = __builtin_bit_cast(i32x4, Wh[k]), l = __builtin_bit_cast(i32x4, Wl[k]);
            asm volatile("" : "+a"(h), "+a"(l));
            Wh[k] = __builtin_bit_cast(f16x8, h); Wl[k] = __builtin_bit_cast(f16x8, l);
        }
#endif
    }
    const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + (lq & 1) * 4);

    // fragment reads (B operand: columns = 16 x-pairs r, k-group (t, p) = (tensor, x parity)): slot 2 r + p (+ 2 s)
    const int frag_lane = (lq >> 1) * C3_TEN + (2 * lr + (lq & 1)) * 16 + wave * 8 * C3_ROW;
    // staging writes: lane = (x, channel half): 8 bytes per piece at slot x + 1 of rows 8 w + 1 ..
    const int st_lane = (wave * 8 + 1) * C3_ROW + 16 + lane * 8;
    // epilogue: this lane's 4 channels of voxel x = 2 r + (q >> 1): float offset 16 r + 4 q inside the x row
    const unsigned epi_lane_f = 16u * lr + 4u * lq;

    const int G = gridDim.x, b0 = blockIdx.x;
    const int np = b0 < a.N ? (a.N - b0 + G - 1) / G : 0;
#ifdef C3_CLK
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif

    f32x4 acc[3][8], accx[3][ONEACC ? 1 : 8];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[s][i] = f32x4{0.f, 0.f, 0.f, 0.f}; if constexpr (!ONEACC) accx[s][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // (read through the constant address space: the index is uniform, so these become scalar loads; as plain global loads they
    // brought 64-bit vector address arithmetic and vector loads into the gap between two steps, where no MFMA covers them.  The
    // arrays are written by earlier launches only.)
    typedef const unsigned __attribute__((address_space(4))) *cu32p;
    const cu32p amaxA_c = (cu32p)(unsigned long long)a.amaxA, amaxB_c = (cu32p)(unsigned long long)a.amaxB;
    auto patch_exp = [&](int p) __attribute__((always_inline)) {      // max |x| < 2^ex -> scale 2^(14 - ex); all-zero patch: 0
        const unsigned fa = amaxA_c[p], fb = amaxB_c[p];
        const unsigned fm = fa > fb ? fa : fb;
        const int ex = (int)((fm >> 23) & 255u) - 126;
        // (a patch whose maximum is below 2^-82 - fp32 subnormals included - keeps the scale 2^96: 2^(14 - ex) would overflow
        // the scale or underflow its inverse; such inputs are then simply small fp16 values)
        const int ce = 14 - ex;
        return fm ? (ce < 96 ? ce : 96) : 0;
    };
    // ONE buffer resource per array for the whole launch: the patch goes into the scalar offset (not range-checked by the
    // hardware), a lane with nothing to load or store aims past the array through its VECTOR offset (loads return 0, stores
    // are dropped).  N < 4096 patches of 1 MiB keep every byte offset below 2^32 (the host checks).
    auto rsrc_of = [&](const void *base, unsigned long long bytes) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(unsigned)bytes, 0x00020000);
    };
    const unsigned OOB = 0xffffff00u;
    const __amdgpu_buffer_rsrc_t inA_rsrc = rsrc_of(a.inA, (unsigned long long)a.N * patch_f * 4u);
    const __amdgpu_buffer_rsrc_t inB_rsrc = rsrc_of(a.inB, (unsigned long long)a.N * patch_f * 4u);
    const __amdgpu_buffer_rsrc_t wd_rsrc = rsrc_of(a.fc_W, (unsigned long long)patch_f * 4u);
    const __amdgpu_buffer_rsrc_t bits_rsrc = rsrc_of(a.fc_bits, (SUMS && a.fc_bits) ? (unsigned long long)a.N * (patch_f >> 2) : 0ull);
    auto patch_of = [&](int pi) __attribute__((always_inline)) { return (unsigned)(b0 + pi * G); };

    // ---- staging: unit u = 0..15 of a plane: row 8 w + (u >> 1), tensor u & 1: 1 KB of fp32 per wave instruction ------------
    // eight units in flight (HBM latency under load is ~2 us = half a step)
    f32x4 R4[C3_PF];
    float sc = 1.f, sc11 = 2048.f;
    auto stage_unit = [&](int wbase, auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        const f32x4 v = R4[u % C3_PF];
        const float x0 = v.x * sc, x1 = v.y * sc, x2 = v.z * sc, x3 = v.w * sc;
        const f16x2 h01 = __builtin_convertvector(f32x2{x0, x1}, f16x2);
        const f16x2 h23 = __builtin_convertvector(f32x2{x2, x3}, f16x2);
        const f32x2 g01 = __builtin_convertvector(h01, f32x2), g23 = __builtin_convertvector(h23, f32x2);
        f16x2 l01, l23;
        if constexpr (ONEACC) {       // x 2^e - h is exact in fp32; its rounding to fp16 is the only one
            l01 = __builtin_convertvector(f32x2{x0 - g01.x, x1 - g01.y}, f16x2);
            l23 = __builtin_convertvector(f32x2{x2 - g23.x, x3 - g23.y}, f16x2);
        } else {
            l01 = __builtin_convertvector(f32x2{__builtin_fmaf(g01.x, -2048.f, v.x * sc11), __builtin_fmaf(g01.y, -2048.f, v.y * sc11)}, f16x2);
            l23 = __builtin_convertvector(f32x2{__builtin_fmaf(g23.x, -2048.f, v.z * sc11), __builtin_fmaf(g23.y, -2048.f, v.w * sc11)}, f16x2);
        }
        char *dst = lds + wbase + (u >> 1) * C3_ROW + (u & 1) * C3_TEN;
        if constexpr (C3_ABL & 64) { asm volatile("" :: "v"(h01), "v"(h23), "v"(l01), "v"(l23), "v"(dst)); return; }
        *reinterpret_cast<uint2 *>(dst) = uint2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
        *reinterpret_cast<uint2 *>(dst + C3_PIECE) = uint2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
    };
    // voff = lane * 16 or OOB (no such plane), soff = byte offset of row 8 w of the plane inside the array
    auto load_unit = [&](unsigned voff, unsigned soff, auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        R4[u % C3_PF] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128((u & 1) ? inB_rsrc : inA_rsrc, (int)voff, (int)(soff + (unsigned)((u % 16) >> 1) * 1024u), 0));
    };
    struct Cur { unsigned voff, soff; };
    auto cursor = [&](int pi, int z) __attribute__((always_inline)) {
        Cur c;
        const bool ok = pi < np;
        c.voff = ok ? (unsigned)lane * 16u : OOB;
        c.soff = ok ? (patch_of(pi) * patch_f + (unsigned)z * plane_f) * 4u + (unsigned)(wave * 8) * 1024u : 0u;
        if constexpr (C3_ABL & 8) c.soff = (unsigned)(wave * 8) * 1024u;
        return c;
    };
    // plane stream cursors: c1 = the plane staged by the running step, c2 = the one after it (its first 8 loads go out early)
    int c1_pi = 0, c1_z = 0, c2_pi = 0, c2_z = 0;
    auto advance = [&](int &pi, int &z) __attribute__((always_inline)) { if (++z == D) { z = 0; ++pi; } };

    float fs = 0.f, sa = 0.f;       // running logit-difference partial / sum of the ReLU'd output of the patch in the epilogue
    f32x4 wdq[C3_WD];      // the head's weight difference, fetched C3_WD rows ahead
#pragma unroll
    for (int i = 0; i < C3_WD; ++i) wdq[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // epilogue constants of a finished plane (patch ordinal pe, plane zo); !valid: inv = 0, zero bias and lane offsets past the
    // arrays turn the epilogue of a step without a finished plane into a no-op
    // tau_u: the marking threshold as an integer key, (bits(tau) - 1) >> 1, 0 when nothing is to be marked (epi_row)
    struct Epi { float inv; f32x4 b4; float tau; unsigned tau_u; unsigned zadj; unsigned off_w, off_b, row_f, bits_s; };
    auto epi_setup = [&](bool valid, int pe, int zo) __attribute__((always_inline)) {
        Epi e;
        const int p = valid ? (int)patch_of(pe) : 0;
        const int ce = valid ? patch_exp(p) : 0;
        e.inv = valid ? __builtin_ldexpf(1.f, -(ce + a.e_w)) : 0.f;
        e.b4 = valid ? bias4 : f32x4{0.f, 0.f, 0.f, 0.f};
        // (an all-zero patch: every pre-activation IS its bias in both arithmetics - nothing to mark)
        e.tau = (valid && (amaxA_c[p] | amaxB_c[p]) != 0u) ? __builtin_ldexpf(a.flip_tau, 14 - ce) : 0.f;
        e.tau_u = e.tau > 0.f ? (__builtin_bit_cast(unsigned, e.tau) - 1u) >> 1 : 0u;
        // a pre-activation that is EXACTLY +0 needs no second look only when this lane's four biases are zero (then it is an
        // all-zero window, +0 in the exact evaluation too); under a non-zero bias +0 can be acc * inv == -bias, or inputs flushed by
        // the fp16 split, with a tiny positive exact value: those are marked (key 0 instead of 0x7fffffff)
        e.zadj = (bias4.x == 0.f && bias4.y == 0.f && bias4.z == 0.f && bias4.w == 0.f) ? 1u : 0u;
        e.off_w = valid ? epi_lane_f * 4u : OOB;
        e.off_b = valid ? (epi_lane_f >> 2) : OOB;
        e.row_f = (unsigned)(zo * 32 + wave * 8) * 256u;       // float offset of row 8 w of that plane inside the patch
        e.bits_s = (unsigned)p * (patch_f >> 2);
        return e;
    };
    auto wd_load = [&](const Epi &E, auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        if constexpr (C3_ABL & 32) return;
        wdq[i % C3_WD] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wd_rsrc, (int)E.off_w, (int)((E.row_f + (unsigned)i * 256u) * 4u), 0));
    };

    auto wd_first = [&](const Epi &E) __attribute__((always_inline)) {      // rows 0 .. C3_WD - 1 of the plane finished next
        wd_load(E, IC<0>{});
        if constexpr (C3_WD > 1) wd_load(E, IC<1>{});
        if constexpr (C3_WD > 2) wd_load(E, IC<2>{});
        if constexpr (C3_WD > 3) wd_load(E, IC<3>{});
        if constexpr (C3_WD > 4) wd_load(E, IC<4>{});
        if constexpr (C3_WD > 5) wd_load(E, IC<5>{});
        if constexpr (C3_WD > 6) wd_load(E, IC<6>{});
        if constexpr (C3_WD > 7) wd_load(E, IC<7>{});
    };
    // ---- epilogue of one finished x row (plane zo, row 8 w + i) of accumulator set S -------------------------------
    auto epi_row = [&](auto S, auto I, const Epi &E) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value, i = decltype(I)::value;
        const f32x4 w4 = wdq[i % C3_WD];
        if constexpr (i + C3_WD < 8) wd_load(E, IC<i + C3_WD>{});
        const f32x4 c = acc[s][i];
        f32x4 val;
        if constexpr (ONEACC) {
            val.x = __builtin_fmaf(c.x, E.inv, E.b4.x); val.y = __builtin_fmaf(c.y, E.inv, E.b4.y);
            val.z = __builtin_fmaf(c.z, E.inv, E.b4.z); val.w = __builtin_fmaf(c.w, E.inv, E.b4.w);
        } else {
            const f32x4 d = accx[s][i];
            val.x = __builtin_fmaf(__builtin_fmaf(d.x, 0x1p-11f, c.x), E.inv, E.b4.x);
            val.y = __builtin_fmaf(__builtin_fmaf(d.y, 0x1p-11f, c.y), E.inv, E.b4.y);
            val.z = __builtin_fmaf(__builtin_fmaf(d.z, 0x1p-11f, c.z), E.inv, E.b4.z);
            val.w = __builtin_fmaf(__builtin_fmaf(d.w, 0x1p-11f, c.w), E.inv, E.b4.w);
        }
        unsigned unsure = 0u;
        if constexpr (SUMS) {
            const float mn = fminf(fminf(__builtin_fabsf(val.x), __builtin_fabsf(val.y)), fminf(__builtin_fabsf(val.z), __builtin_fabsf(val.w)));
#if C3_EPI2
            // mn, tau >= 0: their bit patterns order like the numbers, and the sign of the difference is the answer (no compare, no
            // VCC round trip with its wait states).  A pre-activation that is EXACTLY +0 under ZERO biases is not marked: that is an
            // all-zero window (the reference's initial weights on a zero-padded volume), +0 in the exact evaluation too - marking
            // those filled the list segments of padded patches with groups that need no second look.  The key (bits - zadj) >> 1
            // sends +0 to 0x7fffffff (above every threshold) when zadj = 1 (zero biases) and to 0 (marked) otherwise, and keeps
            // the order of everything else.
            unsure = (((__builtin_bit_cast(unsigned, mn) - E.zadj) >> 1) - E.tau_u) >> 31;
#else
            unsure = (mn < E.tau && mn > 0.f) ? 16u : 0u;
#endif
        }
        val.x = fmaxf(val.x, 0.f); val.y = fmaxf(val.y, 0.f); val.z = fmaxf(val.z, 0.f); val.w = fmaxf(val.w, 0.f);
        fs += __builtin_fmaf(val.y, w4.y, val.x * w4.x) + __builtin_fmaf(val.w, w4.w, val.z * w4.z);
        if constexpr (SUMS) {
            sa += (val.x + val.y) + (val.z + val.w);
#if C3_EPI2
            // after the ReLU a value is +0 or positive (the accumulators start at +0 and never reach -0): the negated bit pattern has
            // its top bit set exactly when x > 0, and v_alignbit shifts it into the byte - two integer instructions per channel, no
            // compare / select with their VCC wait states
            const float r0 = val.x, r1 = val.y, r2 = val.z, r3 = val.w;
            unsigned nib = unsure;
            nib = __builtin_amdgcn_alignbit(nib, 0u - __builtin_bit_cast(unsigned, r3), 31);
            nib = __builtin_amdgcn_alignbit(nib, 0u - __builtin_bit_cast(unsigned, r2), 31);
            nib = __builtin_amdgcn_alignbit(nib, 0u - __builtin_bit_cast(unsigned, r1), 31);
            nib = __builtin_amdgcn_alignbit(nib, 0u - __builtin_bit_cast(unsigned, r0), 31);
#else
            const unsigned nib = (val.x > 0.f ? 1u : 0u) | (val.y > 0.f ? 2u : 0u) | (val.z > 0.f ? 4u : 0u) | (val.w > 0.f ? 8u : 0u) | unsure;
#endif
            if constexpr (!(C3_ABL & 16))
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)nib, bits_rsrc, (int)E.off_b, (int)(E.bits_s + ((E.row_f + (unsigned)i * 256u) >> 2)), 0);
            else asm volatile("" :: "v"(nib));
        }
#if C3_PIN_SUMS
        // The two running sums are needed only when the patch ends, and the optimiser sinks their ~13 instructions per row out of the
        // sweep to the end of the step, where nothing covers them (100 vector instructions with the matrix pipe idle, the row's values
        // alive until then): this pins them to the row's block.
        if constexpr (SUMS) asm volatile("" : "+v"(fs), "+v"(sa));
        else asm volatile("" : "+v"(fs));
#endif
    };
    auto wave_sum = [&](float x) __attribute__((always_inline)) {
        int v = __builtin_bit_cast(int, x);
#define C3_ROW_SHR_ADD(n) v = __builtin_bit_cast(int, __builtin_bit_cast(float, v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x110 + (n), 0xf, 0xf, true)))
        C3_ROW_SHR_ADD(1); C3_ROW_SHR_ADD(2); C3_ROW_SHR_ADD(4); C3_ROW_SHR_ADD(8);
#undef C3_ROW_SHR_ADD
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 15));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 31));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 47));
        const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(v, 63));
        return (r0 + r1) + (r2 + r3);
    };
    auto finish_patch = [&](int p) __attribute__((always_inline)) {
        const float t = wave_sum(fs);
        if (lane == 0) a.fc_part[(size_t)p * 4 + wave] = t;
        if constexpr (SUMS) {
            const float u = wave_sum(sa);
            if (lane == 0 && a.asum_part) a.asum_part[(size_t)p * 4 + wave] = u;
        }
        fs = 0.f; sa = 0.f;
    };

    // ---- one step: input plane z of patch ordinal pi (plane ordinal n), rotation R = z mod 3 ------------------------
    // sets: zo = z - 1 -> (R + 2) % 3, zo = z -> R, zo = z + 1 -> (R + 1) % 3 (restarted row by row, after the epilogue of the
    // plane z - 2 it still holds)
    auto step = [&](auto RR, int pi, int z, long long n) __attribute__((always_inline)) {
        constexpr int R = decltype(RR)::value;
        constexpr int S_lo = (R + 2) % 3, S_mid = R, S_hi = (R + 1) % 3;
        // everything the previous step wrote into this plane's image has landed; nobody still reads the other image
        if constexpr (!(C3_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int abase = frag_lane + (int)(n & 1) * C3_PLANE;
        const int wbase = st_lane + (int)((n + 1) & 1) * C3_PLANE;
        const Cur C1 = cursor(c1_pi, c1_z), C2 = cursor(c2_pi, c2_z);
        {
            const int ce = c1_pi < np ? patch_exp((int)patch_of(c1_pi)) : 0;
            sc = __builtin_ldexpf(1.f, ce); sc11 = __builtin_ldexpf(1.f, ce + 11);
        }
        // finished plane: z - 2 of this patch, or plane D - 1 of the previous one at z = 0 (D - 2 is done by the light step)
        const bool ev = z >= 2 || (z == 0 && pi > 0);
        const Epi E = epi_setup(ev, z >= 2 ? pi : pi - 1, z >= 2 ? z - 2 : D - 1);

        auto frag = [&](int j, int s, f16x8 &bh, f16x8 &bl) __attribute__((always_inline)) {
            const char *p = lds + abase + j * C3_ROW + s * 32;
            bh = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(p));
            bl = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(p + C3_PIECE));
        };
        // A step is 20 blocks (input row j, k-step s), each its own scheduling region (nothing crosses a block boundary): the
        // fragments of block b + 2 are read in block b into one of three register sets, so a read is always a full block of
        // MFMAs (>= 430 cycles) ahead of its use wherever the scheduler puts it inside the block.  Left alone it sinks every
        // read to its first use - the reads cannot pass the staging writes above them, which may alias - and each fragment
        // then pays its LDS latency in front of an MFMA (timing builds: +325 us of 1.8 ms for the staging writes alone).
        // Inside a block: one MFMA, then up to C3_PIPE vector instructions, and so on - the conversion / epilogue arithmetic in
        // the MFMAs' shadow (an MFMA holds the issue port for 8 of its 16 cycles).
        f16x8 Fh[3], Fl[3];
        frag(0, 0, Fh[0], Fl[0]);
        frag(0, 1, Fh[1], Fl[1]);
        auto block = [&](auto J, auto SS) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value, s = decltype(SS)::value, b = 2 * j + s;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (s == 0 && j < 8) {
                // row j of the plane finished two steps ago, then (below) its accumulators restart for plane z + 1
                if constexpr (!(C3_ABL & 2)) epi_row(IC<S_hi>{}, J, E);
            }
            // The staging units of the next plane go where the MFMA stream has room for their arithmetic: rows 0 / 1 issue 18 / 36
            // MFMAs and already carry an epilogue row, rows 8 / 9 carry none - units 0..11 ride on rows 2..7, 12..15 on rows 8, 9
            if constexpr (!(C3_ABL & 1) && j >= 2) {
                constexpr int u = 2 * (j - 2) + s;
                stage_unit(wbase, IC<u>{});
                if constexpr (u + C3_PF < 16) load_unit(C1.voff, C1.soff, IC<u + C3_PF>{}); else load_unit(C2.voff, C2.soff, IC<u + C3_PF - 16>{});
            }
            if constexpr (b + 2 < 20) frag((b + 2) / 2, (b + 2) & 1, Fh[(b + 2) % 3], Fl[(b + 2) % 3]);
            const f16x8 bh = Fh[b % 3], bl = Fl[b % 3];
            constexpr int rows_here = (j < 2 ? j + 1 : 3) < (10 - j) ? (j < 2 ? j + 1 : 3) : (10 - j);
#pragma unroll
            for (int di = 0; di < 3; ++di) {
                const int i = j - di;          // output row; y tap index = di
                if (i < 0 || i > 7) continue;
#pragma unroll
                for (int dzi = 0; dzi < 3; ++dzi) {      // z tap index: output plane z + 1 - dzi
                    const int k = (dzi * 3 + di) * 2 + s;
                    const int set = dzi == 0 ? S_hi : (dzi == 1 ? S_mid : S_lo);
                    const bool start = dzi == 0 && di == 0 && s == 0;       // first contribution to (plane z + 1, row i)
                    f32x4 c = acc[set][i];
                    if (start) c = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (ONEACC) {      // the small products first
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wl[k], bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wh[k], bl, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wh[k], bh, c, 0, 0, 0);
                    } else {
                        f32x4 d = accx[set][i];
                        if (start) d = f32x4{0.f, 0.f, 0.f, 0.f};
                        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wl[k], bh, d, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wh[k], bh, c, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wh[k], bl, d, 0, 0, 0);
                        accx[set][i] = d;
                    }
                    acc[set][i] = c;
                }
            }
#if C3_PIPE
#pragma unroll
            for (int m = 0; m < 9 * rows_here; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(C3_FILL_MASK, C3_PIPE, 0);
            }
#endif
        };
        auto row = [&](auto J) __attribute__((always_inline)) { block(J, IC<0>{}); block(J, IC<1>{}); };
        row(IC<0>{}); row(IC<1>{}); row(IC<2>{}); row(IC<3>{}); row(IC<4>{});
        row(IC<5>{}); row(IC<6>{}); row(IC<7>{}); row(IC<8>{}); row(IC<9>{});
        {   // the weight difference for rows 0 and 1 of the plane the next step finishes: plane z - 1 (z + 1 < D), or plane D - 2 in
            // the light step behind z = D - 1 (the vector is the same for every patch)
            const Epi En = epi_setup(z + 1 < D ? z >= 1 : true, pi, z + 1 < D ? z - 1 : D - 2);
            wd_first(En);
        }
        if constexpr (R == 0) { if (z == 0 && pi > 0) finish_patch((int)patch_of(pi - 1)); }
        c1_pi = c2_pi; c1_z = c2_z;
        advance(c2_pi, c2_z);
    };
    // a step without an input plane: only the epilogue of a finished plane held by set S (plane D - 2 after the last
    // input plane of a patch, plane D - 1 after the last patch), then the set is cleared
    auto light = [&](auto S, int pe, int zo) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        const Epi E = epi_setup(true, pe, zo);
        auto rows = [&](auto I) __attribute__((always_inline)) { epi_row(S, I, E); };
        rows(IC<0>{}); rows(IC<1>{}); rows(IC<2>{}); rows(IC<3>{}); rows(IC<4>{}); rows(IC<5>{}); rows(IC<6>{}); rows(IC<7>{});
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[s][i] = f32x4{0.f, 0.f, 0.f, 0.f}; if constexpr (!ONEACC) accx[s][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        {   // plane D - 1 is finished next (step 0 of the next patch, or the final light step); nothing after that one
            const Epi En = epi_setup(zo == D - 2, pe, D - 1);
            wd_first(En);
        }
    };

    // ---- prologue: plane 0 of the first patch, synchronously; the first 8 loads of plane 1 in flight ------------------
    __syncthreads();       // the zero fill
    if (np > 0) {
        const Cur C0 = cursor(0, 0);
        const int ce = patch_exp((int)patch_of(0));
        sc = __builtin_ldexpf(1.f, ce); sc11 = __builtin_ldexpf(1.f, ce + 11);
        const int wb = st_lane;       // image 0
        auto eight = [&](auto U0) __attribute__((always_inline)) {
            constexpr int u0 = decltype(U0)::value;
            load_unit(C0.voff, C0.soff, IC<u0>{}); load_unit(C0.voff, C0.soff, IC<u0 + 1>{}); load_unit(C0.voff, C0.soff, IC<u0 + 2>{}); load_unit(C0.voff, C0.soff, IC<u0 + 3>{});
            load_unit(C0.voff, C0.soff, IC<u0 + 4>{}); load_unit(C0.voff, C0.soff, IC<u0 + 5>{}); load_unit(C0.voff, C0.soff, IC<u0 + 6>{}); load_unit(C0.voff, C0.soff, IC<u0 + 7>{});
            stage_unit(wb, IC<u0>{}); stage_unit(wb, IC<u0 + 1>{}); stage_unit(wb, IC<u0 + 2>{}); stage_unit(wb, IC<u0 + 3>{});
            stage_unit(wb, IC<u0 + 4>{}); stage_unit(wb, IC<u0 + 5>{}); stage_unit(wb, IC<u0 + 6>{}); stage_unit(wb, IC<u0 + 7>{});
        };
        eight(IC<0>{}); eight(IC<8>{});
        c1_pi = 0; c1_z = 0;
        advance(c1_pi, c1_z);          // plane 1
        c2_pi = c1_pi; c2_z = c1_z;
        advance(c2_pi, c2_z);          // plane 2
        const Cur C1 = cursor(c1_pi, c1_z);
        load_unit(C1.voff, C1.soff, IC<0>{}); load_unit(C1.voff, C1.soff, IC<1>{}); load_unit(C1.voff, C1.soff, IC<2>{}); load_unit(C1.voff, C1.soff, IC<3>{});
        load_unit(C1.voff, C1.soff, IC<4>{}); load_unit(C1.voff, C1.soff, IC<5>{}); load_unit(C1.voff, C1.soff, IC<6>{}); load_unit(C1.voff, C1.soff, IC<7>{});
        if constexpr (C3_PF == 16) {
            load_unit(C1.voff, C1.soff, IC<8>{}); load_unit(C1.voff, C1.soff, IC<9>{}); load_unit(C1.voff, C1.soff, IC<10>{}); load_unit(C1.voff, C1.soff, IC<11>{});
            load_unit(C1.voff, C1.soff, IC<12>{}); load_unit(C1.voff, C1.soff, IC<13>{}); load_unit(C1.voff, C1.soff, IC<14>{}); load_unit(C1.voff, C1.soff, IC<15>{});
        }
    }

    // D = 32: 33 = 3 * 11 steps per patch (the last one light), so the set rotation restarts at every patch
    long long n = 0;
    for (int pi = 0; pi < np; ++pi) {
#if C3_PEEL
        for (int zz = 0; zz < 10; ++zz) {
            step(IC<0>{}, pi, 3 * zz, n); ++n;
            step(IC<1>{}, pi, 3 * zz + 1, n); ++n;
            step(IC<2>{}, pi, 3 * zz + 2, n); ++n;
        }
        step(IC<0>{}, pi, 30, n); ++n;
        step(IC<1>{}, pi, 31, n); ++n;
        light(IC<0>{}, pi, D - 2);
#else
        for (int zz = 0; zz < 11; ++zz) {
            step(IC<0>{}, pi, 3 * zz, n); ++n;
            step(IC<1>{}, pi, 3 * zz + 1, n); ++n;
            if (zz < 10) { step(IC<2>{}, pi, 3 * zz + 2, n); ++n; }
            else light(IC<0>{}, pi, D - 2);
        }
#endif
    }
    if (np > 0) {
        light(IC<1>{}, np - 1, D - 1);
        finish_patch((int)patch_of(np - 1));
    }
#ifdef C3_CLK
    if (a.clk && tid == 0) { a.clk[2 * b0] = __builtin_amdgcn_s_memtime() - clk0; a.clk[2 * b0 + 1] = __builtin_amdgcn_s_memrealtime() - rt0; }
#endif
}

// ======================================================================================================================
// Backward of the same conv (8 -> 16 channels, flipped taps) in a Fisher pass.  Its input is not a tensor: the cotangent of
// the conv's output under the unit cotangent is [sign of output element] x (W0 - W1) of the head - the sign bytes the
// forward kernel left (after the flip fix-up) and ONE vector shared by all patches, kept pre-split as fp16 pairs
// ([voxel][h8 | l8], c3d_presplit_vec).  Same sweep: wave w owns output rows 8 w .. 8 w + 7, an MFMA column block is half an x
// row (16 voxels), the four k-groups of a k-step are the x offsets -1, 0, +1 (and one zero group), one k-step per (dz, dy).
// Output channels 0..7 (the skip source, the first conv): masked by that layer's sign field and only summed per voxel
// (nothing below it needs the cotangent itself); channels 8..15 (the conv_transpose in front): stored + summed per voxel.
namespace {
constexpr int B3_TEN = 34 * 16;          // one piece of one row: slots x = -1 .. 32, 8 fp16 channels each
constexpr int B3_ROW = 2 * B3_TEN;
}  // namespace

struct C3BwdArgs {
    const unsigned char *bits;     // [N][D * 32 * 32 * 2] sign byte per 4 output channels of the forward conv (low nibble)
    const void *vec;               // [D * 32 * 32][h8 | l8] fp16: (W0 - W1) 2^e_in split (c3d_presplit_vec)
    const void *W;                 // [9 k-steps][2 pieces][64 lanes][8] fp16 (c3d_bwd_pack)
    const unsigned char *maskA;    // [N][D * 32 * 32 * 2] sign field of the tensor behind output channels 0..7, or null (no ReLU)
    float *dB;                     // [N, D, 32, 32, 8] cotangent of channels 8..15
    float *sumA, *sumB;            // [N][D * 32 * 32] per-voxel sums of the (masked) channels 0..7 / of channels 8..15
    int N, D;
    int e_in, e_w;                 // scale exponents of the vector and of the packed weights
};

// RW = output rows per wave.  8 (default): one workgroup per patch like the forward kernel - 192 accumulators + 72 weight
// registers; since the sweep loop is peeled (no join inside a patch, so no accumulator copies) it fits without scratch.
// 4 (ALQ_C3D_BWD_ROWS=4): a workgroup owns HALF the rows of a patch (16 + 2 halo rows per plane image, staged 1.125 x), 96
// accumulators, weights pinned to the AGPR half, 128 registers per lane left for co-resident side-stream kernels.
template <bool ONEACC, int RW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void c3d_bwd_kernel(const C3BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    static_assert(ONEACC, "the backward kernel is built for the one-accumulator form");
    static_assert(RW == 4 || RW == 8, "4 waves x RW rows: a half or a whole patch per workgroup");
    constexpr int NR = 4 * RW;                   // output rows of a workgroup
    constexpr int NPARTS = 32 / NR;              // workgroups (work items) per patch
    constexpr int PLANE = (NR + 2) * B3_ROW;     // image rows y0 - 1 .. y0 + NR
    constexpr int DUMMY = 2 * PLANE;             // 2 KB nobody reads: where the idle half of a halo staging unit writes
    constexpr int NU = RW / 2;                   // own staging units (two rows each) per wave and plane
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int D = a.D;
    const unsigned plane_v = 32u * 32u;                      // voxels per plane
    const unsigned patch_v = (unsigned)D * plane_v;

    for (int i = tid * 16; i < DUMMY + 2 * B3_ROW; i += 256 * 16) *reinterpret_cast<i32x4 *>(lds + i) = i32x4{0, 0, 0, 0};

    f16x8 Wh[9], Wl[9];
    {
        const i32x4 *Wg = reinterpret_cast<const i32x4 *>(a.W);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            Wh[k] = __builtin_bit_cast(f16x8, Wg[(k * 2 + 0) * 64 + lane]);
            Wl[k] = __builtin_bit_cast(f16x8, Wg[(k * 2 + 1) * 64 + lane]);
        }
        // RW = 4: the weights start their lives in accumulation registers like the forward kernel's (96 accumulators + 72 weight
        // registers fit the AGPR half); RW = 8: the 192 accumulators fill three quarters of it and the allocator does better alone
        // (timing builds: 1.45 ms per 2000 patches against 1.47 ms with the weights pinned)
        if constexpr (RW == 4) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                i32x4 h = __builtin_bit_cast(i32x4, Wh[k]), l = __builtin_bit_cast(i32x4, Wl[k]);
                asm volatile("" : "+a"(h), "+a"(l));
                Wh[k] = __builtin_bit_cast(f16x8, h); Wl[k] = __builtin_bit_cast(f16x8, l);
            }
        }
    }
    // fragment reads: column j = voxel 16 hx + j, k-group kg = x offset kg - 1 (group 3 has zero weights: it re-reads group 2's slot)
    const int frag_lane = (lj + (lq < 2 ? lq : 2)) * 16 + wave * RW * B3_ROW;
    // staging: a unit = two rows of the wave's strip, lane = (row lane >> 5, x = lane & 31); image row = y - y0 + 1
    const int st_lane = (wave * RW + 1 + (lane >> 5)) * B3_ROW + ((lane & 31) + 1) * 16;
    const float inv = __builtin_ldexpf(1.f, -(a.e_in + a.e_w));

    const int G = gridDim.x, b0 = blockIdx.x;
    const int nitems = a.N * NPARTS;
    const int np = b0 < nitems ? (nitems - b0 + G - 1) / G : 0;       // work items (patch, part) of this workgroup

    f32x4 acc[3][RW][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < RW; ++i) { acc[s][i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[s][i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // ONE buffer resource per array for the whole launch (no per-step descriptor building, few scalar registers): the patch
    // goes into the scalar offset - which the hardware does not range-check - and a lane with nothing to load or store aims
    // past the array through its VECTOR offset (loads then return 0, stores are dropped).  N < 4096 patches keep every
    // offset below 2^32 (the host checks).
    auto rsrc_of = [&](const void *base, unsigned long long bytes) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(unsigned)bytes, 0x00020000);
    };
    const unsigned OOB = 0xffffff00u;
    const __amdgpu_buffer_rsrc_t vec_rsrc = rsrc_of(a.vec, (unsigned long long)patch_v * 32u);
    const __amdgpu_buffer_rsrc_t bits_rsrc = rsrc_of(a.bits, (unsigned long long)a.N * patch_v * 2u);
    const __amdgpu_buffer_rsrc_t mask_rsrc = rsrc_of(a.maskA, a.maskA ? (unsigned long long)a.N * patch_v * 2u : 0ull);
    const __amdgpu_buffer_rsrc_t dB_rsrc = rsrc_of(a.dB, (unsigned long long)a.N * patch_v * 32u);
    const __amdgpu_buffer_rsrc_t sA_rsrc = rsrc_of(a.sumA, (unsigned long long)a.N * patch_v * 4u);
    const __amdgpu_buffer_rsrc_t sB_rsrc = rsrc_of(a.sumB, (unsigned long long)a.N * patch_v * 4u);
    auto item_patch = [&](int pi) __attribute__((always_inline)) { return (unsigned)((b0 + pi * G) / NPARTS); };
    auto item_y0 = [&](int pi) __attribute__((always_inline)) { return ((b0 + pi * G) % NPARTS) * NR; };

    // ---- staging of a plane: own units u = 0 .. NU - 1 (rows y0 + w RW + 2 u, + 1), and - half patches only - one halo row per
    // edge wave (wave 0: row y0 - 1, wave 3: row y0 + NR; lanes 0..31, zeros outside the patch) ----------------------------
    i32x4 Vh[2], Vl[2], Hh, Hl;
    unsigned Sb[2], Hb = 0u;
    struct Src { unsigned voff, vvox, soff; };      // lane offset into the sign bytes (or OOB), lane voxel inside the plane, patch offset
    auto src_of = [&](int pi, bool halo) __attribute__((always_inline)) {
        Src c;
        const bool ok = pi < np;
        const int y0 = ok ? item_y0(pi) : 0;
        int y = y0 + wave * RW + (lane >> 5);
        bool live = ok;
        if (halo) {
            y = wave == 0 ? y0 - 1 : y0 + NR;
            live = ok && lane < 32 && y >= 0 && y < 32;
        }
        c.vvox = (unsigned)((live ? y : 0) * 32 + (lane & 31));
        c.voff = live ? c.vvox * 2u : OOB;
        c.soff = ok ? item_patch(pi) * patch_v * 2u : 0u;
        return c;
    };
    auto mask8 = [&](int b, i32x4 h, i32x4 l, i32x4 *oh, i32x4 *ol) __attribute__((always_inline)) {
        // bits 0..3: channels 0..3, bits 8..11: channels 4..7
        const unsigned m0 = (unsigned)__builtin_amdgcn_sbfe(b, 0, 1), m1 = (unsigned)__builtin_amdgcn_sbfe(b, 1, 1);
        const unsigned m2 = (unsigned)__builtin_amdgcn_sbfe(b, 2, 1), m3 = (unsigned)__builtin_amdgcn_sbfe(b, 3, 1);
        const unsigned m4 = (unsigned)__builtin_amdgcn_sbfe(b, 8, 1), m5 = (unsigned)__builtin_amdgcn_sbfe(b, 9, 1);
        const unsigned m6 = (unsigned)__builtin_amdgcn_sbfe(b, 10, 1), m7 = (unsigned)__builtin_amdgcn_sbfe(b, 11, 1);
        const unsigned k0 = (m0 & 0xffffu) | (m1 & 0xffff0000u), k1 = (m2 & 0xffffu) | (m3 & 0xffff0000u);
        const unsigned k2 = (m4 & 0xffffu) | (m5 & 0xffff0000u), k3 = (m6 & 0xffffu) | (m7 & 0xffff0000u);
        *oh = i32x4{(int)((unsigned)h.x & k0), (int)((unsigned)h.y & k1), (int)((unsigned)h.z & k2), (int)((unsigned)h.w & k3)};
        *ol = i32x4{(int)((unsigned)l.x & k0), (int)((unsigned)l.y & k1), (int)((unsigned)l.z & k2), (int)((unsigned)l.w & k3)};
    };
    auto load_unit = [&](const Src &c, unsigned z, auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        const unsigned so = (z * plane_v + (unsigned)u * 64u);
        // (the vector is patch-independent: a dead lane reads voxel 0 of it and its sign bits come back 0)
        Vh[u & 1] = __builtin_amdgcn_raw_buffer_load_b128(vec_rsrc, (int)(c.vvox * 32u), (int)(so * 32u), 0);
        Vl[u & 1] = __builtin_amdgcn_raw_buffer_load_b128(vec_rsrc, (int)(c.vvox * 32u + 16u), (int)(so * 32u), 0);
        Sb[u & 1] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(bits_rsrc, (int)c.voff, (int)(c.soff + so * 2u), 0);
    };
    auto stage_unit = [&](int wbase, auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        i32x4 oh, ol;
        mask8((int)Sb[u & 1], Vh[u & 1], Vl[u & 1], &oh, &ol);
        char *dst = lds + wbase + (2 * u) * B3_ROW;
        *reinterpret_cast<i32x4 *>(dst) = oh;
        *reinterpret_cast<i32x4 *>(dst + B3_TEN) = ol;
    };
    auto load_halo = [&](const Src &c, unsigned z) __attribute__((always_inline)) {
        Hh = __builtin_amdgcn_raw_buffer_load_b128(vec_rsrc, (int)(c.vvox * 32u), (int)(z * plane_v * 32u), 0);
        Hl = __builtin_amdgcn_raw_buffer_load_b128(vec_rsrc, (int)(c.vvox * 32u + 16u), (int)(z * plane_v * 32u), 0);
        Hb = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(bits_rsrc, (int)c.voff, (int)(c.soff + z * plane_v * 2u), 0);
    };
    auto stage_halo = [&](int image) __attribute__((always_inline)) {      // image = byte offset of the plane image
        i32x4 oh, ol;
        mask8((int)Hb, Hh, Hl, &oh, &ol);
        // lanes 0..31 -> the halo row of this wave (image row 0 or NR + 1), the others into the dummy rows
        const int row = wave == 0 ? 0 : NR + 1;
        char *dst = lds + (lane < 32 ? image + row * B3_ROW + ((lane & 31) + 1) * 16 : DUMMY + (lane & 31) * 16);
        *reinterpret_cast<i32x4 *>(dst) = oh;
        *reinterpret_cast<i32x4 *>(dst + B3_TEN) = ol;
    };
    const bool halo_wave = NPARTS > 1 && (wave == 0 || wave == 3);

    int c1_pi = 0, c1_z = 0, c2_pi = 0, c2_z = 0;
    auto advance = [&](int &pi, int &z) __attribute__((always_inline)) { if (++z == D) { z = 0; ++pi; } };

    // ---- epilogue of one finished x row (plane zo, row y0 + w RW + i) of set S: both halves --------------------------------
    // lane (j, q): channels 4 q .. 4 q + 3 of voxel 16 hx + j; q < 2: masked + summed, q >= 2: stored + summed
    unsigned mkq[2][2] = {{0u, 0u}, {0u, 0u}};      // sign bytes of the masked half, fetched two rows ahead: [row parity][hx]
    struct Epi { unsigned pv; unsigned row_v; float inv; unsigned off_mask, off_dB, off_sum; };
    auto epi_setup = [&](bool valid, int pe, int zo) __attribute__((always_inline)) {
        Epi e;
        e.pv = valid ? item_patch(pe) * patch_v : 0u;          // first voxel of the patch in the batch
        e.row_v = (unsigned)(zo * 32 + (valid ? item_y0(pe) : 0) + wave * RW) * 32u;
        e.inv = valid ? inv : 0.f;
        // lane offsets; an invalid step (no finished plane) sends every access past the arrays
        e.off_mask = (valid && lq < 2) ? (unsigned)(lj * 2 + lq) : OOB;
        e.off_dB = (valid && lq >= 2) ? (unsigned)lj * 32u + (unsigned)(lq - 2) * 16u : OOB;
        // sums: after the cross-row add, lanes q = 0 / 1 hold the masked half's sum of half row 0 / 1 (x = j / 16 + j), lanes
        // q = 2 / 3 the stored half's: one 128-byte store per field and row
        e.off_sum = valid ? (unsigned)((lq & 1) * 16 + lj) * 4u : OOB;
        return e;
    };
    auto mask_load = [&](const Epi &E, auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
#pragma unroll
        for (int hx = 0; hx < 2; ++hx)
            mkq[i & 1][hx] = (unsigned char)__builtin_amdgcn_raw_buffer_load_b8(mask_rsrc, (int)E.off_mask,
                                                                               (int)((E.pv + E.row_v + (unsigned)i * 32u + (unsigned)hx * 16u) * 2u), 0);
    };
    const bool has_mask = a.maskA != nullptr;
    float t_keep = 0.f;      // the 8-channel sums of half row 0 (even 16-lane rows) until half row 1 joins them
    // one half row (16 voxels): the second one fetches the sign bytes two rows ahead and stores the sums of both
    auto epi_half = [&](auto S, auto I, auto HX, const Epi &E) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value, i = decltype(I)::value, hx = decltype(HX)::value;
        const unsigned mk = mkq[i & 1][hx];
        if constexpr (hx == 1 && i + 2 < RW) mask_load(E, IC<i + 2>{});      // (both halves' bytes of row i + 2, after this row's were read)
        const f32x4 c = acc[s][i][hx];
        const int nib = (lq < 2 && has_mask) ? (int)mk : 15;
        // (scalars first: __builtin_bit_cast applied to an element of an ext_vector lvalue reads element 0)
        const float v0 = c.x * E.inv, v1 = c.y * E.inv, v2 = c.z * E.inv, v3 = c.w * E.inv;
        f32x4 val;
        // (the empty asm hides the 0 / -1 masks from the optimiser, which otherwise rewrites "x & sext(bit)" into bit test + compare +
        // select: three vector instructions and a wait state per value instead of v_bfe_i32 + v_and_b32)
        unsigned k0 = (unsigned)__builtin_amdgcn_sbfe(nib, 0, 1), k1 = (unsigned)__builtin_amdgcn_sbfe(nib, 1, 1);
        unsigned k2 = (unsigned)__builtin_amdgcn_sbfe(nib, 2, 1), k3 = (unsigned)__builtin_amdgcn_sbfe(nib, 3, 1);
        asm("" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3));
        val.x = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v0) & k0);
        val.y = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v1) & k1);
        val.z = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v2) & k2);
        val.w = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v3) & k3);
        const unsigned vox = E.pv + E.row_v + (unsigned)i * 32u + (unsigned)hx * 16u;      // first voxel of this half row
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, val), dB_rsrc, (int)E.off_dB, (int)(vox * 32u), 2 /* nt */);
#ifndef C3_NO_HOLD
        ALQ_STORE_HOLD("v"(val));      // (16-byte store with a register soffset: alq_internal.h)
#endif
        const float t = (val.x + val.y) + (val.z + val.w);
        // lanes (j, q) and (j, q ^ 1) hold the two 4-channel groups of one voxel.  v_permlane16_swap(a, b) returns
        // ([a.row0, b.row0, a.row2, b.row2], [a.row1, b.row1, a.row3, b.row3]) (rows of 16 lanes; probed on the device,
        // tools/probe/permlane_probe.hip): with (t, 0) the two results add up to the 8-channel sums in the EVEN rows, with (0, t)
        // in the ODD rows - half row 0 goes to lanes q = 0 / 2, half row 1 to lanes q = 1 / 3, which is how the merged stores
        // below want them.  (Never (t, t): the compiler folds the two results of identical operands into one.)
        const unsigned tu = __builtin_bit_cast(unsigned, t);
        const auto sw = hx == 0 ? __builtin_amdgcn_permlane16_swap(tu, 0u, false, false) : __builtin_amdgcn_permlane16_swap(0u, tu, false, false);
        const unsigned s0 = sw[0], s1 = sw[1];       // (scalars first: see the note on __builtin_bit_cast above)
        const float t2 = __builtin_bit_cast(float, s0) + __builtin_bit_cast(float, s1);
        if constexpr (hx == 0) {
            t_keep = t2;
        } else {
            const float ts = t_keep + t2;
            const unsigned rowv = E.pv + E.row_v + (unsigned)i * 32u;
            // lanes q < 2 -> the masked half's field, q >= 2 -> the stored half's: two stores, each with the other half of the wave off
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ts), sA_rsrc, (int)(lq < 2 ? E.off_sum : OOB), (int)(rowv * 4u), 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ts), sB_rsrc, (int)(lq >= 2 ? E.off_sum : OOB), (int)(rowv * 4u), 0);
        }
    };
    auto epi_row = [&](auto S, auto I, const Epi &E) __attribute__((always_inline)) {
        epi_half(S, I, IC<0>{}, E);
        epi_half(S, I, IC<1>{}, E);
    };

    auto step = [&](auto RR, int pi, int z, long long n) __attribute__((always_inline)) {
        constexpr int R = decltype(RR)::value;
        constexpr int S_lo = (R + 2) % 3, S_mid = R, S_hi = (R + 1) % 3;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int abase = frag_lane + (int)(n & 1) * PLANE;
        const int wimage = (int)((n + 1) & 1) * PLANE;
        const int wbase = st_lane + wimage;
        const Src C1 = src_of(c1_pi, false), C2 = src_of(c2_pi, false);
        const Src H2 = src_of(c2_pi, true);
        const unsigned z1 = (unsigned)c1_z, z2 = (unsigned)c2_z;
        const bool ev = z >= 2 || (z == 0 && pi > 0);
        const Epi E = epi_setup(ev, z >= 2 ? pi : pi - 1, z >= 2 ? z - 2 : D - 1);

        auto frag = [&](int j, int hx, f16x8 &bh, f16x8 &bl) __attribute__((always_inline)) {
            const char *p = lds + abase + j * B3_ROW + hx * 256;
            bh = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(p));
            bl = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(p + B3_TEN));
        };
        // One scheduling region per input row (nothing crosses a row boundary: left alone the scheduler sinks the staging loads to
        // their uses); inside it one MFMA : C3_BPIPE vector instructions.
        f16x8 bh, bl, nh, nl;
        frag(0, 0, bh, bl);
        auto row = [&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j < RW) epi_row(IC<S_hi>{}, J, E);
            // staging of the next plane: RW = 8: unit j / 2 on the even rows below 8; RW = 4: the epilogue rows 0..3 carry enough,
            // the halo row rides on row 3, the two own units on rows 4 and 5.  The unit after the next one goes out behind each.
            if constexpr (RW == 8) {
                if constexpr (j < 8 && (j & 1) == 0) {
                    stage_unit(wbase, IC<j / 2>{});
                    if constexpr (j / 2 + 2 < NU) load_unit(C1, z1, IC<j / 2 + 2>{}); else load_unit(C2, z2, IC<j / 2 + 2 - NU>{});
                }
            } else {
                if constexpr (j == 3) {
                    if (halo_wave) { stage_halo(wimage); load_halo(H2, z2); }
                }
                if constexpr (j >= 4) {
                    stage_unit(wbase, IC<j - 4>{});
                    load_unit(C2, z2, IC<j - 4>{});
                }
            }
#pragma unroll
            for (int hx = 0; hx < 2; ++hx) {
                if (hx == 0) frag(j, 1, nh, nl);
                else if (j < RW + 1) frag(j + 1, 0, nh, nl);
#pragma unroll
                for (int di = 0; di < 3; ++di) {
                    const int i = j - di;
                    if (i < 0 || i > RW - 1) continue;
#pragma unroll
                    for (int dzi = 0; dzi < 3; ++dzi) {
                        const int k = dzi * 3 + di;
                        const int set = dzi == 0 ? S_hi : (dzi == 1 ? S_mid : S_lo);
                        const bool start = dzi == 0 && di == 0;       // first contribution to (plane z + 1, row i, this half)
                        f32x4 c = acc[set][i][hx];
                        if (start) c = f32x4{0.f, 0.f, 0.f, 0.f};
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wl[k], bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wh[k], bl, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wh[k], bh, c, 0, 0, 0);
                        acc[set][i][hx] = c;
                    }
                }
                bh = nh; bl = nl;
            }
#if C3_BPIPE
            {
                constexpr int rows_here = (j < 2 ? j + 1 : 3) < (RW + 2 - j) ? (j < 2 ? j + 1 : 3) : (RW + 2 - j);
#pragma unroll
                for (int m = 0; m < 18 * rows_here; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(C3_FILL_MASK, C3_BPIPE, 0);
                }
            }
#endif
        };
        row(IC<0>{}); row(IC<1>{}); row(IC<2>{}); row(IC<3>{}); row(IC<4>{}); row(IC<5>{});
        if constexpr (RW == 8) { row(IC<6>{}); row(IC<7>{}); row(IC<8>{}); row(IC<9>{}); }
        {   // the sign bytes of rows 0 and 1 of the plane the next step finishes (same item: plane z - 1, or D - 2 in the light step)
            const Epi En = epi_setup(z + 1 < D ? z >= 1 : true, pi, z + 1 < D ? z - 1 : D - 2);
            mask_load(En, IC<0>{});
            mask_load(En, IC<1>{});
        }
        c1_pi = c2_pi; c1_z = c2_z;
        advance(c2_pi, c2_z);
    };
    auto light = [&](auto S, int pe, int zo) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        const Epi E = epi_setup(true, pe, zo);
        auto rows = [&](auto I) __attribute__((always_inline)) { epi_row(S, I, E); };
        rows(IC<0>{}); rows(IC<1>{}); rows(IC<2>{}); rows(IC<3>{});
        if constexpr (RW == 8) { rows(IC<4>{}); rows(IC<5>{}); rows(IC<6>{}); rows(IC<7>{}); }
#pragma unroll
        for (int i = 0; i < RW; ++i) { acc[s][i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[s][i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        {   // plane D - 1 of the same item is finished next (after the D - 2 light step); nothing after the final one
            const Epi En = epi_setup(zo == D - 2, pe, D - 1);
            mask_load(En, IC<0>{});
            mask_load(En, IC<1>{});
        }
    };

    __syncthreads();
    if (np > 0) {
        // plane 0 of the first item synchronously, then the first loads of plane 1 (RW = 8: two units ahead; RW = 4: the whole
        // plane - two units and the halo row)
        const Src C0 = src_of(0, false), H0 = src_of(0, true);
        const int wb = st_lane;
        load_unit(C0, 0u, IC<0>{}); load_unit(C0, 0u, IC<1>{});
        stage_unit(wb, IC<0>{}); stage_unit(wb, IC<1>{});
        if constexpr (RW == 8) {
            load_unit(C0, 0u, IC<2>{}); load_unit(C0, 0u, IC<3>{});
            stage_unit(wb, IC<2>{}); stage_unit(wb, IC<3>{});
        } else if (halo_wave) {
            load_halo(H0, 0u);
            stage_halo(0);
        }
        c1_pi = 0; c1_z = 0;
        advance(c1_pi, c1_z);
        c2_pi = c1_pi; c2_z = c1_z;
        advance(c2_pi, c2_z);
        const Src C1 = src_of(c1_pi, false);
        load_unit(C1, (unsigned)c1_z, IC<0>{}); load_unit(C1, (unsigned)c1_z, IC<1>{});
        if constexpr (RW == 4) { if (halo_wave) load_halo(src_of(c1_pi, true), (unsigned)c1_z); }
    }
    long long n = 0;
    for (int pi = 0; pi < np; ++pi) {
#if C3_PEEL
        for (int zz = 0; zz < 10; ++zz) {
            step(IC<0>{}, pi, 3 * zz, n); ++n;
            step(IC<1>{}, pi, 3 * zz + 1, n); ++n;
            step(IC<2>{}, pi, 3 * zz + 2, n); ++n;
        }
        step(IC<0>{}, pi, 30, n); ++n;
        step(IC<1>{}, pi, 31, n); ++n;
        light(IC<0>{}, pi, D - 2);
#else
        for (int zz = 0; zz < 11; ++zz) {
            step(IC<0>{}, pi, 3 * zz, n); ++n;
            step(IC<1>{}, pi, 3 * zz + 1, n); ++n;
            if (zz < 10) { step(IC<2>{}, pi, 3 * zz + 2, n); ++n; }
            else light(IC<0>{}, pi, D - 2);
        }
#endif
    }
    if (np > 0) light(IC<1>{}, np - 1, D - 1);
}

// ======================================================================================================================
// The same backward launch with the 27 taps PACKED INTO 7 K-STEPS instead of 9 (round 6).  The kernel above spends one k-step per
// (dz, dy) on the three x offsets and a ZERO fourth k-group: a quarter of its MFMAs multiply zeros.  Here a k-group is a TAP
// addressed per lane, and the nine in-plane taps of an output voxel, taken in row-major order p = 3 dy + dx, are cut into
//   A = p 0..3: row y - 1 at x - 1, x, x + 1 and row y at x - 1        (one k-step)
//   B = p 4..7: row y at x, x + 1 and row y + 1 at x - 1, x            (one k-step)
//   p = 8:      row y + 1 at x + 1 - the leftover tap of each of the three planes: ONE k-step S whose k-groups are the PLANES
//               zo - 1, zo, zo + 1 of the output plane zo.
// A and B stay input-stationary in z exactly like the kernel above (a fragment of input plane z feeds the three rotating
// accumulator sets of the output planes z + 1, z, z - 1): 2 fragments x 3 planes x 3 products = 18 MFMAs per half row and step;
// S is output-stationary: at step z it completes output plane z - 1 from the images of planes z - 2, z - 1, z, which a RING OF FOUR
// plane images keeps resident (three + the one being staged): 3 MFMAs.  21 MFMAs per half row and step instead of 27, six
// 16-byte fragment reads per 21 MFMAs.  LDS image: piece-major (h rows, then l rows) so that the row pitch is 34 slots = 2 mod 16 -
// the two k-groups of a 16-lane read group that straddle a row (A: (y - 1, x + 1) | (y, x - 1)) land on disjoint banks - and a
// plane pitch of 0 mod 16 slots for the S fragment's lanes.  The z halo has no image: the S weights of the k-group whose plane
// does not exist (plane -1 at z = 1, plane D in the light step) are zeroed for that step.
namespace {
constexpr int P7_ROW = 34 * 16;               // one piece of one image row: slots x = -1 .. 32
constexpr int P7_PIECE = 34 * P7_ROW;         // rows y = -1 .. 32
constexpr int P7_PLANE = 2 * P7_PIECE + 128;  // h image | l image | pad: 2320 slots
constexpr int P7_LDS = 4 * P7_PLANE;          // 148,480 of the 163,840 bytes
static_assert((P7_ROW / 16) % 16 == 2 && (P7_PLANE / 16) % 16 == 0 && P7_LDS <= 160 * 1024, "bank-conflict-free pitches (see above)");
}  // namespace

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void c3d_bwd7_kernel(const C3BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int RW = 8, NU = 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int D = a.D;
    const unsigned plane_v = 32u * 32u;
    const unsigned patch_v = (unsigned)D * plane_v;

    for (int i = tid * 16; i < P7_LDS; i += 256 * 16) *reinterpret_cast<i32x4 *>(lds + i) = i32x4{0, 0, 0, 0};

    // weights: [A0 A1 A2 B0 B1 B2 S] x [h, l] x [64 lanes] x [8] fp16 (c3d_bwd7_pack)
    f16x8 Wa[3][2], Wb[3][2], Ws[2];
    {
        const i32x4 *Wg = reinterpret_cast<const i32x4 *>(a.W);
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                Wa[dz][pc] = __builtin_bit_cast(f16x8, Wg[((0 + dz) * 2 + pc) * 64 + lane]);
                Wb[dz][pc] = __builtin_bit_cast(f16x8, Wg[((3 + dz) * 2 + pc) * 64 + lane]);
            }
        Ws[0] = __builtin_bit_cast(f16x8, Wg[(6 * 2 + 0) * 64 + lane]);
        Ws[1] = __builtin_bit_cast(f16x8, Wg[(6 * 2 + 1) * 64 + lane]);
#if C3_B7PINW
        // 192 accumulators + the 48 registers of A and B fit the accumulation half (240 of 256): started there, they leave the
        // architectural half to the fragments, the staging and the epilogue (the S weights are masked per step: they stay vector)
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                i32x4 x = __builtin_bit_cast(i32x4, Wa[dz][pc]), y = __builtin_bit_cast(i32x4, Wb[dz][pc]);
                asm volatile("" : "+a"(x), "+a"(y));
                Wa[dz][pc] = __builtin_bit_cast(f16x8, x); Wb[dz][pc] = __builtin_bit_cast(f16x8, y);
            }
#endif
    }
    // fragment reads, lane (column j = voxel 16 hx + j, k-group q): byte offsets inside a plane image, + i rows, + hx * 256
    const int offA = wave * RW * P7_ROW + (lq < 3 ? (lj + lq) * 16 : P7_ROW + lj * 16);
    const int offB = (wave * RW + 1) * P7_ROW + (lq < 2 ? (lj + 1 + lq) * 16 : P7_ROW + (lj + lq - 2) * 16);
    const int offS = (wave * RW + 2) * P7_ROW + (lj + 2) * 16;
    const int ringS = lq < 2 ? lq + 2 : 0;       // ring slot of the lane's plane relative to the current one: z - 2, z - 1, z, (z)
    // staging: a unit = two rows of the wave's strip, lane = (row lane >> 5, x = lane & 31); image row = y + 1
    const int st_lane = (wave * RW + 1 + (lane >> 5)) * P7_ROW + ((lane & 31) + 1) * 16;
    const float inv = __builtin_ldexpf(1.f, -(a.e_in + a.e_w));

    const int G = gridDim.x, b0 = blockIdx.x;
    const int np = b0 < a.N ? (a.N - b0 + G - 1) / G : 0;

    f32x4 acc[3][RW][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < RW; ++i) { acc[s][i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[s][i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    auto rsrc_of = [&](const void *base, unsigned long long bytes) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(unsigned)bytes, 0x00020000);
    };
    const unsigned OOB = 0xffffff00u;
    const __amdgpu_buffer_rsrc_t vec_rsrc = rsrc_of(a.vec, (unsigned long long)patch_v * 32u);
    const __amdgpu_buffer_rsrc_t bits_rsrc = rsrc_of(a.bits, (unsigned long long)a.N * patch_v * 2u);
    const __amdgpu_buffer_rsrc_t mask_rsrc = rsrc_of(a.maskA, a.maskA ? (unsigned long long)a.N * patch_v * 2u : 0ull);
    const __amdgpu_buffer_rsrc_t dB_rsrc = rsrc_of(a.dB, (unsigned long long)a.N * patch_v * 32u);
    const __amdgpu_buffer_rsrc_t sA_rsrc = rsrc_of(a.sumA, (unsigned long long)a.N * patch_v * 4u);
    const __amdgpu_buffer_rsrc_t sB_rsrc = rsrc_of(a.sumB, (unsigned long long)a.N * patch_v * 4u);
    auto item_patch = [&](int pi) __attribute__((always_inline)) { return (unsigned)(b0 + pi * G); };

    // ---- staging of a plane: units u = 0 .. 3 (rows 8 w + 2 u, + 1): as in the kernel above ------------------------------------
    i32x4 Vh[2], Vl[2];
    unsigned Sb[2];
    struct Src { unsigned voff, vvox, soff; };
    auto src_of = [&](int pi) __attribute__((always_inline)) {
        Src c;
        const bool ok = pi < np;
        const int y = wave * RW + (lane >> 5);
        c.vvox = (unsigned)(y * 32 + (lane & 31));
        c.voff = ok ? c.vvox * 2u : OOB;
        c.soff = ok ? item_patch(pi) * patch_v * 2u : 0u;
        return c;
    };
    auto mask8 = [&](int b, i32x4 h, i32x4 l, i32x4 *oh, i32x4 *ol) __attribute__((always_inline)) {
        const unsigned m0 = (unsigned)__builtin_amdgcn_sbfe(b, 0, 1), m1 = (unsigned)__builtin_amdgcn_sbfe(b, 1, 1);
        const unsigned m2 = (unsigned)__builtin_amdgcn_sbfe(b, 2, 1), m3 = (unsigned)__builtin_amdgcn_sbfe(b, 3, 1);
        const unsigned m4 = (unsigned)__builtin_amdgcn_sbfe(b, 8, 1), m5 = (unsigned)__builtin_amdgcn_sbfe(b, 9, 1);
        const unsigned m6 = (unsigned)__builtin_amdgcn_sbfe(b, 10, 1), m7 = (unsigned)__builtin_amdgcn_sbfe(b, 11, 1);
        const unsigned k0 = (m0 & 0xffffu) | (m1 & 0xffff0000u), k1 = (m2 & 0xffffu) | (m3 & 0xffff0000u);
        const unsigned k2 = (m4 & 0xffffu) | (m5 & 0xffff0000u), k3 = (m6 & 0xffffu) | (m7 & 0xffff0000u);
        *oh = i32x4{(int)((unsigned)h.x & k0), (int)((unsigned)h.y & k1), (int)((unsigned)h.z & k2), (int)((unsigned)h.w & k3)};
        *ol = i32x4{(int)((unsigned)l.x & k0), (int)((unsigned)l.y & k1), (int)((unsigned)l.z & k2), (int)((unsigned)l.w & k3)};
    };
    auto load_unit = [&](const Src &c, unsigned z, auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        const unsigned so = (z * plane_v + (unsigned)u * 64u);
        Vh[u & 1] = __builtin_amdgcn_raw_buffer_load_b128(vec_rsrc, (int)(c.vvox * 32u), (int)(so * 32u), 0);
        Vl[u & 1] = __builtin_amdgcn_raw_buffer_load_b128(vec_rsrc, (int)(c.vvox * 32u + 16u), (int)(so * 32u), 0);
        Sb[u & 1] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(bits_rsrc, (int)c.voff, (int)(c.soff + so * 2u), 0);
    };
    auto stage_unit = [&](int wbase, auto U) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        i32x4 oh, ol;
        mask8((int)Sb[u & 1], Vh[u & 1], Vl[u & 1], &oh, &ol);
        char *dst = lds + wbase + (2 * u) * P7_ROW;
        *reinterpret_cast<i32x4 *>(dst) = oh;
        *reinterpret_cast<i32x4 *>(dst + P7_PIECE) = ol;
    };

    int c1_pi = 0, c1_z = 0, c2_pi = 0, c2_z = 0;
    auto advance = [&](int &pi, int &z) __attribute__((always_inline)) { if (++z == D) { z = 0; ++pi; } };

    // ---- epilogue of one finished x row: as in the kernel above ---------------------------------------------------------------
    unsigned mkq[2][2] = {{0u, 0u}, {0u, 0u}};
    struct Epi { unsigned pv; unsigned row_v; float inv; unsigned off_mask, off_dB, off_sum; };
    auto epi_setup = [&](bool valid, int pe, int zo) __attribute__((always_inline)) {
        Epi e;
        e.pv = valid ? item_patch(pe) * patch_v : 0u;
        e.row_v = (unsigned)(zo * 32 + wave * RW) * 32u;
        e.inv = valid ? inv : 0.f;
        e.off_mask = (valid && lq < 2) ? (unsigned)(lj * 2 + lq) : OOB;
        e.off_dB = (valid && lq >= 2) ? (unsigned)lj * 32u + (unsigned)(lq - 2) * 16u : OOB;
        e.off_sum = valid ? (unsigned)((lq & 1) * 16 + lj) * 4u : OOB;
        return e;
    };
    auto mask_load = [&](const Epi &E, auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
#pragma unroll
        for (int hx = 0; hx < 2; ++hx)
            mkq[i & 1][hx] = (unsigned char)__builtin_amdgcn_raw_buffer_load_b8(mask_rsrc, (int)E.off_mask,
                                                                               (int)((E.pv + E.row_v + (unsigned)i * 32u + (unsigned)hx * 16u) * 2u), 0);
    };
    const bool has_mask = a.maskA != nullptr;
    float t_keep = 0.f;
    auto epi_half = [&](auto S, auto I, auto HX, const Epi &E) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value, i = decltype(I)::value, hx = decltype(HX)::value;
        const unsigned mk = mkq[i & 1][hx];
        if constexpr (hx == 1 && i + 2 < RW) mask_load(E, IC<i + 2>{});
        const f32x4 c = acc[s][i][hx];
        const int nib = (lq < 2 && has_mask) ? (int)mk : 15;
        const float v0 = c.x * E.inv, v1 = c.y * E.inv, v2 = c.z * E.inv, v3 = c.w * E.inv;
        f32x4 val;
        unsigned k0 = (unsigned)__builtin_amdgcn_sbfe(nib, 0, 1), k1 = (unsigned)__builtin_amdgcn_sbfe(nib, 1, 1);
        unsigned k2 = (unsigned)__builtin_amdgcn_sbfe(nib, 2, 1), k3 = (unsigned)__builtin_amdgcn_sbfe(nib, 3, 1);
        asm("" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3));
        val.x = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v0) & k0);
        val.y = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v1) & k1);
        val.z = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v2) & k2);
        val.w = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v3) & k3);
        const unsigned vox = E.pv + E.row_v + (unsigned)i * 32u + (unsigned)hx * 16u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, val), dB_rsrc, (int)E.off_dB, (int)(vox * 32u), 2 /* nt */);
        if constexpr (!(C3_B7ABL & 8)) ALQ_STORE_HOLD("v"(val));      // (16-byte store with a register soffset: alq_internal.h)
        const float t = (val.x + val.y) + (val.z + val.w);
        const unsigned tu = __builtin_bit_cast(unsigned, t);
        const auto sw = hx == 0 ? __builtin_amdgcn_permlane16_swap(tu, 0u, false, false) : __builtin_amdgcn_permlane16_swap(0u, tu, false, false);
        const unsigned s0 = sw[0], s1 = sw[1];
        const float t2 = __builtin_bit_cast(float, s0) + __builtin_bit_cast(float, s1);
        if constexpr (hx == 0) {
            t_keep = t2;
        } else {
            const float ts = t_keep + t2;
            const unsigned rowv = E.pv + E.row_v + (unsigned)i * 32u;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ts), sA_rsrc, (int)(lq < 2 ? E.off_sum : OOB), (int)(rowv * 4u), 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ts), sB_rsrc, (int)(lq >= 2 ? E.off_sum : OOB), (int)(rowv * 4u), 0);
        }
    };
    auto epi_row = [&](auto S, auto I, const Epi &E) __attribute__((always_inline)) {
        epi_half(S, I, IC<0>{}, E);
        epi_half(S, I, IC<1>{}, E);
    };
    // the S weights with the k-group of a plane that does not exist zeroed (kill: that k-group, or -1)
    auto s_weights = [&](int kill, f16x8 &sh, f16x8 &sl) __attribute__((always_inline)) {
        const int km = lq == kill ? 0 : -1;
        i32x4 h = __builtin_bit_cast(i32x4, Ws[0]), l = __builtin_bit_cast(i32x4, Ws[1]);
        h.x &= km; h.y &= km; h.z &= km; h.w &= km;
        l.x &= km; l.y &= km; l.z &= km; l.w &= km;
        sh = __builtin_bit_cast(f16x8, h); sl = __builtin_bit_cast(f16x8, l);
    };
    // three products of one k-step into one accumulator, the small ones first
    auto mac3 = [&](f32x4 c, const f16x8 &wh, const f16x8 &wl, const f16x8 &bh, const f16x8 &bl) __attribute__((always_inline)) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh, c, 0, 0, 0);
        return c;
    };

    auto step = [&](auto RR, int pi, int z, long long n) __attribute__((always_inline)) {
        constexpr int R = decltype(RR)::value;
        constexpr int S_lo = (R + 2) % 3, S_mid = R, S_hi = (R + 1) % 3;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int cur = (int)(n & 3) * P7_PLANE;
        const int baseA = offA + cur, baseB = offB + cur;
        const int baseS = offS + (int)((unsigned)((int)(n & 3) + ringS) & 3u) * P7_PLANE;
        const int wbase = st_lane + (int)((n + 1) & 3) * P7_PLANE;
        const Src C1 = src_of(c1_pi), C2 = src_of(c2_pi);
        const unsigned z1 = (unsigned)c1_z, z2 = (unsigned)c2_z;
        const bool ev = z >= 2 || (z == 0 && pi > 0);
        const Epi E = epi_setup(ev, z >= 2 ? pi : pi - 1, z >= 2 ? z - 2 : D - 1);
        f16x8 sh, sl;
        s_weights(z == 1 ? 0 : -1, sh, sl);       // output plane 0 (completed at z = 1): its plane -1 is the z halo

        struct Frag { f16x8 ah, al, bh, bl, sh, sl; };
        auto frag = [&](int i, int hx, Frag &f) __attribute__((always_inline)) {
            const int o = i * P7_ROW + hx * 256;
            f.ah = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(lds + baseA + o));
            f.al = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(lds + baseA + o + P7_PIECE));
            f.bh = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(lds + baseB + o));
            f.bl = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(lds + baseB + o + P7_PIECE));
            f.sh = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(lds + baseS + o));
            f.sl = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(lds + baseS + o + P7_PIECE));
        };
        Frag fc, fn;
        frag(0, 0, fc);
        auto row = [&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(C3_B7ABL & 1)) epi_row(IC<S_hi>{}, I, E);          // row i of the plane finished two steps ago, then its accumulators restart below
            else { const f32x4 k0 = acc[S_hi][i][0], k1 = acc[S_hi][i][1]; asm volatile("" :: "v"(k0), "v"(k1)); }      // (timing build: the MFMAs stay alive)
            if constexpr ((i & 1) == 0 && !(C3_B7ABL & 2)) {
                stage_unit(wbase, IC<i / 2>{});
                if constexpr (i / 2 + 2 < NU) load_unit(C1, z1, IC<i / 2 + 2>{}); else load_unit(C2, z2, IC<i / 2 + 2 - NU>{});
            }
#pragma unroll
            for (int hx = 0; hx < 2; ++hx) {
                if (hx == 0) frag(i, 1, fn);
                else if (i < RW - 1) frag(i + 1, 0, fn);
                // output planes z + 1 (first contribution: starts at zero), z, z - 1
                f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = acc[S_mid][i][hx], c2 = acc[S_lo][i][hx];
                c0 = mac3(c0, Wa[0][0], Wa[0][1], fc.ah, fc.al);
                c1 = mac3(c1, Wa[1][0], Wa[1][1], fc.ah, fc.al);
                c2 = mac3(c2, Wa[2][0], Wa[2][1], fc.ah, fc.al);
                c0 = mac3(c0, Wb[0][0], Wb[0][1], fc.bh, fc.bl);
                c1 = mac3(c1, Wb[1][0], Wb[1][1], fc.bh, fc.bl);
                c2 = mac3(c2, Wb[2][0], Wb[2][1], fc.bh, fc.bl);
                if constexpr (!(C3_B7ABL & 4)) c2 = mac3(c2, sh, sl, fc.sh, fc.sl);
                acc[S_hi][i][hx] = c0; acc[S_mid][i][hx] = c1; acc[S_lo][i][hx] = c2;
                fc = fn;
            }
#if C3_B7PIPE
#pragma unroll
            for (int m = 0; m < 42; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(C3_FILL_MASK, C3_B7PIPE, 0);
            }
#endif
        };
        row(IC<0>{}); row(IC<1>{}); row(IC<2>{}); row(IC<3>{}); row(IC<4>{}); row(IC<5>{}); row(IC<6>{}); row(IC<7>{});
        {
            const Epi En = epi_setup(z + 1 < D ? z >= 1 : true, pi, z + 1 < D ? z - 1 : D - 2);
            mask_load(En, IC<0>{});
            mask_load(En, IC<1>{});
        }
        c1_pi = c2_pi; c1_z = c2_z;
        advance(c2_pi, c2_z);
    };
    // a step without an input plane: the epilogue of the finished plane held by set S, which is then cleared.  LAST = the light step
    // behind plane D - 1 of a patch's sweep (finished plane D - 2): plane D - 1, held by set (S + 1) % 3, still lacks its S k-step
    // (planes D - 2, D - 1 and the z halo), which n - the count of planes staged so far - locates in the ring
    auto light = [&](auto S, auto LAST, int pe, int zo, long long n) __attribute__((always_inline)) {
        constexpr int s = decltype(S)::value;
        constexpr bool last = decltype(LAST)::value != 0;
        const Epi E = epi_setup(true, pe, zo);
        auto rows = [&](auto I) __attribute__((always_inline)) { epi_row(S, I, E); };
        rows(IC<0>{}); rows(IC<1>{}); rows(IC<2>{}); rows(IC<3>{}); rows(IC<4>{}); rows(IC<5>{}); rows(IC<6>{}); rows(IC<7>{});
#pragma unroll
        for (int i = 0; i < RW; ++i) { acc[s][i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[s][i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if constexpr (last) {
            constexpr int sp = (s + 1) % 3;
            f16x8 sh, sl;
            s_weights(2, sh, sl);
            const int baseS = offS + (int)((unsigned)((int)(n & 3) + ringS) & 3u) * P7_PLANE;
#pragma unroll
            for (int i = 0; i < RW; ++i)
#pragma unroll
                for (int hx = 0; hx < 2; ++hx) {
                    const int o = i * P7_ROW + hx * 256;
                    const f16x8 bh = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(lds + baseS + o));
                    const f16x8 bl = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(lds + baseS + o + P7_PIECE));
                    acc[sp][i][hx] = mac3(acc[sp][i][hx], sh, sl, bh, bl);
                }
        }
        {
            const Epi En = epi_setup(zo == D - 2, pe, D - 1);
            mask_load(En, IC<0>{});
            mask_load(En, IC<1>{});
        }
    };

    __syncthreads();
    if (np > 0) {
        const Src C0 = src_of(0);
        const int wb = st_lane;       // ring slot 0
        load_unit(C0, 0u, IC<0>{}); load_unit(C0, 0u, IC<1>{});
        stage_unit(wb, IC<0>{}); stage_unit(wb, IC<1>{});
        load_unit(C0, 0u, IC<2>{}); load_unit(C0, 0u, IC<3>{});
        stage_unit(wb, IC<2>{}); stage_unit(wb, IC<3>{});
        c1_pi = 0; c1_z = 0;
        advance(c1_pi, c1_z);
        c2_pi = c1_pi; c2_z = c1_z;
        advance(c2_pi, c2_z);
        const Src C1 = src_of(c1_pi);
        load_unit(C1, (unsigned)c1_z, IC<0>{}); load_unit(C1, (unsigned)c1_z, IC<1>{});
    }
    long long n = 0;
    for (int pi = 0; pi < np; ++pi) {
        for (int zz = 0; zz < 10; ++zz) {
            step(IC<0>{}, pi, 3 * zz, n); ++n;
            step(IC<1>{}, pi, 3 * zz + 1, n); ++n;
            step(IC<2>{}, pi, 3 * zz + 2, n); ++n;
        }
        step(IC<0>{}, pi, 30, n); ++n;
        step(IC<1>{}, pi, 31, n); ++n;
        light(IC<0>{}, IC<1>{}, pi, D - 2, n);
    }
    if (np > 0) light(IC<1>{}, IC<0>{}, np - 1, D - 1, n);
}

// One MFMA on fp16 SUBNORMAL operands: 32 products 2^-20 * 2^10 per element -> 2^-5 when the matrix core keeps subnormal
// inputs (gfx950 does), 0 when it flushes them.  The one-accumulator form relies on it for the low pieces of small values.
__global__ void c3d_subnormal_probe(float *out) {
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0x1p-20f; b[i] = (_Float16)0x1p10f; }
    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = d.x;
}

// 1: fp16 subnormal MFMA operands are honoured on this context's device, 0: flushed.  Probed once per context (the answer is a
// property of the device the context is bound to); a probe that could not run (launch or copy error) is reported as 0 for
// this call and NOT cached, so a transient failure does not switch the one-accumulator kernels off for good.
int c3d_subnormals_ok(alq_ctx *ctx) {
    if (ctx->f16_subnormal_mfma >= 0) return ctx->f16_subnormal_mfma;
    float *d = nullptr, h = -1.f;
    if (hipMalloc(&d, sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return 0; }
    hipLaunchKernelGGL(c3d_subnormal_probe, dim3(1), dim3(64), 0, ctx->stream, d);
    const bool ok = hipGetLastError() == hipSuccess &&
                    hipMemcpyAsync(&h, d, sizeof(float), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                    hipStreamSynchronize(ctx->stream) == hipSuccess;
    (void)hipFree(d);
    if (!ok) { (void)hipGetLastError(); return 0; }
    ctx->f16_subnormal_mfma = (h == 0x1p-5f) ? 1 : 0;
    return ctx->f16_subnormal_mfma;
}

// ------------------------------------------------------------------------------------------------------ host
int c3d_fwd_build(const View &in, const View &out, const int k[3], const int lo[3], const int s[3], C3dPlan *plan) {
    plan->ok = false;
    if (getenv("ALQ_NO_C3D")) return ALQ_OK;
    if (!(k[0] == 3 && k[1] == 3 && k[2] == 3 && lo[0] == 1 && lo[1] == 1 && lo[2] == 1 && s[0] == 1 && s[1] == 1 && s[2] == 1)) return ALQ_OK;
    if (!(in.D == 32 && in.H == 32 && in.W == 32 && out.D == 32 && out.H == 32 && out.W == 32)) return ALQ_OK;
    if (!(in.C == 16 && in.split == 8 && in.cs == 8 && in.c0 == 0 && out.C == 8)) return ALQ_OK;
    plan->D = in.D;
    plan->oneacc = getenv("ALQ_C3D_TWOACC") ? 0 : 1;
    plan->flops_per_patch = 2.0 * 27 * 16 * 8 * (double)in.vox();
    plan->ok = true;
    return ALQ_OK;
}

// Bmat [(tap, ci)][co] (TF conv layout), tap = (dz * 3 + dy) * 3 + dx.  k-step (dz, dy, s): lane (i = lane & 15 -> output
// voxel v = i >> 3 of the x pair, channel co = i & 7; k-group kg = lane >> 4 -> tensor t = kg >> 1, x parity p = kg & 1):
// window position q = 2 s + p holds input x = 2 r - 1 + q, i.e. x tap index q - v of output voxel 2 r + v.
void c3d_fwd_pack(C3dPlan *plan, const std::vector<float> &Bmat) {
    float amax = 0.f;
    for (float w : Bmat) amax = std::max(amax, std::fabs(w));
    int ex = 0;
    if (amax > 0.f) (void)std::frexp(amax, &ex);
    plan->w_exp = 14 - ex;
    plan->h_W.assign((size_t)18 * 2 * 64 * 8, 0);
    for (int dzi = 0; dzi < 3; ++dzi)
        for (int dyi = 0; dyi < 3; ++dyi)
            for (int s = 0; s < 2; ++s)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 15, kg = lane >> 4, v = i >> 3, co = i & 7, t = kg >> 1, p = kg & 1;
                    const int dxi = 2 * s + p - v;
                    const int ks = (dzi * 3 + dyi) * 2 + s;
                    for (int c = 0; c < 8; ++c) {
                        float w = 0.f;
                        if (dxi >= 0 && dxi <= 2) w = Bmat[((size_t)((dzi * 3 + dyi) * 3 + dxi) * 16 + 8 * t + c) * 8 + co];
                        const float ws = std::ldexp(w, plan->w_exp);
                        const _Float16 h = (_Float16)ws;
                        const _Float16 l = (_Float16)std::ldexp(ws - (float)h, plan->oneacc ? 0 : 11);
                        unsigned short hb, lb;
                        std::memcpy(&hb, &h, 2);
                        std::memcpy(&lb, &l, 2);
                        plan->h_W[((size_t)(ks * 2 + 0) * 64 + lane) * 8 + c] = hb;
                        plan->h_W[((size_t)(ks * 2 + 1) * 64 + lane) * 8 + c] = lb;
                    }
                }
}

int c3d_fwd_launch(alq_ctx *ctx, const C3dPlan &plan, const View &in, const float *bias, int N, const unsigned *amaxA, const unsigned *amaxB,
                   const float *fc_W, float *fc_part, float *asum_part, unsigned char *fc_bits, float flip_tau) {
    ALQ_REQUIRE(plan.ok && plan.d_W, ALQ_EINVAL, "c3d: weights not set");
    ALQ_REQUIRE(in.split == 8 && in.cs == 8 && in.C == 16 && in.D == plan.D && in.H == 32 && in.W == 32 && plan.D == 32, ALQ_EINVAL, "c3d: input view mismatch");
    ALQ_REQUIRE(amaxA && amaxB && fc_W && fc_part && bias, ALQ_EINVAL, "c3d: missing argument");
    ALQ_REQUIRE(N < 4096, ALQ_EUNSUPPORTED, "c3d: 32-bit byte offsets hold fewer than 4096 patches per pass");
    if (N <= 0) return ALQ_OK;
    C3FwdArgs a;
    a.inA = in.p; a.inB = in.p + in.delta; a.W = plan.d_W; a.bias = bias; a.amaxA = amaxA; a.amaxB = amaxB;
    a.fc_W = fc_W; a.fc_part = fc_part; a.asum_part = asum_part; a.fc_bits = fc_bits; a.N = N; a.D = plan.D; a.e_w = plan.w_exp;
    a.flip_tau = flip_tau; a.clk = nullptr;
    const unsigned grid = (unsigned)std::min(N, 256);
    ProfScope ps(ctx, PROF_IGEMM_F16, plan.flops_per_patch * N);
    auto go = [&](auto kfn) -> int {
        ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, C3_LDS));
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), C3_LDS, ctx->stream, a);
        return ALQ_OK;
    };
    if (fc_bits) ALQ_TRY(plan.oneacc ? go(c3d_fwd_kernel<true, true>) : go(c3d_fwd_kernel<true, false>));
    else ALQ_TRY(plan.oneacc ? go(c3d_fwd_kernel<false, true>) : go(c3d_fwd_kernel<false, false>));
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

// ---- backward ------------------------------------------------------------------------------------------------------------
int c3d_bwd_build(const View &fwd_in, const View &fwd_out, const int k[3], const int lo[3], const int s[3], C3dPlan *plan) {
    C3dPlan f;
    ALQ_TRY(c3d_fwd_build(fwd_in, fwd_out, k, lo, s, &f));      // same geometry rules
    plan->ok = f.ok;
    plan->D = f.D;
    plan->oneacc = 1;
    plan->flops_per_patch = f.flops_per_patch;
    return ALQ_OK;
}

// Bmat = the FORWARD conv's [(tap, ci)][co].  k-step (dzi, di) = input plane / row offsets (dzi - 1, di - 1) relative to the
// output voxel, k-group kg = x offset kg - 1 (kg = 3: zero): the forward tap that links them is (2 - dzi, 2 - di, 2 - kg).
// A-operand row i = output channel ci (0..15), elements c = the 8 channels co of the cotangent.
void c3d_bwd_pack(C3dPlan *plan, const std::vector<float> &Bmat) {
    float amax = 0.f;
    for (float w : Bmat) amax = std::max(amax, std::fabs(w));
    int ex = 0;
    if (amax > 0.f) (void)std::frexp(amax, &ex);
    plan->w_exp = 14 - ex;
    plan->h_W.assign((size_t)9 * 2 * 64 * 8, 0);
    for (int dzi = 0; dzi < 3; ++dzi)
        for (int di = 0; di < 3; ++di)
            for (int lane = 0; lane < 64; ++lane) {
                const int ci = lane & 15, kg = lane >> 4;
                const int ks = dzi * 3 + di;
                for (int c = 0; c < 8; ++c) {
                    float w = 0.f;
                    if (kg < 3) w = Bmat[((size_t)(((2 - dzi) * 3 + (2 - di)) * 3 + (2 - kg)) * 16 + ci) * 8 + c];
                    const float ws = std::ldexp(w, plan->w_exp);
                    const _Float16 h = (_Float16)ws;
                    const _Float16 l = (_Float16)(ws - (float)h);
                    unsigned short hb, lb;
                    std::memcpy(&hb, &h, 2);
                    std::memcpy(&lb, &l, 2);
                    plan->h_W[((size_t)(ks * 2 + 0) * 64 + lane) * 8 + c] = hb;
                    plan->h_W[((size_t)(ks * 2 + 1) * 64 + lane) * 8 + c] = lb;
                }
            }
}

// The 7-k-step form (c3d_bwd7_kernel): fragments [A0 A1 A2 B0 B1 B2 S] x [h, l]; lane (ci = lane & 15, k-group q = lane >> 4) of
// fragment f holds the weights of ONE tap (dz, dy, dx) - offsets (dz - 1, dy - 1, dx - 1) of the cotangent voxel relative to the
// output voxel; the forward tap that links them is (2 - dz, 2 - dy, 2 - dx) - or zeros:
//   A_dz: q < 3 -> (dz, 0, q), q = 3 -> (dz, 1, 0);   B_dz: (dz, 1, 1), (dz, 1, 2), (dz, 2, 0), (dz, 2, 1);   S: q < 3 -> (q, 2, 2).
void c3d_bwd7_pack(C3dPlan *plan, const std::vector<float> &Bmat) {
    plan->h_W7.assign((size_t)7 * 2 * 64 * 8, 0);
    for (int f = 0; f < 7; ++f)
        for (int lane = 0; lane < 64; ++lane) {
            const int ci = lane & 15, q = lane >> 4;
            int dz = -1, dy = 0, dx = 0;
            if (f < 3) { dz = f; if (q < 3) { dy = 0; dx = q; } else { dy = 1; dx = 0; } }
            else if (f < 6) { dz = f - 3; const int p = 4 + q; dy = p / 3; dx = p % 3; }
            else if (q < 3) { dz = q; dy = 2; dx = 2; }
            for (int c = 0; c < 8; ++c) {
                float w = 0.f;
                if (dz >= 0) w = Bmat[((size_t)(((2 - dz) * 3 + (2 - dy)) * 3 + (2 - dx)) * 16 + ci) * 8 + c];
                const float ws = std::ldexp(w, plan->w_exp);
                const _Float16 h = (_Float16)ws;
                const _Float16 l = (_Float16)(ws - (float)h);
                unsigned short hb, lb;
                std::memcpy(&hb, &h, 2);
                std::memcpy(&lb, &l, 2);
                plan->h_W7[((size_t)(f * 2 + 0) * 64 + lane) * 8 + c] = hb;
                plan->h_W7[((size_t)(f * 2 + 1) * 64 + lane) * 8 + c] = lb;
            }
        }
}

// the head's weight difference as fp16 pairs at their true scale: per voxel (8 channels) [h8 | l8], x 2^e = h + l
void c3d_presplit_vec(const float *v, long long F, int e, std::vector<unsigned short> *out) {
    out->assign((size_t)F * 2, 0);
    for (long long vox = 0; vox < F / 8; ++vox)
        for (int c = 0; c < 8; ++c) {
            const float xs = std::ldexp(v[vox * 8 + c], e);
            const _Float16 h = (_Float16)xs;
            const _Float16 l = (_Float16)(xs - (float)h);
            unsigned short hb, lb;
            std::memcpy(&hb, &h, 2);
            std::memcpy(&lb, &l, 2);
            (*out)[(size_t)vox * 16 + c] = hb;
            (*out)[(size_t)vox * 16 + 8 + c] = lb;
        }
}

int c3d_bwd_launch(alq_ctx *ctx, const C3dPlan &plan, int N, const unsigned char *bits, const void *vec16, int e_in, const unsigned char *maskA,
                   float *dB, float *sumA, float *sumB, int rows_per_wave) {
    ALQ_REQUIRE(plan.ok && plan.d_W && plan.D == 32, ALQ_EINVAL, "c3d: backward weights not set");
    ALQ_REQUIRE(bits && vec16 && dB && sumA && sumB, ALQ_EINVAL, "c3d: missing argument");
    ALQ_REQUIRE(N < 4096, ALQ_EUNSUPPORTED, "c3d: 32-bit byte offsets hold fewer than 4096 patches per pass");
    if (N <= 0) return ALQ_OK;
    C3BwdArgs a;
    a.bits = bits; a.vec = vec16; a.W = plan.d_W; a.maskA = maskA; a.dB = dB; a.sumA = sumA; a.sumB = sumB;
    a.N = N; a.D = plan.D; a.e_in = e_in; a.e_w = plan.w_exp;
    ProfScope ps(ctx, PROF_IGEMM_F16, plan.flops_per_patch * N);
    // a workgroup per patch (8 rows per wave; with the peeled sweep loop nothing spills: 1.45 ms per 2000 patches against 1.63 ms)
    // unless the model was created under ALQ_C3D_BWD_ROWS=4: the half-patch form (4 rows per wave, 128 registers per lane left to
    // co-resident kernels)
    const int rows8 = rows_per_wave == 4 ? 0 : 1;
    if (rows_per_wave == 7 && plan.d_W7) {      // the 27 taps in 7 k-steps (default since round 6; ALQ_C3D_BWD_ROWS=8: the 9-k-step kernel)
        a.W = plan.d_W7;
        auto kfn = c3d_bwd7_kernel;
        ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, P7_LDS));
        hipLaunchKernelGGL(kfn, dim3((unsigned)std::min(N, 256)), dim3(256), P7_LDS, ctx->stream, a);
    } else if (rows8) {
        auto kfn = c3d_bwd_kernel<true, 8>;
        const int ldsb = 2 * (32 + 2) * B3_ROW + 2 * B3_ROW;
        ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
        hipLaunchKernelGGL(kfn, dim3((unsigned)std::min(N, 256)), dim3(256), ldsb, ctx->stream, a);
    } else {
        auto kfn = c3d_bwd_kernel<true, 4>;
        const int ldsb = 2 * (16 + 2) * B3_ROW + 2 * B3_ROW;
        ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
        hipLaunchKernelGGL(kfn, dim3((unsigned)std::min(2 * N, 256)), dim3(256), ldsb, ctx->stream, a);
    }
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

}  // namespace alq
