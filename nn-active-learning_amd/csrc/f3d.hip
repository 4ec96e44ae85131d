// Forward of NET-C's `enc2` (3x3x3 conv 8 -> 16 channels at 16^3 + bias + ReLU; NN_extended.py:416-426) in a Fisher pass, with the
// 2x2x2 max-pool behind it (`pool2`, NN_extended.py:428-441) in the same launch.  On the two-slot engine the conv took 0.32 ms per 2047
// patches whether it multiplied bf16 triples or fp16 pairs (its tiles at 16^3 are prologue-bound), the pool another 0.17 ms
// to read the tensor back.  The plane sweep of d3d.hip at this layer's shape:
//   * fp16 pairs at their true scale in ONE accumulator (the input bound is the MEASURED per-patch maximum of the first layer - the pool in
//     between cannot raise it - so the lo pieces of typical values are normal fp16 numbers; the matrix cores keep the subnormal rest): the round-5 accuracy
//     study (tools/gpu_accuracy_stats.py, ACC_EXTRA_MASK=68) counts 120 / 113 / 56 patches beyond 2e-6 / 1e-5 / 1e-4 of the exact-fp32
//     engine with this launch on the pair split against 122 / 116 / 58 without and 118 / 112 / 56 for bf16 triples everywhere;
//   * a workgroup sweeps the 16 planes of a patch, wave w owns output rows 4 w .. 4 w + 3; K = 32 = three x offsets x 8 input
//     channels (+ one k-group of zeros): one K step per (dz, dy), 9 x 3 products per output row; an activation fragment is read once
//     per input row and feeds up to 3 (dy) x 3 (dz) x 3 products = 27 MFMAs; the 18 weight fragments stay in registers;
//   * epilogue: bias, ReLU, 16-byte stores, sign bytes, channel sums - and the pool: the ReLU'd rows of an even plane wait in
//     registers for the odd plane above them; window maxima over dz and dy in the lane, over dx with the neighbouring lane; the first
//     maximum in window order wins (pool_fwd_vox_kernel's rule); pooled tensor, arg-max bytes and pooled channel sums are written by
//     the even lanes.  Plane parity and accumulator-set rotation are compile-time: the sweep loop is unrolled by six.
#include "alq_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace alq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct F3Args {
    const float *in;              // [N][16^3][8] (dense)
    const unsigned *amax;         // per patch: bit pattern of a bound on max |x|
    const unsigned short *Whi;    // [9 (dz, dy)][64 lanes][8] fp16 bits: hi pieces (f3d_pack)
    const unsigned short *Wlo;    // the lo pieces (x 2^11), same layout
    const float *bias;            // [16]
    float *out;                   // [N][16^3][16]
    unsigned char *sg;            // sign field of out: byte (voxel * 16 + c) / 4, bit c & 3  (or nullptr)
    float *osum;                  // [N][16^3] channel sums of out (or nullptr)
    float *pout;                  // [N][8^3][16] the 2x2x2 max-pool of out
    unsigned char *parg;          // [N][8^3][16] window index (dz * 2 + dy) * 2 + dx of the maximum
    float *posum;                 // [N][8^3] channel sums of pout (or nullptr)
    int e_w;                      // weights were scaled by 2^e_w
    int N;
};

constexpr unsigned F3_OOB = 0xffffff00u;
constexpr int F3_ROWB = 18 * 16;              // one piece of a row: 18 voxel slots (x = -1 .. 16) x 8 channels x 2 B
constexpr int F3_SLOT = 2 * F3_ROWB;          // a row: pieces h, l
constexpr int F3_PLANE = 18 * F3_SLOT;        // image of a plane: rows y = -1 .. 16: 10,368 B
constexpr int F3_STRIP = 2 * F3_PLANE;

__device__ inline __amdgpu_buffer_rsrc_t f3_rsrc(const void *base, unsigned long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ inline int f3_s(unsigned v) { return __builtin_amdgcn_readfirstlane((int)v); }
template <int V> struct F3IC { static constexpr int value = V; };
#ifndef F3_PIPE
#define F3_PIPE 2
#endif
#ifndef F3_ONEACC
#define F3_ONEACC 1      // 1: lo pieces at their true scale, all three products in one accumulator (needs fp16 subnormals in the matrix cores: c3d_subnormals_ok)
#endif
constexpr int F3_FILL_MASK = 0x096;       // what may fill the gap behind an MFMA: VALU, SALU, VMEM, DS

#ifndef F3_WGS
#define F3_WGS 2      // workgroups per CU: with one accumulator per output the kernel fits 224 registers, and a second workgroup issues MFMAs while the first one's
                      // wave is in its epilogue / pool arithmetic: 292 -> 222 us per 2047 patches (two accumulators: 256 registers + spills, 378 us)
#endif
#if F3_WGS == 1
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void f3d_fwd_kernel(const F3Args a) {
#else
__global__ __launch_bounds__(256, 2) void f3d_fwd_kernel(const F3Args a) {
#endif
    extern __shared__ __attribute__((aligned(16))) char f3lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char *strip = f3lds;
    for (int i = threadIdx.x; i < F3_STRIP / 16; i += 256) reinterpret_cast<i32x4 *>(strip)[i] = i32x4{0, 0, 0, 0};
    __syncthreads();
    const int n = lane & 15, kg = lane >> 4;
    f16x8 wh[9], wl[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        wh[c] = *reinterpret_cast<const f16x8 *>(a.Whi + ((size_t)c * 64 + lane) * 8);
        wl[c] = *reinterpret_cast<const f16x8 *>(a.Wlo + ((size_t)c * 64 + lane) * 8);
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) asm volatile("" : "+v"(wh[c]), "+v"(wl[c]));      // arrived before the loop
    f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + 4 * kg);
    asm volatile("" : "+v"(bias4));

    typedef const unsigned __attribute__((address_space(4))) *cu32p;
    const cu32p amax_c = (cu32p)(unsigned long long)a.amax;
    auto patch_exp = [&](int p) __attribute__((always_inline)) {      // max |x| < 2^ex -> scale 2^(14 - ex); all-zero patch: 0  (c3d_fwd_kernel)
        const unsigned fm = amax_c[p];
        const int ex = (int)((fm >> 23) & 255u) - 126;
        const int ce = 14 - ex;
        return fm ? (ce < 96 ? ce : 96) : 0;
    };

    const __amdgpu_buffer_rsrc_t i_rsrc = f3_rsrc(a.in, (unsigned long long)a.N * 4096 * 32);
    const __amdgpu_buffer_rsrc_t o_rsrc = f3_rsrc(a.out, (unsigned long long)a.N * 4096 * 64);
    const __amdgpu_buffer_rsrc_t s_rsrc = f3_rsrc(a.sg, a.sg ? (unsigned long long)a.N * 4096 * 4 : 0ull);
    const __amdgpu_buffer_rsrc_t u_rsrc = f3_rsrc(a.osum, a.osum ? (unsigned long long)a.N * 4096 * 4 : 0ull);
    const __amdgpu_buffer_rsrc_t po_rsrc = f3_rsrc(a.pout, (unsigned long long)a.N * 512 * 64);
    const __amdgpu_buffer_rsrc_t pa_rsrc = f3_rsrc(a.parg, (unsigned long long)a.N * 512 * 16);
    const __amdgpu_buffer_rsrc_t pu_rsrc = f3_rsrc(a.posum, a.posum ? (unsigned long long)a.N * 512 * 4 : 0ull);

    // staging lane roles: a 16-byte load covers 4 channels of one voxel; two rows (512 B each) per load: row lane >> 5, voxel (lane & 31) >> 1, channels 4 (lane & 1) ..
    const int srow = lane >> 5, sx = (lane & 31) >> 1, cq = lane & 1;
    char *const w_base = strip + (4 * wave + 1 + srow) * F3_SLOT + (sx + 1) * 16 + cq * 8;      // unit u: + 2 u rows
    const unsigned ldA = (unsigned)lane * 16u;
    // fragment lane roles: column n = voxel x, k-group kg = x offset kg - 1 (slot n + kg; group 3 has zero weights and re-reads group 2's slot); input row y = 4 w - 1 + r = image row 4 w + r
    const char *const f_base = strip + 4 * wave * F3_SLOT + (n + (kg < 2 ? kg : 2)) * 16;
    // epilogue lane roles: voxel x = n of the output row, channels 4 kg .. + 3
    const unsigned e_out = (unsigned)n * 64u + (unsigned)kg * 16u, e_sg = (unsigned)n * 4u + (unsigned)kg;
    const unsigned e_sum = kg == 0 ? (unsigned)n * 4u : F3_OOB;
    // pool: the even lanes (x even) write pooled voxel x / 2
    const bool pl_w = (n & 1) == 0;
    const unsigned p_out = pl_w ? (unsigned)(n >> 1) * 64u + (unsigned)kg * 16u : F3_OOB;
    const unsigned p_arg = pl_w ? (unsigned)(n >> 1) * 16u + (unsigned)kg * 4u : F3_OOB;
    const unsigned p_sum = (pl_w && kg == 0) ? (unsigned)(n >> 1) * 4u : F3_OOB;

    const int G8 = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, jb = (int)blockIdx.x >> 3;
    const int npx = a.N > xcd ? (a.N - xcd + 7) >> 3 : 0;
    const int npw = npx > jb ? (npx - jb + G8 - 1) / G8 : 0;
    auto patch_of = [&](int i) __attribute__((always_inline)) { return 8 * (jb + (i < npw ? i : npw - 1) * G8) + xcd; };

    f32x4 RA[2];
    auto fetch = [&](int q) __attribute__((always_inline)) {      // this wave's four rows of plane q of the workgroup's sequence (patch q >> 4, plane q & 15)
        const unsigned row0 = ((unsigned)patch_of(q >> 4) * 16u + (unsigned)(q & 15)) * 16u + 4u * (unsigned)wave;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            RA[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(i_rsrc, (int)ldA, f3_s((row0 + 2u * u) * 512u), 0));
    };
    auto stage_unit = [&](auto U, int img, float sc, float sc11) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value;
        char *dst = w_base + img * F3_PLANE + 2 * u * F3_SLOT;
        const f32x4 g = RA[u];
        const f16x2 h01 = __builtin_convertvector(f32x2{g.x * sc, g.y * sc}, f16x2), h23 = __builtin_convertvector(f32x2{g.z * sc, g.w * sc}, f16x2);
#if F3_ONEACC
        const f16x2 l01 = __builtin_convertvector(f32x2{__builtin_fmaf(g.x, sc, -(float)h01.x), __builtin_fmaf(g.y, sc, -(float)h01.y)}, f16x2);
        const f16x2 l23 = __builtin_convertvector(f32x2{__builtin_fmaf(g.z, sc, -(float)h23.x), __builtin_fmaf(g.w, sc, -(float)h23.y)}, f16x2);
#else
        const f16x2 l01 = __builtin_convertvector(f32x2{__builtin_fmaf((float)h01.x, -2048.f, g.x * sc11), __builtin_fmaf((float)h01.y, -2048.f, g.y * sc11)}, f16x2);
        const f16x2 l23 = __builtin_convertvector(f32x2{__builtin_fmaf((float)h23.x, -2048.f, g.z * sc11), __builtin_fmaf((float)h23.y, -2048.f, g.w * sc11)}, f16x2);
#endif
        *reinterpret_cast<i32x2 *>(dst) = i32x2{__builtin_bit_cast(int, h01), __builtin_bit_cast(int, h23)};
        *reinterpret_cast<i32x2 *>(dst + F3_ROWB) = i32x2{__builtin_bit_cast(int, l01), __builtin_bit_cast(int, l23)};
    };

    f32x4 acc[3][4], acx[3][4];
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
        for (int ry = 0; ry < 4; ++ry) { acc[st][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; acx[st][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    f32x4 keep[4];      // the ReLU'd rows of the even plane of a pair
    f32x4 hold;         // row 0 / 2 of the odd plane until row 1 / 3 is there
#pragma unroll
    for (int ry = 0; ry < 4; ++ry) keep[ry] = f32x4{0.f, 0.f, 0.f, 0.f};
    hold = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 pkeep = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 okeep[2];
    okeep[0] = okeep[1] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one channel of a pooled voxel: the window in order (dz, dy, dx); the first maximum wins (strict >, as pool_fwd_vox_kernel).  Returns the maximum, idx = its position
    auto pool1 = [&](float e00, float e01, float e10, float e11, float o00, float o01, float o10, float o11, unsigned *idx) __attribute__((always_inline)) {
        float b = e00;
        unsigned k = 0u;
        if (e01 > b) { b = e01; k = 1u; }
        if (e10 > b) { b = e10; k = 2u; }
        if (e11 > b) { b = e11; k = 3u; }
        if (o00 > b) { b = o00; k = 4u; }
        if (o01 > b) { b = o01; k = 5u; }
        if (o10 > b) { b = o10; k = 6u; }
        if (o11 > b) { b = o11; k = 7u; }
        *idx = k;
        return b;
    };
    auto nb = [&](float v) __attribute__((always_inline)) {       // the value of the lane holding voxel x ^ 1
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));      // quad_perm [1, 0, 3, 2]
    };
    // pooled row (rows ry0, ry0 + 1 of planes zo - 1 (keep) and zo (r0, r1)) of pooled plane zo / 2
    auto pool_rows = [&](const f32x4 &e0, const f32x4 &e1, const f32x4 &r0, const f32x4 &r1, int p, int zo, int prow, bool tv) __attribute__((always_inline)) {
        unsigned i0, i1, i2, i3;
        const float m0 = pool1(e0.x, nb(e0.x), e1.x, nb(e1.x), r0.x, nb(r0.x), r1.x, nb(r1.x), &i0);
        const float m1 = pool1(e0.y, nb(e0.y), e1.y, nb(e1.y), r0.y, nb(r0.y), r1.y, nb(r1.y), &i1);
        const float m2 = pool1(e0.z, nb(e0.z), e1.z, nb(e1.z), r0.z, nb(r0.z), r1.z, nb(r1.z), &i2);
        const float m3 = pool1(e0.w, nb(e0.w), e1.w, nb(e1.w), r0.w, nb(r0.w), r1.w, nb(r1.w), &i3);
        const unsigned pv = (unsigned)f3_s((((unsigned)p * 8u + (unsigned)(tv ? (zo >> 1) : 0)) * 8u + (unsigned)prow) * 8u);      // first pooled voxel of the row
        const f32x4 o = f32x4{m0, m1, m2, m3};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o), po_rsrc, (int)(tv ? p_out : F3_OOB), (int)(pv * 64u), 0);
        ALQ_STORE_HOLD("v"(o));      // (a 16-byte store with a scalar offset reads its data late: nothing may write these registers in the next cycles, t3d_fwd_kernel)
        __builtin_amdgcn_raw_buffer_store_b32((int)(i0 | (i1 << 8) | (i2 << 16) | (i3 << 24)), pa_rsrc, (int)(tv ? p_arg : F3_OOB), (int)(pv * 16u), 0);
        float s_ = (m0 + m1) + (m2 + m3);
        s_ += __shfl_xor(s_, 16, 64);
        s_ += __shfl_xor(s_, 32, 64);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, s_), pu_rsrc, (int)(tv ? p_sum : F3_OOB), (int)(pv * 4u), 0);
        pkeep = o;      // (16-byte store data is read late by the hardware: the registers stay the store's until the end of the step, t3d_fwd_kernel)
    };

    // finish row ry of the plane held by set ST: plane zo (parity ODD compile-time) of patch p (tv: it exists)
    auto epi_row = [&](auto ST, auto RY, auto ODD, int p, int zo, bool tv, float inv) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value, ry = decltype(RY)::value;
        constexpr bool odd = decltype(ODD)::value != 0;
        const unsigned vrow = (unsigned)f3_s((((unsigned)p * 16u + (unsigned)(tv ? zo : 0)) * 16u + (unsigned)(4 * wave + ry)) * 16u);      // first voxel of the row
#if F3_ONEACC
        const f32x4 c = acc[st][ry], cx = f32x4{0.f, 0.f, 0.f, 0.f};
#else
        const f32x4 c = acc[st][ry], cx = acx[st][ry];      // (not cleared: the first MFMA of the set's next plane starts from zero)
#endif
        const float v0 = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(cx.x, 0x1p-11f, c.x), inv, bias4.x), 0.f);
        const float v1 = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(cx.y, 0x1p-11f, c.y), inv, bias4.y), 0.f);
        const float v2 = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(cx.z, 0x1p-11f, c.z), inv, bias4.z), 0.f);
        const float v3 = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(cx.w, 0x1p-11f, c.w), inv, bias4.w), 0.f);
        const f32x4 o = f32x4{v0, v1, v2, v3};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o), o_rsrc, (int)(tv ? e_out : F3_OOB), (int)(vrow * 64u), 0);
        ALQ_STORE_HOLD("v"(o));
        const unsigned bits = (v0 > 0.f ? 1u : 0u) | (v1 > 0.f ? 2u : 0u) | (v2 > 0.f ? 4u : 0u) | (v3 > 0.f ? 8u : 0u);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)bits, s_rsrc, (int)(tv ? e_sg : F3_OOB), (int)(vrow * 4u), 0);
        float s_ = (v0 + v1) + (v2 + v3);
        s_ += __shfl_xor(s_, 16, 64);
        s_ += __shfl_xor(s_, 32, 64);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, s_), u_rsrc, (int)(tv ? e_sum : F3_OOB), (int)(vrow * 4u), 0);
        if constexpr (!odd) {
            keep[ry] = o;      // (kept alive by the pool of the plane above: also what the late-reading 16-byte store needs)
        } else if constexpr ((ry & 1) == 0) {
            hold = o;
        } else {
            pool_rows(keep[ry - 1], keep[ry], hold, o, p, zo, 2 * wave + (ry >> 1), tv);
            okeep[ry >> 1] = o;      // (its 16-byte store reads the registers late - with a second wave on the SIMD later than the pool's arithmetic lasts)
        }
    };

    // Step s of patch i, J = s % 6 compile-time (accumulator sets rotate with s % 3, the pool pairs planes with s % 2): image q & 1 holds input plane s (q = 16 i + s).
    // Block r = input row 4 w - 1 + r: block 0: row 3 of plane s - 2 (set J + 1), block 1: the two staging units of plane q + 1 (RA) into the other image,
    // block 2: the loads of plane q + 2, blocks 3, 4, 5: rows 0, 1, 2 of plane s - 1 (set J + 2)
    auto step = [&](auto JJ, int i, int s) __attribute__((always_inline)) {
        constexpr int J6 = decltype(JJ)::value, J = J6 % 3;
        constexpr int S0 = (J + 1) % 3, S1 = J, S2 = (J + 2) % 3;      // sets of the output planes s + 1 (dz = 0), s (dz = 1), s - 1 (dz = 2)
        constexpr int odd1 = (J6 + 1) & 1, odd2 = J6 & 1;               // parity of planes s - 1, s - 2
        __builtin_amdgcn_sched_barrier(0);
        const int q = 16 * i + s;
        const int p = patch_of(i);
        const float inv = __builtin_ldexpf(1.f, -(patch_exp(p) + a.e_w));
        const int cen = patch_exp(patch_of((q + 1) >> 4));
        const bool nv = ((q + 1) >> 4) < npw;
        const float sc = nv ? __builtin_ldexpf(1.f, cen) : 0.f, sc11 = nv ? __builtin_ldexpf(1.f, cen + 11) : 0.f;
        const int img = q & 1;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const char *fb = f_base + img * F3_PLANE;
        f16x8 Fh[2], Fl[2];
        auto frag = [&](auto R, f16x8 *fh, f16x8 *fl) __attribute__((always_inline)) {
            constexpr int r = decltype(R)::value;
            *fh = *reinterpret_cast<const f16x8 *>(fb + r * F3_SLOT);
            *fl = *reinterpret_cast<const f16x8 *>(fb + r * F3_SLOT + F3_ROWB);
        };
        frag(F3IC<0>{}, &Fh[0], &Fl[0]);
        auto block = [&](auto R) __attribute__((always_inline)) {
            constexpr int r = decltype(R)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (r + 1 < 6) frag(F3IC<r + 1>{}, &Fh[(r + 1) % 2], &Fl[(r + 1) % 2]);
            if constexpr (r == 0) epi_row(F3IC<S0>{}, F3IC<3>{}, F3IC<odd2>{}, p, s - 2, s >= 2, inv);
            if constexpr (r == 1) { stage_unit(F3IC<0>{}, img ^ 1, sc, sc11); stage_unit(F3IC<1>{}, img ^ 1, sc, sc11); }
            if constexpr (r == 2) fetch(q + 2);
            if constexpr (r >= 3) epi_row(F3IC<S2>{}, F3IC<r - 3>{}, F3IC<odd1>{}, p, s - 1, s >= 1, inv);
            const f16x8 xh = Fh[r % 2], xl = Fl[r % 2];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int ry = r - dy;        // output row 4 w + ry takes input row 4 w - 1 + r through tap dy
                if (ry < 0 || ry > 3) continue;
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    const int st = dz == 0 ? S0 : (dz == 1 ? S1 : S2);
                    const int kk = dz * 3 + dy;
                    const bool first = dz == 0 && dy == 0;      // the first contribution to row ry of plane s + 1 starts its accumulators
                    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
#if F3_ONEACC
                    f32x4 c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[kk], xh, first ? zero : acc[st][ry], 0, 0, 0);      // the small products first
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[kk], xl, c1, 0, 0, 0);
                    acc[st][ry] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[kk], xh, c1, 0, 0, 0);
#else
                    acx[st][ry] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[kk], xh, first ? zero : acx[st][ry], 0, 0, 0);      // (l, h) + (h, l) at 2^11, (h, h)
                    acc[st][ry] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[kk], xh, first ? zero : acc[st][ry], 0, 0, 0);
                    acx[st][ry] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[kk], xl, acx[st][ry], 0, 0, 0);
#endif
                }
            }
#pragma unroll
            for (int m = 0; m < 9 * ((r == 0 || r == 5) ? 1 : ((r == 1 || r == 4) ? 2 : 3)); ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(F3_FILL_MASK, F3_PIPE, 0);
            }
        };
        block(F3IC<0>{}); block(F3IC<1>{}); block(F3IC<2>{}); block(F3IC<3>{}); block(F3IC<4>{}); block(F3IC<5>{});
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 k0 = pkeep, k1 = okeep[0], k2 = okeep[1];
        asm volatile("" :: "v"(k0), "v"(k1), "v"(k2));
    };

    if (npw > 0) {
        {   // plane 0 of the first patch into image 0, the loads of plane 1
            fetch(0);
            const int ce = patch_exp(patch_of(0));
            const float sc = __builtin_ldexpf(1.f, ce), sc11 = __builtin_ldexpf(1.f, ce + 11);
            stage_unit(F3IC<0>{}, 0, sc, sc11); stage_unit(F3IC<1>{}, 0, sc, sc11);
            __builtin_amdgcn_sched_barrier(0);
            fetch(1);
        }
        for (int i = 0; i < npw; ++i) {
            for (int k = 0; k < 2; ++k) {
                step(F3IC<0>{}, i, 6 * k); step(F3IC<1>{}, i, 6 * k + 1); step(F3IC<2>{}, i, 6 * k + 2);
                step(F3IC<3>{}, i, 6 * k + 3); step(F3IC<4>{}, i, 6 * k + 4); step(F3IC<5>{}, i, 6 * k + 5);
            }
            step(F3IC<0>{}, i, 12); step(F3IC<1>{}, i, 13); step(F3IC<2>{}, i, 14); step(F3IC<3>{}, i, 15);
            // behind the last input plane: row 3 of plane 14 (set 2, even), plane 15 (set 0, odd) - cleared: it is plane 0 of the next patch
            __builtin_amdgcn_sched_barrier(0);
            const int p = patch_of(i);
            const float inv = __builtin_ldexpf(1.f, -(patch_exp(p) + a.e_w));
            epi_row(F3IC<2>{}, F3IC<3>{}, F3IC<0>{}, p, 14, true, inv);
            epi_row(F3IC<0>{}, F3IC<0>{}, F3IC<1>{}, p, 15, true, inv);
            epi_row(F3IC<0>{}, F3IC<1>{}, F3IC<1>{}, p, 15, true, inv);
            epi_row(F3IC<0>{}, F3IC<2>{}, F3IC<1>{}, p, 15, true, inv);
            epi_row(F3IC<0>{}, F3IC<3>{}, F3IC<1>{}, p, 15, true, inv);
#pragma unroll
            for (int ry = 0; ry < 4; ++ry) { acc[0][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; acx[0][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7" :: "v"(pkeep), "v"(okeep[0]), "v"(okeep[1]) : "memory");
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------- host
int f3d_build(const View &in, const View &out, const int k[3], const int lo[3], const int s[3], F3dPlan *plan) {
    plan->ok = false;
    if (getenv("ALQ_NO_F3D")) return ALQ_OK;
    if (!(k[0] == 3 && k[1] == 3 && k[2] == 3 && s[0] == 1 && s[1] == 1 && s[2] == 1 && lo[0] == 1 && lo[1] == 1 && lo[2] == 1)) return ALQ_OK;
    if (!(in.D == 16 && in.H == 16 && in.W == 16 && out.D == 16 && out.H == 16 && out.W == 16 && in.C == 8 && in.split == 0 && in.cs == 8 && in.c0 == 0 &&
          out.C == 16 && out.cs == 16 && out.c0 == 0 && out.split == 0)) return ALQ_OK;
    plan->flops_per_patch = 2.0 * 27 * 8 * 16 * 4096.0;
    plan->ok = true;
    return ALQ_OK;
}

// W: TF conv filter [tap = (tz * 3 + ty) * 3 + tx][ci (8)][co (16)].  Fragment of (dz, dy): lane -> row co = lane & 15, k-group kg = lane >> 4 = tx (3: zeros), ci = c.
void f3d_pack(F3dPlan *plan, const float *W) {
    float amax = 0.f;
    for (size_t i = 0; i < (size_t)27 * 8 * 16; ++i) amax = std::max(amax, std::fabs(W[i]));
    int ex = 0;
    if (amax > 0.f) (void)std::frexp(amax, &ex);
    plan->w_exp = 14 - ex;
    plan->h_Whi.assign((size_t)9 * 64 * 8, 0);
    plan->h_Wlo.assign((size_t)9 * 64 * 8, 0);
    for (int kk = 0; kk < 9; ++kk)
        for (int lane = 0; lane < 64; ++lane) {
            const int co = lane & 15, tx = lane >> 4;
            for (int c = 0; c < 8; ++c) {
                const float w = tx < 3 ? W[((size_t)(kk * 3 + tx) * 8 + c) * 16 + co] : 0.f;
                const float ws = std::ldexp(w, plan->w_exp);
                const _Float16 h = (_Float16)ws;
                const _Float16 l = (_Float16)std::ldexp(ws - (float)h, F3_ONEACC ? 0 : 11);
                unsigned short hb, lb;
                std::memcpy(&hb, &h, 2);
                std::memcpy(&lb, &l, 2);
                plan->h_Whi[((size_t)kk * 64 + lane) * 8 + c] = hb;
                plan->h_Wlo[((size_t)kk * 64 + lane) * 8 + c] = lb;
            }
        }
}

int f3d_fwd_launch(alq_ctx *ctx, const F3dPlan &plan, int N, const float *in, const unsigned *amax, const float *bias, float *out, unsigned char *sg,
                   float *osum, float *pout, unsigned char *parg, float *posum) {
    ALQ_REQUIRE(plan.ok && plan.d_Whi && plan.d_Wlo, ALQ_EINVAL, "f3d: weights not set");
    ALQ_REQUIRE(in && amax && bias && out && pout && parg, ALQ_EINVAL, "f3d: missing argument");
    ALQ_REQUIRE(N < 4096, ALQ_EUNSUPPORTED, "f3d: 32-bit byte offsets hold fewer than 4096 patches per pass");
    if (N <= 0) return ALQ_OK;
    F3Args a;
    a.in = in; a.amax = amax; a.Whi = reinterpret_cast<const unsigned short *>(plan.d_Whi); a.Wlo = reinterpret_cast<const unsigned short *>(plan.d_Wlo);
    a.bias = bias; a.out = out; a.sg = sg; a.osum = osum; a.pout = pout; a.parg = parg; a.posum = posum; a.e_w = plan.w_exp; a.N = N;
    const int cus = ctx->num_cus;
    long long g = std::min<long long>((long long)F3_WGS * cus, (long long)N);
    g = std::max<long long>(8, (g + 7) / 8 * 8);
    ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(f3d_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)F3_STRIP));
    ProfScope ps(ctx, PROF_IGEMM_F16, plan.flops_per_patch * N);
    hipLaunchKernelGGL(f3d_fwd_kernel, dim3((unsigned)g), dim3(256), F3_STRIP, ctx->stream, a);
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

}  // namespace alq
