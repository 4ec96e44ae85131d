// Forward of NET-C's `dec1` (3x3x3 conv over the concat [up1 output | enc2 skip], 32 -> 16 channels at 16^3, + bias + ReLU;
// NN_extended.py:416-426) in a Fisher pass: the largest launch the two-slot engine still ran (0.79 ms per 2047 patches at 36 % of
// the fp16 matrix-core rate).  Plane sweep at ONE wave per SIMD, the scheme of c3d.hip at this layer's shape:
//   * a workgroup owns a patch and sweeps z; wave w owns output rows 4 w .. 4 w + 3 of every plane.  Step s has input plane s in
//     LDS (an image of 18 rows x 18 voxel slots, fp16 pairs; y and x halo = zero rows / slots) and contracts it into the THREE
//     output planes it touches (s + 1, s, s - 1: three accumulator sets in registers, rotating with s, compile-time indices in a
//     loop unrolled by three), so an activation fragment (input row, dx; hi and lo piece) is read ONCE and feeds up to
//     3 (dy) x 3 (dz) x 3 products = 27 MFMAs.  A first version swept y with four planes per workgroup: 36 fragment pairs per
//     162 MFMAs, two void steps in ten, and the same 0.8 ms as the two-slot engine;
//   * MFMA 16 x 16 x 32 (f16): rows = 16 output channels, columns = the 16 voxels of an x row, K = the 32 input channels of one tap;
//     all 27 x 2 weight fragments stay in registers (216 of the 512 a lone wave has);
//   * while plane s is contracted the wave converts its four rows of plane s + 1 (loaded during step s - 1) into the other image,
//     and finishes output plane s - 1 row by row as its last contributions arrive (row 3 at the start of the next step): bias,
//     ReLU, 16-byte stores (a lane holds 4 channels of one voxel), the sign byte of those 4 channels, the voxel's channel sum;
//     one barrier per step; six scheduling blocks (input row) per step, fragments read one block ahead of their MFMAs, one
//     MFMA then up to two other instructions (c3d_fwd_kernel);
//   * fp16 pairs x 2^e = h + l 2^-11, lo pieces scaled up, their two products in a second accumulator (the per-patch input bounds
//     are derived ones - fwd_bounds_kernel - and loose: at their true scale the lo pieces would be fp16 subnormals);
//   * LDS row = [4 k-groups][18 voxel slots (x = -1 .. 16)][8 channels x 2 B]: the 16 lanes of a k-group read 256 contiguous
//     bytes (conflict-free), an x shift is +16 B;
//   * the first and last input plane of a patch also run the MFMAs of the output planes outside the volume (4 % of the work; their
//     sets are discarded) - no special steps, no control flow inside a patch except the loop of five times three steps.
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

struct D3Args {
    const float *inA, *inB;       // [N][16^3][16]: channels 0 .. 15 / 16 .. 31 of the concat input (dense)
    const unsigned *amaxA, *amaxB;      // per patch: bit pattern of a bound on max |x| of each half
    const unsigned short *Whi;    // [27 taps][64 lanes][8] fp16 bits: hi pieces (d3d_pack)
    const unsigned short *Wlo;    // the lo pieces (x 2^11), same layout
    const float *bias;            // [16]
    float *out;                   // [N][16^3][16]
    unsigned char *sg;            // sign field of out: byte (voxel * 16 + c) / 4, bit c & 3  (or nullptr)
    float *osum;                  // [N][16^3] channel sums of out (or nullptr)
    int e_w;                      // weights were scaled by 2^e_w
    int relu;
    int N;
};

constexpr unsigned D3_OOB = 0xffffff00u;
constexpr int D3_KG = 18 * 16;                // a k-group block of a row: 18 voxel slots x 8 channels x 2 B
constexpr int D3_ROWB = 4 * D3_KG;            // one piece of a row: 1152 B
constexpr int D3_SLOT = 2 * D3_ROWB;          // a row: pieces h, l
constexpr int D3_PLANE = 18 * D3_SLOT;        // image of a plane: rows y = -1 .. 16: 41,472 B
constexpr int D3_STRIP = 2 * D3_PLANE;        // the plane being contracted and the one being staged: 82,944 B

__device__ inline __amdgpu_buffer_rsrc_t d3_rsrc(const void *base, unsigned long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ inline int d3_s(unsigned v) { return __builtin_amdgcn_readfirstlane((int)v); }

template <int V> struct D3IC { static constexpr int value = V; };
// what may fill the gap behind an MFMA: VALU (2), SALU (4), VMEM (0x10), DS (0x80)
constexpr int D3_FILL_MASK = 0x096;
#ifndef D3_PIPE
#define D3_PIPE 2
#endif
#ifndef D3_START
#define D3_START 1
#endif

struct D3Row { f32x4 a, b; };      // what staging one row needs, per lane: 4 channels of one voxel from each half

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void d3d_fwd_kernel(const D3Args a) {
    extern __shared__ __attribute__((aligned(16))) char d3lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char *strip = d3lds;
    // zero fill, once: the images' rows y = -1, 16 and the slots x = -1, 16 of every row are never written again
    for (int i = threadIdx.x; i < D3_STRIP / 16; i += 256) reinterpret_cast<i32x4 *>(strip)[i] = i32x4{0, 0, 0, 0};
    __syncthreads();
    const int n = lane & 15, kg = lane >> 4;
    f16x8 wh[27], wl[27];
#pragma unroll
    for (int c = 0; c < 27; ++c) {
        wh[c] = *reinterpret_cast<const f16x8 *>(a.Whi + ((size_t)c * 64 + lane) * 8);
        wl[c] = *reinterpret_cast<const f16x8 *>(a.Wlo + ((size_t)c * 64 + lane) * 8);
    }
#pragma unroll
    for (int c = 0; c < 27; ++c) asm volatile("" : "+a"(wh[c]), "+a"(wl[c]));      // arrived before the loop, and in the accumulator half of the file
    f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + 4 * kg);
    asm volatile("" : "+v"(bias4));
    const float relu_floor = a.relu ? 0.f : -__builtin_inff();

    typedef const unsigned __attribute__((address_space(4))) *cu32p;
    const cu32p amaxA_c = (cu32p)(unsigned long long)a.amaxA, amaxB_c = (cu32p)(unsigned long long)a.amaxB;
    auto patch_exp = [&](int p) __attribute__((always_inline)) {      // max |x| < 2^ex -> scale 2^(14 - ex); all-zero patch: 0  (c3d_fwd_kernel)
        const unsigned fa = amaxA_c[p], fb = amaxB_c[p];
        const unsigned fm = fa > fb ? fa : fb;
        const int ex = (int)((fm >> 23) & 255u) - 126;
        const int ce = 14 - ex;
        return fm ? (ce < 96 ? ce : 96) : 0;
    };

    const __amdgpu_buffer_rsrc_t ia_rsrc = d3_rsrc(a.inA, (unsigned long long)a.N * 4096 * 64);
    const __amdgpu_buffer_rsrc_t ib_rsrc = d3_rsrc(a.inB, (unsigned long long)a.N * 4096 * 64);
    const __amdgpu_buffer_rsrc_t o_rsrc = d3_rsrc(a.out, (unsigned long long)a.N * 4096 * 64);
    const __amdgpu_buffer_rsrc_t s_rsrc = d3_rsrc(a.sg, a.sg ? (unsigned long long)a.N * 4096 * 4 : 0ull);
    const __amdgpu_buffer_rsrc_t u_rsrc = d3_rsrc(a.osum, a.osum ? (unsigned long long)a.N * 4096 * 4 : 0ull);

    // staging lane roles: voxel x = lane >> 2 of a row, channels 4 cq .. + 3 of each half: k-group 2 h + (cq >> 1), bytes 8 (cq & 1) ..; image row 4 w + 1 + r
    const int sx = lane >> 2, cq = lane & 3;
    char *const w_base = strip + (4 * wave + 1) * D3_SLOT + (cq >> 1) * D3_KG + (sx + 1) * 16 + (cq & 1) * 8;      // half A of image 0; half B: + 2 k-group blocks
    const unsigned ldA = (unsigned)lane * 16u;
    // fragment lane roles: column n = voxel x (slot n + dx), k-group kg; input row y = 4 w - 1 + r = image row 4 w + r
    const char *const f_base = strip + 4 * wave * D3_SLOT + kg * D3_KG + n * 16;
    // epilogue lane roles: voxel x = n of the output row, channels 4 kg .. + 3
    const unsigned e_out = (unsigned)n * 64u + (unsigned)kg * 16u, e_sg = (unsigned)n * 4u + (unsigned)kg;
    const unsigned e_sum = kg == 0 ? (unsigned)n * 4u : D3_OOB;

    // patches of this workgroup (XCD-aware as in t3d.hip): workgroup b of XCD b % 8 takes k = b / 8, b / 8 + G / 8, ... of the patches 8 k + b % 8
    const int G8 = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, jb = (int)blockIdx.x >> 3;
    const int npx = a.N > xcd ? (a.N - xcd + 7) >> 3 : 0;
    const int npw = npx > jb ? (npx - jb + G8 - 1) / G8 : 0;
    auto patch_of = [&](int i) __attribute__((always_inline)) { return 8 * (jb + (i < npw ? i : npw - 1) * G8) + xcd; };

    D3Row RA[4];
    // loads of this wave's four rows of plane q of the workgroup's sequence (patch q >> 4, plane q & 15)
    auto fetch = [&](int q) __attribute__((always_inline)) {
        const unsigned row0 = ((unsigned)patch_of(q >> 4) * 16u + (unsigned)(q & 15)) * 16u + 4u * (unsigned)wave;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            RA[r].a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ia_rsrc, (int)ldA, d3_s((row0 + r) * 1024u), 0));
            RA[r].b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ib_rsrc, (int)ldA, d3_s((row0 + r) * 1024u), 0));
        }
    };
    // unit u = 2 r + h: half h of row r of RA -> image `img` (scale 2^ce)
    auto stage_unit = [&](auto U, int img, float sc, float sc11) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value, r = u >> 1, h = u & 1;
        char *dst = w_base + img * D3_PLANE + r * D3_SLOT + h * 2 * D3_KG;
        const f32x4 g = h ? RA[r].b : RA[r].a;
        const f16x2 h01 = __builtin_convertvector(f32x2{g.x * sc, g.y * sc}, f16x2), h23 = __builtin_convertvector(f32x2{g.z * sc, g.w * sc}, f16x2);
        const f16x2 l01 = __builtin_convertvector(f32x2{__builtin_fmaf((float)h01.x, -2048.f, g.x * sc11), __builtin_fmaf((float)h01.y, -2048.f, g.y * sc11)}, f16x2);
        const f16x2 l23 = __builtin_convertvector(f32x2{__builtin_fmaf((float)h23.x, -2048.f, g.z * sc11), __builtin_fmaf((float)h23.y, -2048.f, g.w * sc11)}, f16x2);
        *reinterpret_cast<i32x2 *>(dst) = i32x2{__builtin_bit_cast(int, h01), __builtin_bit_cast(int, h23)};
        *reinterpret_cast<i32x2 *>(dst + D3_ROWB) = i32x2{__builtin_bit_cast(int, l01), __builtin_bit_cast(int, l23)};
    };

    // three accumulator sets x four rows x (hi products, lo products at 2^11)
    f32x4 acc[3][4], acx[3][4];
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
        for (int ry = 0; ry < 4; ++ry) { acc[st][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; acx[st][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    f32x4 okeep[2];
    okeep[0] = okeep[1] = f32x4{0.f, 0.f, 0.f, 0.f};

    // finish row ry of the plane held by set ST: plane zo of patch p (tv: it exists), then clear the set's row
    auto epi_row = [&](auto ST, auto RY, int p, int zo, bool tv, float inv) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value, ry = decltype(RY)::value;
        const unsigned vrow = (unsigned)d3_s((((unsigned)p * 16u + (unsigned)(tv ? zo : 0)) * 16u + (unsigned)(4 * wave + ry)) * 16u);      // first voxel of the row
        const f32x4 c = acc[st][ry], cx = acx[st][ry];      // (not cleared: the first MFMA of the set's next plane starts from zero)
        if (!D3_START) { acc[st][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; acx[st][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const float v0 = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(cx.x, 0x1p-11f, c.x), inv, bias4.x), relu_floor);
        const float v1 = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(cx.y, 0x1p-11f, c.y), inv, bias4.y), relu_floor);
        const float v2 = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(cx.z, 0x1p-11f, c.z), inv, bias4.z), relu_floor);
        const float v3 = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(cx.w, 0x1p-11f, c.w), inv, bias4.w), relu_floor);
        const f32x4 o = f32x4{v0, v1, v2, v3};
        okeep[ry & 1] = o;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o), o_rsrc, (int)(tv ? e_out : D3_OOB), (int)(vrow * 64u), 0);
        ALQ_STORE_HOLD("v"(o));      // (a 16-byte store with a scalar offset reads its data late: nothing may write these registers in the next cycles, t3d_fwd_kernel)
        const unsigned bits = (v0 > 0.f ? 1u : 0u) | (v1 > 0.f ? 2u : 0u) | (v2 > 0.f ? 4u : 0u) | (v3 > 0.f ? 8u : 0u);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)bits, s_rsrc, (int)(tv ? e_sg : D3_OOB), (int)(vrow * 4u), 0);
        float s_ = (v0 + v1) + (v2 + v3);
        s_ += __shfl_xor(s_, 16, 64);
        s_ += __shfl_xor(s_, 32, 64);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, s_), u_rsrc, (int)(tv ? e_sum : D3_OOB), (int)(vrow * 4u), 0);
    };

    // Step s (J = s % 3) of patch i: image q & 1 holds input plane s (q = 16 i + s).  Block r = input row 4 w - 1 + r:
    //   block 0: row 3 of plane s - 2 (set J + 1, its last contributions came at the end of step s - 1);  blocks 1, 2: the eight staging units of
    //   plane q + 1 (RA, loaded during step s - 1) into the other image;  block 3: the loads of plane q + 2;  blocks 3, 4, 5: rows 0, 1, 2 of plane
    //   s - 1 (set J + 2), finished by blocks 2, 3, 4.
    auto step = [&](auto JJ, int i, int s) __attribute__((always_inline)) {
        constexpr int J = decltype(JJ)::value;
        constexpr int S0 = (J + 1) % 3, S1 = J, S2 = (J + 2) % 3;      // sets of the output planes s + 1 (dz = 0), s (dz = 1), s - 1 (dz = 2)
        __builtin_amdgcn_sched_barrier(0);
        const int q = 16 * i + s;
        const int p = patch_of(i);
        const float inv = __builtin_ldexpf(1.f, -(patch_exp(p) + a.e_w));
        const int pn = patch_of((q + 1) >> 4);
        const int cen = patch_exp(pn);
        const bool nv = ((q + 1) >> 4) < npw;      // (behind the last plane of the workgroup: zeros into the image nobody reads)
        const float sc = nv ? __builtin_ldexpf(1.f, cen) : 0.f, sc11 = nv ? __builtin_ldexpf(1.f, cen + 11) : 0.f;
        const int img = q & 1;
        // every wave's rows of plane q are in image q & 1, and every wave is done reading the image this step overwrites
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const char *fb = f_base + img * D3_PLANE;
        f16x8 Fh[2][3], Fl[2][3];
        auto frag = [&](auto R, f16x8 *fh, f16x8 *fl) __attribute__((always_inline)) {
            constexpr int r = decltype(R)::value;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                fh[dx] = *reinterpret_cast<const f16x8 *>(fb + r * D3_SLOT + dx * 16);
                fl[dx] = *reinterpret_cast<const f16x8 *>(fb + r * D3_SLOT + dx * 16 + D3_ROWB);
            }
        };
        frag(D3IC<0>{}, Fh[0], Fl[0]);
        auto block = [&](auto R) __attribute__((always_inline)) {
            constexpr int r = decltype(R)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (r + 1 < 6) frag(D3IC<r + 1>{}, Fh[(r + 1) % 2], Fl[(r + 1) % 2]);
            if constexpr (r == 0) epi_row(D3IC<S0>{}, D3IC<3>{}, p, s - 2, s >= 2, inv);
            if constexpr (r == 1) { stage_unit(D3IC<0>{}, img ^ 1, sc, sc11); stage_unit(D3IC<1>{}, img ^ 1, sc, sc11); stage_unit(D3IC<2>{}, img ^ 1, sc, sc11); stage_unit(D3IC<3>{}, img ^ 1, sc, sc11); }
            if constexpr (r == 2) { stage_unit(D3IC<4>{}, img ^ 1, sc, sc11); stage_unit(D3IC<5>{}, img ^ 1, sc, sc11); stage_unit(D3IC<6>{}, img ^ 1, sc, sc11); stage_unit(D3IC<7>{}, img ^ 1, sc, sc11); }
            if constexpr (r == 3) {
                fetch(q + 2);
                epi_row(D3IC<S2>{}, D3IC<0>{}, p, s - 1, s >= 1, inv);
            }
            if constexpr (r == 4) epi_row(D3IC<S2>{}, D3IC<1>{}, p, s - 1, s >= 1, inv);
            if constexpr (r == 5) epi_row(D3IC<S2>{}, D3IC<2>{}, p, s - 1, s >= 1, inv);
            int nm = 0;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const f16x8 xh = Fh[r % 2][dx], xl = Fl[r % 2][dx];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int ry = r - dy;        // output row 4 w + ry takes input row 4 w - 1 + r through tap dy
                    if (ry < 0 || ry > 3) continue;
#pragma unroll
                    for (int dz = 0; dz < 3; ++dz) {
                        const int st = dz == 0 ? S0 : (dz == 1 ? S1 : S2);
                        const int tap = (dz * 3 + dy) * 3 + dx;
                        // the first contribution to row ry of plane s + 1 starts its accumulators (no clearing pass over the set)
                        const bool first = D3_START && dz == 0 && dy == 0 && dx == 0;
                        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
                        acx[st][ry] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[tap], xh, first ? zero : acx[st][ry], 0, 0, 0);      // (l, h) + (h, l) at 2^11, (h, h)
                        acc[st][ry] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[tap], xh, first ? zero : acc[st][ry], 0, 0, 0);
                        acx[st][ry] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[tap], xl, acx[st][ry], 0, 0, 0);
                        nm += 3;
                    }
                }
            }
            (void)nm;
#pragma unroll
            for (int m = 0; m < 27 * ((r == 0 || r == 5) ? 1 : ((r == 1 || r == 4) ? 2 : 3)); ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(D3_FILL_MASK, D3_PIPE, 0);
            }
        };
        block(D3IC<0>{}); block(D3IC<1>{}); block(D3IC<2>{}); block(D3IC<3>{}); block(D3IC<4>{}); block(D3IC<5>{});
        __builtin_amdgcn_sched_barrier(0);
        // (16-byte store data is read late by the hardware: the last rows' registers stay theirs until here, t3d_fwd_kernel)
        const f32x4 k0 = okeep[0], k1 = okeep[1];
        asm volatile("" :: "v"(k0), "v"(k1));
    };

    if (npw > 0) {
        {   // plane 0 of the first patch into image 0, the loads of plane 1
            fetch(0);
            const int ce = patch_exp(patch_of(0));
            const float sc = __builtin_ldexpf(1.f, ce), sc11 = __builtin_ldexpf(1.f, ce + 11);
            stage_unit(D3IC<0>{}, 0, sc, sc11); stage_unit(D3IC<1>{}, 0, sc, sc11); stage_unit(D3IC<2>{}, 0, sc, sc11); stage_unit(D3IC<3>{}, 0, sc, sc11);
            stage_unit(D3IC<4>{}, 0, sc, sc11); stage_unit(D3IC<5>{}, 0, sc, sc11); stage_unit(D3IC<6>{}, 0, sc, sc11); stage_unit(D3IC<7>{}, 0, sc, sc11);
            __builtin_amdgcn_sched_barrier(0);
            fetch(1);
        }
        for (int i = 0; i < npw; ++i) {
            for (int k = 0; k < 5; ++k) {
                step(D3IC<0>{}, i, 3 * k);
                step(D3IC<1>{}, i, 3 * k + 1);
                step(D3IC<2>{}, i, 3 * k + 2);
            }
            step(D3IC<0>{}, i, 15);
            // behind the last input plane: row 3 of plane 14 (set 2), plane 15 (set 0) - cleared: it is plane 0 of the next patch, whose first
            // contributions come from its own input plane, not from a plane in front of it
            __builtin_amdgcn_sched_barrier(0);
            const int p = patch_of(i);
            const float inv = __builtin_ldexpf(1.f, -(patch_exp(p) + a.e_w));
            epi_row(D3IC<2>{}, D3IC<3>{}, p, 14, true, inv);
            epi_row(D3IC<0>{}, D3IC<0>{}, p, 15, true, inv);
            epi_row(D3IC<0>{}, D3IC<1>{}, p, 15, true, inv);
            epi_row(D3IC<0>{}, D3IC<2>{}, p, 15, true, inv);
            epi_row(D3IC<0>{}, D3IC<3>{}, p, 15, true, inv);
#pragma unroll
            for (int ry = 0; ry < 4; ++ry) { acc[0][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; acx[0][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7" :: "v"(okeep[0]), "v"(okeep[1]) : "memory");
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------- backward
// dec1's backward-data pass (16 -> 32 channels at 16^3: the cotangent of the concat input, fp16 pairs under the STATIC cotangent
// bound) on the same plane sweep.  din[u] = sum_d dout[u + d - 1] W[2 - d]^T: a forward conv with flipped taps, but its K is 16
// channels, so an MFMA K step holds TWO taps - k-groups 0, 1 one window position, 2, 3 another (each lane's LDS address carries
// its own position).  Per output row and dz five K steps: (dy, dx = 0 | dx = 1) for dy = 0 .. 2, (dy = 0 | dy = 1, dx = 2) and
// (dy = 2, dx = 2 | nothing): 15 x 3 products against the 13.5 x 3 of the real taps.  32 output channels = two MFMA row blocks: a
// wave takes ONE block (wave & 1) of EIGHT output rows (wave >> 1) - its 30 weight fragments (120 registers) and 3 x 8 x 2
// accumulators (192) fit the 512 registers of a lone wave; the two waves of a row group read the same activation fragments
// (40 reads per 360 MFMAs each).  Ten blocks (input rows) per step; the block that completes an output row of plane s - 1
// (row ry: block ry + 2) is followed by its epilogue - 16-byte stores into the wave's half of the concat cotangent, channel sums
// for the half that has a parameterised producer (up1).
struct D3BArgs {
    const float *dout;            // [N][16^3][16] cotangent of dec1's output, ReLU mask applied (dense)
    const unsigned short *Whi;    // [2 row blocks][3 dz][5 K steps][64 lanes][8] fp16 bits: hi pieces (d3d_bwd_pack)
    const unsigned short *Wlo;    // the lo pieces (x 2^11), same layout
    float *dinA, *dinB;           // [N][16^3][16]: channels 0 .. 15 / 16 .. 31 of the concat cotangent (dense)
    float *sumB;                  // [N][16^3] channel sums of dinB (or nullptr)
    float scale, scale11, inv;    // 2^e_in, 2^(e_in + 11), 2^-(e_in + e_w)
    int N;
};
constexpr int DB_KG = 18 * 16;                // a k-group block of a row: 18 voxel slots x 8 channels x 2 B
constexpr int DB_ROWB = 2 * DB_KG;            // one piece of a row: 576 B
constexpr int DB_SLOT = 2 * DB_ROWB;          // a row: pieces h, l
constexpr int DB_PLANE = 19 * DB_SLOT;        // image of a plane: rows y = -1 .. 16 and one more zero row (the pair fragment of the last row reads it): 21,888 B
constexpr int DB_STRIP = 2 * DB_PLANE;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void d3d_bwd_kernel(const D3BArgs a) {
    extern __shared__ __attribute__((aligned(16))) char d3lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cb = wave & 1, rg = wave >> 1;      // row block of the 32 output channels, group of eight output rows
    char *strip = d3lds;
    for (int i = threadIdx.x; i < DB_STRIP / 16; i += 256) reinterpret_cast<i32x4 *>(strip)[i] = i32x4{0, 0, 0, 0};
    __syncthreads();
    const int n = lane & 15, kg = lane >> 4;
    f16x8 wh[15], wl[15];
#pragma unroll
    for (int c = 0; c < 15; ++c) {
        wh[c] = *reinterpret_cast<const f16x8 *>(a.Whi + ((size_t)(cb * 15 + c) * 64 + lane) * 8);
        wl[c] = *reinterpret_cast<const f16x8 *>(a.Wlo + ((size_t)(cb * 15 + c) * 64 + lane) * 8);
    }
#pragma unroll
    for (int c = 0; c < 15; ++c) asm volatile("" : "+a"(wh[c]), "+a"(wl[c]));

    const __amdgpu_buffer_rsrc_t i_rsrc = d3_rsrc(a.dout, (unsigned long long)a.N * 4096 * 64);
    const __amdgpu_buffer_rsrc_t o_rsrc = d3_rsrc(cb ? a.dinB : a.dinA, (unsigned long long)a.N * 4096 * 64);
    const __amdgpu_buffer_rsrc_t u_rsrc = d3_rsrc(a.sumB, a.sumB ? (unsigned long long)a.N * 4096 * 4 : 0ull);

    // staging lane roles: voxel x = lane >> 2 of a row, channels 4 cq .. + 3: k-group cq >> 1, bytes 8 (cq & 1) ..; image row 4 w + 1 + r
    const int sx = lane >> 2, cq = lane & 3;
    char *const w_base = strip + (4 * wave + 1) * DB_SLOT + (cq >> 1) * DB_KG + (sx + 1) * 16 + (cq & 1) * 8;
    const unsigned ldA = (unsigned)lane * 16u;
    // fragment lane roles: column n = voxel x, k-group kg: channel half kg & 1 of window position kg >> 1.  Pair (dx = 0 | dx = 1) of input row r:
    // slot n + (kg >> 1);  pair (row r | row r + 1) at dx = 2: slot n + 2 of row r + (kg >> 1).  Input row y = 8 rg - 1 + r = image row 8 rg + r
    const char *const f01 = strip + 8 * rg * DB_SLOT + (kg & 1) * DB_KG + (n + (kg >> 1)) * 16;
    const char *const f22 = strip + 8 * rg * DB_SLOT + (kg >> 1) * DB_SLOT + (kg & 1) * DB_KG + (n + 2) * 16;
    // epilogue lane roles: voxel x = n of the output row, channels 16 cb + 4 kg .. + 3
    const unsigned e_out = (unsigned)n * 64u + (unsigned)kg * 16u;
    const unsigned e_sum = (kg == 0 && cb == 1) ? (unsigned)n * 4u : D3_OOB;

    const int G8 = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, jb = (int)blockIdx.x >> 3;
    const int npx = a.N > xcd ? (a.N - xcd + 7) >> 3 : 0;
    const int npw = npx > jb ? (npx - jb + G8 - 1) / G8 : 0;
    auto patch_of = [&](int i) __attribute__((always_inline)) { return 8 * (jb + (i < npw ? i : npw - 1) * G8) + xcd; };

    f32x4 RA[4];
    auto fetch = [&](int q) __attribute__((always_inline)) {
        const unsigned row0 = ((unsigned)patch_of(q >> 4) * 16u + (unsigned)(q & 15)) * 16u + 4u * (unsigned)wave;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            RA[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(i_rsrc, (int)ldA, d3_s((row0 + r) * 1024u), 0));
    };
    auto stage_unit = [&](auto U, int img, float sc, float sc11) __attribute__((always_inline)) {
        constexpr int r = decltype(U)::value;
        char *dst = w_base + img * DB_PLANE + r * DB_SLOT;
        const f32x4 g = RA[r];
        const f16x2 h01 = __builtin_convertvector(f32x2{g.x * sc, g.y * sc}, f16x2), h23 = __builtin_convertvector(f32x2{g.z * sc, g.w * sc}, f16x2);
        const f16x2 l01 = __builtin_convertvector(f32x2{__builtin_fmaf((float)h01.x, -2048.f, g.x * sc11), __builtin_fmaf((float)h01.y, -2048.f, g.y * sc11)}, f16x2);
        const f16x2 l23 = __builtin_convertvector(f32x2{__builtin_fmaf((float)h23.x, -2048.f, g.z * sc11), __builtin_fmaf((float)h23.y, -2048.f, g.w * sc11)}, f16x2);
        *reinterpret_cast<i32x2 *>(dst) = i32x2{__builtin_bit_cast(int, h01), __builtin_bit_cast(int, h23)};
        *reinterpret_cast<i32x2 *>(dst + DB_ROWB) = i32x2{__builtin_bit_cast(int, l01), __builtin_bit_cast(int, l23)};
    };

    f32x4 acc[3][8], acx[3][8];
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
        for (int ry = 0; ry < 8; ++ry) { acc[st][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; acx[st][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    f32x4 okeep[2];
    okeep[0] = okeep[1] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto epi_row = [&](auto ST, auto RY, int p, int zo, bool tv) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value, ry = decltype(RY)::value;
        const unsigned vrow = (unsigned)d3_s((((unsigned)p * 16u + (unsigned)(tv ? zo : 0)) * 16u + (unsigned)(8 * rg + ry)) * 16u);
        const f32x4 c = acc[st][ry], cx = acx[st][ry];
        if (!D3_START) { acc[st][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; acx[st][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const float v0 = __builtin_fmaf(cx.x, 0x1p-11f, c.x) * a.inv, v1 = __builtin_fmaf(cx.y, 0x1p-11f, c.y) * a.inv;
        const float v2 = __builtin_fmaf(cx.z, 0x1p-11f, c.z) * a.inv, v3 = __builtin_fmaf(cx.w, 0x1p-11f, c.w) * a.inv;
        const f32x4 o = f32x4{v0, v1, v2, v3};
        okeep[ry & 1] = o;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o), o_rsrc, (int)(tv ? e_out : D3_OOB), (int)(vrow * 64u), 0);
        ALQ_STORE_HOLD("v"(o));      // (a 16-byte store with a scalar offset reads its data late: nothing may write these registers in the next cycles, t3d_fwd_kernel)
        float s_ = (v0 + v1) + (v2 + v3);
        s_ += __shfl_xor(s_, 16, 64);
        s_ += __shfl_xor(s_, 32, 64);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, s_), u_rsrc, (int)(tv ? e_sum : D3_OOB), (int)(vrow * 4u), 0);
    };

    // Step s (J = s % 3) of patch i: image q & 1 holds input plane s (q = 16 i + s); block r = input row 8 rg - 1 + r (r = 0 .. 9).
    //   block 0: row 7 of plane s - 2 (set J + 1);  blocks 2, 3: the four staging units of plane q + 1 into the other image;  block 4: the loads of plane
    //   q + 2;  block ry + 3 (ry = 0 .. 6): row ry of plane s - 1 (set J + 2), finished by block ry + 2
    auto step = [&](auto JJ, int i, int s) __attribute__((always_inline)) {
        constexpr int J = decltype(JJ)::value;
        constexpr int S0 = (J + 1) % 3, S1 = J, S2 = (J + 2) % 3;
        __builtin_amdgcn_sched_barrier(0);
        const int q = 16 * i + s;
        const int p = patch_of(i);
        const bool nv = ((q + 1) >> 4) < npw;
        const float sc = nv ? a.scale : 0.f, sc11 = nv ? a.scale11 : 0.f;
        const int img = q & 1;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const char *fa = f01 + img * DB_PLANE, *fb = f22 + img * DB_PLANE;
        f16x8 Fh[2][2], Fl[2][2];       // [set][pair kind: (dx 0 | dx 1), (row r | row r + 1 at dx 2)]
        auto frag = [&](auto R, f16x8 *fh, f16x8 *fl) __attribute__((always_inline)) {
            constexpr int r = decltype(R)::value;
            fh[0] = *reinterpret_cast<const f16x8 *>(fa + r * DB_SLOT);
            fl[0] = *reinterpret_cast<const f16x8 *>(fa + r * DB_SLOT + DB_ROWB);
            fh[1] = *reinterpret_cast<const f16x8 *>(fb + r * DB_SLOT);
            fl[1] = *reinterpret_cast<const f16x8 *>(fb + r * DB_SLOT + DB_ROWB);
        };
        frag(D3IC<0>{}, Fh[0], Fl[0]);
        auto mm = [&](auto ST, auto RY, auto KS, const f16x8 &xh, const f16x8 &xl) __attribute__((always_inline)) {
            constexpr int st = decltype(ST)::value, ry = decltype(RY)::value, ks = decltype(KS)::value;
            // K step 0 of dz = 0 is the first contribution to its row of plane s + 1: it starts the accumulators (no clearing pass over the set)
            constexpr bool first = D3_START && ks == 0;
            const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
            acx[st][ry] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ks], xh, first ? zero : acx[st][ry], 0, 0, 0);
            acc[st][ry] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ks], xh, first ? zero : acc[st][ry], 0, 0, 0);
            acx[st][ry] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ks], xl, acx[st][ry], 0, 0, 0);
        };
        auto block = [&](auto R) __attribute__((always_inline)) {
            constexpr int r = decltype(R)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (r + 1 < 10) frag(D3IC<r + 1>{}, Fh[(r + 1) % 2], Fl[(r + 1) % 2]);
            if constexpr (r == 0) epi_row(D3IC<S0>{}, D3IC<7>{}, p, s - 2, s >= 2);
            if constexpr (r == 2) { stage_unit(D3IC<0>{}, img ^ 1, sc, sc11); stage_unit(D3IC<1>{}, img ^ 1, sc, sc11); }
            if constexpr (r == 3) { stage_unit(D3IC<2>{}, img ^ 1, sc, sc11); stage_unit(D3IC<3>{}, img ^ 1, sc, sc11); }
            if constexpr (r == 4) fetch(q + 2);
            if constexpr (r >= 3) epi_row(D3IC<S2>{}, D3IC<r - 3>{}, p, s - 1, s >= 1);
            const f16x8 ph = Fh[r % 2][0], pl = Fl[r % 2][0], qh = Fh[r % 2][1], ql = Fl[r % 2][1];
            // K step index of (dz, pi): dz * 5 + pi;  sets: dz = 0 -> S0, 1 -> S1, 2 -> S2
#define D3B_ALLDZ(RYV, PIV, XH, XL)                                                   \
            mm(D3IC<S0>{}, D3IC<RYV>{}, D3IC<0 * 5 + PIV>{}, XH, XL);                 \
            mm(D3IC<S1>{}, D3IC<RYV>{}, D3IC<1 * 5 + PIV>{}, XH, XL);                 \
            mm(D3IC<S2>{}, D3IC<RYV>{}, D3IC<2 * 5 + PIV>{}, XH, XL);
            // (dx 0 | dx 1) of input row r: output rows r - dy, K step pi = dy
            if constexpr (r <= 7) { D3B_ALLDZ((r <= 7 ? r : 0), 0, ph, pl) }
            if constexpr (r >= 1 && r <= 8) { D3B_ALLDZ((r >= 1 && r <= 8 ? r - 1 : 0), 1, ph, pl) }
            if constexpr (r >= 2) { D3B_ALLDZ((r >= 2 ? r - 2 : 0), 2, ph, pl) }
            // (dy 0 | dy 1) at dx 2: output row r (pi = 3);  (dy 2 | -) at dx 2: output row r - 2 (pi = 4)
            if constexpr (r <= 7) { D3B_ALLDZ((r <= 7 ? r : 0), 3, qh, ql) }
            if constexpr (r >= 2) { D3B_ALLDZ((r >= 2 ? r - 2 : 0), 4, qh, ql) }
#undef D3B_ALLDZ
#pragma unroll
            for (int m = 0; m < 9 * ((r == 0 || r == 9) ? 2 : ((r == 1 || r == 8) ? 3 : 5)); ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(D3_FILL_MASK, D3_PIPE, 0);
            }
        };
        block(D3IC<0>{}); block(D3IC<1>{}); block(D3IC<2>{}); block(D3IC<3>{}); block(D3IC<4>{});
        block(D3IC<5>{}); block(D3IC<6>{}); block(D3IC<7>{}); block(D3IC<8>{}); block(D3IC<9>{});
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 k0 = okeep[0], k1 = okeep[1];
        asm volatile("" :: "v"(k0), "v"(k1));
    };

    if (npw > 0) {
        {
            fetch(0);
            stage_unit(D3IC<0>{}, 0, a.scale, a.scale11); stage_unit(D3IC<1>{}, 0, a.scale, a.scale11);
            stage_unit(D3IC<2>{}, 0, a.scale, a.scale11); stage_unit(D3IC<3>{}, 0, a.scale, a.scale11);
            __builtin_amdgcn_sched_barrier(0);
            fetch(1);
        }
        for (int i = 0; i < npw; ++i) {
            for (int k = 0; k < 5; ++k) {
                step(D3IC<0>{}, i, 3 * k);
                step(D3IC<1>{}, i, 3 * k + 1);
                step(D3IC<2>{}, i, 3 * k + 2);
            }
            step(D3IC<0>{}, i, 15);
            // behind the last input plane: row 7 of plane 14 (set 2), plane 15 (set 0) - cleared for plane 0 of the next patch
            __builtin_amdgcn_sched_barrier(0);
            const int p = patch_of(i);
            epi_row(D3IC<2>{}, D3IC<7>{}, p, 14, true);
            epi_row(D3IC<0>{}, D3IC<0>{}, p, 15, true); epi_row(D3IC<0>{}, D3IC<1>{}, p, 15, true);
            epi_row(D3IC<0>{}, D3IC<2>{}, p, 15, true); epi_row(D3IC<0>{}, D3IC<3>{}, p, 15, true);
            epi_row(D3IC<0>{}, D3IC<4>{}, p, 15, true); epi_row(D3IC<0>{}, D3IC<5>{}, p, 15, true);
            epi_row(D3IC<0>{}, D3IC<6>{}, p, 15, true); epi_row(D3IC<0>{}, D3IC<7>{}, p, 15, true);
#pragma unroll
            for (int ry = 0; ry < 8; ++ry) { acc[0][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; acx[0][ry] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7" :: "v"(okeep[0]), "v"(okeep[1]) : "memory");
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------- host
int d3d_build(const View &in, const View &out, const int k[3], const int lo[3], const int s[3], D3dPlan *plan) {
    plan->ok = false;
    if (getenv("ALQ_NO_D3D")) return ALQ_OK;
    if (!(k[0] == 3 && k[1] == 3 && k[2] == 3 && s[0] == 1 && s[1] == 1 && s[2] == 1 && lo[0] == 1 && lo[1] == 1 && lo[2] == 1)) return ALQ_OK;
    if (!(in.D == 16 && in.H == 16 && in.W == 16 && out.D == 16 && out.H == 16 && out.W == 16 && in.C == 32 && in.split == 16 && in.cs == 16 && in.c0 == 0 &&
          out.C == 16 && out.cs == 16 && out.c0 == 0 && out.split == 0)) return ALQ_OK;
    plan->flops_per_patch = 2.0 * 27 * 32 * 16 * 4096.0;
    plan->ok = true;
    return ALQ_OK;
}

// W: TF conv filter [tap = (tz * 3 + ty) * 3 + tx][ci (32)][co (16)].  Fragment of a tap: lane -> row co = lane & 15, k-group kg = lane >> 4: ci = 8 kg + c.
void d3d_pack(D3dPlan *plan, const float *W) {
    float amax = 0.f;
    for (size_t i = 0; i < (size_t)27 * 32 * 16; ++i) amax = std::max(amax, std::fabs(W[i]));
    int ex = 0;
    if (amax > 0.f) (void)std::frexp(amax, &ex);
    plan->w_exp = 14 - ex;
    plan->h_Whi.assign((size_t)27 * 64 * 8, 0);
    plan->h_Wlo.assign((size_t)27 * 64 * 8, 0);
    for (int tap = 0; tap < 27; ++tap)
        for (int lane = 0; lane < 64; ++lane) {
            const int co = lane & 15, kg = lane >> 4;
            for (int c = 0; c < 8; ++c) {
                const float w = W[((size_t)tap * 32 + 8 * kg + c) * 16 + co];
                const float ws = std::ldexp(w, plan->w_exp);
                const _Float16 h = (_Float16)ws;
                const _Float16 l = (_Float16)std::ldexp(ws - (float)h, 11);
                unsigned short hb, lb;
                std::memcpy(&hb, &h, 2);
                std::memcpy(&lb, &l, 2);
                const size_t o = ((size_t)tap * 64 + lane) * 8 + c;
                plan->h_Whi[o] = hb;
                plan->h_Wlo[o] = lb;
            }
        }
}

// Backward weights from the TF conv filter W [tap = (tz * 3 + ty) * 3 + tx][ci (32)][co (16)]: fragment (row block cb, dz, K step pi), lane -> row
// ci = 16 cb + (lane & 15), k-group kg = lane >> 4: window position half h = kg >> 1, co = 8 (kg & 1) + c; the position (dy, dx) of (pi, h): pi < 3: (pi, h);
// pi = 3: (h, 2); pi = 4: (2, 2) for h = 0, none for h = 1.  din[u] = sum_d dout[u + d - 1] W[2 - d]: tap = 2 - position per dimension.
void d3d_bwd_pack(D3dPlan *plan, const float *W) {
    plan->h_Bhi.assign((size_t)2 * 15 * 64 * 8, 0);
    plan->h_Blo.assign((size_t)2 * 15 * 64 * 8, 0);
    for (int cb = 0; cb < 2; ++cb)
        for (int dz = 0; dz < 3; ++dz)
            for (int pi = 0; pi < 5; ++pi)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 15, kg = lane >> 4, h = kg >> 1;
                    int dy, dx;
                    bool any = true;
                    if (pi < 3) { dy = pi; dx = h; } else if (pi == 3) { dy = h; dx = 2; } else { dy = 2; dx = 2; any = h == 0; }
                    for (int c = 0; c < 8; ++c) {
                        const int co = 8 * (kg & 1) + c, ci = 16 * cb + r;
                        const int tap = ((2 - dz) * 3 + (2 - dy)) * 3 + (2 - dx);
                        const float w = any ? W[((size_t)tap * 32 + ci) * 16 + co] : 0.f;
                        const float ws = std::ldexp(w, plan->w_exp);
                        const _Float16 hh = (_Float16)ws;
                        const _Float16 ll = (_Float16)std::ldexp(ws - (float)hh, 11);
                        unsigned short hb, lb;
                        std::memcpy(&hb, &hh, 2);
                        std::memcpy(&lb, &ll, 2);
                        const size_t o = ((size_t)((cb * 3 + dz) * 5 + pi) * 64 + lane) * 8 + c;
                        plan->h_Bhi[o] = hb;
                        plan->h_Blo[o] = lb;
                    }
                }
}

int d3d_bwd_launch(alq_ctx *ctx, const D3dPlan &plan, int N, const float *dout, float in_bound, float *dinA, float *dinB, float *sumB) {
    ALQ_REQUIRE(plan.ok && plan.d_Bhi && plan.d_Blo, ALQ_EINVAL, "d3d: backward weights not set");
    ALQ_REQUIRE(dout && dinA && dinB && in_bound > 0.f, ALQ_EINVAL, "d3d: missing argument");
    ALQ_REQUIRE(N < 4096, ALQ_EUNSUPPORTED, "d3d: 32-bit byte offsets hold fewer than 4096 patches per pass");
    if (N <= 0) return ALQ_OK;
    int ex = 0;
    (void)std::frexp(in_bound, &ex);
    const int e_in = 14 - ex;
    D3BArgs a;
    a.dout = dout; a.Whi = reinterpret_cast<const unsigned short *>(plan.d_Bhi); a.Wlo = reinterpret_cast<const unsigned short *>(plan.d_Blo);
    a.dinA = dinA; a.dinB = dinB; a.sumB = sumB;
    a.scale = std::ldexp(1.f, e_in); a.scale11 = std::ldexp(1.f, e_in + 11); a.inv = std::ldexp(1.f, -(e_in + plan.w_exp)); a.N = N;
    const int cus = ctx->num_cus;
    long long g = std::min<long long>((long long)cus, (long long)N);
    g = std::max<long long>(8, (g + 7) / 8 * 8);
    const size_t lds = DB_STRIP;
    ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(d3d_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope ps(ctx, PROF_IGEMM_F16, plan.flops_per_patch * N);
    hipLaunchKernelGGL(d3d_bwd_kernel, dim3((unsigned)g), dim3(256), lds, ctx->stream, a);
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

int d3d_fwd_launch(alq_ctx *ctx, const D3dPlan &plan, int N, const float *inA, const float *inB, const unsigned *amaxA, const unsigned *amaxB,
                   const float *bias, int relu, float *out, unsigned char *sg, float *osum) {
    ALQ_REQUIRE(plan.ok && plan.d_Whi && plan.d_Wlo, ALQ_EINVAL, "d3d: weights not set");
    ALQ_REQUIRE(inA && inB && amaxA && amaxB && bias && out, ALQ_EINVAL, "d3d: missing argument");
    ALQ_REQUIRE(N < 4096, ALQ_EUNSUPPORTED, "d3d: 32-bit byte offsets hold fewer than 4096 patches per pass");
    if (N <= 0) return ALQ_OK;
    D3Args a;
    a.inA = inA; a.inB = inB; a.amaxA = amaxA; a.amaxB = amaxB;
    a.Whi = reinterpret_cast<const unsigned short *>(plan.d_Whi); a.Wlo = reinterpret_cast<const unsigned short *>(plan.d_Wlo);
    a.bias = bias; a.out = out; a.sg = sg; a.osum = osum; a.e_w = plan.w_exp; a.relu = relu; a.N = N;
    const int cus = ctx->num_cus;
    long long g = std::min<long long>((long long)cus, (long long)N);
    g = std::max<long long>(8, (g + 7) / 8 * 8);
    const size_t lds = D3_STRIP;
    ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(d3d_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope ps(ctx, PROF_IGEMM_F16, plan.flops_per_patch * N);
    hipLaunchKernelGGL(d3d_fwd_kernel, dim3((unsigned)g), dim3(256), lds, ctx->stream, a);
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

}  // namespace alq
