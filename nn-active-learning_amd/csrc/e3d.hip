// Backward of NET-C's `enc2` (3x3x3 conv 8 -> 16 channels at 16^3, NN_extended.py:416-426) in a Fisher pass, fused with the two
// max-pool backward steps on either side of it.  Until round 5 that stretch of the pass was three launches and two tensors that
// only they touched:
//     pool2 backward (scatter the pooled cotangent to the arg-max voxels, add the skip cotangent, ReLU mask, channel sums)
//  -> enc2 backward-data (two-slot engine)  ->  pool1 backward (arg-max scatter, ReLU mask, channel sums of enc1's cotangent)
// 0.72 ms per 2047 patches, of which 0.47 ms in the two HBM round trips of the masked 16-channel cotangent (256 KB per patch,
// written and read) and of the 8-channel result (128 KB, written and read).  Here one kernel reads the skip cotangent, the pooled
// cotangent, the two arg-max fields and the two sign fields, and writes only the two channel-sum fields the box-filter dot
// products need (enc2's: 16 KB per patch, enc1's: += 128 KB).
//
// Machine mapping (the row sweep of t3d.hip, with the staged rows SHARED by the four waves of a workgroup):
//   * a workgroup owns four consecutive z planes of one patch (wave w: plane 4 q + w) and sweeps y in steps of two rows; a step
//     STAGES rows 2 t - 1, 2 t of the six planes 4 q - 1 .. 4 q + 4 - three rows per wave: scatter + skip + mask, channel sums for
//     the workgroup's own planes, fp16-pair split at the static scale of the cotangent bound - into an LDS ring of six row slots
//     per plane, then (one barrier) every wave contracts its tile y0 = 2 t - 2 (rows y0, y0 + 1: 32 voxels x 8 channels = one
//     16 x 16 MFMA tile in pair form: rows = (x parity, 8 input channels), columns = (row, x pair), K step = two window positions
//     x 16 output channels; 9 (dz, dy) x 2 K steps x 3 products).  Six slots: the rows of step t + 1 land in slots no tile of step
//     t reads, so ONE barrier per step is enough.  A first version let every wave stage the three planes it needs itself: 460
//     vector instructions per 54 MFMAs, slower than the launches it replaced.
//   * fp16 pairs x 2^e = h + l 2^-11 (lo pieces scaled up, their two products in a second accumulator: the static bound is loose
//     down here - a product of L1 norms over four layers - and at their true scale the lo pieces of typical values were fp16
//     subnormals with a handful of bits); the hi pieces of the weights live in registers (72), the lo pieces in LDS;
//   * the epilogue routes the 4 channels a lane holds to the eight positions of their pool1 windows (arg-max byte, pooled sign),
//     swaps half of the partial sums with the lane holding the other 4 channels and adds two 8-byte pieces per lane into enc1's
//     channel-sum field (c3d_bwd wrote the skip part of it earlier on the same stream);
//   * rows outside the volume are fetched from the nearest row inside and staged with scale 0 - no per-lane predication.
// Loads run two steps ahead (one for the small, cache-resident pooled tensors); loop body straight-line, first pass void: see
// t3d_fwd_kernel for why.
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

struct E3Args {
    const float *skip;            // [N][16^3][16] cotangent of enc2's output from its skip consumer (dense)
    const float *dpool;           // [N][8^3][16] cotangent of pool2's output (dense)
    const unsigned char *am2;     // [N][8^3][16] arg-max window index (dz * 2 + dy) * 2 + dx of pool2
    const unsigned char *sg2;     // enc2's output sign field: byte (voxel * 16 + c) / 4, bit c & 3
    const unsigned short *Whi;    // [9 (dz, dy)][2 K steps][64 lanes][8] fp16 bits: hi pieces (e3d_pack)
    const unsigned short *Wlo;    // the lo pieces (x 2^11), same layout
    const unsigned char *am1;     // [N][16^3][8] arg-max of pool1
    const unsigned char *sg1;     // pool1's output sign field: byte (voxel * 8 + c) / 4
    float *dsum2;                 // [N][16^3] channel sums of enc2's masked cotangent (out)
    float *dsum1;                 // [N][32^3] channel sums of enc1's cotangent (+=)
    float scale, inv;             // 2^e_in, 2^-(e_in + e_w)
    int N;
};

constexpr unsigned E3_OOB = 0xffffff00u;
constexpr int E3_ROWB = 18 * 32 + 16;         // 18 voxel slots (x = -1 .. 16) x 16 channels x 2 B, + 16: a row slot (2 pieces) is 32 mod 64 bytes, so the two tile rows of a fragment read land on disjoint banks
constexpr int E3_SLOT = 2 * E3_ROWB;          // pieces h, l
constexpr int E3_RING = 6;                    // row slots per plane
constexpr int E3_PLANE = E3_RING * E3_SLOT;
constexpr int E3_STRIP = 6 * E3_PLANE;        // planes 4 q - 1 .. 4 q + 4
constexpr int E3_WLO = 9 * 2 * 1024;          // lo weight fragments
constexpr int E3_SINK = E3_SLOT;              // behind the ring: where the void steps' rows go (see the staging)

__device__ inline __amdgpu_buffer_rsrc_t e3_rsrc(const void *base, unsigned long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(unsigned)bytes, 0x00020000);
}

// a scalar the compiler must keep scalar: with ~100 live SGPRs it moved some row offsets to vector registers and wrapped every
// buffer access that used them in a waterfall loop
__device__ inline int e3_s(unsigned v) { return __builtin_amdgcn_readfirstlane((int)v); }

struct E3RowA { f32x4 sk; unsigned sg; };      // what staging one row needs, per lane: skip cotangent + enc2's sign byte,
struct E3RowB { f32x4 dp; unsigned am; };      // pooled cotangent + pool2's arg-max bytes

__global__ __launch_bounds__(256, 2) void e3d_bwd_kernel(const E3Args a) {
    extern __shared__ __attribute__((aligned(16))) char e3lds[];
    {   // lo weight fragments into LDS, once
        const i32x4 *src = reinterpret_cast<const i32x4 *>(a.Wlo);
        i32x4 *dst = reinterpret_cast<i32x4 *>(e3lds);
        for (int i = threadIdx.x; i < E3_WLO / 16; i += 256) dst[i] = src[i];
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char *strip = e3lds + E3_WLO;
    // the zero slots (x = -1: slot 0, x = 16: slot 17) of every row and piece, once: 72 piece rows x 2 slots x 32 B
    for (int i = threadIdx.x; i < 6 * E3_RING * 2 * 2 * 2; i += 256) {
        const int row = i >> 2, s = (i >> 1) & 1, half = i & 1;
        *reinterpret_cast<i32x4 *>(strip + row * E3_ROWB + (s ? 17 : 0) * 32 + half * 16) = i32x4{0, 0, 0, 0};
    }
    __syncthreads();
    const int n = lane & 15, kg = lane >> 4, ry = n >> 3, j = n & 7;
    f16x8 wh[9][2];
#pragma unroll
    for (int c = 0; c < 9; ++c)
#pragma unroll
        for (int s = 0; s < 2; ++s) wh[c][s] = *reinterpret_cast<const f16x8 *>(a.Whi + ((size_t)(c * 2 + s) * 64 + lane) * 8);
#pragma unroll
    for (int c = 0; c < 9; ++c)
#pragma unroll
        for (int s = 0; s < 2; ++s) asm volatile("" : "+v"(wh[c][s]));      // arrived before the loop (t3d_fwd_kernel)
    const char *wl = e3lds + lane * 16;

    const __amdgpu_buffer_rsrc_t sk_rsrc = e3_rsrc(a.skip, (unsigned long long)a.N * 4096 * 64);
    const __amdgpu_buffer_rsrc_t dp_rsrc = e3_rsrc(a.dpool, (unsigned long long)a.N * 512 * 64);
    const __amdgpu_buffer_rsrc_t a2_rsrc = e3_rsrc(a.am2, (unsigned long long)a.N * 512 * 16);
    const __amdgpu_buffer_rsrc_t s2_rsrc = e3_rsrc(a.sg2, (unsigned long long)a.N * 4096 * 4);
    const __amdgpu_buffer_rsrc_t a1_rsrc = e3_rsrc(a.am1, (unsigned long long)a.N * 4096 * 8);
    const __amdgpu_buffer_rsrc_t s1_rsrc = e3_rsrc(a.sg1, (unsigned long long)a.N * 4096 * 2);
    const __amdgpu_buffer_rsrc_t d2_rsrc = e3_rsrc(a.dsum2, (unsigned long long)a.N * 4096 * 4);
    const __amdgpu_buffer_rsrc_t d1_rsrc = e3_rsrc(a.dsum1, (unsigned long long)a.N * 32768 * 4);

    // staging lane roles: voxel x = lane >> 2 of a row, channels 4 cq .. + 3
    const int sx = lane >> 2, cq = lane & 3;
    const int w_off = (sx + 1) * 32 + cq * 8;
    const unsigned ldA = (unsigned)lane * 16u, ldS = (unsigned)lane;
    const unsigned ldP = (unsigned)(sx >> 1) * 64u + (unsigned)cq * 16u, ldM = (unsigned)(sx >> 1) * 16u + (unsigned)cq * 4u;
    const unsigned lanepar4 = (unsigned)(sx & 1) * 0x01010101u;
    const unsigned st2 = cq == 0 ? (unsigned)sx * 4u : E3_OOB;
    // fragment lane roles: column (ry, j), K step s: window position q = 2 s + (kg >> 1) -> voxel x = 2 j - 1 + q = slot 2 j + q, channel half kg & 1
    const int f_off = (2 * j + (kg >> 1)) * 32 + (kg & 1) * 16 + wave * E3_PLANE;      // (+ the wave's first plane of the six)
    // epilogue lane roles: pooled voxel (z, y0 + ry, x = 2 j + (kg >> 1)) of pool1, channels 4 (kg & 1) .. + 3; window sums shared with lane ^ 16
    const bool hi = (kg & 1) != 0;      // this lane adds into plane 2 z + 1 of enc1's grid (window positions 4 .. 7), else 2 z
    const unsigned e_am = (unsigned)(ry * 16 + 2 * j + (kg >> 1)) * 8u + (unsigned)(kg & 1) * 4u;
    const unsigned e_sg = (unsigned)(ry * 16 + 2 * j + (kg >> 1)) * 2u + (unsigned)(kg & 1);
    const unsigned e_d1 = (unsigned)(hi ? 32 * 128 : 0) + (unsigned)(2 * ry) * 128u + (unsigned)(2 * (2 * j + (kg >> 1))) * 4u;      // + dy2 * 128

    // unit order as in t3d.hip: workgroup b of XCD b % 8 takes units v = b / 8, b / 8 + G / 8, ...: patch 8 (v >> 2) + b % 8, planes 4 (v & 3) + wave
    const int G8 = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, jb = (int)blockIdx.x >> 3;
    const int np = a.N > xcd ? (a.N - xcd + 7) >> 3 : 0;
    const int nv = 4 * np;
    const int nunits = nv > jb ? (nv - jb + G8 - 1) / G8 : 0;
    const int total = nunits * 9;       // nine steps per unit: t = 0 .. 8 stage rows 2 t - 1, 2 t and contract tile y0 = 2 t - 2 (t = 0: nothing to contract)

    // step S (clamped into the stream) -> patch, first plane of the workgroup, step t of the unit
    auto coords = [&](int S, int *p, int *z0, int *t) __attribute__((always_inline)) {
        const int Sc = S < total ? S : total - 1;
        const int u = Sc / 9;
        const int v = jb + u * G8;
        *p = 8 * (v >> 2) + xcd;
        *z0 = 4 * (v & 3);
        *t = Sc - u * 9;
    };
    // the three rows this wave stages in step S: list index i = 3 wave + r -> plane pi = i >> 1 of the six (zz = z0 - 1 + pi), row yy = 2 t - 1 + (i & 1);
    // clamped into the volume (the staged values of a row outside are multiplied by 0)
    auto rowof = [&](int S, int r, int *p, int *zz, int *yy, int *t) __attribute__((always_inline)) {
        int z0;
        coords(S, p, &z0, t);
        const int i = 3 * wave + r;
        *zz = z0 - 1 + (i >> 1);
        *yy = 2 * *t - 1 + (i & 1);
    };
    auto clampi = [](int v) { return v < 0 ? 0 : (v > 15 ? 15 : v); };
    auto fetchA = [&](int S, E3RowA *R) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            int p, zz, yy, t;
            rowof(S, r, &p, &zz, &yy, &t);
            const unsigned row = ((unsigned)p * 16u + (unsigned)clampi(zz)) * 16u + (unsigned)clampi(yy);             // row index of the 16^3 grid
            R[r].sk = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sk_rsrc, (int)ldA, e3_s(row * 1024u), 0));
            R[r].sg = (unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(s2_rsrc, (int)ldS, e3_s(row * 64u), 0);
        }
    };
    auto fetchB = [&](int S, E3RowB *R) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            int p, zz, yy, t;
            rowof(S, r, &p, &zz, &yy, &t);
            const unsigned prow = ((unsigned)p * 8u + (unsigned)(clampi(zz) >> 1)) * 8u + (unsigned)(clampi(yy) >> 1);   // row index of the 8^3 grid
            R[r].dp = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dp_rsrc, (int)ldP, e3_s(prow * 512u), 0));
            R[r].am = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(a2_rsrc, (int)ldM, e3_s(prow * 128u), 0);
        }
    };

    E3RowA RA[2][3];
    E3RowB RB[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        RB[i].dp = f32x4{0.f, 0.f, 0.f, 0.f}; RB[i].am = 0u;
#pragma unroll
        for (int b = 0; b < 2; ++b) { RA[b][i].sk = f32x4{0.f, 0.f, 0.f, 0.f}; RA[b][i].sg = 0u; }
    }

    if (total > 0)
    for (int S0 = -2; S0 < total; S0 += 2) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            __builtin_amdgcn_sched_barrier(0);
            const int S = S0 + b;
            const bool live = S0 >= 0 && S < total;      // (the stream may end on the first step of a pass: the second one replays it without stores)
            int p, z0, t;
            coords(S0 >= 0 ? S : 0, &p, &z0, &t);
            const int z = z0 + wave;
            // what the epilogue of this step's tile (y0 = 2 t - 2) needs, requested now: pool1's arg-max word and sign byte of the lane's
            // pooled voxel and the two 8-byte pieces of enc1's field it adds to (no other wave touches them meanwhile)
            const int y0 = 2 * t - 2;
            const bool tv = live && t >= 1;
            const unsigned trow = ((unsigned)p * 16u + (unsigned)z) * 16u + (unsigned)(tv ? y0 : 0);          // first row of the tile in pool1's grid
            const unsigned am1w = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(a1_rsrc, (int)e_am, e3_s(trow * 128u), 0);
            const unsigned sg1b = (unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(s1_rsrc, (int)e_sg, e3_s(trow * 32u), 0);
            const unsigned d1row = (unsigned)e3_s((((unsigned)p * 32u + (unsigned)(2 * z)) * 32u + (unsigned)(tv ? 2 * y0 : 0)) * 128u);     // row (2 z, 2 y0) of enc1's grid, bytes
            const f32x2 old0 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(d1_rsrc, (int)e_d1, (int)d1row, 0));
            const f32x2 old1 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(d1_rsrc, (int)(e_d1 + 128u), (int)d1row, 0));
            // ---- stage this wave's three rows into ring slot (yy + 6) % 6 of their planes
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const E3RowA &ra = RA[b][r];
                const E3RowB &rb = RB[r];
                const int i = 3 * wave + r, pi = i >> 1;
                const int zz = z0 - 1 + pi, yy = 2 * t - 1 + (i & 1);
                const bool rv = live && zz >= 0 && zz < 16 && yy >= 0 && yy < 16;
                const float scr = rv ? a.scale : 0.f;
                const unsigned x4 = rb.am ^ (lanepar4 ^ ((unsigned)((((zz & 1) * 2 + (yy & 1)) * 2)) * 0x01010101u));      // byte c is 0 where channel c's arg-max is this voxel
                float g0 = ra.sk.x + (((x4 & 0xffu) == 0u) ? rb.dp.x : 0.f);
                float g1 = ra.sk.y + (((x4 & 0xff00u) == 0u) ? rb.dp.y : 0.f);
                float g2 = ra.sk.z + (((x4 & 0xff0000u) == 0u) ? rb.dp.z : 0.f);
                float g3 = ra.sk.w + (((x4 & 0xff000000u) == 0u) ? rb.dp.w : 0.f);
                // ReLU mask from the sign nibble: sign-extended bit & value (no compare / select)
                unsigned k0 = (unsigned)__builtin_amdgcn_sbfe((int)ra.sg, 0, 1), k1 = (unsigned)__builtin_amdgcn_sbfe((int)ra.sg, 1, 1);
                unsigned k2 = (unsigned)__builtin_amdgcn_sbfe((int)ra.sg, 2, 1), k3 = (unsigned)__builtin_amdgcn_sbfe((int)ra.sg, 3, 1);
                asm("" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3));
                g0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, g0) & k0); g1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, g1) & k1);
                g2 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, g2) & k2); g3 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, g3) & k3);
                {   // a plane the workgroup owns (pi = 1 .. 4): channel sums of the masked cotangent (four lanes per voxel)
                    float s_ = (g0 + g1) + (g2 + g3);
                    s_ += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s_), 0xB1, 0xf, 0xf, true));      // quad_perm [1, 0, 3, 2]
                    s_ += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s_), 0x4E, 0xf, 0xf, true));      // quad_perm [2, 3, 0, 1]
                    const bool own = rv && pi >= 1 && pi <= 4;
                    const unsigned row = ((unsigned)p * 16u + (unsigned)clampi(zz)) * 16u + (unsigned)clampi(yy);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, s_), d2_rsrc, (int)(own ? st2 : E3_OOB), e3_s(row * 64u), 0);
                }
                const float x0 = g0 * scr, x1 = g1 * scr, x2 = g2 * scr, x3 = g3 * scr;
                const f16x2 h01 = __builtin_convertvector(f32x2{x0, x1}, f16x2), h23 = __builtin_convertvector(f32x2{x2, x3}, f16x2);
                const float sc11 = scr * 2048.f;
                const f16x2 l01 = __builtin_convertvector(f32x2{__builtin_fmaf((float)h01.x, -2048.f, g0 * sc11), __builtin_fmaf((float)h01.y, -2048.f, g1 * sc11)}, f16x2);
                const f16x2 l23 = __builtin_convertvector(f32x2{__builtin_fmaf((float)h23.x, -2048.f, g2 * sc11), __builtin_fmaf((float)h23.y, -2048.f, g3 * sc11)}, f16x2);
                // a void step stages into the sink behind the ring, never into the ring: the step that replays the stream's last one has
                // that step's slots, which slower waves of the workgroup may still be contracting (no barrier separates them)
                char *dst = strip + (live ? pi * E3_PLANE + ((yy + 6) % 6) * E3_SLOT : E3_STRIP) + w_off;
                *reinterpret_cast<i32x2 *>(dst) = i32x2{__builtin_bit_cast(int, h01), __builtin_bit_cast(int, h23)};
                *reinterpret_cast<i32x2 *>(dst + E3_ROWB) = i32x2{__builtin_bit_cast(int, l01), __builtin_bit_cast(int, l23)};
            }
            fetchB(S + 1, RB);      // (first: the wait for these, one step from now, must not cover the two-step requests behind them - vmcnt retires in order)
            fetchA(S + 2, RA[b]);
            // every wave's rows of this step are in the ring before any tile of the step reads them; the six slots make this the only barrier
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // ---- contract tile y0 = 2 t - 2 (rows y0 + ry): input rows y0 + ry + dy - 1, dy = 0 .. 2 -> ring slot (2 t - 3 + ry + dy + 6) % 6
            f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f}, cl = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int s0 = ((2 * t + 3 + dy) % 6) * E3_SLOT, s1 = ((2 * t + 4 + dy) % 6) * E3_SLOT;
                const int rowo = (ry ? s1 : s0) + f_off;
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    const char *src = strip + dz * E3_PLANE + rowo;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const f16x8 xh = *reinterpret_cast<const f16x8 *>(src + s * 64);
                        const f16x8 xl = *reinterpret_cast<const f16x8 *>(src + s * 64 + E3_ROWB);
                        const f16x8 wlo = *reinterpret_cast<const f16x8 *>(wl + ((dz * 3 + dy) * 2 + s) * 1024);
                        const f16x8 whi = wh[dz * 3 + dy][s];
                        cl = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, xh, cl, 0, 0, 0);      // (l, h) + (h, l) at 2^11, (h, h)
                        cl = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, xl, cl, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, xh, c, 0, 0, 0);
                    }
                }
            }
            // ---- epilogue: this lane holds channels 4 (kg & 1) .. + 3 of pooled voxel (z, y0 + ry, x = 2 j + (kg >> 1)) of pool1's output
            const float r0 = __builtin_fmaf(cl.x, 0x1p-11f, c.x) * a.inv, r1 = __builtin_fmaf(cl.y, 0x1p-11f, c.y) * a.inv;
            const float r2 = __builtin_fmaf(cl.z, 0x1p-11f, c.z) * a.inv, r3 = __builtin_fmaf(cl.w, 0x1p-11f, c.w) * a.inv;
            unsigned m0 = (unsigned)__builtin_amdgcn_sbfe((int)sg1b, 0, 1), m1 = (unsigned)__builtin_amdgcn_sbfe((int)sg1b, 1, 1);
            unsigned m2 = (unsigned)__builtin_amdgcn_sbfe((int)sg1b, 2, 1), m3 = (unsigned)__builtin_amdgcn_sbfe((int)sg1b, 3, 1);
            asm("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3));
            const float v0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r0) & m0), v1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & m1);
            const float v2 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r2) & m2), v3 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r3) & m3);
            // the eight window sums of this lane's 4 channels by compare / select (LDS float adds at the arg-max positions were tried:
            // four ds_add_f32 per step cost 750 LDS cycles, three times everything else the step does in LDS), then half of them
            // swapped with the lane holding the other 4 channels (lane ^ 16): lanes kg & 1 = 0 keep window positions 0 .. 3, the others 4 .. 7
            float sw[8];
#pragma unroll
            for (int w = 0; w < 8; ++w)
                sw[w] = ((((am1w & 255u) == (unsigned)w) ? v0 : 0.f) + ((((am1w >> 8) & 255u) == (unsigned)w) ? v1 : 0.f)) +
                        (((((am1w >> 16) & 255u) == (unsigned)w) ? v2 : 0.f) + (((am1w >> 24) == (unsigned)w) ? v3 : 0.f));
            f32x4 mine;
            {
                const float g0 = __shfl_xor(hi ? sw[0] : sw[4], 16, 64), g1 = __shfl_xor(hi ? sw[1] : sw[5], 16, 64);
                const float g2 = __shfl_xor(hi ? sw[2] : sw[6], 16, 64), g3 = __shfl_xor(hi ? sw[3] : sw[7], 16, 64);
                mine = f32x4{(hi ? sw[4] : sw[0]) + g0, (hi ? sw[5] : sw[1]) + g1, (hi ? sw[6] : sw[2]) + g2, (hi ? sw[7] : sw[3]) + g3};
            }
            // enc1's grid: voxel (2 z + dz, 2 (y0 + ry) + dy, 2 x + dx); the lane's four sums = (dy, dx) of its dz
            const f32x2 n0 = f32x2{old0.x + mine.x, old0.y + mine.y}, n1 = f32x2{old1.x + mine.z, old1.y + mine.w};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, n0), d1_rsrc, (int)(tv ? e_d1 : E3_OOB), (int)d1row, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, n1), d1_rsrc, (int)(tv ? e_d1 + 128u : E3_OOB), (int)d1row, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------- host
int e3d_build(const View &in, const View &out, const int k[3], const int lo[3], const int s[3], E3dPlan *plan) {
    plan->ok = false;
    if (getenv("ALQ_NO_E3D")) return ALQ_OK;
    if (!(k[0] == 3 && k[1] == 3 && k[2] == 3 && s[0] == 1 && s[1] == 1 && s[2] == 1 && lo[0] == 1 && lo[1] == 1 && lo[2] == 1)) return ALQ_OK;
    if (!(in.D == 16 && in.H == 16 && in.W == 16 && out.D == 16 && out.H == 16 && out.W == 16 && in.C == 8 && out.C == 16 && in.split == 0 && out.split == 0)) return ALQ_OK;
    plan->flops_per_patch = 2.0 * 27 * 8 * 16 * 4096.0;
    plan->ok = true;
    return ALQ_OK;
}

// W: TF conv filter [tap = (tz * 3 + ty) * 3 + tx][ci (8)][co (16)].  A fragment of (dz, dy, K step s): lane -> row r = lane & 15 (x parity px = r >> 3,
// ci = r & 7), k-group kg = lane >> 4: window position q = 2 s + (kg >> 1), co = 8 (kg & 1) + c.  The input row of (dz, dy) lies at
// offset (dz - 1, dy - 1) from the output row, the window position q at x offset q - 1 - px: tap = (1 - offset) per dimension.
void e3d_pack(E3dPlan *plan, const float *W) {
    float amax = 0.f;
    for (size_t i = 0; i < (size_t)27 * 8 * 16; ++i) amax = std::max(amax, std::fabs(W[i]));
    int ex = 0;
    if (amax > 0.f) (void)std::frexp(amax, &ex);
    plan->w_exp = 14 - ex;
    plan->h_Whi.assign((size_t)9 * 2 * 64 * 8, 0);
    plan->h_Wlo.assign((size_t)9 * 2 * 64 * 8, 0);
    for (int dz = 0; dz < 3; ++dz)
        for (int dy = 0; dy < 3; ++dy)
            for (int s = 0; s < 2; ++s)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 15, kg = lane >> 4, px = r >> 3, ci = r & 7, q = 2 * s + (kg >> 1);
                    const int tz = 2 - dz, ty = 2 - dy, tx = 2 - q + px;
                    for (int c = 0; c < 8; ++c) {
                        const int co = 8 * (kg & 1) + c;
                        const float w = (tx >= 0 && tx <= 2) ? W[((size_t)((tz * 3 + ty) * 3 + tx) * 8 + ci) * 16 + co] : 0.f;
                        const float ws = std::ldexp(w, plan->w_exp);
                        const _Float16 h = (_Float16)ws;
                        const _Float16 l = (_Float16)std::ldexp(ws - (float)h, 11);
                        unsigned short hb, lb;
                        std::memcpy(&hb, &h, 2);
                        std::memcpy(&lb, &l, 2);
                        const size_t o = ((size_t)((dz * 3 + dy) * 2 + s) * 64 + lane) * 8 + c;
                        plan->h_Whi[o] = hb;
                        plan->h_Wlo[o] = lb;
                    }
                }
}

int e3d_bwd_launch(alq_ctx *ctx, const E3dPlan &plan, int N, const float *skip, const float *dpool, const unsigned char *am2, const unsigned char *sg2,
                   const unsigned char *am1, const unsigned char *sg1, float *dsum2, float *dsum1, float in_bound) {
    ALQ_REQUIRE(plan.ok && plan.d_Whi && plan.d_Wlo, ALQ_EINVAL, "e3d: weights not set");
    ALQ_REQUIRE(skip && dpool && am2 && sg2 && am1 && sg1 && dsum2 && dsum1 && in_bound > 0.f, ALQ_EINVAL, "e3d: missing argument");
    ALQ_REQUIRE(N < 4096, ALQ_EUNSUPPORTED, "e3d: 32-bit byte offsets hold fewer than 4096 patches per pass");
    if (N <= 0) return ALQ_OK;
    int ex = 0;
    (void)std::frexp(in_bound, &ex);
    const int e_in = 14 - ex;
    E3Args a;
    a.skip = skip; a.dpool = dpool; a.am2 = am2; a.sg2 = sg2; a.Whi = reinterpret_cast<const unsigned short *>(plan.d_Whi);
    a.Wlo = reinterpret_cast<const unsigned short *>(plan.d_Wlo); a.am1 = am1; a.sg1 = sg1; a.dsum2 = dsum2; a.dsum1 = dsum1;
    a.scale = std::ldexp(1.f, e_in); a.inv = std::ldexp(1.f, -(e_in + plan.w_exp)); a.N = N;
    const int cus = ctx->num_cus;
    long long g = std::min<long long>(2LL * cus, (long long)N * 4);
    g = std::max<long long>(8, (g + 7) / 8 * 8);
    const size_t lds = E3_WLO + E3_STRIP + E3_SINK;
    ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(e3d_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope ps(ctx, PROF_IGEMM_F16, plan.flops_per_patch * N);
    hipLaunchKernelGGL(e3d_bwd_kernel, dim3((unsigned)g), dim3(256), lds, ctx->stream, a);
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

}  // namespace alq
