// Backward-data pass of NET-C's `up1` (3x3x3 / stride-2 conv_transpose 32 -> 16 channels, 8^3 -> 16^3; reference call site
// NN_extended.py:574-587) in a Fisher pass: din[q] = sum_t dout[2 q + t] W[t] - a stride-2 conv of the 16-channel cotangent at 16^3
// into 32 channels at 8^3, masked by the sign field of the producer's (bott's) ReLU'd output, + the channel sums of the result.
// 12 % of a patch's bytes and 2 % of its flops; the two-slot engine ran it at 0.32 - 0.35 ms per 2047 patches, three times what
// its 0.65 GB cost at the HBM rate.  Here: two workgroups per CU, a workgroup sweeps the eight output planes of a patch; per plane
//   * the two new input planes (2 zq + 1, 2 zq + 2; plane 2 zq is the previous step's last) are converted to fp16 pairs
//     x 2^e = h + l 2^-11 under the static cotangent bound and written to an LDS ring of three planes (the loads of the next
//     step's planes are in flight meanwhile);
//   * MFMA 16 x 16 x 32 (f16): rows = 16 of the 32 input channels (wave & 1), columns = two output rows x 8 voxels (wave >> 1 and
//     + 2: two tiles per wave), K = two window positions x 16 output channels; per tile and input plane five K steps
//     (ty, tx = 0 | 1) for ty = 0 .. 2, (ty = 0 | 1, tx = 2), (ty = 2, tx = 2 | nothing): 15 K steps x 3 products; the 30 weight
//     fragments of a wave's channel block stay in registers;
//   * a lane's LDS address carries its own window position: voxel (2 qy + ty, 2 qx + tx).  The two output rows of a tile are two
//     input rows apart; rows are stored with a 16-byte shift on every other row pair so that the sixteen 16-byte reads of a
//     k-group fall on disjoint banks;
//   * epilogue: ReLU mask from the sign byte, 16-byte stores (a lane holds 4 channels of one voxel), channel sums: 16 channels in
//     the wave, the other 16 from the partner wave through LDS.
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

struct T8BwdArgs {
    const float *dout;            // [N][16^3][16] cotangent of up1's output (dense)
    float *din;                   // [N][8^3][32] cotangent of its input, masked (dense)
    const unsigned short *W;      // [2 pieces][2 channel blocks][3 tz][5 K steps][64 lanes][8] fp16 bits (t3d8_bwd_pack)
    const unsigned char *mask_bits;     // sign field of the input activation: byte (voxel * 32 + c) / 4, bit c & 3; or null
    float *dsum;                  // [N][8^3] channel sums of the (masked) result, or null
    float scale, scale11, inv;    // 2^e_in, 2^(e_in + 11), 2^-(e_in + e_w)
    int N;
};

constexpr unsigned T8B_OOB = 0xffffff00u;
constexpr int T8B_KG = 17 * 16;               // a k-group block of a row: 17 voxel slots (x = 0 .. 16) x 8 channels x 2 B
constexpr int T8B_PIECE = 2 * T8B_KG;         // one piece of a row: 544 B
constexpr int T8B_ROW = 2 * T8B_PIECE + 16;   // 1104 B: + 16 so that a shifted row ends in front of the next one
constexpr int T8B_PLANE = 18 * T8B_ROW + 32;  // rows y = 0 .. 17 (16, 17: zeros): 19,904 B
constexpr int T8B_RING = 3 * T8B_PLANE;
constexpr int T8B_XCH = 4 * 16 * 4;           // partial channel sums of the odd channel block: 4 tiles x 16 voxels
constexpr int T8B_LDS = T8B_RING + T8B_XCH;

__device__ inline __amdgpu_buffer_rsrc_t t8b_rsrc(const void *base, unsigned long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ inline int t8b_s(unsigned v) { return __builtin_amdgcn_readfirstlane((int)v); }
// byte offset of row y inside a plane image: every other row PAIR is shifted by 16 bytes
__device__ __host__ inline int t8b_row(int y) { return y * T8B_ROW + ((y >> 1) & 1) * 16; }

__global__ __launch_bounds__(256, 2) void t3d8_bwd_kernel(const T8BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char t8lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cb = wave & 1, tp = wave >> 1;      // channel block; tiles tp and tp + 2 (output rows 2 tile, 2 tile + 1)
    char *ring = t8lds;
    float *xch = reinterpret_cast<float *>(t8lds + T8B_RING);
    for (int i = threadIdx.x; i < T8B_LDS / 16; i += 256) reinterpret_cast<i32x4 *>(t8lds)[i] = i32x4{0, 0, 0, 0};
    __syncthreads();
    const int n = lane & 15, kg = lane >> 4;
    f16x8 wh[15], wl[15];
#pragma unroll
    for (int c = 0; c < 15; ++c) {
        wh[c] = *reinterpret_cast<const f16x8 *>(a.W + ((size_t)((0 * 2 + cb) * 15 + c) * 64 + lane) * 8);
        wl[c] = *reinterpret_cast<const f16x8 *>(a.W + ((size_t)((1 * 2 + cb) * 15 + c) * 64 + lane) * 8);
    }
#pragma unroll
    for (int c = 0; c < 15; ++c) asm volatile("" : "+v"(wh[c]), "+v"(wl[c]));      // arrived before the loop

    const __amdgpu_buffer_rsrc_t i_rsrc = t8b_rsrc(a.dout, (unsigned long long)a.N * 4096 * 64);
    const __amdgpu_buffer_rsrc_t o_rsrc = t8b_rsrc(a.din, (unsigned long long)a.N * 512 * 128);
    const __amdgpu_buffer_rsrc_t m_rsrc = t8b_rsrc(a.mask_bits, a.mask_bits ? (unsigned long long)a.N * 512 * 8 : 0ull);
    const __amdgpu_buffer_rsrc_t u_rsrc = t8b_rsrc(a.dsum, a.dsum ? (unsigned long long)a.N * 512 * 4 : 0ull);

    // staging lane roles: voxel x = lane >> 2 of a row, channels 4 cq .. + 3: k-group cq >> 1, bytes 8 (cq & 1) ..
    const int sx = lane >> 2, cq = lane & 3;
    const int w_off = (cq >> 1) * T8B_KG + sx * 16 + (cq & 1) * 8;
    const unsigned ldA = (unsigned)lane * 16u;
    // fragment lane roles: column n = output voxel (row 2 tile + (n >> 3), x = n & 7), k-group kg: channel half kg & 1 of window position kg >> 1.
    //   (tx 0 | 1) at ty: input row 2 qy + ty, slot 2 qx + (kg >> 1);   (ty 0 | 1) at tx 2: row 2 qy + (kg >> 1), slot 2 qx + 2;   (ty 2 | -) at tx 2: row 2 qy + 2 + (kg >> 1)
    int f01[2][3], f22[2][2];       // [tile of the wave][...] byte offsets inside a plane image (row shift included)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qy = 2 * (tp + 2 * j) + (n >> 3), qx = n & 7, h = kg >> 1;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) f01[j][ty] = t8b_row(2 * qy + ty) + (kg & 1) * T8B_KG + (2 * qx + h) * 16;
        f22[j][0] = t8b_row(2 * qy + h) + (kg & 1) * T8B_KG + (2 * qx + 2) * 16;
        f22[j][1] = t8b_row(2 * qy + 2 + h) + (kg & 1) * T8B_KG + (2 * qx + 2) * 16;
    }
    // epilogue lane roles: output voxel n of the tile, channels 16 cb + 4 kg .. + 3
    const unsigned e_out = (unsigned)(cb * 64 + kg * 16), e_msk = (unsigned)(cb * 4 + kg);

    // patches of this workgroup (XCD-aware as in t3d.hip)
    const int G8 = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, jb = (int)blockIdx.x >> 3;
    const int npx = a.N > xcd ? (a.N - xcd + 7) >> 3 : 0;
    const int npw = npx > jb ? (npx - jb + G8 - 1) / G8 : 0;
    auto patch_of = [&](int i) __attribute__((always_inline)) { return 8 * (jb + (i < npw ? i : npw - 1) * G8) + xcd; };

    // this wave's rows of a staging group: 32 rows (two planes x 16) over 4 waves = 8 rows: list index k = 8 wave + j -> plane pa + (k >> 4), row k & 15
    f32x4 RA[8];
    auto fetch = [&](int p, int pa) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * wave + j;
            const int pz = pa + (k >> 4);
            const unsigned row = ((unsigned)p * 16u + (unsigned)(pz < 16 ? pz : 15)) * 16u + (unsigned)(k & 15);
            RA[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(i_rsrc, (int)ldA, t8b_s(row * 1024u), 0));
        }
    };
    auto stage = [&](int pa, bool on) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * wave + j;
            const int pz = pa + (k >> 4), y = k & 15;
            const bool rv = on && pz < 16;
            const float sc = rv ? a.scale : 0.f, sc11 = rv ? a.scale11 : 0.f;
            char *dst = ring + (pz % 3) * T8B_PLANE + t8b_row(y) + w_off;
            const f32x4 g = RA[j];
            const f16x2 h01 = __builtin_convertvector(f32x2{g.x * sc, g.y * sc}, f16x2), h23 = __builtin_convertvector(f32x2{g.z * sc, g.w * sc}, f16x2);
            const f16x2 l01 = __builtin_convertvector(f32x2{__builtin_fmaf((float)h01.x, -2048.f, g.x * sc11), __builtin_fmaf((float)h01.y, -2048.f, g.y * sc11)}, f16x2);
            const f16x2 l23 = __builtin_convertvector(f32x2{__builtin_fmaf((float)h23.x, -2048.f, g.z * sc11), __builtin_fmaf((float)h23.y, -2048.f, g.w * sc11)}, f16x2);
            *reinterpret_cast<i32x2 *>(dst) = i32x2{__builtin_bit_cast(int, h01), __builtin_bit_cast(int, h23)};
            *reinterpret_cast<i32x2 *>(dst + T8B_PIECE) = i32x2{__builtin_bit_cast(int, l01), __builtin_bit_cast(int, l23)};
        }
    };

    // the sequence of steps: step g = 8 i + zq needs planes 2 zq .. 2 zq + 2 of patch i; group(g) = the two planes it adds to the ring (2 zq + 1, 2 zq + 2).
    // Plane 0 of a patch is staged on its own at the patch's first step (by wave rows k < 16 of a group whose first plane is -1: skipped rows).
    const int total = 8 * npw;
    if (total > 0) fetch(patch_of(0), 1);
    for (int g = 0; g < total; ++g) {
        const int i = g >> 3, zq = g & 7;
        const int p = patch_of(i);
        if (zq == 0) {      // plane 0 of the patch: 16 rows, 4 per wave (the group loads are in flight behind these)
            f32x4 R0[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                R0[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(i_rsrc, (int)ldA, t8b_s((((unsigned)p * 16u) * 16u + (unsigned)(4 * wave + j)) * 1024u), 0));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                char *dst = ring + t8b_row(4 * wave + j) + w_off;
                const f32x4 gq = R0[j];
                const float sc = a.scale, sc11 = a.scale11;
                const f16x2 h01 = __builtin_convertvector(f32x2{gq.x * sc, gq.y * sc}, f16x2), h23 = __builtin_convertvector(f32x2{gq.z * sc, gq.w * sc}, f16x2);
                const f16x2 l01 = __builtin_convertvector(f32x2{__builtin_fmaf((float)h01.x, -2048.f, gq.x * sc11), __builtin_fmaf((float)h01.y, -2048.f, gq.y * sc11)}, f16x2);
                const f16x2 l23 = __builtin_convertvector(f32x2{__builtin_fmaf((float)h23.x, -2048.f, gq.z * sc11), __builtin_fmaf((float)h23.y, -2048.f, gq.w * sc11)}, f16x2);
                *reinterpret_cast<i32x2 *>(dst) = i32x2{__builtin_bit_cast(int, h01), __builtin_bit_cast(int, h23)};
                *reinterpret_cast<i32x2 *>(dst + T8B_PIECE) = i32x2{__builtin_bit_cast(int, l01), __builtin_bit_cast(int, l23)};
            }
        }
        stage(2 * zq + 1, true);
        // the sign bytes of this step's two tiles (used after the MFMAs)
        unsigned mb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned vox = ((unsigned)p * 8u + (unsigned)zq) * 64u + (unsigned)(tp + 2 * j) * 16u;      // first voxel of the tile
            mb[j] = a.mask_bits ? (unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(m_rsrc, (int)((unsigned)n * 8u + e_msk), t8b_s(vox * 8u), 0) : 0xfu;
        }
        __syncthreads();
        if (g + 1 < total) fetch(patch_of((g + 1) >> 3), 2 * ((g + 1) & 7) + 1);
        f32x4 c[2], cx[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { c[j] = f32x4{0.f, 0.f, 0.f, 0.f}; cx[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int tz = 0; tz < 3; ++tz) {
            const char *pl = ring + ((2 * zq + tz) % 3) * T8B_PLANE;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pi = 0; pi < 5; ++pi) {
                    const char *src = pl + (pi < 3 ? f01[j][pi] : f22[j][pi - 3]);
                    const f16x8 xh = *reinterpret_cast<const f16x8 *>(src);
                    const f16x8 xl = *reinterpret_cast<const f16x8 *>(src + T8B_PIECE);
                    cx[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[tz * 5 + pi], xh, cx[j], 0, 0, 0);
                    c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[tz * 5 + pi], xh, c[j], 0, 0, 0);
                    cx[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[tz * 5 + pi], xl, cx[j], 0, 0, 0);
                }
        }
        // ---- epilogue: this lane holds channels 16 cb + 4 kg .. + 3 of voxel n of its tiles
        float part[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned vox = (unsigned)t8b_s((((unsigned)p * 8u + (unsigned)zq) * 64u + (unsigned)(tp + 2 * j) * 16u));
            const unsigned m = mb[j];
            const float v0 = (m & 1u) ? __builtin_fmaf(cx[j].x, 0x1p-11f, c[j].x) * a.inv : 0.f, v1 = (m & 2u) ? __builtin_fmaf(cx[j].y, 0x1p-11f, c[j].y) * a.inv : 0.f;
            const float v2 = (m & 4u) ? __builtin_fmaf(cx[j].z, 0x1p-11f, c[j].z) * a.inv : 0.f, v3 = (m & 8u) ? __builtin_fmaf(cx[j].w, 0x1p-11f, c[j].w) * a.inv : 0.f;
            const f32x4 o = f32x4{v0, v1, v2, v3};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o), o_rsrc, (int)((unsigned)n * 128u + e_out), (int)(vox * 128u), 0);
            float s_ = (v0 + v1) + (v2 + v3);
            s_ += __shfl_xor(s_, 16, 64);
            s_ += __shfl_xor(s_, 32, 64);
            part[j] = s_;
            if (cb == 1 && kg == 0) xch[(tp + 2 * j) * 16 + n] = s_;
            asm volatile("s_nop 3" :: "v"(o) : "memory");      // (16-byte store data is read late by the hardware: t3d_fwd_kernel)
        }
        __syncthreads();      // the partner's partial sums are there; everyone is done reading the ring planes the next step overwrites
        if (cb == 0 && a.dsum) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned vox = (unsigned)t8b_s((((unsigned)p * 8u + (unsigned)zq) * 64u + (unsigned)(tp + 2 * j) * 16u));
                const float s_ = part[j] + xch[(tp + 2 * j) * 16 + n];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, s_), u_rsrc, (int)(kg == 0 ? (unsigned)n * 4u : T8B_OOB), (int)(vox * 4u), 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------- host
// W: TF conv_transpose filter [tap = (tz * 3 + ty) * 3 + tx][co (16)][ci (32)].  Fragment (piece, channel block cb, tz, K step pi): lane -> row ci = 16 cb + (lane & 15),
// k-group kg = lane >> 4: window position half h = kg >> 1, co = 8 (kg & 1) + c; the position (ty, tx) of (pi, h): pi < 3: (pi, h); pi = 3: (h, 2); pi = 4: (2, 2) for
// h = 0, none for h = 1.
void t3d8_bwd_pack(T3dPlan *plan, const float *W) {
    float amax = 0.f;
    for (size_t i = 0; i < (size_t)27 * 16 * 32; ++i) amax = std::max(amax, std::fabs(W[i]));
    int ex = 0;
    if (amax > 0.f) (void)std::frexp(amax, &ex);
    plan->w_exp = 14 - ex;
    plan->h_W.assign((size_t)2 * 2 * 15 * 64 * 8, 0);
    for (int cb = 0; cb < 2; ++cb)
        for (int tz = 0; tz < 3; ++tz)
            for (int pi = 0; pi < 5; ++pi)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 15, kg = lane >> 4, h = kg >> 1;
                    int ty, tx;
                    bool any = true;
                    if (pi < 3) { ty = pi; tx = h; } else if (pi == 3) { ty = h; tx = 2; } else { ty = 2; tx = 2; any = h == 0; }
                    for (int c = 0; c < 8; ++c) {
                        const int co = 8 * (kg & 1) + c, ci = 16 * cb + r;
                        const float w = any ? W[((size_t)((tz * 3 + ty) * 3 + tx) * 16 + co) * 32 + ci] : 0.f;
                        const float ws = std::ldexp(w, plan->w_exp);
                        const _Float16 hh = (_Float16)ws;
                        const _Float16 ll = (_Float16)std::ldexp(ws - (float)hh, 11);
                        unsigned short hb, lb;
                        std::memcpy(&hb, &hh, 2);
                        std::memcpy(&lb, &ll, 2);
                        const size_t f = (size_t)(cb * 15 + tz * 5 + pi);
                        plan->h_W[(((size_t)0 * 30 + f) * 64 + lane) * 8 + c] = hb;
                        plan->h_W[(((size_t)1 * 30 + f) * 64 + lane) * 8 + c] = lb;
                    }
                }
}

int t3d8_bwd_launch(alq_ctx *ctx, const T3dPlan &plan, const View &dout, const View &din, int N, float in_bound, const unsigned char *mask_bits, float *dsum) {
    ALQ_REQUIRE(plan.ok && plan.d_W && plan.kind == 8, ALQ_EINVAL, "t3d8: backward weights not set");
    ALQ_REQUIRE(dout.cs == 16 && dout.c0 == 0 && dout.split == 0 && dout.D == 16 && din.cs == 32 && din.c0 == 0 && din.split == 0 && din.D == 8 && in_bound > 0.f,
                ALQ_EINVAL, "t3d8: view mismatch");
    ALQ_REQUIRE(N < 4096, ALQ_EUNSUPPORTED, "t3d8: 32-bit byte offsets hold fewer than 4096 patches per pass");
    if (N <= 0) return ALQ_OK;
    int ex = 0;
    (void)std::frexp(in_bound, &ex);
    const int e_in = 14 - ex;
    T8BwdArgs a;
    a.dout = dout.p; a.din = din.p; a.W = reinterpret_cast<const unsigned short *>(plan.d_W); a.mask_bits = mask_bits; a.dsum = dsum;
    a.scale = std::ldexp(1.f, e_in); a.scale11 = std::ldexp(1.f, e_in + 11); a.inv = std::ldexp(1.f, -(e_in + plan.w_exp)); a.N = N;
    const int cus = ctx->num_cus;
    long long g = std::min<long long>(2LL * cus, (long long)N);
    g = std::max<long long>(8, (g + 7) / 8 * 8);
    ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(t3d8_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T8B_LDS));
    ProfScope ps(ctx, PROF_IGEMM_F16, plan.flops_per_patch * N);
    hipLaunchKernelGGL(t3d8_bwd_kernel, dim3((unsigned)g), dim3(256), T8B_LDS, ctx->stream, a);
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

}  // namespace alq
