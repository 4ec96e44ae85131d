// Row-sweep engine for the stride-2 3x3x3 conv_transpose (NET-C's `up2`: 16 -> 8 channels, 16^3 -> 32^3) and its backward-data
// pass.  Replaces, for that layer, the tf.nn.conv3d_transpose call site NN_extended.py:574-587 (forward) and its gradient.
//
// TF geometry (hard parts): with output_shape = 2 * in the op is the FULL transposed result of length 2 in + 1
// cropped to [0, 2 in): out[o] = sum over (i, t) with 2 i + t = o of in[i] W[t], t in {0, 1, 2} per dimension.  Per dimension an
// even output o = 2 i sees taps t = 0 (input i) and t = 2 (input i - 1), an odd output o = 2 i + 1 sees tap t = 1 (input i): the
// 2 x 2 x 2 output block of input cell i needs the 8 cells i - d, d in {0, 1}^3, and 27 of the 64 (parity, neighbour) pairs carry a tap.
//
// These launches are HBM-bound (forward: 256 KB read, 1.13 MB written per patch; backward the mirror image): the two-slot tile
// engine ran them at 3.8 - 4.0 TB/s with 19 - 38 % of the matrix pipe busy (profiles/r04fin_pmc_summary.json).  Here every wave
// works alone - no workgroup barrier, no shared tile slot:
//   * a wave owns ONE input plane iz of one patch and sweeps its 16 rows; an MFMA column block is one x row of 16 cells;
//   * the MFMA A operand is the WEIGHTS (16 rows = x parity x 8 output channels, "pair form"; K = 2 x neighbours x 16 input
//     channels), resident in registers for the whole launch, the B operand the activations: the accumulator of a (z parity, y
//     parity) pair is then one complete output row - 32 voxels x 8 channels = 1 KB contiguous, ONE 16-byte store per lane;
//   * a row is fetched ONCE per use with one coalesced 16-byte load per lane, split (bf16 triples forward, fp16 pairs at their
//     true scale backward) and transposed into fragment order through a wave-private strip of LDS (in-order DS execution makes the
//     write -> read hand-over inside a wave safe without a barrier); the fragments of row iy serve tile iy (as the d_y = 0
//     neighbour) and tile iy + 1 (d_y = 1) from registers;
//   * 9 (z, y) tap pairs x 6 (3) piece products = 54 (27) MFMAs per tile of 16 cells, 25 % of the K slots zero (x parity 1 has
//     no d_x = 1 tap; backward: the 4th x tap slot) - irrelevant under the HBM bound.
// Two 256-thread workgroups per CU (two waves per SIMD): one wave's loads / stores / split run beside the other's MFMAs.
#include "alq_internal.h"
#ifndef T3_CLOBBER
#define T3_CLOBBER : "memory"
#endif

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
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct T3FwdArgs {
    const float *in;             // [N][16][16][16][16] dense
    float *out;                  // [N][32][32][32][8] dense
    const unsigned short *W;     // [9 (cz, cy)][3 pieces][64 lanes][8] bf16 bits (t3d_fwd_pack)
    const float *bias;           // [8]
    float *osum;                 // [N][32^3] channel sums of the output, or null
    unsigned *out_amax;          // [N][16] max |out| per (patch, input plane) as float bits (every entry written), or null
    int N;
};
struct T3BwdArgs {
    const float *dout;           // [N][32][32][32][8] dense: cotangent of the layer's output
    float *din;                  // [N][16][16][16][16] dense: cotangent of its input (masked when mask_bits)
    const unsigned short *W;     // [9 (tz, ty)][2 pieces][64 lanes][8] fp16 bits (t3d_bwd_pack)
    const unsigned char *mask_bits;   // sign field of the input activation: byte (voxel * 16 + 4 g) / 4, bit j = channel 4 g + j > 0; or null
    float *dsum;                 // [N][16^3] channel sums of the (masked) result, or null
    float inv;                   // 2^-(e_in + e_w)
    float scale;                 // 2^e_in
    int N;
};

constexpr unsigned T3_OOB = 0xffffff00u;
// forward LDS strip of a wave: [row 0: plane iz, row 1: plane iz - 1][3 pieces][17 voxel slots (slot 0 = x -1: zero)][16 ch x 2 B]
constexpr int T3F_ROWB = 17 * 32, T3F_WAVE = 2 * 3 * T3F_ROWB;
// backward: [6 rows][2 pieces][34 voxel slots (32, 33: zero)][8 ch x 2 B]
constexpr int T3B_ROWB = 34 * 16, T3B_WAVE = 6 * 2 * T3B_ROWB;

__device__ inline __amdgpu_buffer_rsrc_t t3_rsrc(const void *base, unsigned long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(unsigned)bytes, 0x00020000);
}
// x -> (hi, rem): hi = bf16(x) round-to-nearest packed pairwise, rem = x - hi (exact) - the split of igemm4.hip
__device__ inline unsigned t3_split2(float &a, float &b) {
    const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    a -= __builtin_bit_cast(float, hb << 16);
    b -= __builtin_bit_cast(float, hb & 0xffff0000u);
    return hb;
}
__device__ inline unsigned t3_pack2(float a, float b) {      // last piece: at most 8 significant bits left, the high halves are exact
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}

// XCD-aware unit order: workgroups b and b + 8 share an XCD (speed only, MI355X_MICROARCH.md): the four quarter-patch units
// of a patch go to workgroups of ONE XCD, so the plane a quarter shares with its neighbour is read from that XCD's L2.
// unit v of XCD x (v = j, j + G/8, ...): patch 8 (v / 4) + x, quarter v % 4; a wave's work is the stream of the 16 row tiles of
// each of its workgroup's units, tile T = 16 r + (sweep position).
struct T3Cur { int p, iz, s; bool ok; };
__device__ inline T3Cur t3_tile(int T, int total, int wave) {
    const int G8 = (int)gridDim.x >> 3, x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    const int v = j + (T >> 4) * G8;
    T3Cur c;
    c.p = 8 * (v >> 2) + x;
    c.iz = 4 * (v & 3) + wave;
    c.s = T & 15;
    c.ok = T < total;
    return c;
}
__device__ inline int t3_total_tiles(int N) {      // 16 x the units of this workgroup: v = j + r G8 with 8 (v >> 2) + x < N
    const int G8 = (int)gridDim.x >> 3, x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    const int np = N > x ? (N - x + 7) >> 3 : 0;          // patches of this XCD
    const int nv = 4 * np;                                // its units: v < nv
    return nv > j ? 16 * ((nv - j + G8 - 1) / G8) : 0;
}

// ---------------------------------------------------------------------------------------------------------------- forward
constexpr int T3F_PF = 4;      // tiles of loads in flight per wave (2 KB each)
template <bool SUMS, bool AMAX>
__global__ __launch_bounds__(256, 2) void t3d_fwd_kernel(const T3FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char t3lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char *strip = t3lds + wave * T3F_WAVE;
    // zero slots (x = -1) of the six piece rows, once
    if (lane < 12) {
        const int r = lane >> 1;
        *reinterpret_cast<i32x4 *>(strip + r * T3F_ROWB + (lane & 1) * 16) = i32x4{0, 0, 0, 0};
    }
    const int n = lane & 15, kg = lane >> 4;
    // staging: lane -> voxel lane >> 2, channels 4 (lane & 3) .. + 3 of a 1 KB row; fragments: cell n, d_x = kg >> 1, channel half kg & 1
    const int w_off = ((lane >> 2) + 1) * 32 + (lane & 3) * 8;
    const int f_off = (n + 1 - (kg >> 1)) * 32 + (kg & 1) * 16;
    bf16x8 wr[9][3];
#pragma unroll
    for (int c = 0; c < 9; ++c)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
            wr[c][pc] = *reinterpret_cast<const bf16x8 *>(a.W + ((size_t)(c * 3 + pc) * 64 + lane) * 8);
    f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + 4 * (kg & 1));
    // the launch constants must have ARRIVED before the tile loop: a first use inside the loop would put a vmcnt wait there that
    // (sized for the pass that follows these loads) also waits for the tile stores of every later pass
#pragma unroll
    for (int c = 0; c < 9; ++c)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) asm volatile("" : "+v"(wr[c][pc]));
    asm volatile("" : "+v"(bias4));
    const __amdgpu_buffer_rsrc_t in_rsrc = t3_rsrc(a.in, (unsigned long long)a.N * 16 * 16 * 16 * 16 * 4);
    const __amdgpu_buffer_rsrc_t out_rsrc = t3_rsrc(a.out, (unsigned long long)a.N * 32 * 32 * 32 * 8 * 4);
    const __amdgpu_buffer_rsrc_t sum_rsrc = t3_rsrc(a.osum, SUMS ? (unsigned long long)a.N * 32 * 32 * 32 * 4 : 0ull);
    const unsigned ld_off = (unsigned)lane * 16u;
    const unsigned st_off = (unsigned)(2 * n + (kg >> 1)) * 32u + (unsigned)(kg & 1) * 16u;      // inside a 1 KB output row
    const unsigned sm_off = (kg & 1) ? T3_OOB : (unsigned)(2 * n + (kg >> 1)) * 4u;              // lanes with the channel-half 0 store the sums
    const int total = t3_total_tiles(a.N);

    f32x4 R0[T3F_PF], R1[T3F_PF];      // rows (iz, iy) and (iz - 1, iy) of the tiles in flight
    auto fetch = [&](int T, f32x4 &r0, f32x4 &r1) __attribute__((always_inline)) {
        const T3Cur c = t3_tile(T, total, wave);
        const unsigned row0 = ((unsigned)(c.p * 16 + c.iz) * 16u + (unsigned)c.s) * 1024u;      // byte offset of row (p, iz, iy)
        r0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)(c.ok ? ld_off : T3_OOB), (int)row0, 0));
        r1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)((c.ok && c.iz > 0) ? ld_off : T3_OOB), (int)(row0 - 16u * 1024u), 0));
    };
#pragma unroll
    for (int k = 0; k < T3F_PF; ++k) { R0[k] = f32x4{0.f, 0.f, 0.f, 0.f}; R1[k] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    bf16x8 fo[2][3];      // fragments of row iy - 1 (d_y = 1), planes iz (0) and iz - 1 (1)
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) fo[d][pc] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    float amx = 0.f;
    const __amdgpu_buffer_rsrc_t amx_rsrc = t3_rsrc(a.out_amax, AMAX ? (unsigned long long)a.N * 16 * 4 : 0ull);
    // The body is straight-line (no branch, no atomic) and there are no loads in front of the loop: the first pass over the body
    // (T0 = -T3F_PF) only issues the first fetches - its own tiles are void (zero rows, stores aimed past the arrays).  With that
    // shape the compiler's vmcnt counting lets the loads of the next T3F_PF - 1 tiles and the stores of the last ones stay in
    // flight; with a break or an atomic inside, or with prologue loads in front, it drained everything at the loop head.  The
    // stream is a multiple of 16 tiles and 16 % T3F_PF = 0: every tile of a pass over the body is real or none is.
    static_assert(16 % T3F_PF == 0, "a sweep must start at k = 0");
    if (total > 0)
    for (int T0 = -T3F_PF; T0 < total; T0 += T3F_PF) {
        const bool live = T0 >= 0;
        const unsigned st_o = live ? st_off : T3_OOB, sm_o = live ? sm_off : T3_OOB;
#pragma unroll
        for (int k = 0; k < T3F_PF; ++k) {
            const int T = T0 + k;
            // (nothing moves across a tile boundary: left alone the scheduler starts the next tile's split early - and with it the
            // wait for loads that were meant to stay in flight for another tile)
            __builtin_amdgcn_sched_barrier(0);
            const T3Cur cur = t3_tile(live ? T : 0, total, wave);
            const int p = cur.p, iz = cur.iz, iy = cur.s;
            // split the two rows of this tile into bf16 triples, into the strip
            {
                auto split_row = [&](int base, const f32x4 &R) __attribute__((always_inline)) {
                    float x = R.x, y = R.y, z = R.z, w = R.w;
                    const unsigned h0 = t3_split2(x, y), h1 = t3_split2(z, w);
                    const unsigned m0 = t3_split2(x, y), m1 = t3_split2(z, w);
                    *reinterpret_cast<i32x2 *>(strip + (base + 0) * T3F_ROWB + w_off) = i32x2{(int)h0, (int)h1};
                    *reinterpret_cast<i32x2 *>(strip + (base + 1) * T3F_ROWB + w_off) = i32x2{(int)m0, (int)m1};
                    *reinterpret_cast<i32x2 *>(strip + (base + 2) * T3F_ROWB + w_off) = i32x2{(int)t3_pack2(x, y), (int)t3_pack2(z, w)};
                };
                split_row(0, R0[k]);
                split_row(3, R1[k]);
            }
            fetch(T + T3F_PF, R0[k], R1[k]);      // (beyond the stream: lanes aim past the array)
            bf16x8 fn[2][3];
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
                    fn[d][pc] = *reinterpret_cast<const bf16x8 *>(strip + (d * 3 + pc) * T3F_ROWB + f_off);
            if (k == 0) {      // a new sweep (iy = 0, only ever at k = 0): no row above - clear the carried fragments, branch-free
                const int keep = iy == 0 ? 0 : -1;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) {
                        i32x4 t = __builtin_bit_cast(i32x4, fo[d][pc]);
                        t.x &= keep; t.y &= keep; t.z &= keep; t.w &= keep;
                        fo[d][pc] = __builtin_bit_cast(bf16x8, t);
                    }
            }
            f32x4 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // (cz, cy): 0 = (parity 0, d 0: tap 0), 1 = (parity 0, d 1: tap 2), 2 = (parity 1, d 0: tap 1)
#pragma unroll
            for (int cz = 0; cz < 3; ++cz)
#pragma unroll
                for (int cy = 0; cy < 3; ++cy) {
                    const int pz = cz == 2, dz = cz == 1, py = cy == 2, dy = cy == 1;
                    const bf16x8 *w = wr[cz * 3 + cy];
                    const bf16x8 *f = dy ? fo[dz] : fn[dz];
                    f32x4 c = acc[pz][py];
                    // small products first: (lo, hi) (hi, lo) (mid, mid) (mid, hi) (hi, mid) (hi, hi)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], f[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], f[2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], f[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], f[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], f[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], f[0], c, 0, 0, 0);
                    acc[pz][py] = c;
                }
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) fo[d][pc] = fn[d][pc];
            // four finished output rows: (2 iz + pz, 2 iy + py), each 1 KB.
            // The four value vectors stay ALIVE until the end of the tile (the empty asm below): a 16-byte buffer store reads its data
            // registers some cycles after it issues, and with a second wave on the SIMD a vector instruction of this wave that reused
            // them right behind the store (the |v| of the maximum, two instructions later) reached the registers first - the stored
            // row then held max(|v.x|, |v.y|) in channel 0 of its last lanes (seen only in workgroups dispatched as the second one
            // of a CU; the compiler's hazard table covers this case only for stores without a scalar offset).
            f32x4 vv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = acc[i >> 1][i & 1];
                v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                vv[i] = v;
                if constexpr (AMAX)
                    amx = fmaxf(fmaxf(amx, fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y))), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pz = i >> 1, py = i & 1;
                const f32x4 v = vv[i];
                const unsigned orow = ((unsigned)p * 32u + (unsigned)(2 * iz + pz)) * 32u + (unsigned)(2 * iy + py);      // output row index
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), out_rsrc, (int)st_o, (int)(orow * 1024u), 0);
                if constexpr (SUMS) {
                    const float s = (v.x + v.y) + (v.z + v.w);
                    const float t = s + __shfl_xor(s, 16, 64);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, t), sum_rsrc, (int)sm_o, (int)(orow * 128u), 0);
                }
            }
            if constexpr (AMAX) {
                if (k == T3F_PF - 1) {      // (iy = 15 only ever at the last k) the sweep of this plane is done: its maximum, one store
                    float mx = amx;
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
                    const unsigned off = (live && iy == 15 && lane == 0) ? 0u : T3_OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, mx), amx_rsrc, (int)off, (int)((unsigned)(p * 16 + iz) * 4u), 0);
                    amx = iy == 15 ? 0.f : amx;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 3" :: "v"(vv[0]), "v"(vv[1]), "v"(vv[2]), "v"(vv[3]) T3_CLOBBER);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- backward
// d_in[i][ci] = sum over t in {0,1,2}^3, co of W[t][co][ci] d_out[2 i + t][co] (2 i + t inside the output).  Tile = one row of 16
// cells; MFMA rows = 16 input channels, K = 4 x positions (2 n .. 2 n + 3; the 4th has zero weights) x 8 output channels per
// (t_z, t_y) pair; fp16 pairs at their true scale, three products in one accumulator (c3d.hip's one-accumulator form: the
// matrix cores keep fp16 subnormals, probed by c3d_subnormals_ok).  The sweep runs DOWNWARD (iy = 15 .. 0): a tile fetches the
// rows t_y = 0, 1 of its three planes and takes t_y = 2 (row 2 iy + 2 = the t_y = 0 row of tile iy + 1) from the tile before it.
constexpr int T3B_PF = 2;      // tiles of loads in flight per wave (6 KB each)
template <bool MASK, bool SUMS>
__global__ __launch_bounds__(256, 2) void t3d_bwd_kernel(const T3BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char t3lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char *strip = t3lds + wave * T3B_WAVE;
    if (lane < 24) {      // voxel slots 32, 33 of the twelve piece rows: zero, once
        const int r = lane >> 1;
        *reinterpret_cast<i32x4 *>(strip + r * T3B_ROWB + (32 + (lane & 1)) * 16) = i32x4{0, 0, 0, 0};
    }
    const int n = lane & 15, kg = lane >> 4;
    // staging: lane -> voxel lane >> 1, channels 4 (lane & 1) .. + 3 of a 1 KB row (32 voxels x 8 channels)
    const int w_off = (lane >> 1) * 16 + (lane & 1) * 8;
    const int f_off = (2 * n + kg) * 16;
    f16x8 wr[9][2];
#pragma unroll
    for (int c = 0; c < 9; ++c)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
            wr[c][pc] = *reinterpret_cast<const f16x8 *>(a.W + ((size_t)(c * 2 + pc) * 64 + lane) * 8);
#pragma unroll
    for (int c = 0; c < 9; ++c)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) asm volatile("" : "+v"(wr[c][pc]));      // arrived before the tile loop (see the forward kernel)
    const __amdgpu_buffer_rsrc_t in_rsrc = t3_rsrc(a.dout, (unsigned long long)a.N * 32 * 32 * 32 * 8 * 4);
    const __amdgpu_buffer_rsrc_t out_rsrc = t3_rsrc(a.din, (unsigned long long)a.N * 16 * 16 * 16 * 16 * 4);
    const __amdgpu_buffer_rsrc_t sum_rsrc = t3_rsrc(a.dsum, SUMS ? (unsigned long long)a.N * 16 * 16 * 16 * 4 : 0ull);
    const __amdgpu_buffer_rsrc_t msk_rsrc = t3_rsrc(a.mask_bits, MASK ? (unsigned long long)a.N * 16 * 16 * 16 * 4 : 0ull);
    const unsigned ld_off = (unsigned)lane * 16u;
    const unsigned st_off = (unsigned)n * 64u + (unsigned)kg * 16u;       // cell n, channels 4 kg .. + 3 inside a 1 KB row of d_in
    const unsigned mk_off = (unsigned)n * 4u + (unsigned)kg;
    const unsigned sm_off = kg == 0 ? (unsigned)n * 4u : T3_OOB;
    const float sc = a.scale;
    const int total = t3_total_tiles(a.N);

    auto stage = [&](int slot, const f32x4 &R) __attribute__((always_inline)) {      // one row -> fp16 pair (h, l) in strip rows 2 slot, 2 slot + 1
        const float x0 = R.x * sc, x1 = R.y * sc, x2 = R.z * sc, x3 = R.w * sc;
        const f16x2 h01 = __builtin_convertvector(f32x2{x0, x1}, f16x2), h23 = __builtin_convertvector(f32x2{x2, x3}, f16x2);
        const f16x2 l01 = __builtin_convertvector(f32x2{x0 - (float)h01.x, x1 - (float)h01.y}, f16x2);
        const f16x2 l23 = __builtin_convertvector(f32x2{x2 - (float)h23.x, x3 - (float)h23.y}, f16x2);
        *reinterpret_cast<i32x2 *>(strip + (2 * slot) * T3B_ROWB + w_off) = i32x2{__builtin_bit_cast(int, h01), __builtin_bit_cast(int, h23)};
        *reinterpret_cast<i32x2 *>(strip + (2 * slot + 1) * T3B_ROWB + w_off) = i32x2{__builtin_bit_cast(int, l01), __builtin_bit_cast(int, l23)};
    };
    f32x4 R[T3B_PF][6];      // rows (tz, ty) = (0,0) (0,1) (1,0) (1,1) (2,0) (2,1) of the tiles in flight
    unsigned MK[T3B_PF];     // ... and their mask bytes
    auto fetch = [&](int T, f32x4 *r, unsigned &mk) __attribute__((always_inline)) {
        const T3Cur c = t3_tile(T, total, wave);
        const int iy = 15 - c.s;
        // source rows (z = 2 iz + tz, y = 2 iy + ty) at ((p * 32 + z) * 32 + y) * 1024; z = 32 does not exist
        const unsigned b0 = (((unsigned)c.p * 32u + (unsigned)(2 * c.iz)) * 32u + (unsigned)(2 * iy)) * 1024u;
        const unsigned okl = c.ok ? ld_off : T3_OOB, ok2 = (c.ok && 2 * c.iz + 2 < 32) ? ld_off : T3_OOB;
#pragma unroll
        for (int tz = 0; tz < 3; ++tz)
#pragma unroll
            for (int ty = 0; ty < 2; ++ty)
                r[tz * 2 + ty] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)(tz < 2 ? okl : ok2),
                                                                                                 (int)(b0 + (unsigned)tz * 32768u + (unsigned)ty * 1024u), 0));
        if constexpr (MASK)
            mk = (unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(msk_rsrc, (int)(c.ok ? mk_off : T3_OOB),
                                                                             (int)((((unsigned)c.p * 16u + (unsigned)c.iz) * 16u + (unsigned)iy) * 64u), 0);
    };
#pragma unroll
    for (int k = 0; k < T3B_PF; ++k) {
        MK[k] = 0u;
#pragma unroll
        for (int s = 0; s < 6; ++s) R[k][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    f16x8 fo[3][2];   // fragments of row ty = 2 (the previous tile's ty = 0) per tz
#pragma unroll
    for (int tz = 0; tz < 3; ++tz)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) fo[tz][pc] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    static_assert(16 % T3B_PF == 0, "a sweep must start at k = 0");
    // (straight-line body, first pass void: see the forward kernel)
    if (total > 0)
    for (int T0 = -T3B_PF; T0 < total; T0 += T3B_PF) {
        const bool live = T0 >= 0;
        const unsigned st_o = live ? st_off : T3_OOB, sm_o = live ? sm_off : T3_OOB;
#pragma unroll
        for (int k = 0; k < T3B_PF; ++k) {
            const int T = T0 + k;
            __builtin_amdgcn_sched_barrier(0);      // (see the forward kernel)
            const T3Cur cur = t3_tile(live ? T : 0, total, wave);
            const int p = cur.p, iz = cur.iz, iy = 15 - cur.s;
#pragma unroll
            for (int s = 0; s < 6; ++s) stage(s, R[k][s]);
            const unsigned nb = MK[k];
            fetch(T + T3B_PF, R[k], MK[k]);
            f16x8 fn[3][2][2];      // [tz][ty][piece]
#pragma unroll
            for (int tz = 0; tz < 3; ++tz)
#pragma unroll
                for (int ty = 0; ty < 2; ++ty)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc)
                        fn[tz][ty][pc] = *reinterpret_cast<const f16x8 *>(strip + ((2 * tz + ty) * 2 + pc) * T3B_ROWB + f_off);
            if (k == 0) {      // a new sweep starts at iy = 15 (only ever at k = 0): row y = 32 does not exist
                const int keep = cur.s == 0 ? 0 : -1;
#pragma unroll
                for (int tz = 0; tz < 3; ++tz)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        i32x4 t = __builtin_bit_cast(i32x4, fo[tz][pc]);
                        t.x &= keep; t.y &= keep; t.z &= keep; t.w &= keep;
                        fo[tz][pc] = __builtin_bit_cast(f16x8, t);
                    }
            }
            f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tz = 0; tz < 3; ++tz)
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) {
                    const f16x8 *w = wr[tz * 3 + ty];
                    const f16x8 *f = ty == 2 ? fo[tz] : fn[tz][ty];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[1], f[0], c, 0, 0, 0);      // (l, h) (h, l) (h, h)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], f[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], f[0], c, 0, 0, 0);
                }
#pragma unroll
            for (int tz = 0; tz < 3; ++tz)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) fo[tz][pc] = fn[tz][0][pc];
            // epilogue: cell row (iz, iy): 16 cells x 16 channels = 1 KB
            const unsigned crow = ((unsigned)p * 16u + (unsigned)iz) * 16u + (unsigned)iy;
            f32x4 v = f32x4{c.x * a.inv, c.y * a.inv, c.z * a.inv, c.w * a.inv};
            if constexpr (MASK) {
                v.x = (nb & 1u) ? v.x : 0.f; v.y = (nb & 2u) ? v.y : 0.f;
                v.z = (nb & 4u) ? v.z : 0.f; v.w = (nb & 8u) ? v.w : 0.f;
            }
            // (opaque: in the peeled void first pass v is a constant, and the compiler would otherwise materialise one copy for the
            // store and another for the hold below - the hold must name the registers the store reads, tools/isa_store_hazard.py)
#ifndef T3_NO_OPAQUE
            asm volatile("" : "+v"(v));
#endif
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), out_rsrc, (int)st_o, (int)(crow * 1024u), 0);
            if constexpr (SUMS) {
                float s = (v.x + v.y) + (v.z + v.w);
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, s), sum_rsrc, (int)sm_o, (int)(crow * 64u), 0);
            }
            // (the store's data registers stay alive for a while: see the forward kernel's epilogue)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 3" :: "v"(v) T3_CLOBBER);
        }
    }
}

// ------------------------------------------------------------------------------------------- forward, 32 -> 16 channels at 8^3
// NET-C's `up1` (8^3 -> 16^3).  32 input channels are exactly one MFMA K block: a B fragment is (cell n, channels 8 kg .. + 7 of
// ONE neighbour cell), so there are 27 (parity, neighbour) MFMA groups per tile and 8 parity accumulators [16 output channels x 16
// cells]; the 27 x 3 weight fragments (81 KB of bf16 triples) do not fit registers: resident in LDS, ONE 512-thread workgroup per
// CU, wave w = input plane w of the patch, tile = 16 cells = rows 2 j, 2 j + 1 of that plane (four tiles per plane, unrolled: the
// ring slots of the rows are compile-time constants).  Rows are split ONCE (bf16 triples) into a wave-private strip of three row
// slots per plane (rows 2 j - 1, 2 j, 2 j + 1: the first is the previous tile's last); a first version took the B fragments
// straight from global memory and split every voxel eight times - 640 vector instructions per 162 MFMAs, matrix pipe 33 % busy.
// x = -1 has no slot (81 KB + 8 strips fill the 160 KB): the d_x = 1 fragments of the lanes ix = 0 are cleared by a select.
struct T8FwdArgs {
    const float *in;             // [N][8][8][8][32] dense
    float *out;                  // [N][16][16][16][16] dense
    const unsigned short *W;     // [27 (cz, cy, cx)][3 pieces][64 lanes][8] bf16 bits (t3d8_fwd_pack)
    const float *bias;           // [16]
    float *osum;                 // [N][16^3] channel sums of the output, or null
    int N;
};
constexpr int T8_WBYTES = 27 * 3 * 1024;
constexpr int T8_ROWB = 8 * 64 + 32;               // 8 voxels x 32 channels x 2 B, + 32: rows r = 0 / 1 of a fragment read land on disjoint banks
constexpr int T8_SLOT = 3 * T8_ROWB;               // the three pieces of a row
constexpr int T8_PLANE = 3 * T8_SLOT;              // three row slots
constexpr int T8_WAVE = 2 * T8_PLANE;              // planes iz, iz - 1

template <bool SUMS>
__global__ __launch_bounds__(512, 2) void t3d8_fwd_kernel(const T8FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char t3lds[];
    {   // the weight fragments into LDS, once
        const i32x4 *src = reinterpret_cast<const i32x4 *>(a.W);
        i32x4 *dst = reinterpret_cast<i32x4 *>(t3lds);
        for (int i = threadIdx.x; i < T8_WBYTES / 16; i += 512) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = lane & 15, kg = lane >> 4, r = n >> 3, ix = n & 7;
    char *strip = t3lds + T8_WBYTES + wave * T8_WAVE;
    f32x4 bias4 = *reinterpret_cast<const f32x4 *>(a.bias + 4 * kg);
    asm volatile("" : "+v"(bias4));
    const __amdgpu_buffer_rsrc_t in_rsrc = t3_rsrc(a.in, (unsigned long long)a.N * 8 * 8 * 8 * 32 * 4);
    const __amdgpu_buffer_rsrc_t out_rsrc = t3_rsrc(a.out, (unsigned long long)a.N * 16 * 16 * 16 * 16 * 4);
    const __amdgpu_buffer_rsrc_t sum_rsrc = t3_rsrc(a.osum, SUMS ? (unsigned long long)a.N * 16 * 16 * 16 * 4 : 0ull);
    const int iz = wave;
    const char *wl = t3lds + lane * 16;
    const int npw = a.N > (int)blockIdx.x ? (a.N - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;      // patches of this workgroup
    // staging: a row = 8 voxels x 32 channels = 1 KB: lane -> voxel lane >> 3, channels 4 (lane & 7) .. + 3
    const unsigned ld_off = (unsigned)lane * 16u;
    const int w_off = (lane >> 3) * 64 + (lane & 7) * 8;
    // fragments: cell (r, ix), neighbour d_x: voxel ix - d_x (ix = 0, d_x = 1: cleared), channels 8 kg .. + 7
    const int f_off0 = ix * 64 + kg * 16, f_off1 = (ix > 0 ? ix - 1 : 0) * 64 + kg * 16;
    const int keepx = ix > 0 ? -1 : 0;

    f32x4 R[2][4];      // [buffer][plane iz: rows 2 j, 2 j + 1; plane iz - 1: rows 2 j, 2 j + 1]
    auto fetch = [&](int u, int j, f32x4 *rr) __attribute__((always_inline)) {      // the four rows of tile j of unit u (patch blockIdx + gridDim u)
        const bool ok = u >= 0 && u < npw;
        const int p = (int)blockIdx.x + (int)gridDim.x * (ok ? u : 0);
        const unsigned pb = (((unsigned)p * 8u + (unsigned)iz) * 8u + (unsigned)(2 * j)) * 1024u;      // row (p, iz, 2 j)
        const unsigned o0 = ok ? ld_off : T3_OOB, o1 = (ok && iz > 0) ? ld_off : T3_OOB;
        rr[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)o0, (int)pb, 0));
        rr[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)o0, (int)(pb + 1024u), 0));
        rr[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)o1, (int)(pb - 8u * 1024u), 0));
        rr[3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)o1, (int)(pb - 8u * 1024u + 1024u), 0));
    };
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) R[b][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one pass of the loop = the four tiles of one plane (unit u = one patch), straight-line; pass u = -1 is void and only fetches
    // (see t3d_fwd_kernel); tile j uses buffer j & 1 and refills it with tile j + 2 (of the next unit for j = 2, 3)
    if (npw > 0)
    for (int u = -1; u < npw; ++u) {
        const bool live = u >= 0;
        const int p = (int)blockIdx.x + (int)gridDim.x * (live ? u : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            const int b = j & 1;
            // ring: row y of the plane sits in slot (y + 1) % 3, so tile j has its rows 2 j - 1, 2 j, 2 j + 1 in slots s0, s0 + 1, s0 + 2 (mod 3)
            const int s0 = (2 * j) % 3;
            if (j == 0) {      // row -1 does not exist: its slot holds zeros for this tile
                for (int i = lane; i < 2 * 3 * (T8_ROWB / 16); i += 64) {
                    const int pl = i / (3 * (T8_ROWB / 16)), q = i % (3 * (T8_ROWB / 16));
                    *reinterpret_cast<i32x4 *>(strip + pl * T8_PLANE + s0 * T8_SLOT + q * 16) = i32x4{0, 0, 0, 0};
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {      // split rows (plane i >> 1, row 2 j + (i & 1)) into slots s0 + 1, s0 + 2
                const int pl = i >> 1, sl = (s0 + 1 + (i & 1)) % 3;
                float x = R[b][i].x, y = R[b][i].y, z = R[b][i].z, w = R[b][i].w;
                const unsigned h0 = t3_split2(x, y), h1 = t3_split2(z, w);
                const unsigned m0 = t3_split2(x, y), m1 = t3_split2(z, w);
                char *dst = strip + pl * T8_PLANE + sl * T8_SLOT + w_off;
                *reinterpret_cast<i32x2 *>(dst) = i32x2{(int)h0, (int)h1};
                *reinterpret_cast<i32x2 *>(dst + T8_ROWB) = i32x2{(int)m0, (int)m1};
                *reinterpret_cast<i32x2 *>(dst + 2 * T8_ROWB) = i32x2{(int)t3_pack2(x, y), (int)t3_pack2(z, w)};
            }
            fetch(j < 2 ? u : u + 1, (j + 2) & 3, R[b]);
            // per lane: the slot of row (2 j + r - d_y): d_y = 0 -> s0 + 1 + r, d_y = 1 -> s0 + r (mod 3)
            const int ro0 = (r ? (s0 + 2) % 3 : (s0 + 1) % 3) * T8_SLOT, ro1 = (r ? (s0 + 1) % 3 : s0 % 3) * T8_SLOT;
            f32x4 acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            // (tried on the device and dropped, each within +-4 % of this form: requesting the weight fragments of the next one / three
            // groups ahead of the MFMAs by hand, interleaving the MFMA chains of two groups; without the stores the kernel takes 170 us)
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const int dz = d >> 2, dy = (d >> 1) & 1, dx = d & 1;
                const char *src = strip + dz * T8_PLANE + (dy ? ro1 : ro0) + (dx ? f_off1 : f_off0);
                i32x4 q0 = *reinterpret_cast<const i32x4 *>(src), q1 = *reinterpret_cast<const i32x4 *>(src + T8_ROWB), q2 = *reinterpret_cast<const i32x4 *>(src + 2 * T8_ROWB);
                if (dx) {
                    q0.x &= keepx; q0.y &= keepx; q0.z &= keepx; q0.w &= keepx;
                    q1.x &= keepx; q1.y &= keepx; q1.z &= keepx; q1.w &= keepx;
                    q2.x &= keepx; q2.y &= keepx; q2.z &= keepx; q2.w &= keepx;
                }
                const bf16x8 f0 = __builtin_bit_cast(bf16x8, q0), f1 = __builtin_bit_cast(bf16x8, q1), f2 = __builtin_bit_cast(bf16x8, q2);
                // every output parity this neighbour feeds: per dimension d = 0 -> parities 0 (tap 0: c = 0) and 1 (tap 1: c = 2); d = 1 -> parity 0 (tap 2: c = 1)
#pragma unroll
                for (int qz = 0; qz < (dz ? 1 : 2); ++qz)
#pragma unroll
                    for (int qy = 0; qy < (dy ? 1 : 2); ++qy)
#pragma unroll
                        for (int qx = 0; qx < (dx ? 1 : 2); ++qx) {
                            const int cz = dz ? 1 : (qz ? 2 : 0), cy = dy ? 1 : (qy ? 2 : 0), cx = dx ? 1 : (qx ? 2 : 0);
                            const int pz = cz == 2, py = cy == 2, px = cx == 2;
                            const int combo = (cz * 3 + cy) * 3 + cx;
                            const bf16x8 w0 = *reinterpret_cast<const bf16x8 *>(wl + (combo * 3 + 0) * 1024);
                            const bf16x8 w1 = *reinterpret_cast<const bf16x8 *>(wl + (combo * 3 + 1) * 1024);
                            const bf16x8 w2 = *reinterpret_cast<const bf16x8 *>(wl + (combo * 3 + 2) * 1024);
                            f32x4 c = acc[(pz * 2 + py) * 2 + px];
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, f0, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, f2, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, f1, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, f0, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, f1, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, f0, c, 0, 0, 0);
                            acc[(pz * 2 + py) * 2 + px] = c;
                        }
            }
            // epilogue: lane = (cell (r, ix), output channels 4 kg .. + 3); output voxel (2 iz + pz, 2 (2 j + r) + py, 2 ix + px), 64 B per voxel
            const unsigned vb = (((unsigned)p * 16u + (unsigned)(2 * iz)) * 16u + (unsigned)(2 * (2 * j + r))) * 16u + (unsigned)(2 * ix);      // voxel index at parity 0
            f32x4 vv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x4 v = acc[i];
                v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                vv[i] = v;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pz = i >> 2, py = (i >> 1) & 1, px = i & 1;
                const unsigned vox = vb + (unsigned)(pz * 256 + py * 16 + px);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, vv[i]), out_rsrc, (int)(live ? vox * 64u + (unsigned)kg * 16u : T3_OOB), 0, 0);
            }
            if constexpr (SUMS) {      // channel sums: the two cross-lane steps of all eight parities side by side (one LDS round trip each, not sixteen)
                float s1[8], s2[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) s1[i] = (vv[i].x + vv[i].y) + (vv[i].z + vv[i].w);
#pragma unroll
                for (int i = 0; i < 8; ++i) s2[i] = s1[i] + __shfl_xor(s1[i], 16, 64);
#pragma unroll
                for (int i = 0; i < 8; ++i) s1[i] = s2[i] + __shfl_xor(s2[i], 32, 64);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int pz = i >> 2, py = (i >> 1) & 1, px = i & 1;
                    const unsigned vox = vb + (unsigned)(pz * 256 + py * 16 + px);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, s1[i]), sum_rsrc, (int)((live && kg == 0) ? vox * 4u : T3_OOB), 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 3" :: "v"(vv[0]), "v"(vv[1]), "v"(vv[2]), "v"(vv[3]), "v"(vv[4]), "v"(vv[5]), "v"(vv[6]), "v"(vv[7]) T3_CLOBBER);      // (store data stays alive: t3d_fwd_kernel)
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------- host
static bool t3_geometry(const View &in, const View &out, const int k[3], const int lo[3], const int s[3]) {
    return k[0] == 3 && k[1] == 3 && k[2] == 3 && s[0] == 2 && s[1] == 2 && s[2] == 2 && lo[0] == 0 && lo[1] == 0 && lo[2] == 0 &&
           in.D == 16 && in.H == 16 && in.W == 16 && out.D == 32 && out.H == 32 && out.W == 32 && in.C == 16 && out.C == 8 &&
           in.split == 0 && out.split == 0 && in.cs == 16 && in.c0 == 0 && out.cs == 8 && out.c0 == 0;
}

static bool t8_geometry(const View &in, const View &out, const int k[3], const int lo[3], const int s[3]) {
    return k[0] == 3 && k[1] == 3 && k[2] == 3 && s[0] == 2 && s[1] == 2 && s[2] == 2 && lo[0] == 0 && lo[1] == 0 && lo[2] == 0 &&
           in.D == 8 && in.H == 8 && in.W == 8 && out.D == 16 && out.H == 16 && out.W == 16 && in.C == 32 && out.C == 16 &&
           in.split == 0 && out.split == 0 && in.cs == 32 && in.c0 == 0 && out.cs == 16 && out.c0 == 0;
}

int t3d_build(const View &in, const View &out, const int k[3], const int lo[3], const int s[3], T3dPlan *fwd, T3dPlan *bwd) {
    fwd->ok = bwd->ok = false;
    fwd->kind = bwd->kind = 0;
    if (getenv("ALQ_NO_T3D")) return ALQ_OK;
    if (t8_geometry(in, out, k, lo, s) && !getenv("ALQ_NO_T3D8")) {      // 32 -> 16 channels at 8^3: forward here, backward in t3d8b.hip
        fwd->kind = bwd->kind = 8;
        fwd->flops_per_patch = bwd->flops_per_patch = 2.0 * 27 * 32 * 16 * (double)in.vox();
        fwd->ok = true;
        bwd->ok = !getenv("ALQ_NO_T3D8B");
        return ALQ_OK;
    }
    if (!t3_geometry(in, out, k, lo, s)) return ALQ_OK;
    fwd->flops_per_patch = bwd->flops_per_patch = 2.0 * 27 * 16 * 8 * (double)in.vox();
    fwd->ok = bwd->ok = true;
    return ALQ_OK;
}

static unsigned short t3_bf16_rne(float x) {
    unsigned u;
    std::memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float t3_bf16_f(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// W: TF conv3d_transpose filter [tap = (tz * 3 + ty) * 3 + tx][co (8)][ci (16)].
// Forward A fragment of (cz, cy): lane -> row r = lane & 15 (x parity r >> 3, co = r & 7), k-group kg = lane >> 4 (d_x = kg >> 1,
// ci = 8 (kg & 1) + c).  Tap per dimension: (parity 0, d 0) -> 0, (parity 0, d 1) -> 2, (parity 1, d 0) -> 1, (parity 1, d 1) -> none.
void t3d_fwd_pack(T3dPlan *plan, const float *W) {
    plan->h_W.assign((size_t)9 * 3 * 64 * 8, 0);
    auto tap = [](int par, int d) { return par ? (d ? -1 : 1) : (d ? 2 : 0); };
    for (int cz = 0; cz < 3; ++cz)
        for (int cy = 0; cy < 3; ++cy) {
            const int tz = cz == 0 ? 0 : (cz == 1 ? 2 : 1), ty = cy == 0 ? 0 : (cy == 1 ? 2 : 1);
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, kg = lane >> 4, px = r >> 3, co = r & 7, dx = kg >> 1;
                const int tx = tap(px, dx);
                for (int c = 0; c < 8; ++c) {
                    const int ci = 8 * (kg & 1) + c;
                    float w = tx >= 0 ? W[((size_t)((tz * 3 + ty) * 3 + tx) * 8 + co) * 16 + ci] : 0.f;
                    for (int pc = 0; pc < 3; ++pc) {
                        const unsigned short h = pc < 2 ? t3_bf16_rne(w) : t3_bf16_rne(w);      // (the last remainder has <= 8 bits: exact)
                        plan->h_W[((size_t)((cz * 3 + cy) * 3 + pc) * 64 + lane) * 8 + c] = h;
                        w -= t3_bf16_f(h);
                    }
                }
            }
        }
}

// kind 8 (32 -> 16 channels): W [tap][co (16)][ci (32)]; A fragment of (cz, cy, cx): lane -> row co = lane & 15, k = ci = 8 (lane >> 4) + c;
// tap per dimension from c: 0 -> tap 0, 1 -> tap 2, 2 -> tap 1.
void t3d8_fwd_pack(T3dPlan *plan, const float *W) {
    plan->h_W.assign((size_t)27 * 3 * 64 * 8, 0);
    auto tap = [](int c) { return c == 0 ? 0 : (c == 1 ? 2 : 1); };
    for (int cz = 0; cz < 3; ++cz)
        for (int cy = 0; cy < 3; ++cy)
            for (int cx = 0; cx < 3; ++cx) {
                const int combo = (cz * 3 + cy) * 3 + cx, t = (tap(cz) * 3 + tap(cy)) * 3 + tap(cx);
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = lane & 15, kg = lane >> 4;
                    for (int c = 0; c < 8; ++c) {
                        float w = W[((size_t)t * 16 + co) * 32 + 8 * kg + c];
                        for (int pc = 0; pc < 3; ++pc) {
                            const unsigned short h = t3_bf16_rne(w);
                            plan->h_W[((size_t)(combo * 3 + pc) * 64 + lane) * 8 + c] = h;
                            w -= t3_bf16_f(h);
                        }
                    }
                }
            }
}

// Backward A fragment of (tz, ty): lane -> row ci = lane & 15, k-group = x tap tx = lane >> 4 (3: zero), k = co.  fp16 pairs of
// w * 2^e_w at their true scale.
void t3d_bwd_pack(T3dPlan *plan, const float *W) {
    float amax = 0.f;
    for (size_t i = 0; i < (size_t)27 * 8 * 16; ++i) amax = std::max(amax, std::fabs(W[i]));
    int ex = 0;
    if (amax > 0.f) (void)std::frexp(amax, &ex);
    plan->w_exp = 14 - ex;
    plan->h_W.assign((size_t)9 * 2 * 64 * 8, 0);
    for (int tz = 0; tz < 3; ++tz)
        for (int ty = 0; ty < 3; ++ty)
            for (int lane = 0; lane < 64; ++lane) {
                const int ci = lane & 15, tx = lane >> 4;
                for (int co = 0; co < 8; ++co) {
                    const float w = tx < 3 ? W[((size_t)((tz * 3 + ty) * 3 + tx) * 8 + co) * 16 + ci] : 0.f;
                    const float ws = std::ldexp(w, plan->w_exp);
                    const _Float16 h = (_Float16)ws;
                    const _Float16 l = (_Float16)(ws - (float)h);
                    unsigned short hb, lb;
                    std::memcpy(&hb, &h, 2);
                    std::memcpy(&lb, &l, 2);
                    plan->h_W[((size_t)((tz * 3 + ty) * 2 + 0) * 64 + lane) * 8 + co] = hb;
                    plan->h_W[((size_t)((tz * 3 + ty) * 2 + 1) * 64 + lane) * 8 + co] = lb;
                }
            }
}

static unsigned t3_grid(alq_ctx *ctx, int N) {
    // two workgroups per CU; units = 4 per patch, dealt per XCD (t3_unit): a multiple of 8 workgroups
    const int cus = ctx->num_cus;
    const long long units = (long long)N * 4;
    int per_cu = 2;
    if (const char *e = getenv("ALQ_T3D_WGS")) per_cu = std::max(1, std::min(2, atoi(e)));      // (tuning / diagnostics)
    long long g = std::min<long long>((long long)per_cu * cus, units);
    g = std::max<long long>(8, (g + 7) / 8 * 8);
    return (unsigned)g;
}

int t3d_fwd_launch(alq_ctx *ctx, const T3dPlan &plan, const View &in, const View &out, const float *bias, int N, float *osum, unsigned *out_amax) {
    ALQ_REQUIRE(plan.ok && plan.d_W, ALQ_EINVAL, "t3d: weights not set");
    if (plan.kind == 8) {
        ALQ_REQUIRE(in.cs == 32 && in.c0 == 0 && in.split == 0 && out.cs == 16 && out.c0 == 0 && out.split == 0 && in.D == 8 && in.H == 8 && in.W == 8 &&
                    out.D == 16 && bias && !out_amax, ALQ_EINVAL, "t3d8: view mismatch");
        ALQ_REQUIRE(N < 4096, ALQ_EUNSUPPORTED, "t3d: 32-bit byte offsets hold fewer than 4096 patches per pass");
        if (N <= 0) return ALQ_OK;
        T8FwdArgs a8;
        a8.in = in.p; a8.out = out.p; a8.W = reinterpret_cast<const unsigned short *>(plan.d_W); a8.bias = bias; a8.osum = osum; a8.N = N;
        const int cus = ctx->num_cus;
        const dim3 grid8((unsigned)std::min(N, cus));
        ProfScope ps8(ctx, PROF_IGEMM3_FWD, plan.flops_per_patch * N);
        auto go = [&](auto kfn) -> int {
            ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, T8_WBYTES + 8 * T8_WAVE));
            hipLaunchKernelGGL(kfn, grid8, dim3(512), T8_WBYTES + 8 * T8_WAVE, ctx->stream, a8);
            return ALQ_OK;
        };
        ALQ_TRY(osum ? go(t3d8_fwd_kernel<true>) : go(t3d8_fwd_kernel<false>));
        ALQ_HIP(hipGetLastError());
        return ALQ_OK;
    }
    ALQ_REQUIRE(in.cs == 16 && in.c0 == 0 && in.split == 0 && out.cs == 8 && out.c0 == 0 && out.split == 0 && in.D == 16 && out.D == 32 && bias,
                ALQ_EINVAL, "t3d: view mismatch");
    ALQ_REQUIRE(N < 4096, ALQ_EUNSUPPORTED, "t3d: 32-bit byte offsets hold fewer than 4096 patches per pass");
    if (N <= 0) return ALQ_OK;
    T3FwdArgs a;
    a.in = in.p; a.out = out.p; a.W = reinterpret_cast<const unsigned short *>(plan.d_W); a.bias = bias; a.osum = osum; a.out_amax = out_amax; a.N = N;
    ProfScope ps(ctx, PROF_IGEMM3_FWD, plan.flops_per_patch * N);
    const dim3 grid(t3_grid(ctx, N));
    if (osum && out_amax) hipLaunchKernelGGL((t3d_fwd_kernel<true, true>), grid, dim3(256), 4 * T3F_WAVE, ctx->stream, a);
    else if (osum) hipLaunchKernelGGL((t3d_fwd_kernel<true, false>), grid, dim3(256), 4 * T3F_WAVE, ctx->stream, a);
    else if (out_amax) hipLaunchKernelGGL((t3d_fwd_kernel<false, true>), grid, dim3(256), 4 * T3F_WAVE, ctx->stream, a);
    else hipLaunchKernelGGL((t3d_fwd_kernel<false, false>), grid, dim3(256), 4 * T3F_WAVE, ctx->stream, a);
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

int t3d_bwd_launch(alq_ctx *ctx, const T3dPlan &plan, const View &dout, const View &din, int N, float in_bound, const unsigned char *mask_bits, float *dsum) {
    ALQ_REQUIRE(plan.ok && plan.d_W, ALQ_EINVAL, "t3d: weights not set");
    if (plan.kind == 8) return t3d8_bwd_launch(ctx, plan, dout, din, N, in_bound, mask_bits, dsum);
    ALQ_REQUIRE(dout.cs == 8 && dout.c0 == 0 && dout.split == 0 && din.cs == 16 && din.c0 == 0 && din.split == 0 && din.D == 16 && dout.D == 32 && in_bound > 0.f,
                ALQ_EINVAL, "t3d: view mismatch");
    ALQ_REQUIRE(N < 4096, ALQ_EUNSUPPORTED, "t3d: 32-bit byte offsets hold fewer than 4096 patches per pass");
    if (N <= 0) return ALQ_OK;
    int ex = 0;
    (void)std::frexp(in_bound, &ex);
    const int e_in = 14 - ex;
    T3BwdArgs a;
    a.dout = dout.p; a.din = din.p; a.W = reinterpret_cast<const unsigned short *>(plan.d_W); a.mask_bits = mask_bits; a.dsum = dsum;
    a.scale = std::ldexp(1.f, e_in); a.inv = std::ldexp(1.f, -(e_in + plan.w_exp)); a.N = N;
    ProfScope ps(ctx, PROF_IGEMM_F16, plan.flops_per_patch * N);
    const dim3 grid(t3_grid(ctx, N));
    if (mask_bits && dsum) hipLaunchKernelGGL((t3d_bwd_kernel<true, true>), grid, dim3(256), 4 * T3B_WAVE, ctx->stream, a);
    else if (mask_bits) hipLaunchKernelGGL((t3d_bwd_kernel<true, false>), grid, dim3(256), 4 * T3B_WAVE, ctx->stream, a);
    else if (dsum) hipLaunchKernelGGL((t3d_bwd_kernel<false, true>), grid, dim3(256), 4 * T3B_WAVE, ctx->stream, a);
    else hipLaunchKernelGGL((t3d_bwd_kernel<false, false>), grid, dim3(256), 4 * T3B_WAVE, ctx->stream, a);
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

}  // namespace alq
