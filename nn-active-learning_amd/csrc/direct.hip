// Direct (VALU) convolution for the first layer of a patch network: 1-4 input channels.
//
// With Ci = 1 a 3x3x3 conv has K = 27: far too little contraction depth to pay for the MFMA
// engine's staging (the general kernel sat at 9 % matrix-pipe use on it), and on gfx950 the fp32
// MFMA shares the vector ALUs anyway.  Here one thread owns one output voxel and CO accumulators;
// the halo'd input block and the whole weight tensor sit in LDS (weights are read as wave-uniform
// broadcasts), the output row of a voxel is CO contiguous floats -> 16-byte stores that are
// contiguous across the lanes of a wave; bias / ReLU / channel sum are fused.
#include "alq_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace alq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int CO>
__global__ __launch_bounds__(256) void direct_conv_kernel(const DirectArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int K = a.ntaps * a.Ci;
    float *Wl = lds;                       // [K][CO]
    float *Al = lds + ((K * CO + 3) & ~3); // [PT][HZ][HY][HX][Ci]
    for (int i = tid; i < K * CO; i += 256) Wl[i] = a.W[i];

    int t = blockIdx.x;
    const int tx = t % a.tilesX; t /= a.tilesX;
    const int ty = t % a.tilesY; t /= a.tilesY;
    const int tz = t % a.tilesZ; t /= a.tilesZ;
    const int p0 = t * a.PT;
    const int mz0 = tz * a.TZ, my0 = ty * a.TY, mx0 = tx * a.TX;
    const int nhv = a.PT * a.HZ * a.HY * a.HX;
    for (int i = tid; i < nhv * a.Ci; i += 256) {
        int r = i / a.Ci;
        const int c = i - r * a.Ci;
        const int hx = r % a.HX; r /= a.HX;
        const int hy = r % a.HY; r /= a.HY;
        const int hz = r % a.HZ; r /= a.HZ;
        const int patch = p0 + r;
        const int iz = mz0 + a.minz + hz, iy = my0 + a.miny + hy, ix = mx0 + a.minx + hx;
        float v = 0.f;
        if (patch < a.N && iz >= 0 && iz < a.ID && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW)
            v = a.in[((((long long)patch * a.ID + iz) * a.IH + iy) * a.IW + ix) * a.in_cs + a.in_c0 + c];
        Al[i] = v;
    }
    __syncthreads();

    // this thread's output voxel
    const int TV = a.TZ * a.TY * a.TX;
    const int v = tid;
    const int pt = v / TV;
    int q = v - pt * TV;
    const int x = q % a.TX; q /= a.TX;
    const int y = q % a.TY;
    const int z = q / a.TY;
    const bool live = v < a.rows && p0 + pt < a.N && mz0 + z < a.MD && my0 + y < a.MH && mx0 + x < a.MW;
    const int base = (((pt * a.HZ + z) * a.HY + y) * a.HX + x) * a.Ci;

    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = a.bias ? a.bias[c] : 0.f;
    int kk = 0;
    for (int dz = 0; dz < a.tnz; ++dz)
        for (int dy = 0; dy < a.tny; ++dy) {
            const int rowoff = base + ((dz * a.HY + dy) * a.HX) * a.Ci;
            for (int dxc = 0; dxc < a.tnx * a.Ci; ++dxc, ++kk) {      // (dx, ci) are contiguous in the halo row
                const float xv = Al[rowoff + dxc];
                const float *w = Wl + kk * CO;
#pragma unroll
                for (int c = 0; c < CO; c += 4) {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + c);      // wave-uniform address: LDS broadcast
                    acc[c] = fmaf(xv, wv.x, acc[c]);
                    acc[c + 1] = fmaf(xv, wv.y, acc[c + 1]);
                    acc[c + 2] = fmaf(xv, wv.z, acc[c + 2]);
                    acc[c + 3] = fmaf(xv, wv.w, acc[c + 3]);
                }
            }
        }
    if (!live) return;
    const long long ovox = (((long long)(p0 + pt) * a.OD + mz0 + z) * a.OH + my0 + y) * a.OW + mx0 + x;
    float *orow = a.out + ovox * a.out_cs + a.out_c0;
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CO; c += 4) {
        f32x4 o = f32x4{acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
        if (a.relu) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        *reinterpret_cast<f32x4 *>(orow + c) = o;
        sum += (o.x + o.y) + (o.z + o.w);
    }
    if (a.osum) a.osum[ovox] = sum;
}

// ---- first conv (1 input channel, 3x3x3, 8 output channels) fused with the 2x2x2 max-pool behind it --------
// One thread owns one pooling window: 2x2x2 output voxels x 8 channels = 64 accumulators over a 4x4x4 input
// block held in registers (14.6 FMAs per LDS read instead of 2.7 in the voxel-per-thread kernel above), so the
// pool is a register-level max and the 32^3 x 8 activation is written once and never read back for pooling.
// Accumulation order per output equals direct_conv_kernel's (bias, then taps z, y, x ascending): same bits.
struct DirectPoolArgs {
    const float *in;           // [N, D, H, W] (one channel)
    float *out;                // conv output view
    float *pout;               // pooled output view
    unsigned *argmax;          // [N * pooled voxels][2] packed window indices (4 channels per word)
    float *osum, *posum;       // channel sums of the conv output / pooled output (or null)
    unsigned *amax;            // [N] max |conv output| per patch as float bits (atomic max; zeroed by the caller) or null
    unsigned char *sg;         // sign field of the conv output (View::sg: one byte per 4 channels, bit k = (value k > 0)) or null
    unsigned char *psg;        // sign field of the pooled output or null
    const float *W;            // [27][8]
    const float *bias;
    int out_cs, out_c0, po_cs, po_c0;
    int D, H, Wd, N, relu;
    int tilesZ, tilesY, tilesX;
};

// Schedule: ONE workgroup barrier (after the halo'd input block is in LDS); from there every wave works alone on its
// slab of 1 x 8 x 8 windows = 2 x 16 x 16 voxels, one z plane (4 voxels x 8 channels per thread) per pass; 33 KB of
// LDS and <= 128 registers: four workgroups (16 waves) per CU.
// The kernel is VALU-issue bound (PMC: SQ_ACTIVE_INST_VALU 60 % of the SIMD time, 2.4 k vector instructions per wave of
// which 864 are the packed FMAs), so everything around the FMAs is written for instruction count:
//  * halo'd block: rows of 16 aligned floats + two edge values -> 3 float4 + 2 scalar loads per thread with one
//    division per item (13 scalar loads with two divisions and 64-bit addresses each were 690 instructions, 29 %);
//  * v_pk_fma_f32 takes the broadcast input straight from one half of an aligned register pair (op_sel), so the
//    64 inputs live in 32 pairs instead of 64 duplicated pairs (no copies, 64 registers less);
//  * channel sums and the patch maximum come from the registers that hold the results, not from the store loop;
//  * half a plane goes through a wave-private 4 KB LDS buffer that turns the window-per-lane register layout (64 B lane
//    stride) into 1 KB runs per store instruction.  16-byte chunk q of a row (32 chunks) sits at position
//    q ^ ((q >> 3) & 3): ds_write_b128 serves 8 consecutive lanes per cycle over 32 banks, and the 8 windows of a row
//    (64 B apart) then hit 8 different 16-byte bank groups instead of 2 (4-way conflicts: SQ_LDS_BANK_CONFLICT was
//    40 % of the LDS cycles); the reading lanes take consecutive positions, and since the XOR only permutes chunks
//    inside a 64-byte segment a store instruction still covers a contiguous 1 KB.
// Accumulation order per output is unchanged (bias, then taps z, y, x ascending, one fma each): same bits.
// the weight pair is a scalar register pair: the weights come through the scalar cache (s_load), not through LDS
// broadcasts (108 ds_read_b128 per wave were a quarter of the LDS cycles) and hold no vector registers
static __device__ __forceinline__ void dcp_fma_lo(f32x2 &acc, const f32x2 x, const f32x2 w) {     // acc += x.lo * w
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(x), "s"(w));
}
static __device__ __forceinline__ void dcp_fma_hi(f32x2 &acc, const f32x2 x, const f32x2 w) {     // acc += x.hi * w
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "s"(w));
}
typedef const __attribute__((address_space(4))) f32x4 *dcp_cw4;

template <bool ALIGNED>      // ALIGNED: the row length is a multiple of 4 voxels (float4 loads of the block's rows)
__global__ __launch_bounds__(256, 4) void direct_conv_pool_kernel(const DirectPoolArgs a) {
    constexpr int TWZ = 4, TWY = 8, TWX = 8;                 // windows per workgroup
    constexpr int HZ = 2 * TWZ + 2, HY = 2 * TWY + 2, HX = 2 * TWX + 2;
    constexpr int AX = 24;                                   // LDS row: x = -1 at column 3, the 16 aligned voxels from column 4
    extern __shared__ __attribute__((aligned(16))) float dcp_lds[];
    float *Al = dcp_lds;                                   // HZ * HY rows of AX
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float *Ol = dcp_lds + HZ * HY * AX + wv * 1024;  // this wave's half plane: 8 x 16 voxels x 8 channels
    const dcp_cw4 Wc = (dcp_cw4)(a.W);          // [27][8], read-only for the launch: constant address space -> s_load
    int t = blockIdx.x;
    const int tx = t % a.tilesX; t /= a.tilesX;
    const int ty = t % a.tilesY; t /= a.tilesY;
    const int tz = t % a.tilesZ; t /= a.tilesZ;
    const long long n = t;
    const int z0 = tz * 2 * TWZ, y0 = ty * 2 * TWY, x0 = tx * 2 * TWX;
    const long long pv0 = n * (long long)a.D * a.H * a.Wd;
    const float *src = a.in + pv0;
    if constexpr (ALIGNED) {
        // all loads first, then the LDS writes: one memory latency per workgroup
        f32x4 mid[3];
        float edge[2];
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int k = tid + it * 256, row = k >> 2, part = k & 3;
            const int hz = row / HY, hy = row - hz * HY;
            const int iz = z0 + hz - 1, iy = y0 + hy - 1, ix = x0 + 4 * part;
            mid[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k < HZ * HY * 4 && (unsigned)iz < (unsigned)a.D && (unsigned)iy < (unsigned)a.H && ix < a.Wd)
                mid[it] = *reinterpret_cast<const f32x4 *>(src + (iz * a.H + iy) * a.Wd + ix);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = tid + it * 256, row = k >> 1, side = k & 1;
            const int hz = row / HY, hy = row - hz * HY;
            const int iz = z0 + hz - 1, iy = y0 + hy - 1, ix = x0 + (side ? HX - 2 : -1);
            edge[it] = 0.f;
            if (k < HZ * HY * 2 && (unsigned)iz < (unsigned)a.D && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.Wd)
                edge[it] = src[(iz * a.H + iy) * a.Wd + ix];
        }
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int k = tid + it * 256;
            if (k < HZ * HY * 4) *reinterpret_cast<f32x4 *>(Al + (k >> 2) * AX + 4 + 4 * (k & 3)) = mid[it];
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = tid + it * 256;
            if (k < HZ * HY * 2) Al[(k >> 1) * AX + ((k & 1) ? HX + 2 : 3)] = edge[it];
        }
    } else {
        constexpr int NIT = (HZ * HY * HX + 255) / 256;
        float stg[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            int r = i;
            const int hx = r % HX; r /= HX;
            const int hy = r % HY;
            const int hz = r / HY;
            const int iz = z0 + hz - 1, iy = y0 + hy - 1, ix = x0 + hx - 1;
            float v = 0.f;
            if (i < HZ * HY * HX && iz >= 0 && iz < a.D && iy >= 0 && iy < a.H && ix >= 0 && ix < a.Wd)
                v = src[(iz * a.H + iy) * a.Wd + ix];
            stg[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            if (i < HZ * HY * HX) Al[(i / HX) * AX + 3 + i % HX] = stg[it];
        }
    }
    __syncthreads();
    const int wx = lane % TWX, wy = lane / TWX, wz = wv;
    const int pz = tz * TWZ + wz, py = ty * TWY + wy, px = tx * TWX + wx;      // pooled coordinates
    const int PD = a.D / 2, PH = a.H / 2, PW = a.Wd / 2;
    const bool wlive = pz < PD && py < PH && px < PW;
    // 4x4x4 input block of the thread's window in 32 register pairs
    f32x2 in[4][4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float *row = Al + ((2 * wz + p) * HY + (2 * wy + q)) * AX + 3 + 2 * wx;
            in[p][q][0] = f32x2{row[0], row[1]};
            in[p][q][1] = f32x2{row[2], row[3]};
        }
    const unsigned pvox = (unsigned)((pz * PH + py) * PW + px);                // pooled voxel inside the patch
    const long long pp0 = n * (long long)PD * PH * PW;
    char *pbase = reinterpret_cast<char *>(a.pout + pp0 * a.po_cs + a.po_c0);    // uniform bases, 32-bit lane offsets
    char *abase = reinterpret_cast<char *>(a.argmax + pp0 * 2);
    float best[8];
    unsigned bidx0 = 0u, bidx1 = 0u;
#pragma unroll
    for (int c = 0; c < 8; ++c) best[c] = -INFINITY;
    f32x2 bias2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bias2[c] = a.bias ? *reinterpret_cast<const f32x2 *>(a.bias + 2 * c) : f32x2{0.f, 0.f};
    float amx = 0.f;
    char *obase = reinterpret_cast<char *>(a.out + pv0 * a.out_cs + a.out_c0);
    char *sbase = a.osum ? reinterpret_cast<char *>(a.osum + pv0) : nullptr;
    unsigned char *gbase = a.sg ? a.sg + ((pv0 * a.out_cs + a.out_c0) >> 2) : nullptr;       // out_cs, out_c0 multiples of 4 (launch)
    const bool full = z0 + 2 * TWZ <= a.D && y0 + 2 * TWY <= a.H && x0 + 2 * TWX <= a.Wd;      // uniform
    const int vox00 = (((z0 + 2 * wz) * a.H + y0 + 2 * wy) * a.Wd + x0 + 2 * wx);      // first voxel of the window, inside the patch
    const int sw = (wx >> 1) & 3;               // chunk swizzle: position = chunk ^ ((chunk >> 3) & 3), chunk >> 2 == wx
    // store loop: lane -> chunk position lp of the half plane's row 2 * it + lb (every other row of the plane); the
    // chunk there is lq
    const int lp = lane & 31, lb = lane >> 5, lq = lp ^ ((lp >> 3) & 3);
    const int voff = (2 * lb * a.Wd + (lq >> 1)) * a.out_cs + (lq & 1) * 4;
#pragma unroll
    for (int vz = 0; vz < 2; ++vz) {
        f32x2 acc[4][4];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[v][c] = bias2[c];
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f32x4 w0 = Wc[((dz * 3 + dy) * 3 + dx) * 2], w1 = Wc[((dz * 3 + dy) * 3 + dx) * 2 + 1];
                    const f32x2 w2[4] = {f32x2{w0.x, w0.y}, f32x2{w0.z, w0.w}, f32x2{w1.x, w1.y}, f32x2{w1.z, w1.w}};
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int xi = (v & 1) + dx;
                        const f32x2 xp = in[vz + dz][(v >> 1) + dy][xi >> 1];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (xi & 1) dcp_fma_hi(acc[v][c], xp, w2[c]);
                            else dcp_fma_lo(acc[v][c], xp, w2[c]);
                        }
                    }
                }
        const int z = z0 + 2 * wz + vz;
#pragma unroll
        for (int vy = 0; vy < 2; ++vy) {       // half a plane (the rows 2 * wy + vy) at a time through the wave's 4 KB of LDS
#pragma unroll
            for (int vx = 0; vx < 2; ++vx) {
                const int v = 2 * vy + vx;
                float o[8];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    o[2 * c] = a.relu ? fmaxf(acc[v][c].x, 0.f) : acc[v][c].x;
                    o[2 * c + 1] = a.relu ? fmaxf(acc[v][c].y, 0.f) : acc[v][c].y;
                }
                const int ly = 2 * wy + vy, lx = 2 * wx + vx, q0 = 2 * lx;
                float *orow = Ol + wy * 128;
                *reinterpret_cast<f32x4 *>(orow + ((q0 ^ sw) << 2)) = f32x4{o[0], o[1], o[2], o[3]};
                *reinterpret_cast<f32x4 *>(orow + (((q0 + 1) ^ sw) << 2)) = f32x4{o[4], o[5], o[6], o[7]};
                const bool vin = full || (z < a.D && y0 + ly < a.H && x0 + lx < a.Wd);
                if (vin) {
                    if (gbase) {      // 8 channels = two sign bytes, written as one 16-bit store
                        const unsigned lo4 = (o[0] > 0.f ? 1u : 0u) | (o[1] > 0.f ? 2u : 0u) | (o[2] > 0.f ? 4u : 0u) | (o[3] > 0.f ? 8u : 0u);
                        const unsigned hi4 = (o[4] > 0.f ? 1u : 0u) | (o[5] > 0.f ? 2u : 0u) | (o[6] > 0.f ? 4u : 0u) | (o[7] > 0.f ? 8u : 0u);
                        *reinterpret_cast<unsigned short *>(gbase + (((unsigned)(vox00 + ((vz * a.H + vy) * a.Wd + vx)) * (unsigned)a.out_cs) >> 2)) =
                            (unsigned short)(lo4 | (hi4 << 8));
                    }
                    if (sbase)
                        *reinterpret_cast<float *>(sbase + (unsigned)(vox00 + ((vz * a.H + vy) * a.Wd + vx)) * 4u) =
                            ((o[0] + o[1]) + (o[2] + o[3])) + ((o[4] + o[5]) + (o[6] + o[7]));
                    if (!a.relu)       // after a ReLU the patch maximum is the maximum of the pooled values (below)
                        amx = fmaxf(amx, fmaxf(fmaxf(fmaxf(__builtin_fabsf(o[0]), __builtin_fabsf(o[1])), fmaxf(__builtin_fabsf(o[2]), __builtin_fabsf(o[3]))),
                                               fmaxf(fmaxf(__builtin_fabsf(o[4]), __builtin_fabsf(o[5])), fmaxf(__builtin_fabsf(o[6]), __builtin_fabsf(o[7])))));
                }
                const unsigned wi = (unsigned)(vz * 4 + v);        // window order (dz, dy, dx): first maximum wins
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (o[c] > best[c]) { best[c] = o[c]; bidx0 = (bidx0 & ~(255u << (8 * c))) | (wi << (8 * c)); }
                    if (o[c + 4] > best[c + 4]) { best[c + 4] = o[c + 4]; bidx1 = (bidx1 & ~(255u << (8 * c))) | (wi << (8 * c)); }
                }
            }
            // the wave's LDS operations execute in issue order: the reads below see the writes above once both are
            // issued in this order (the wave barrier keeps the compiler from moving them)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int orow0 = ((z * a.H + y0 + vy) * a.Wd + x0) * a.out_cs;
            if (full) {
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    __builtin_nontemporal_store(*reinterpret_cast<const f32x4 *>(Ol + (it * 64 + lane) * 4),
                                                reinterpret_cast<f32x4 *>(obase + (unsigned)(orow0 + 4 * it * a.Wd * a.out_cs + voff) * 4u));
            } else {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = 2 * (2 * it + lb) + vy;
                    if (z < a.D && y0 + row < a.H && x0 + (lq >> 1) < a.Wd)
                        *reinterpret_cast<f32x4 *>(obase + (unsigned)(orow0 + 4 * it * a.Wd * a.out_cs + voff) * 4u) =
                            *reinterpret_cast<const f32x4 *>(Ol + (it * 64 + lane) * 4);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    if (wlive) {
        char *prow = pbase + pvox * (unsigned)a.po_cs * 4u;
        *reinterpret_cast<f32x4 *>(prow) = f32x4{best[0], best[1], best[2], best[3]};
        *reinterpret_cast<f32x4 *>(prow + 16) = f32x4{best[4], best[5], best[6], best[7]};
        *reinterpret_cast<uint2 *>(abase + pvox * 8u) = uint2{bidx0, bidx1};
        if (a.psg) {
            const unsigned lo4 = (best[0] > 0.f ? 1u : 0u) | (best[1] > 0.f ? 2u : 0u) | (best[2] > 0.f ? 4u : 0u) | (best[3] > 0.f ? 8u : 0u);
            const unsigned hi4 = (best[4] > 0.f ? 1u : 0u) | (best[5] > 0.f ? 2u : 0u) | (best[6] > 0.f ? 4u : 0u) | (best[7] > 0.f ? 8u : 0u);
            *reinterpret_cast<unsigned short *>(a.psg + (((pp0 + pvox) * a.po_cs + a.po_c0) >> 2)) = (unsigned short)(lo4 | (hi4 << 8));
        }
        if (a.posum)
            *reinterpret_cast<float *>(reinterpret_cast<char *>(a.posum + pp0) + pvox * 4u) =
                ((best[0] + best[1]) + (best[2] + best[3])) + ((best[4] + best[5]) + (best[6] + best[7]));
        // dims are even, so a window is inside the volume with all of its 8 voxels: after a ReLU (values >= 0) the
        // largest |output| of the patch is the largest pooled value
        if (a.relu)
            amx = fmaxf(fmaxf(fmaxf(best[0], best[1]), fmaxf(best[2], best[3])), fmaxf(fmaxf(best[4], best[5]), fmaxf(best[6], best[7])));
    }
    if (a.amax) {      // 64 waves per 32^3 patch: one atomic each
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) amx = fmaxf(amx, __shfl_xor(amx, off, 64));
        if (lane == 0) atomicMax(a.amax + n, __builtin_bit_cast(unsigned, amx));
    }
}

// eligibility is checked by the caller (model.hip): 3x3x3 SAME conv of one channel into 8, 2x2x2 pool, even dims
int direct_conv_pool_launch(alq_ctx *ctx, const float *d_W, const View &in, const View &out, const View &pout,
                            const float *bias, int relu, uint8_t *argmax, float *osum, float *posum, int N,
                            double flops_per_patch, unsigned *amax, unsigned char *sg, unsigned char *psg) {
    ALQ_REQUIRE(in.C == 1 && in.cs == 1 && in.c0 == 0 && out.C == 8 && pout.C == 8 && d_W, ALQ_EINVAL, "direct conv+pool: bad views");
    ALQ_REQUIRE(((out.cs | out.c0 | pout.cs | pout.c0) & 3) == 0 && in.D % 2 == 0 && in.H % 2 == 0 && in.W % 2 == 0 &&
                    pout.D * 2 == in.D && pout.H * 2 == in.H && pout.W * 2 == in.W,
                ALQ_EUNSUPPORTED, "direct conv+pool: unsupported geometry");
    DirectPoolArgs a;
    a.in = in.p; a.out = out.p; a.pout = pout.p; a.argmax = reinterpret_cast<unsigned *>(argmax);
    a.osum = osum; a.posum = posum; a.W = d_W; a.bias = bias; a.amax = amax; a.sg = sg; a.psg = psg;
    a.out_cs = out.cs; a.out_c0 = out.c0; a.po_cs = pout.cs; a.po_c0 = pout.c0;
    a.D = in.D; a.H = in.H; a.Wd = in.W; a.N = N; a.relu = relu;
    a.tilesZ = (pout.D + 3) / 4; a.tilesY = (pout.H + 7) / 8; a.tilesX = (pout.W + 7) / 8;
    ProfScope ps(ctx, PROF_DIRECT, flops_per_patch * N);
    const size_t lds = (10 * 18 * 24 + 4 * 1024) * sizeof(float);
    // in-patch offsets are 32-bit in the kernel
    ALQ_REQUIRE((long long)in.D * in.H * in.W * std::max(std::max(out.cs, pout.cs), 1) < (1LL << 29), ALQ_EUNSUPPORTED,
                "direct conv+pool: volume too large");
    const dim3 grid((unsigned)((long long)N * a.tilesZ * a.tilesY * a.tilesX));
    const bool aligned = in.W % 4 == 0 && (reinterpret_cast<uintptr_t>(in.p) & 15) == 0;
    if (aligned) {
        static bool attr_a = false;
        if (!attr_a) {
            ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(direct_conv_pool_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_a = true;
        }
        hipLaunchKernelGGL(direct_conv_pool_kernel<true>, grid, dim3(256), lds, ctx->stream, a);
    } else {
        static bool attr_u = false;
        if (!attr_u) {
            ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(direct_conv_pool_kernel<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_u = true;
        }
        hipLaunchKernelGGL(direct_conv_pool_kernel<false>, grid, dim3(256), lds, ctx->stream, a);
    }
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

int direct_build_plan(const IgemmPlan &p1, DirectPlan *dp) {
    dp->ok = false;
    const IgemmArgs &g = p1.a;
    if (!p1.smallc || g.Ci > 4 || g.sm != 1 || g.so != 1 || g.ooffz || g.ooffy || g.ooffx) return ALQ_OK;
    if (!(g.Co == 8 || g.Co == 16 || g.Co == 24 || g.Co == 32)) return ALQ_OK;
    if (g.rows != 256) return ALQ_OK;
    // the taps must be the full box [min, min+n) in x-fastest order (a forward conv's are)
    const int tnx = g.HX - g.TX + 1, tny = g.HY - g.TY + 1, tnz = g.HZ - g.TZ + 1;
    if (tnx * tny * tnz != g.ntaps) return ALQ_OK;
    for (int t = 0; t < g.ntaps; ++t) {
        const int ix = t % tnx, iy = (t / tnx) % tny, iz = t / (tnx * tny);
        if (p1.h_koff[t * g.Ci] != ((iz * g.HY + iy) * g.HX + ix) * g.Ci) return ALQ_OK;
    }
    DirectArgs &a = dp->a;
    std::memset(&a, 0, sizeof(a));
    a.Ci = g.Ci; a.ID = g.ID; a.IH = g.IH; a.IW = g.IW;
    a.Co = g.Co; a.OD = g.OD; a.OH = g.OH; a.OW = g.OW;
    a.MD = g.MD; a.MH = g.MH; a.MW = g.MW;
    a.PT = g.PT; a.TZ = g.TZ; a.TY = g.TY; a.TX = g.TX; a.HZ = g.HZ; a.HY = g.HY; a.HX = g.HX; a.rows = g.rows;
    a.minz = g.minz; a.miny = g.miny; a.minx = g.minx;
    a.ntaps = g.ntaps; a.tnx = tnx; a.tny = tny; a.tnz = tnz;
    a.tilesZ = g.tilesZ; a.tilesY = g.tilesY; a.tilesX = g.tilesX;
    const int K = g.ntaps * g.Ci;
    dp->lds_bytes = ((size_t)((K * g.Co + 3) & ~3) + (size_t)g.PT * g.HZ * g.HY * g.HX * g.Ci) * 4;
    if (dp->lds_bytes > 60 * 1024) return ALQ_OK;
    dp->flops_per_patch = p1.flops_per_patch;
    dp->ok = true;
    return ALQ_OK;
}

int direct_launch(alq_ctx *ctx, const DirectPlan &dp, const View &in, const View &out, const float *bias, int relu,
                  int N, float *osum, int prof_cls) {
    DirectArgs a = dp.a;
    ALQ_REQUIRE(in.C == a.Ci && in.D == a.ID && in.H == a.IH && in.W == a.IW && out.C == a.Co && out.D == a.OD &&
                    out.H == a.OH && out.W == a.OW,
                ALQ_EINVAL, "direct conv: views do not match the plan");
    ALQ_REQUIRE(out.cs % 4 == 0 && out.c0 % 4 == 0 && dp.d_W, ALQ_EINVAL, "direct conv: bad output slice or weights");
    a.in = in.p; a.in_cs = in.cs; a.in_c0 = in.c0;
    a.out = out.p; a.out_cs = out.cs; a.out_c0 = out.c0;
    a.W = dp.d_W; a.bias = bias; a.relu = relu; a.N = N; a.osum = osum;
    const int pgroups = (N + a.PT - 1) / a.PT;
    const dim3 grid((unsigned)(pgroups * a.tilesZ * a.tilesY * a.tilesX));
    ProfScope ps(ctx, prof_cls, dp.flops_per_patch * N);
    switch (a.Co) {
        case 8: hipLaunchKernelGGL(direct_conv_kernel<8>, grid, dim3(256), dp.lds_bytes, ctx->stream, a); break;
        case 16: hipLaunchKernelGGL(direct_conv_kernel<16>, grid, dim3(256), dp.lds_bytes, ctx->stream, a); break;
        case 24: hipLaunchKernelGGL(direct_conv_kernel<24>, grid, dim3(256), dp.lds_bytes, ctx->stream, a); break;
        case 32: hipLaunchKernelGGL(direct_conv_kernel<32>, grid, dim3(256), dp.lds_bytes, ctx->stream, a); break;
        default: set_error("direct conv: Co=%d", a.Co); return ALQ_EUNSUPPORTED;
    }
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

}  // namespace alq
