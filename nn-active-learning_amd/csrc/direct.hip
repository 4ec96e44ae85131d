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
    const float *W;            // [27][8]
    const float *bias;
    int out_cs, out_c0, po_cs, po_c0;
    int D, H, Wd, N, relu;
    int tilesZ, tilesY, tilesX;
};

__global__ __launch_bounds__(256, 2) void direct_conv_pool_kernel(const DirectPoolArgs a) {
    constexpr int TWZ = 4, TWY = 8, TWX = 8;                 // windows per workgroup
    constexpr int HZ = 2 * TWZ + 2, HY = 2 * TWY + 2, HX = 2 * TWX + 2;
    // LDS: weights, the halo'd input block, and the whole 8 x 16 x 16 x 8 output block: threads own windows
    // (x stride of 2 voxels between lanes), so direct stores would touch a new cache line per lane; the block is
    // written to LDS in window order and stored by voxel-consecutive lanes as full 2 KB runs
    extern __shared__ __attribute__((aligned(16))) float dcp_lds[];
    float *Wl = dcp_lds;                                   // 27 * 8
    float *Al = dcp_lds + 224;                             // HZ * HY * HX = 3240
    float *Ol = dcp_lds + 224 + ((HZ * HY * HX + 3) & ~3); // 2048 voxels x 8
    const int tid = threadIdx.x;
    for (int i = tid; i < 27 * 8; i += 256) Wl[i] = a.W[i];
    int t = blockIdx.x;
    const int tx = t % a.tilesX; t /= a.tilesX;
    const int ty = t % a.tilesY; t /= a.tilesY;
    const int tz = t % a.tilesZ; t /= a.tilesZ;
    const long long n = t;
    const int z0 = tz * 2 * TWZ, y0 = ty * 2 * TWY, x0 = tx * 2 * TWX;
    const float *src = a.in + n * (long long)a.D * a.H * a.Wd;
    {   // all loads of the halo'd block first (13 in flight per thread), then the LDS writes: one memory latency
        // per workgroup instead of one per element (the rolled loop made this phase the whole kernel time)
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
                v = src[((long long)iz * a.H + iy) * a.Wd + ix];
            stg[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            if (i < HZ * HY * HX) Al[i] = stg[it];
        }
    }
    __syncthreads();
    const int wx = tid % TWX, wy = (tid / TWX) % TWY, wz = tid / (TWX * TWY);
    const int pz = tz * TWZ + wz, py = ty * TWY + wy, px = tx * TWX + wx;      // pooled coordinates
    const int PD = a.D / 2, PH = a.H / 2, PW = a.Wd / 2;
    const bool wlive = pz < PD && py < PH && px < PW;
    // 4x4x4 input block in registers for the whole thread; the 8 output channels in two halves, so that only
    // 8 voxels x 4 channels of accumulators are live at a time (2 waves per SIMD)
    float in[4][4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float *row = Al + ((2 * wz + p) * HY + (2 * wy + q)) * HX + 2 * wx;
            const f32x2 lo = *reinterpret_cast<const f32x2 *>(row), hi = *reinterpret_cast<const f32x2 *>(row + 2);
            in[p][q][0] = lo.x; in[p][q][1] = lo.y; in[p][q][2] = hi.x; in[p][q][3] = hi.y;
        }
    float psum = 0.f;
    const long long pvox0 = ((n * PD + pz) * PH + py) * PW + px;
#pragma unroll 1
    for (int ch = 0; ch < 2; ++ch) {
        float acc[8][4];
#pragma unroll
        for (int v = 0; v < 8; ++v)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[v][c] = a.bias ? a.bias[ch * 4 + c] : 0.f;
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(Wl + ((dz * 3 + dy) * 3 + dx) * 8 + ch * 4);
                    const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int v = 0; v < 8; ++v) {
                        const float xv = in[(v >> 2) + dz][((v >> 1) & 1) + dy][(v & 1) + dx];
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[v][c] = fmaf(xv, wv[c], acc[v][c]);
                    }
                }
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned bidx = 0u;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const int z = 2 * pz + (v >> 2), y = 2 * py + ((v >> 1) & 1), x = 2 * px + (v & 1);
            const long long vox = ((n * a.D + z) * a.H + y) * a.Wd + x;
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = a.relu ? fmaxf(acc[v][c], 0.f) : acc[v][c];
            (void)vox;
            const int lv = ((2 * wz + (v >> 2)) * (2 * TWY) + 2 * wy + ((v >> 1) & 1)) * (2 * TWX) + 2 * wx + (v & 1);
            *reinterpret_cast<f32x4 *>(Ol + lv * 8 + ch * 4) = f32x4{o[0], o[1], o[2], o[3]};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (o[c] > best[c]) {               // window order (dz, dy, dx): first maximum wins
                    best[c] = o[c];
                    bidx = (bidx & ~(255u << (8 * c))) | ((unsigned)v << (8 * c));
                }
        }
        if (wlive) {
            *reinterpret_cast<f32x4 *>(a.pout + pvox0 * a.po_cs + a.po_c0 + ch * 4) = f32x4{best[0], best[1], best[2], best[3]};
            a.argmax[pvox0 * 2 + ch] = bidx;
        }
        psum += (best[0] + best[1]) + (best[2] + best[3]);
    }
    if (wlive && a.posum) a.posum[pvox0] = psum;
    __syncthreads();
    // voxel-consecutive lanes: 32 B per lane, 2 KB contiguous per wave along x
    float amx = 0.f;
    for (int lv = tid; lv < 2 * TWZ * 2 * TWY * 2 * TWX; lv += 256) {
        const int lx = lv % (2 * TWX), ly = (lv / (2 * TWX)) % (2 * TWY), lz = lv / (2 * TWX * 2 * TWY);
        const int z = z0 + lz, y = y0 + ly, x = x0 + lx;
        if (z >= a.D || y >= a.H || x >= a.Wd) continue;
        const long long vox = ((n * a.D + z) * a.H + y) * a.Wd + x;
        const f32x4 g0 = *reinterpret_cast<const f32x4 *>(Ol + lv * 8), g1 = *reinterpret_cast<const f32x4 *>(Ol + lv * 8 + 4);
        float *orow = a.out + vox * a.out_cs + a.out_c0;
        *reinterpret_cast<f32x4 *>(orow) = g0;
        *reinterpret_cast<f32x4 *>(orow + 4) = g1;
        if (a.osum) {
            float sum = 0.f;
            sum += (g0.x + g0.y) + (g0.z + g0.w);
            sum += (g1.x + g1.y) + (g1.z + g1.w);
            a.osum[vox] = sum;
        }
        if (a.amax)
            amx = fmaxf(amx, fmaxf(fmaxf(fmaxf(__builtin_fabsf(g0.x), __builtin_fabsf(g0.y)), fmaxf(__builtin_fabsf(g0.z), __builtin_fabsf(g0.w))),
                                   fmaxf(fmaxf(__builtin_fabsf(g1.x), __builtin_fabsf(g1.y)), fmaxf(__builtin_fabsf(g1.z), __builtin_fabsf(g1.w)))));
    }
    if (a.amax) {      // 16 workgroups per 32^3 patch: one atomic each
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) amx = fmaxf(amx, __shfl_xor(amx, off, 64));
        __syncthreads();                 // Ol is read no more: its first words carry the four wave maxima
        if ((tid & 63) == 0) Ol[tid >> 6] = amx;
        __syncthreads();
        if (tid == 0) atomicMax(a.amax + n, __builtin_bit_cast(unsigned, fmaxf(fmaxf(Ol[0], Ol[1]), fmaxf(Ol[2], Ol[3]))));
    }
}

// eligibility is checked by the caller (model.hip): 3x3x3 SAME conv of one channel into 8, 2x2x2 pool, even dims
int direct_conv_pool_launch(alq_ctx *ctx, const float *d_W, const View &in, const View &out, const View &pout,
                            const float *bias, int relu, uint8_t *argmax, float *osum, float *posum, int N,
                            double flops_per_patch, unsigned *amax) {
    ALQ_REQUIRE(in.C == 1 && in.cs == 1 && in.c0 == 0 && out.C == 8 && pout.C == 8 && d_W, ALQ_EINVAL, "direct conv+pool: bad views");
    ALQ_REQUIRE(((out.cs | out.c0 | pout.cs | pout.c0) & 3) == 0 && in.D % 2 == 0 && in.H % 2 == 0 && in.W % 2 == 0 &&
                    pout.D * 2 == in.D && pout.H * 2 == in.H && pout.W * 2 == in.W,
                ALQ_EUNSUPPORTED, "direct conv+pool: unsupported geometry");
    DirectPoolArgs a;
    a.in = in.p; a.out = out.p; a.pout = pout.p; a.argmax = reinterpret_cast<unsigned *>(argmax);
    a.osum = osum; a.posum = posum; a.W = d_W; a.bias = bias; a.amax = amax;
    a.out_cs = out.cs; a.out_c0 = out.c0; a.po_cs = pout.cs; a.po_c0 = pout.c0;
    a.D = in.D; a.H = in.H; a.Wd = in.W; a.N = N; a.relu = relu;
    a.tilesZ = (pout.D + 3) / 4; a.tilesY = (pout.H + 7) / 8; a.tilesX = (pout.W + 7) / 8;
    ProfScope ps(ctx, PROF_DIRECT, flops_per_patch * N);
    const size_t lds = (224 + 3240 + 2048 * 8) * sizeof(float);
    ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(direct_conv_pool_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(direct_conv_pool_kernel, dim3((unsigned)((long long)N * a.tilesZ * a.tilesY * a.tilesX)), dim3(256), lds,
                       ctx->stream, a);
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
